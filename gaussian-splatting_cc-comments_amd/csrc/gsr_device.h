// gsr_device.h -- device-side helpers shared by the gfx950 kernels.
//
// Per-Gaussian arithmetic follows the reference expression by expression so that integer
// outcomes (radii, tile rectangles, keys) are bit-comparable with the CPU oracle; the files
// that include the "exact" helpers are compiled with -ffp-contract=off.
// Reference paths are relative to submodules/diff-gaussian-rasterization/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GSR_TILE_X 16  // cuda_rasterizer/config.h:16
#define GSR_TILE_Y 16  // cuda_rasterizer/config.h:17
#define GSR_TILE_PIX 256

// cuda_rasterizer/auxiliary.h:22-39
#define GSR_SH_C0 0.28209479177387814f
#define GSR_SH_C1 0.4886025119029199f
__device__ static const float GSR_SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                              -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float GSR_SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                              0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                              -0.5900435899266435f};

// One 48-byte record per Gaussian: everything the blend kernels gather per (Gaussian, tile) instance, so a batch load is three
// 16-byte reads of one contiguous record.  Two kernels fill it -- the geometry kernel the first 32 bytes (two 16-byte stores),
// the SH colour kernel (helper stream) the last 16 (one store).  Round 2 had the colour at words 6..8: its 8 + 4-byte stores into
// lines the other kernel was writing cost 1.74x the bytes (85 MB written for 49).  Measured alternative, not kept: colour in a
// dense array of its own writes the minimum (53 MB) but every gathered 16-byte colour record then costs the blend kernels a
// line of its own (forward 393 -> 629 MB, backward 960 -> 1 237 MB of fabric traffic, both 1 % slower).
struct __attribute__((aligned(16))) GsrSplat {
	float x, y;           // pixel-space mean (forward.cu:294)
	float ca, cb;         // conic a, b
	float cc, opacity;    // conic c, opacity (forward.cu:320)
	uint32_t rect_min;    // tile rect min: x | y << 16
	uint32_t rect_wh;     // tile rect size: w | h << 16
	float r, g, b;        // colour (SH colour clamped at 0, or colors_precomp)
	uint32_t unused;
};
static_assert(sizeof(GsrSplat) == 48, "splat record must be 48 bytes");

// Per-(Gaussian,tile) gradient record written by the backward blend (12 floats, 9 used).
struct __attribute__((aligned(16))) GsrGradSlot {
	float dmx, dmy;       // dL/dmean2D.xy (already scaled by 0.5*W, 0.5*H)
	float dca, dcb;       // dL/dconic .x .y
	float dcc, dop;       // dL/dconic .w, dL/dopacity
	float dr, dg;
	float db, pad0, pad1, pad2;
};
static_assert(sizeof(GsrGradSlot) == 48, "grad slot must be 48 bytes");

struct GsrMat3 { float m[3][3]; };  // m[col][row] as glm
struct GsrVec3 { float x, y, z; };

// glm operator*(mat3,mat3), third_party/glm/glm/detail/type_mat3x3.inl:486-519
__device__ __forceinline__ GsrMat3 gsr_mat3_mul(const GsrMat3& a, const GsrMat3& b)
{
	GsrMat3 r;
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int w = 0; w < 3; w++)
			r.m[c][w] = a.m[0][w] * b.m[c][0] + a.m[1][w] * b.m[c][1] + a.m[2][w] * b.m[c][2];
	return r;
}

__device__ __forceinline__ GsrMat3 gsr_mat3_transpose(const GsrMat3& a)
{
	GsrMat3 r;
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int w = 0; w < 3; w++)
			r.m[c][w] = a.m[w][c];
	return r;
}

// float -> int with the device's saturating semantics (NaN -> 0), as cvt.rzi in the reference
__device__ __forceinline__ int gsr_f2i(float f)
{
	if (f != f) return 0;
	if (f >= 2147483648.0f) return 2147483647;
	if (f <= -2147483648.0f) return (-2147483647 - 1);
	return (int)f;
}

// auxiliary.h:42-45 (double arithmetic)
__device__ __forceinline__ float gsr_ndc2pix(float v, int S)
{
	return (float)(((v + 1.0) * S - 1.0) * 0.5);
}

// auxiliary.h:48-58
__device__ __forceinline__ void gsr_get_rect(float px, float py, int max_radius, int gx, int gy, int& minx,
                                             int& miny, int& maxx, int& maxy)
{
	minx = min(gx, max(0, gsr_f2i((px - max_radius) / GSR_TILE_X)));
	miny = min(gy, max(0, gsr_f2i((py - max_radius) / GSR_TILE_Y)));
	maxx = min(gx, max(0, gsr_f2i((px + max_radius + GSR_TILE_X - 1) / GSR_TILE_X)));
	maxy = min(gy, max(0, gsr_f2i((py + max_radius + GSR_TILE_Y - 1) / GSR_TILE_Y)));
}

// auxiliary.h:60-69
__device__ __forceinline__ GsrVec3 gsr_transform_point_4x3(const GsrVec3& p, const float* m)
{
	GsrVec3 r = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
	             m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
	return r;
}

// forward.cu:146-180 rotation part (quaternion used unnormalised)
__device__ __forceinline__ GsrMat3 gsr_build_R(float r, float x, float y, float z)
{
	GsrMat3 R;
	R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z); R.m[0][2] = 2.f * (x * z + r * y);
	R.m[1][0] = 2.f * (x * y + r * z); R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
	R.m[2][0] = 2.f * (x * z - r * y); R.m[2][1] = 2.f * (y * z + r * x); R.m[2][2] = 1.f - 2.f * (x * x + y * y);
	return R;
}

__device__ __forceinline__ GsrMat3 gsr_diag3(float a, float b, float c)
{
	GsrMat3 S;
#pragma unroll
	for (int i = 0; i < 3; i++)
#pragma unroll
		for (int j = 0; j < 3; j++) S.m[i][j] = 0.f;
	S.m[0][0] = a; S.m[1][1] = b; S.m[2][2] = c;
	return S;
}

// forward.cu:146-180 computeCov3D
__device__ __forceinline__ void gsr_cov3d(const float* scale, float mod, const float* q, float* cov3D)
{
	GsrMat3 S = gsr_diag3(mod * scale[0], mod * scale[1], mod * scale[2]);
	GsrMat3 R = gsr_build_R(q[0], q[1], q[2], q[3]);
	GsrMat3 M = gsr_mat3_mul(S, R);
	GsrMat3 Sigma = gsr_mat3_mul(gsr_mat3_transpose(M), M);
	cov3D[0] = Sigma.m[0][0]; cov3D[1] = Sigma.m[0][1]; cov3D[2] = Sigma.m[0][2];
	cov3D[3] = Sigma.m[1][1]; cov3D[4] = Sigma.m[1][2]; cov3D[5] = Sigma.m[2][2];
}

// ---- leaf-parameter activations (scene/gaussian_model.py:114-135 property getters) ----
// In "leaf" mode the per-Gaussian kernels read the optimiser's raw tensors (_scaling, _rotation,
// _opacity, _features_dc, _features_rest) and apply these themselves, in the arithmetic PyTorch's
// elementwise kernels use, so no activated copy ever exists in HBM.
__device__ __forceinline__ float gsr_act_exp(float x) { return expf(x); }                        // torch.exp
__device__ __forceinline__ float gsr_act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }  // torch.sigmoid
// torch.nn.functional.normalize(q): q / max(||q||_2, 1e-12); returns the denominator.  The pairwise
// order of the sum of squares is the one PyTorch-ROCm's row reduction produces for 4 elements
// (measured on MI355X with tools/norm_probe.py: bit-identical on 200 000 random rows).
__device__ __forceinline__ float gsr_act_normalize4(const float* q, float* out)
{
	const float n = sqrtf((q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]));
	const float d = fmaxf(n, 1e-12f);
	out[0] = q[0] / d; out[1] = q[1] / d; out[2] = q[2] / d; out[3] = q[3] / d;
	return d;
}

// ---- wave-cooperative staging of 64 Gaussians x 16 SH coefficients through LDS -----------------------
// Row layout in LDS: 13 float4 per Gaussian (48 floats used + 1 float4 pad).  A lane's own row is then 12
// conflict-free 16-byte accesses (13 is odd: the rows of a ds_read_b128 / ds_write_b128 lane group start in distinct
// bank quads).  The wave's block of 64 x 48 floats is contiguous in HBM; it is moved with 16-byte accesses in
// 12 wave instructions, each covering 16 rows x 64 bytes: lane = 4 * quad + e moves float4 number 4 * chunk + e of
// row 16 * rowblock + rowmap(quad).  The two row maps make the LDS side of the copy conflict-free as well
// (a plain "lane f moves float4 f" copy collides wherever a row boundary -- the pad -- falls inside a lane group:
// measured 3.5M conflict cycles per launch in the per-Gaussian backward at 1M Gaussians):
//   * stores into LDS (ds_write_b128: groups of 8 consecutive lanes = quads 2k, 2k+1, 8 bank quads): the two rows of a
//     group are 4 apart, 13 * 4 = 4 (mod 8), so the two 4-float4 chunks fall into different halves of the banks;
//   * loads from LDS (ds_read_b128: lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32, 16 bank
//     quads): the four rows of a group are r, r+4, r+8, r+12, 13 * 4k = 4k (mod 16).
#define GSR_SH_ROW4 13
#define GSR_SH_ROWF (4 * GSR_SH_ROW4)

__device__ __forceinline__ int gsr_sh_rowmap_store(int quad) { return ((quad >> 1) & 3) + 4 * (quad & 1) + 8 * (quad >> 3); }
__device__ __forceinline__ int gsr_sh_rowmap_load(int quad) { return (int)((0xFEAB6732DC894510ull >> (4 * quad)) & 15ull); }

// packed (P,16,3) tensor -> rows, in two steps so that a kernel can put other loads between the global loads (fetch)
// and their first use (commit)
__device__ __forceinline__ void gsr_sh_rows_fetch(float4* __restrict__ v, const float* __restrict__ shs, int wave_first, int nrows, int lane)
{
	const float4* src = reinterpret_cast<const float4*>(shs + (size_t)wave_first * 48);
	const int e = lane & 3, r0 = gsr_sh_rowmap_store(lane >> 2);
#pragma unroll
	for (int it = 0; it < 12; it++) {
		const int row = 16 * (it / 3) + r0, col = 4 * (it % 3) + e;
		v[it] = (row < nrows) ? src[row * 12 + col] : make_float4(0.f, 0.f, 0.f, 0.f);
	}
}
__device__ __forceinline__ void gsr_sh_rows_commit(float4* __restrict__ rows, const float4* __restrict__ v, int nrows, int lane)
{
	const int e = lane & 3, r0 = gsr_sh_rowmap_store(lane >> 2);
#pragma unroll
	for (int it = 0; it < 12; it++) {
		const int row = 16 * (it / 3) + r0, col = 4 * (it % 3) + e;
		if (row < nrows) rows[row * GSR_SH_ROW4 + col] = v[it];
	}
}
// one half of the fetched block (v[6 * half .. 6 * half + 5] = rows 32 * half .. 32 * half + 31) into LDS rows 0 .. 31
__device__ __forceinline__ void gsr_sh_rows_commit_half(float4* __restrict__ rows, const float4* __restrict__ v, int nrows, int lane, int half)
{
	const int e = lane & 3, r0 = gsr_sh_rowmap_store(lane >> 2);
#pragma unroll
	for (int it = 0; it < 6; it++) {
		const int lrow = 16 * (it / 3) + r0, col = 4 * (it % 3) + e;
		if (32 * half + lrow < nrows) rows[lrow * GSR_SH_ROW4 + col] = v[6 * half + it];
	}
}
__device__ __forceinline__ void gsr_sh_rows_load(float4* __restrict__ rows, const float* __restrict__ shs, int wave_first, int nrows, int lane)
{
	float4 v[12];
	gsr_sh_rows_fetch(v, shs, wave_first, nrows, lane);  // all twelve loads are in flight before the first LDS store
	gsr_sh_rows_commit(rows, v, nrows, lane);
}
__device__ __forceinline__ void gsr_sh_rows_store(const float4* __restrict__ rows, float* __restrict__ dst_shs, int wave_first, int nrows, int lane)
{
	float4* dst = reinterpret_cast<float4*>(dst_shs + (size_t)wave_first * 48);
	const int e = lane & 3, r0 = gsr_sh_rowmap_load(lane >> 2);
#pragma unroll
	for (int it = 0; it < 12; it++) {
		const int row = 16 * (it / 3) + r0, col = 4 * (it % 3) + e;
		if (row < nrows) dst[row * 12 + col] = rows[row * GSR_SH_ROW4 + col];
	}
}
// One half of the wave's block (rows 32 * half .. 32 * half + 31, staged at LDS rows 0 .. 31): six wave instructions.
// Staging the output in two halves halves the LDS a wave holds (6.6 KB: 4 waves per SIMD fit beside the registers).
__device__ __forceinline__ void gsr_sh_rows_store_half(const float4* __restrict__ rows, float* __restrict__ dst_shs, int wave_first, int nrows, int lane, int half)
{
	float4* dst = reinterpret_cast<float4*>(dst_shs + (size_t)wave_first * 48);
	const int e = lane & 3, r0 = gsr_sh_rowmap_load(lane >> 2);
#pragma unroll
	for (int it = 0; it < 6; it++) {
		const int lrow = 16 * (it / 3) + r0, row = 32 * half + lrow, col = 4 * (it % 3) + e;
		if (row < nrows) dst[row * 12 + col] = rows[lrow * GSR_SH_ROW4 + col];
	}
}
// the lane's own 48 coefficients out of / into its row
__device__ __forceinline__ void gsr_sh_row_get(const float4* __restrict__ rows, int lane, float* __restrict__ row48)
{
#pragma unroll
	for (int j = 0; j < 12; j++) {
		const float4 v = rows[lane * GSR_SH_ROW4 + j];
		row48[4 * j] = v.x; row48[4 * j + 1] = v.y; row48[4 * j + 2] = v.z; row48[4 * j + 3] = v.w;
	}
}
// Split leaf tensors _features_dc (P,1,3) + _features_rest (P,15,3) (the torch.cat of
// gaussian_model.py:124-127 done on the fly).  The wave's two blocks are copied LINEARLY into LDS
// (coalesced 16-byte global accesses, conflict-free ds_*_b128): floats [0, 64*45) = the _features_rest
// block, [64*45, 64*48) = the _features_dc block.  A lane then reads / writes its own row with scalar LDS
// accesses at strides 45 and 3 -- both odd, so the 32 banks are hit once each.  (Scattering the elements
// into 48-float rows instead costs 4-way bank conflicts on every access: measured +20 us in preprocess.)
#define GSR_SH_LIN_DC (64 * 45)
__device__ __forceinline__ void gsr_sh_lin_load(float* __restrict__ lin, const float* __restrict__ dc, const float* __restrict__ rest,
                                                int wave_first, int nrows, int lane)
{
	const float* dcw = dc + (size_t)wave_first * 3;
#pragma unroll
	for (int i = 0; i < 3; i++) {
		const int e = lane + 64 * i;
		if (e < nrows * 3) lin[GSR_SH_LIN_DC + e] = dcw[e];
	}
	const float* rw = rest + (size_t)wave_first * 45;
	const int n = nrows * 45;
#pragma unroll
	for (int it = 0; it < 12; it++) {
		const int e0 = (it * 64 + lane) * 4;
		if (e0 + 3 < n) {
			*reinterpret_cast<float4*>(lin + e0) = *reinterpret_cast<const float4*>(rw + e0);
		} else {
#pragma unroll
			for (int j = 0; j < 4; j++) if (e0 + j < n) lin[e0 + j] = rw[e0 + j];
		}
	}
}
// The same in two halves of 32 Gaussians, for a kernel that only READS the rows (the forward's colour kernel): every load of the
// wave's two blocks is issued up front into registers (fetch), then each half is committed to a 6-KB LDS area -- floats
// [0, 32 * 45) = the half's _features_rest rows, [32 * 45, 32 * 48) = its _features_dc rows; both boundaries are multiples of 16
// bytes -- and read back by its 32 lanes.  Half the LDS per wave: six workgroups per CU instead of three.
#define GSR_SH_LINH_DC (32 * 45)
struct GsrShLinFetch { float4 r[12]; float dc[3]; };
__device__ __forceinline__ void gsr_sh_lin_fetch(GsrShLinFetch& v, const float* __restrict__ dc, const float* __restrict__ rest, int wave_first, int nrows, int lane)
{
	const float* dcw = dc + (size_t)wave_first * 3;
#pragma unroll
	for (int i = 0; i < 3; i++) {
		const int e = lane + 64 * i;
		v.dc[i] = (e < nrows * 3) ? dcw[e] : 0.f;
	}
	const float* rw = rest + (size_t)wave_first * 45;
	const int n = nrows * 45;
#pragma unroll
	for (int it = 0; it < 12; it++) {
		const int e0 = (it * 64 + lane) * 4;
		if (e0 + 3 < n) {
			v.r[it] = *reinterpret_cast<const float4*>(rw + e0);
		} else {
			v.r[it] = make_float4(e0 < n ? rw[e0] : 0.f, e0 + 1 < n ? rw[e0 + 1] : 0.f, e0 + 2 < n ? rw[e0 + 2] : 0.f, 0.f);
		}
	}
}
__device__ __forceinline__ void gsr_sh_lin_commit_half(float* __restrict__ lin, const GsrShLinFetch& v, int lane, int half)
{
#pragma unroll
	for (int i = 0; i < 3; i++) {
		const int e = lane + 64 * i;   // element of the wave's 192 dc floats: rows 32 * half .. hold [96 half, 96 half + 96)
		if (e / 96 == half) lin[GSR_SH_LINH_DC + e - 96 * half] = v.dc[i];
	}
#pragma unroll
	for (int it = 0; it < 12; it++) {
		const int f = it * 64 + lane;    // float4 of the wave's 720 rest float4: a half holds 360
		if (f / 360 == half) *reinterpret_cast<float4*>(lin + 4 * (f - 360 * half)) = v.r[it];
	}
}
// lane l (0 .. 31 inside its half) reads its row out of the half's area: the strides of gsr_sh_lin_row_get
__device__ __forceinline__ void gsr_sh_linh_row_get(const float* __restrict__ lin, int l, float* __restrict__ row48)
{
#pragma unroll
	for (int c = 0; c < 3; c++) row48[c] = lin[GSR_SH_LINH_DC + 3 * l + c];
#pragma unroll
	for (int j = 0; j < 45; j++) row48[3 + j] = lin[45 * l + j];
}
__device__ __forceinline__ void gsr_sh_lin_store(const float* __restrict__ lin, float* __restrict__ dc, float* __restrict__ rest,
                                                 int wave_first, int nrows, int lane)
{
	float* dcw = dc + (size_t)wave_first * 3;
#pragma unroll
	for (int i = 0; i < 3; i++) {
		const int e = lane + 64 * i;
		if (e < nrows * 3) dcw[e] = lin[GSR_SH_LIN_DC + e];
	}
	float* rw = rest + (size_t)wave_first * 45;
	const int n = nrows * 45;
#pragma unroll
	for (int it = 0; it < 12; it++) {
		const int e0 = (it * 64 + lane) * 4;
		if (e0 + 3 < n) {
			*reinterpret_cast<float4*>(rw + e0) = *reinterpret_cast<const float4*>(lin + e0);
		} else {
#pragma unroll
			for (int j = 0; j < 4; j++) if (e0 + j < n) rw[e0 + j] = lin[e0 + j];
		}
	}
}
// the lane's own 16 x 3 coefficients (k-major: [3k + c]) out of / into the linear blocks
__device__ __forceinline__ void gsr_sh_lin_row_get(const float* __restrict__ lin, int lane, float* __restrict__ row48)
{
#pragma unroll
	for (int c = 0; c < 3; c++) row48[c] = lin[GSR_SH_LIN_DC + 3 * lane + c];
#pragma unroll
	for (int j = 0; j < 45; j++) row48[3 + j] = lin[45 * lane + j];
}
__device__ __forceinline__ void gsr_sh_lin_row_put(float* __restrict__ lin, int lane, const float* __restrict__ row48)
{
#pragma unroll
	for (int c = 0; c < 3; c++) lin[GSR_SH_LIN_DC + 3 * lane + c] = row48[c];
#pragma unroll
	for (int j = 0; j < 45; j++) lin[45 * lane + j] = row48[3 + j];
}

// d(colour channel c) / d(unit view direction), c = 0..2: the nine sums of backward.cu:98-132, evaluated by the FORWARD
// kernel while the Gaussian's SH row is in registers and kept (GsrGeometry::sh_ddir) for the backward, which then never
// reads the 192-byte SH row again.  Written as sum_k (d basis_k / d{x,y,z}) * sh[k][c] with the 33 non-zero basis
// derivatives formed once and explicit FMAs (the reference multiplies each term out per channel: 3x the
// instructions; the two differ by fp32 rounding only, ~1e-7 relative, far inside the gradient bars).
__device__ __forceinline__ void gsr_sh_ddir9(int deg, const float* sh, float x, float y, float z, float* d9)
{
#pragma unroll
	for (int i = 0; i < 9; i++) d9[i] = 0.f;
	if (deg < 1) return;
#define ACC(comp, coef, k)                                                                 \
	d9[comp] = __builtin_fmaf((coef), sh[(k) * 3], d9[comp]);                              \
	d9[3 + comp] = __builtin_fmaf((coef), sh[(k) * 3 + 1], d9[3 + comp]);                  \
	d9[6 + comp] = __builtin_fmaf((coef), sh[(k) * 3 + 2], d9[6 + comp]);
	ACC(0, -GSR_SH_C1, 3) ACC(1, -GSR_SH_C1, 1) ACC(2, GSR_SH_C1, 2)
	if (deg > 1) {
		ACC(0, GSR_SH_C2[0] * y, 4) ACC(0, GSR_SH_C2[2] * -2.f * x, 6) ACC(0, GSR_SH_C2[3] * z, 7) ACC(0, GSR_SH_C2[4] * 2.f * x, 8)
		ACC(1, GSR_SH_C2[0] * x, 4) ACC(1, GSR_SH_C2[1] * z, 5) ACC(1, GSR_SH_C2[2] * -2.f * y, 6) ACC(1, GSR_SH_C2[4] * -2.f * y, 8)
		ACC(2, GSR_SH_C2[1] * y, 5) ACC(2, GSR_SH_C2[2] * 4.f * z, 6) ACC(2, GSR_SH_C2[3] * x, 7)
		if (deg > 2) {
			const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
			ACC(0, GSR_SH_C3[0] * 6.f * xy, 9) ACC(0, GSR_SH_C3[1] * yz, 10) ACC(0, GSR_SH_C3[2] * -2.f * xy, 11)
			ACC(0, GSR_SH_C3[3] * -6.f * xz, 12) ACC(0, GSR_SH_C3[4] * (-3.f * xx + 4.f * zz - yy), 13) ACC(0, GSR_SH_C3[5] * 2.f * xz, 14)
			ACC(0, GSR_SH_C3[6] * 3.f * (xx - yy), 15)
			ACC(1, GSR_SH_C3[0] * 3.f * (xx - yy), 9) ACC(1, GSR_SH_C3[1] * xz, 10) ACC(1, GSR_SH_C3[2] * (-3.f * yy + 4.f * zz - xx), 11)
			ACC(1, GSR_SH_C3[3] * -6.f * yz, 12) ACC(1, GSR_SH_C3[4] * -2.f * xy, 13) ACC(1, GSR_SH_C3[5] * -2.f * yz, 14)
			ACC(1, GSR_SH_C3[6] * -6.f * xy, 15)
			ACC(2, GSR_SH_C3[1] * xy, 10) ACC(2, GSR_SH_C3[2] * 8.f * yz, 11) ACC(2, GSR_SH_C3[3] * 3.f * (2.f * zz - xx - yy), 12)
			ACC(2, GSR_SH_C3[4] * 8.f * xz, 13) ACC(2, GSR_SH_C3[5] * (xx - yy), 14)
		}
	}
#undef ACC
}

// forward.cu:84-140 computeCov2D, also recomputed by backward.cu:144-199
struct GsrCov2D {
	GsrMat3 T, W, Vrk;
	GsrVec3 t;
	float txtz, tytz, limx, limy;
	float a, b, c;
};

__device__ __forceinline__ void gsr_cov2d(const GsrVec3& mean, float focal_x, float focal_y, float tan_fovx,
                                          float tan_fovy, const float* cov3D, const float* vm, GsrCov2D& o)
{
	GsrVec3 t = gsr_transform_point_4x3(mean, vm);
	o.limx = 1.3f * tan_fovx;
	o.limy = 1.3f * tan_fovy;
	o.txtz = t.x / t.z;
	o.tytz = t.y / t.z;
	t.x = fminf(o.limx, fmaxf(-o.limx, o.txtz)) * t.z;
	t.y = fminf(o.limy, fmaxf(-o.limy, o.tytz)) * t.z;
	o.t = t;
	GsrMat3 J;
	J.m[0][0] = focal_x / t.z; J.m[0][1] = 0.0f; J.m[0][2] = -(focal_x * t.x) / (t.z * t.z);
	J.m[1][0] = 0.0f; J.m[1][1] = focal_y / t.z; J.m[1][2] = -(focal_y * t.y) / (t.z * t.z);
	J.m[2][0] = 0.f; J.m[2][1] = 0.f; J.m[2][2] = 0.f;
	o.W.m[0][0] = vm[0]; o.W.m[0][1] = vm[4]; o.W.m[0][2] = vm[8];
	o.W.m[1][0] = vm[1]; o.W.m[1][1] = vm[5]; o.W.m[1][2] = vm[9];
	o.W.m[2][0] = vm[2]; o.W.m[2][1] = vm[6]; o.W.m[2][2] = vm[10];
	o.T = gsr_mat3_mul(o.W, J);
	o.Vrk.m[0][0] = cov3D[0]; o.Vrk.m[0][1] = cov3D[1]; o.Vrk.m[0][2] = cov3D[2];
	o.Vrk.m[1][0] = cov3D[1]; o.Vrk.m[1][1] = cov3D[3]; o.Vrk.m[1][2] = cov3D[4];
	o.Vrk.m[2][0] = cov3D[2]; o.Vrk.m[2][1] = cov3D[4]; o.Vrk.m[2][2] = cov3D[5];
	GsrMat3 cov = gsr_mat3_mul(gsr_mat3_mul(gsr_mat3_transpose(o.T), gsr_mat3_transpose(o.Vrk)), o.T);
	o.a = cov.m[0][0] + 0.3f;
	o.b = cov.m[0][1];
	o.c = cov.m[1][1] + 0.3f;
}

// ---- wave64 cross-lane helpers (DPP) -------------------------------------------------------
template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND>
__device__ __forceinline__ float gsr_dpp_mov(float v)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, BOUND));
}

// Sum over the 64 lanes of a wave; the total is valid in lane 63 (other lanes hold partials).
__device__ __forceinline__ float gsr_wave_sum_to_lane63(float v)
{
	v += gsr_dpp_mov<0xB1, 0xF, 0xF, true>(v);   // quad_perm [1,0,3,2]
	v += gsr_dpp_mov<0x4E, 0xF, 0xF, true>(v);   // quad_perm [2,3,0,1]
	v += gsr_dpp_mov<0x141, 0xF, 0xF, true>(v);  // row_half_mirror
	v += gsr_dpp_mov<0x140, 0xF, 0xF, true>(v);  // row_mirror  -> every lane holds its row's sum
	v += gsr_dpp_mov<0x142, 0xA, 0xF, false>(v); // row_bcast:15 into rows 1,3
	v += gsr_dpp_mov<0x143, 0xC, 0xF, false>(v); // row_bcast:31 into rows 2,3 -> lane 63 = total
	return v;
}
