// preprocess.hip -- per-Gaussian forward stage (compiled with -ffp-contract=off).
//
// Replaces preprocessCUDA (cuda_rasterizer/forward.cu:192-324) and checkFrustum
// (cuda_rasterizer/rasterizer_impl.cu:56-69).  One Gaussian per lane, 256-thread workgroups.
// Outputs go into one 48-byte splat record per Gaussian instead of five separate arrays, and the
// kernel also accumulates the instance count (num_rendered) -- one atomic add per workgroup into 64
// partial counters -- so no scan over P follows.
#include "gsr_internal.h"
#include <hip/hip_ext.h>
#include "gsr_rect_trim.h"

// forward.cu:21-81 computeColorFromSH, one channel at a time in the glm::vec3 expression order
__device__ __forceinline__ float gsr_sh_channel(int deg, const float* sh, int ch, float x, float y, float z)
{
#define SH(k) sh[(k) * 3 + ch]
	float result = GSR_SH_C0 * SH(0);
	if (deg > 0) {
		result = result - GSR_SH_C1 * y * SH(1) + GSR_SH_C1 * z * SH(2) - GSR_SH_C1 * x * SH(3);
		if (deg > 1) {
			float xx = x * x, yy = y * y, zz = z * z;
			float xy = x * y, yz = y * z, xz = x * z;
			result = result + GSR_SH_C2[0] * xy * SH(4) + GSR_SH_C2[1] * yz * SH(5) +
			         GSR_SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) + GSR_SH_C2[3] * xz * SH(7) +
			         GSR_SH_C2[4] * (xx - yy) * SH(8);
			if (deg > 2) {
				result = result + GSR_SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + GSR_SH_C3[1] * xy * z * SH(10) +
				         GSR_SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
				         GSR_SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
				         GSR_SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + GSR_SH_C3[5] * z * (xx - yy) * SH(14) +
				         GSR_SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
			}
		}
	}
#undef SH
	return result + 0.5f;
}

// The stage is two kernels over the same Gaussians, independent of each other:
//   gsr_preprocess_kernel        geometry: projection, covariance, conic, radius, tile rectangle, depth key, instance count
//                                (44 B in, 56 B out per Gaussian; the depth sort and the instance count wait for it)
//   gsr_preprocess_color_kernel  view-dependent colour from the SH rows (204 B in, 49 B out): needed by nothing before the
//                                blend, so api.hip runs it on a helper stream beside the geometry kernel and the depth sort
// Both write disjoint 16-byte-aligned parts of the 48-byte splat record.  Precomputed colours need no second kernel.
//
// LEAF: the inputs are the optimiser's raw leaves (gsr_internal.h); activations happen here.
template <bool LEAF>
__global__ void __launch_bounds__(GSR_PREPROCESS_BLOCK) gsr_preprocess_kernel(GsrPreprocessArgs a, uint32_t* __restrict__ clear, size_t clear_words,
                                                                              uint32_t* __restrict__ clear2, size_t clear2_words)
{
	const int idx = blockIdx.x * GSR_PREPROCESS_BLOCK + threadIdx.x;
	// The Gaussian's inputs, unconditionally (a culled Gaussian wastes 44 bytes): every load of the wave is in flight at
	// once -- one memory round trip instead of three dependent ones (position -> cull test -> scale / rotation / opacity)
	GsrVec3 p_orig = {0.f, 0.f, 0.f};
	float sc[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f}, cov_in[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	float col_in[3] = {0.f, 0.f, 0.f}, opac = 0.f;
	if (idx < a.P) {
		p_orig.x = a.means3D[3 * idx]; p_orig.y = a.means3D[3 * idx + 1]; p_orig.z = a.means3D[3 * idx + 2];
		if (a.cov3D_precomp) {
#pragma unroll
			for (int k = 0; k < 6; k++) cov_in[k] = a.cov3D_precomp[6 * (size_t)idx + k];
		} else {
			sc[0] = a.scales[3 * idx]; sc[1] = a.scales[3 * idx + 1]; sc[2] = a.scales[3 * idx + 2];
			q[0] = a.rotations[4 * idx]; q[1] = a.rotations[4 * idx + 1]; q[2] = a.rotations[4 * idx + 2]; q[3] = a.rotations[4 * idx + 3];
		}
		opac = a.opacities[idx];
		if (a.colors_precomp) { col_in[0] = a.colors_precomp[3 * idx]; col_in[1] = a.colors_precomp[3 * idx + 1]; col_in[2] = a.colors_precomp[3 * idx + 2]; }
	}
	uint32_t tiles = 0;
	uint2 rect = make_uint2(0u, 0u);
	uint2 rshape = make_uint2(GSR_RECT_NONE, 0u);   // {rectangle in one word, trim word}: what the binning reads (gsr_rect_trim.h)
	int radius_out = 0;
	uint32_t depth_key = 0xFFFFFFFFu;  // culled Gaussians sort behind every visible one
	const int gx = (a.W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (a.H + GSR_TILE_Y - 1) / GSR_TILE_Y;

	if (idx < a.P) {
		do {
			// in_frustum, auxiliary.h:144-175
			GsrVec3 p_view = gsr_transform_point_4x3(p_orig, a.viewmatrix);
			if (p_view.z <= 0.2f) {
				if (a.prefiltered) atomicOr(&a.g.status[0], 1u);
				break;
			}
			const float* pm = a.projmatrix;
			float hx = pm[0] * p_orig.x + pm[4] * p_orig.y + pm[8] * p_orig.z + pm[12];
			float hy = pm[1] * p_orig.x + pm[5] * p_orig.y + pm[9] * p_orig.z + pm[13];
			float hw = pm[3] * p_orig.x + pm[7] * p_orig.y + pm[11] * p_orig.z + pm[15];
			float p_w = 1.0f / (hw + 0.0000001f);
			float p_proj_x = hx * p_w, p_proj_y = hy * p_w;

			float cov3D[6];
			if (a.cov3D_precomp) {
#pragma unroll
				for (int k = 0; k < 6; k++) cov3D[k] = cov_in[k];
			} else {
				if (LEAF) {
					sc[0] = gsr_act_exp(sc[0]); sc[1] = gsr_act_exp(sc[1]); sc[2] = gsr_act_exp(sc[2]);
					gsr_act_normalize4(q, q);
				}
				gsr_cov3d(sc, a.scale_modifier, q, cov3D);
			}
			GsrCov2D c2;
			gsr_cov2d(p_orig, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, cov3D, a.viewmatrix, c2);
			const float cx = c2.a, cy = c2.b, cz = c2.c;
			float det = (cx * cz - cy * cy);
			if (det == 0.0f) break;
			float det_inv = 1.f / det;
			float conic_a = cz * det_inv, conic_b = -cy * det_inv, conic_c = cx * det_inv;
			float mid = 0.5f * (cx + cz);
			float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
			float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
			float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
			float pix = gsr_ndc2pix(p_proj_x, a.W), piy = gsr_ndc2pix(p_proj_y, a.H);
			int minx, miny, maxx, maxy;
			gsr_get_rect(pix, piy, gsr_f2i(my_radius), gx, gy, minx, miny, maxx, maxy);
			if ((maxx - minx) * (maxy - miny) == 0) break;

			tiles = (uint32_t)((maxy - miny) * (maxx - minx));
			radius_out = gsr_f2i(my_radius);
			depth_key = __float_as_uint(p_view.z);  // > 0.2, so unsigned order == float order
			rect = make_uint2((uint32_t)minx | ((uint32_t)miny << 16), (uint32_t)(maxx - minx) | ((uint32_t)(maxy - miny) << 16));
			// the geometry part of the record (GsrSplat: x, y, conic a, b | conic c, opacity, rect | colour): two 16-byte stores, 32 contiguous bytes
			float4* rec = reinterpret_cast<float4*>(a.g.splat + idx);
			rec[0] = make_float4(pix, piy, conic_a, conic_b);
			const float opacity = LEAF ? gsr_act_sigmoid(opac) : opac;
			rec[1] = make_float4(conic_c, opacity, __uint_as_float(rect.x), __uint_as_float(rect.y));
			rshape.x = gsr_rect_pack((uint32_t)minx, (uint32_t)miny, (uint32_t)(maxx - minx), (uint32_t)(maxy - miny));
			if (a.trim) rshape.y = gsr_rect_trim(pix, piy, conic_a, conic_b, conic_c, opacity, minx, miny, maxx - minx, maxy - miny);
			if (a.colors_precomp) {
				rec[2] = make_float4(col_in[0], col_in[1], col_in[2], 0.f);
				a.g.clamped[idx] = 0;
			}
		} while (0);
		if (a.radii) a.radii[idx] = radius_out;  // optional, cuda_rasterizer/rasterizer.h:52
		a.g.tiles_touched[idx] = tiles;
		a.g.rect[idx] = rect;             // dense copy: the depth-ordered kernels gather 8 bytes, not a 48-byte record
		a.g.rshape[idx] = rshape;
		a.g.depth_keys[idx] = depth_key;  // inputs of the depth sort (sort.hip)
		a.g.perm[idx] = (uint32_t)idx;
	}

	// workgroup sum of tiles_touched -> one atomic add into one of the partial instance counts (status words,
	// zeroed by the host side before the launch; the host adds the parts): the count is all the forward needs from this order of the Gaussians -- the
	// reference's inclusive scan over P (rasterizer_impl.cu:323) would give offsets in ORIGINAL order, and the
	// offsets that are used here are those of the depth order (gsr_sorted_block_sums_kernel)
	// ... and, the same way, the range of the depth keys of the visible Gaussians: 64-way partial maxima of ~key and key
	// (sort.hip orders key - min, so the depth sort only needs passes for the bits of max - min)
	// ... and each Gaussian's first gradient slot inside its workgroup: the exclusive scan of tiles_touched over the workgroup's 256
	// Gaussians in INDEX order (slot_base, made global by the depth sort's first kernel from the workgroups' totals in block_tiles:
	// sort.hip).  Index order, not depth order: the per-Gaussian backward walks the Gaussians by index, so a wave's slots and their
	// validity bytes then form ONE contiguous range instead of 64 runs on lines of their own (round 4; what the reference's
	// InclusiveSum over tiles_touched gives, rasterizer_impl.cu:323)
	__shared__ uint32_t wsum[GSR_PREPROCESS_BLOCK / 64], wneg[GSR_PREPROCESS_BLOCK / 64], wmax[GSR_PREPROCESS_BLOCK / 64];
	uint32_t incl = tiles;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t t = __shfl_up(incl, off, 64);
		if ((threadIdx.x & 63) >= (unsigned)off) incl += t;
	}
	uint32_t kneg = tiles ? ~depth_key : 0u, kmax = tiles ? depth_key : 0u;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		kneg = max(kneg, (uint32_t)__shfl_xor(kneg, off, 64));
		kmax = max(kmax, (uint32_t)__shfl_xor(kmax, off, 64));
	}
	if ((threadIdx.x & 63) == 63) { wsum[threadIdx.x >> 6] = incl; wneg[threadIdx.x >> 6] = kneg; wmax[threadIdx.x >> 6] = kmax; }
	__syncthreads();
	{
		uint32_t before = 0, t = 0;
#pragma unroll
		for (int w = 0; w < GSR_PREPROCESS_BLOCK / 64; w++) {
			if (w < (int)(threadIdx.x >> 6)) before += wsum[w];
			t += wsum[w];
		}
		if (idx < a.P) a.g.slot_base[idx] = before + incl - tiles;
		if (threadIdx.x == 0) {
			a.g.block_tiles[blockIdx.x] = t;
			uint32_t n2 = 0, m2 = 0;
#pragma unroll
			for (int w = 0; w < GSR_PREPROCESS_BLOCK / 64; w++) { n2 = max(n2, wneg[w]); m2 = max(m2, wmax[w]); }
			if (t) {
				const int part = blockIdx.x & (GSR_COUNT_PARTS - 1);
				atomicAdd(&a.g.status[4 + part], t);
				atomicMax(&a.g.status[GSR_STATUS_NEGMIN + part], n2);
				atomicMax(&a.g.status[GSR_STATUS_MAX + part], m2);
			}
		}
	}
	// zero the chunk sums of the depth sort's passes (sort.hip): one word per thread of the first workgroups
	for (size_t w = (size_t)blockIdx.x * GSR_PREPROCESS_BLOCK + threadIdx.x; w < clear_words; w += (size_t)gridDim.x * GSR_PREPROCESS_BLOCK)
		clear[w] = 0u;
	// ... and those of the column-pair binning's first pass (tilebin.hip), whose histogram runs behind the depth sort
	for (size_t w = (size_t)blockIdx.x * GSR_PREPROCESS_BLOCK + threadIdx.x; w < clear2_words; w += (size_t)gridDim.x * GSR_PREPROCESS_BLOCK)
		clear2[w] = 0u;
}

// View-dependent colour (forward.cu:21-81, called at :306-312): colour = clamp0(SH(dir) + 0.5) into the record, the
// clamp flags, and the nine derivatives d colour / d direction the backward needs (so that it never reads the 192-byte row
// again).  Done for every Gaussian in front of the near plane -- a superset of those the geometry kernel keeps (it
// also drops det == 0 and empty rectangles); what is written for the difference is never read.
template <bool LEAF>
__global__ void __launch_bounds__(GSR_PREPROCESS_BLOCK) gsr_preprocess_color_kernel(GsrPreprocessArgs a, int sh_via_lds)
{
	// staging of the wave's SH block in two halves of 32 rows (6.6 KB per wave, so that 4 waves per SIMD fit): the packed layout
	// transposed, the split leaf tensors linearly (round 4: they used to pass as one 12-KB block -- three workgroups per CU, 0.097 ms
	// against the packed kernel's 0.060, and the training iteration's binning waited for it)
	__shared__ float4 s_sh[GSR_PREPROCESS_BLOCK / 64][32 * GSR_SH_ROW4];
	const int idx = blockIdx.x * GSR_PREPROCESS_BLOCK + threadIdx.x;
	GsrVec3 p_orig = {0.f, 0.f, 0.f};
	if (idx < a.P) { p_orig.x = a.means3D[3 * idx]; p_orig.y = a.means3D[3 * idx + 1]; p_orig.z = a.means3D[3 * idx + 2]; }
	// The wave's 64 x 48 SH floats are contiguous in HBM: stage them into LDS with coalesced float4
	// loads (a lane reading its own 192-byte row makes every load instruction touch 64 lines)
	float row[48];  // the lane's own SH row (registers: only ever indexed with constants)
	if (sh_via_lds) {
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		const int wave_first = blockIdx.x * GSR_PREPROCESS_BLOCK + wave * 64;
		const int nrows = min(64, a.P - wave_first);
		if (LEAF) {
			GsrShLinFetch v;
			gsr_sh_lin_fetch(v, a.shs, a.shs_rest, wave_first, max(nrows, 0), lane);  // every load in flight at once
			float* lin = reinterpret_cast<float*>(s_sh[wave]);
#pragma unroll
			for (int half = 0; half < 2; half++) {
				gsr_sh_lin_commit_half(lin, v, lane, half);
				__builtin_amdgcn_wave_barrier();
				if ((lane >> 5) == half) gsr_sh_linh_row_get(lin, lane & 31, row);
				__builtin_amdgcn_wave_barrier();
			}
		} else {
			float4 v[12];
			gsr_sh_rows_fetch(v, a.shs, wave_first, max(nrows, 0), lane);  // all twelve loads in flight at once
#pragma unroll
			for (int half = 0; half < 2; half++) {
				gsr_sh_rows_commit_half(s_sh[wave], v, nrows, lane, half);
				__builtin_amdgcn_wave_barrier();
				if ((lane >> 5) == half) gsr_sh_row_get(s_sh[wave], lane & 31, row);
				__builtin_amdgcn_wave_barrier();
			}
		}
	}
	if (idx >= a.P) return;
	if (gsr_transform_point_4x3(p_orig, a.viewmatrix).z <= 0.2f) return;  // the geometry kernel's near-plane test, same expression

	float dx = p_orig.x - a.cam_pos[0], dy = p_orig.y - a.cam_pos[1], dz = p_orig.z - a.cam_pos[2];
	float len = sqrtf(dx * dx + dy * dy + dz * dz);
	dx = dx / len; dy = dy / len; dz = dz / len;
	const float* sh = a.shs + (size_t)idx * a.M * 3;
	float raw[3], ddir9[9];
	if (sh_via_lds) {
#pragma unroll
		for (int ch = 0; ch < 3; ch++) raw[ch] = gsr_sh_channel(a.D, row, ch, dx, dy, dz);
		gsr_sh_ddir9(a.D, row, dx, dy, dz, ddir9);
	} else {
		float sh_local[48];
		if (LEAF) {  // generic M / unaligned leaves: gather the used rows (rare path)
			const int used = (a.D + 1) * (a.D + 1);
#pragma unroll
			for (int k = 0; k < 16; k++)  // constant indices (registers, no scratch); (D+1)^2 <= 16
				if (k < used) {
#pragma unroll
					for (int ch = 0; ch < 3; ch++)
						sh_local[k * 3 + ch] = k == 0 ? a.shs[3 * (size_t)idx + ch] : a.shs_rest[((size_t)idx * (a.M - 1) + (k - 1)) * 3 + ch];
				}
			sh = sh_local;
		}
#pragma unroll
		for (int ch = 0; ch < 3; ch++) raw[ch] = gsr_sh_channel(a.D, sh, ch, dx, dy, dz);
		gsr_sh_ddir9(a.D, sh, dx, dy, dz, ddir9);
	}
#pragma unroll
	for (int k = 0; k < 9; k++) a.g.sh_ddir[(size_t)k * a.P + idx] = ddir9[k];  // nine planes: each store instruction is one contiguous run
	uint8_t clamp_bits = 0;
	float rgb[3];
#pragma unroll
	for (int ch = 0; ch < 3; ch++) {
		const float v = raw[ch];
		if (v < 0) clamp_bits |= (uint8_t)(1u << ch);
		rgb[ch] = fmaxf(v, 0.0f);
	}
	a.g.clamped[idx] = clamp_bits;
	reinterpret_cast<float4*>(a.g.splat + idx)[2] = make_float4(rgb[0], rgb[1], rgb[2], 0.f);   // the record's last 16 bytes: one aligned store per lane
}

// The status words (instance count and depth range as 64-way partial sums / maxima, the prefiltered flag) start at zero.  A kernel
// of ours instead of hipMemsetAsync (which is a kernel launch too) because its dispatch packet can signal an event
// (hipExtLaunchKernelGGL): `done` = the fork event of the helper stream -- a hipEventRecord in front of the first kernel is a
// barrier packet of its own, ~6 us of every forward call.
__global__ void gsr_zero_status_kernel(uint32_t* __restrict__ status)
{
	if (threadIdx.x < GSR_STATUS_WORDS) status[threadIdx.x] = 0u;
}
void gsr_launch_zero_status(uint32_t* status, hipStream_t s, hipEvent_t done)
{
	static_assert(GSR_STATUS_WORDS <= 256, "one workgroup zeroes the status words");
	if (done) hipExtLaunchKernelGGL(gsr_zero_status_kernel, dim3(1), dim3(256), 0, s, nullptr, done, 0, status);
	else hipLaunchKernelGGL(gsr_zero_status_kernel, dim3(1), dim3(256), 0, s, status);
}

// done: optional event signalled by the kernel's own dispatch packet when it has finished (hipExtLaunchKernelGGL): a separate
// hipEventRecord behind the kernel is a barrier packet of its own and costs the stream's next launch ~8 us
void gsr_launch_preprocess(const GsrPreprocessArgs& a, hipStream_t s, hipEvent_t done)
{
	const int nb = (a.P + GSR_PREPROCESS_BLOCK - 1) / GSR_PREPROCESS_BLOCK;
	uint32_t* clear = (uint32_t*)a.g.sort_table;
	const size_t clear_words = gsr_radix_clear_words((size_t)a.P);
	uint32_t* clear2 = (uint32_t*)a.g.col_table;
	const size_t clear2_words = gsr_tilebin_col_clear_words((size_t)a.P);
	if (done) {
		if (a.leaf) hipExtLaunchKernelGGL(gsr_preprocess_kernel<true>, dim3(nb), dim3(GSR_PREPROCESS_BLOCK), 0, s, nullptr, done, 0, a, clear, clear_words, clear2, clear2_words);
		else hipExtLaunchKernelGGL(gsr_preprocess_kernel<false>, dim3(nb), dim3(GSR_PREPROCESS_BLOCK), 0, s, nullptr, done, 0, a, clear, clear_words, clear2, clear2_words);
		return;
	}
	if (a.leaf) hipLaunchKernelGGL(gsr_preprocess_kernel<true>, dim3(nb), dim3(GSR_PREPROCESS_BLOCK), 0, s, a, clear, clear_words, clear2, clear2_words);
	else hipLaunchKernelGGL(gsr_preprocess_kernel<false>, dim3(nb), dim3(GSR_PREPROCESS_BLOCK), 0, s, a, clear, clear_words, clear2, clear2_words);
}

// the colour kernel exists only for SH colours
bool gsr_preprocess_needs_color(const GsrPreprocessArgs& a) { return a.shs && !a.colors_precomp; }

// wgs_per_cu: 0 = as many workgroups per CU as fit; else the kernel is held to that many by (unused) dynamic LDS on top of its own
// staging area (26 KB) -- while it runs beside the depth sort on the helper stream (api.hip)
void gsr_launch_preprocess_color(const GsrPreprocessArgs& a, hipStream_t s, int wgs_per_cu)
{
	const int nb = (a.P + GSR_PREPROCESS_BLOCK - 1) / GSR_PREPROCESS_BLOCK;
	// LDS-transposed SH path: the flagship layout (16 coefficients), 16-byte aligned tensor
	int sh_via_lds = (a.M == 16 && ((uintptr_t)a.shs & 15u) == 0) ? 1 : 0;
	const size_t own = (size_t)(GSR_PREPROCESS_BLOCK / 64) * 32 * GSR_SH_ROW4 * sizeof(float4);
	const size_t share = wgs_per_cu > 0 ? (size_t)160 * 1024 / (size_t)wgs_per_cu : 0;
	const size_t throttle = share > own + 1024 ? share - own - 1024 : 0;   // (1 KB of slack for allocation granularity)
	if (a.leaf) {
		if (((uintptr_t)a.shs_rest & 15u) != 0) sh_via_lds = 0;
		hipLaunchKernelGGL(gsr_preprocess_color_kernel<true>, dim3(nb), dim3(GSR_PREPROCESS_BLOCK), throttle, s, a, sh_via_lds);
	} else {
		hipLaunchKernelGGL(gsr_preprocess_color_kernel<false>, dim3(nb), dim3(GSR_PREPROCESS_BLOCK), throttle, s, a, sh_via_lds);
	}
}

// rasterizer_impl.cu:56-69 checkFrustum: only the view-space z test of in_frustum survives
__global__ void gsr_mark_visible_kernel(int P, const float* means3D, const float* viewmatrix, uint8_t* present)
{
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= P) return;
	GsrVec3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
	GsrVec3 pv = gsr_transform_point_4x3(p, viewmatrix);
	present[idx] = !(pv.z <= 0.2f);
}

void gsr_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s)
{
	hipLaunchKernelGGL(gsr_mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
}
