// gaussian_backward.hip -- per-Gaussian backward (compiled with -ffp-contract=off).
//
// Fuses, in one pass over P: the fixed-order sum of a Gaussian's per-tile gradient slots (what
// the reference accumulates with atomics in backward.cu:561-598), computeCov2DCUDA
// (backward.cu:144-277), preprocessCUDA backward (backward.cu:349-399) with its SH
// (backward.cu:20-139) and covariance (backward.cu:281-345) parts, and the zero fill of
// rasterize_points.cu:168-178: every output element is written exactly once.
#include "gsr_internal.h"

__device__ __forceinline__ GsrVec3 gsr_dnormvdv(GsrVec3 v, GsrVec3 dv)  // auxiliary.h:109-120
{
	float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
	float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
	GsrVec3 r;
	r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
	r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
	r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
	return r;
}

// dRGB/dsh_k for k < (deg+1)^2: the SH basis at direction (x,y,z), expressions of backward.cu:45-96
__device__ __forceinline__ void gsr_sh_basis(int deg, float x, float y, float z, float* b)
{
	b[0] = GSR_SH_C0;
	if (deg > 0) {
		b[1] = -GSR_SH_C1 * y;
		b[2] = GSR_SH_C1 * z;
		b[3] = -GSR_SH_C1 * x;
		if (deg > 1) {
			float xx = x * x, yy = y * y, zz = z * z;
			float xy = x * y, yz = y * z, xz = x * z;
			b[4] = GSR_SH_C2[0] * xy;
			b[5] = GSR_SH_C2[1] * yz;
			b[6] = GSR_SH_C2[2] * (2.f * zz - xx - yy);
			b[7] = GSR_SH_C2[3] * xz;
			b[8] = GSR_SH_C2[4] * (xx - yy);
			if (deg > 2) {
				b[9] = GSR_SH_C3[0] * y * (3.f * xx - yy);
				b[10] = GSR_SH_C3[1] * xy * z;
				b[11] = GSR_SH_C3[2] * y * (4.f * zz - xx - yy);
				b[12] = GSR_SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
				b[13] = GSR_SH_C3[4] * x * (4.f * zz - xx - yy);
				b[14] = GSR_SH_C3[5] * z * (xx - yy);
				b[15] = GSR_SH_C3[6] * x * (xx - 3.f * yy);
			}
		}
	}
}

// backward.cu:20-139.  dL_dRGB = dL_dcolor with clamped channels zeroed (returned in dL_dRGB_out).
// ddir9: d(colour channel)/d(unit direction) as the forward kernel left it (gsr_sh_dcolor_ddir): the SH row itself is
// not needed here.  write_dsh: writes dL_dsh rows [0, (D+1)^2) and zeros the rest up to M; otherwise only the
// view-direction term of dL_dmean is produced (view-parallel mode, see gsr_sh_grad_from_views).
__device__ __forceinline__ void gsr_sh_backward(int deg, int M, GsrVec3 pos, const float* campos, const float* ddir9,
                                                uint8_t clamp_bits, const float* dL_dcolor, float* dL_dmean,
                                                float* dL_dsh, bool write_dsh, float* dL_dRGB_out, float* basis_out = nullptr)
{
	GsrVec3 dir_orig = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
	float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
	float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
	float dd0 = 0.f, dd1 = 0.f, dd2 = 0.f;
	const int used = (deg + 1) * (deg + 1);
	float basis[16];
	if (write_dsh || basis_out) gsr_sh_basis(deg, x, y, z, basis);
	if (basis_out) {
#pragma unroll
		for (int k = 0; k < 16; k++) basis_out[k] = (k < used) ? basis[k] : 0.f;  // entries >= used are never set by gsr_sh_basis
	}
#pragma unroll
	for (int ch = 0; ch < 3; ch++) {
		const float g = dL_dcolor[ch] * (((clamp_bits >> ch) & 1) ? 0.f : 1.f);
		dL_dRGB_out[ch] = g;
		if (write_dsh) {
#pragma unroll
			for (int k = 0; k < 16; k++)
				if (k < used) dL_dsh[k * 3 + ch] = basis[k] * g;
			for (int k = used; k < M; k++) dL_dsh[k * 3 + ch] = 0.f;
		}
		dd0 += ddir9[3 * ch] * g; dd1 += ddir9[3 * ch + 1] * g; dd2 += ddir9[3 * ch + 2] * g;
	}
	GsrVec3 ddir = {dd0, dd1, dd2};
	GsrVec3 dm = gsr_dnormvdv(dir_orig, ddir);
	dL_dmean[0] += dm.x; dL_dmean[1] += dm.y; dL_dmean[2] += dm.z;
}

// backward.cu:281-345
__device__ __forceinline__ void gsr_cov3d_backward(const float* scale, float mod, const float* rot,
                                                   const float* dL_dcov3D, float* dL_dscale, float* dL_drot)
{
	const float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
	GsrMat3 R = gsr_build_R(r, x, y, z);
	const float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
	GsrMat3 S = gsr_diag3(s[0], s[1], s[2]);
	GsrMat3 M = gsr_mat3_mul(S, R);
	GsrMat3 dS;
	dS.m[0][0] = dL_dcov3D[0]; dS.m[0][1] = 0.5f * dL_dcov3D[1]; dS.m[0][2] = 0.5f * dL_dcov3D[2];
	dS.m[1][0] = 0.5f * dL_dcov3D[1]; dS.m[1][1] = dL_dcov3D[3]; dS.m[1][2] = 0.5f * dL_dcov3D[4];
	dS.m[2][0] = 0.5f * dL_dcov3D[2]; dS.m[2][1] = 0.5f * dL_dcov3D[4]; dS.m[2][2] = dL_dcov3D[5];
	GsrMat3 M2;
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int w = 0; w < 3; w++) M2.m[c][w] = 2.0f * M.m[c][w];
	GsrMat3 dL_dM = gsr_mat3_mul(M2, dS);
	GsrMat3 Rt = gsr_mat3_transpose(R);
	GsrMat3 dMt = gsr_mat3_transpose(dL_dM);
#pragma unroll
	for (int k = 0; k < 3; k++)
		dL_dscale[k] = Rt.m[k][0] * dMt.m[k][0] + Rt.m[k][1] * dMt.m[k][1] + Rt.m[k][2] * dMt.m[k][2];
#pragma unroll
	for (int k = 0; k < 3; k++)
#pragma unroll
		for (int w = 0; w < 3; w++) dMt.m[k][w] *= s[k];
#define Dm(a, b) dMt.m[a][b]
	dL_drot[0] = 2 * z * (Dm(0, 1) - Dm(1, 0)) + 2 * y * (Dm(2, 0) - Dm(0, 2)) + 2 * x * (Dm(1, 2) - Dm(2, 1));
	dL_drot[1] = 2 * y * (Dm(1, 0) + Dm(0, 1)) + 2 * z * (Dm(2, 0) + Dm(0, 2)) + 2 * r * (Dm(1, 2) - Dm(2, 1)) - 4 * x * (Dm(2, 2) + Dm(1, 1));
	dL_drot[2] = 2 * x * (Dm(1, 0) + Dm(0, 1)) + 2 * r * (Dm(2, 0) - Dm(0, 2)) + 2 * z * (Dm(1, 2) + Dm(2, 1)) - 4 * y * (Dm(2, 2) + Dm(0, 0));
	dL_drot[3] = 2 * r * (Dm(0, 1) - Dm(1, 0)) + 2 * x * (Dm(2, 0) + Dm(0, 2)) + 2 * y * (Dm(1, 2) + Dm(2, 1)) - 4 * z * (Dm(1, 1) + Dm(0, 0));
#undef Dm
}

// Fixed-order sum of one Gaussian's contiguous run of per-tile gradient slots.  Runs of up to
// GSR_SLOT_COOP slots are added by the owning lane; longer runs (a big splat can own > 1000) are
// added by the whole wave, lanes striding over the run, then reduced with DPP -- so the wave's
// time no longer follows its single most-loaded lane.  The order is fixed: bitwise reproducible.
#ifndef GSR_SLOT_COOP
#define GSR_SLOT_COOP 30   // swept on MI355X at C3: 12 / 18 / 24 / 36 -> 0.172 / 0.164 / 0.166 / 0.171 ms with the slots in depth order (round 2);
                           // in index order (round 4) a lane's records lie next to its neighbours': 18 / 30 / 58 -> 0.130 / 0.120 / 0.118 ms at C3,
                           // 0.745 / 0.626 / 0.628 at C5
#endif
#define GSR_NACC 9
#ifndef GSR_SLOT_ROUND
#define GSR_SLOT_ROUND 6   // slot records requested per round of the per-lane sum
#endif
#define GSR_VALID_WORDS ((GSR_SLOT_COOP + 3 + 3) / 4)  // aligned dwords that cover GSR_SLOT_COOP bytes at any byte offset
// one bit per slot of a lane's run: 32 bits up to 30 slots, 64 beyond (4 GSR_VALID_WORDS <= 64 bits: at most 58 slots)
#if GSR_SLOT_COOP <= 30
typedef uint32_t GsrSlotMask;
__device__ __forceinline__ uint32_t gsr_slot_mask_first(uint32_t m) { return (uint32_t)__builtin_ctz(m | 0x80000000u); }
#else
static_assert(GSR_SLOT_COOP <= 58, "the validity bits of a lane's run must fit 64 bits");
typedef unsigned long long GsrSlotMask;
__device__ __forceinline__ uint32_t gsr_slot_mask_first(unsigned long long m) { return (uint32_t)__builtin_ctzll(m | 0x8000000000000000ull); }
#endif

// Measured in round 4 and rejected, all with bit-identical results:
// * the wave's 64 runs -- contiguous in memory with the slots in index order -- read as one flat stream of 16-byte loads into LDS,
//   nine words per record, every lane then adding its own records from there in the same order: 0.124 -> 0.184 ms at C3 with four
//   loads in flight per lane, 0.207 with eight (138 VGPRs: a wave per SIMD less), no change at C5 -- the LDS round trip and its
//   index arithmetic come on top of the memory one;
// * the chunks of ALL the wave's long runs taken four at a time, validity bytes requested together, then the records together (two
//   round trips per batch instead of two per chunk; nine waves in ten hold long runs at C3, 2.7 each): 0.119 -> 0.123 ms at C3,
//   0.629 -> 0.631 at C5 (two at a time 0.120, eight 0.122 / 0.679) -- the chain of a wave's round trips is hidden by the other waves.
// What the time goes to is bytes: timing-only ablations at C3 -- no slot sums at all 0.122 -> 0.076 ms, no long runs 0.103, no
// dL/dsh block (192 of the kernel's 480 MB) 0.081, none of the other stores 0.104 -- at 4.0 TB/s of mixed reads and writes, 64 % of
// what a copy reaches on this chip.
// One slot of a long run (added by the whole wave, lanes striding over the run).  The validity byte is awaited before the record is
// requested: asking for both at once -- the record of an invalid slot is readable garbage -- saves a round trip per 64 slots and
// was measured in round 4: per-Gaussian backward 0.120 -> 0.129 ms at C3, 0.63 -> 0.80 at C5 (the records of the invalid slots are
// traffic, and four more 16-byte registers in flight cost a wave per SIMD).
__device__ __forceinline__ void gsr_add_slot(const GsrGradSlot* __restrict__ slots, const uint8_t* __restrict__ valid,
                                             uint32_t s, float* acc)
{
	if (!valid[s]) return;
	const float4* sl = reinterpret_cast<const float4*>(slots + s);
	const float4 s0 = sl[0], s1 = sl[1];
	const float s2 = sl[2].x;
	acc[0] += s0.x; acc[1] += s0.y; acc[2] += s0.z; acc[3] += s0.w; acc[4] += s1.x;
	acc[5] += s1.y; acc[6] += s1.z; acc[7] += s1.w; acc[8] += s2;
}

// LEAF: inputs are the optimiser's raw leaves and the outputs are gradients w.r.t. them: the backward
// of exp / sigmoid / normalize / cat (gaussian_model.py:114-135) is applied in the epilogue.
// workgroup = one wave: waves of a CU then start and retire independently (phases of different waves mix)
#define GSR_GB_THREADS 64
template <bool LEAF>
__global__ void __launch_bounds__(GSR_GB_THREADS) gsr_gaussian_backward_kernel(GsrGaussianBackwardArgs a, int sh_via_lds, int skip_dsh)
{
	// staging of the dL/dsh output block: rows of 13 float4; the packed layout goes out in two halves of 32 rows
	// (6.6 KB per wave), the split leaf tensors as one linear 12 KB block
	__shared__ float4 s_sh[GSR_GB_THREADS / 64][(LEAF ? 64 : 32) * GSR_SH_ROW4];
	// the launch covers the Gaussians [first, first + count): the whole scene, or one part of it when the caller
	// pipelines the gradient exchange of a finished part under the kernel of the next (gsr_backward_gaussians)
	const int idx = a.first + blockIdx.x * GSR_GB_THREADS + threadIdx.x;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int M = a.M;
	const int end = a.first + a.count;
	const bool in_range = idx < end;
	// ---- (A) the Gaussian's own inputs, unconditionally for every Gaussian of the range: issued first, so that the
	//      three dependent steps below (these -> slot validity bytes -> slot records) are the only memory round
	//      trips of the wave.  The SH row is NOT among them: the forward left the nine derivatives the backward needs
	//      (GsrGeometry::sh_ddir), so LDS only stages the dL/dsh OUTPUT block.
	uint32_t tiles = 0, base = 0;
	int radius = 0;
	GsrVec3 mean = {0.f, 0.f, 0.f};
	float sc[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f}, cov_in[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	uint8_t clamp_bits = 0;
	float ddir9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	float leaf_opacity = 0.f;
	if (in_range) {
		tiles = a.g.tiles_touched[idx];
		base = a.g.slot_base[idx];
		if (a.radii) radius = a.radii[idx];
		mean.x = a.means3D[3 * idx]; mean.y = a.means3D[3 * idx + 1]; mean.z = a.means3D[3 * idx + 2];
		if (a.cov3D_precomp) {
#pragma unroll
			for (int k = 0; k < 6; k++) cov_in[k] = a.cov3D_precomp[6 * (size_t)idx + k];
		} else {
			sc[0] = a.scales[3 * idx]; sc[1] = a.scales[3 * idx + 1]; sc[2] = a.scales[3 * idx + 2];
			q[0] = a.rotations[4 * idx]; q[1] = a.rotations[4 * idx + 1]; q[2] = a.rotations[4 * idx + 2]; q[3] = a.rotations[4 * idx + 3];
		}
		if (a.shs) {
			clamp_bits = a.g.clamped[idx];
#pragma unroll
			for (int k = 0; k < 9; k++) ddir9[k] = a.g.sh_ddir[(size_t)k * a.P + idx];  // nine planes; garbage for culled Gaussians: unused
		}
		if (LEAF) leaf_opacity = a.g.splat[idx].opacity;
	}
	const int wave_first = a.first + blockIdx.x * GSR_GB_THREADS + wave * 64;
	const int nrows = min(64, end - wave_first);  // Gaussians of this wave (<= 0: none)

	// radii > 0 <=> tiles_touched > 0 (forward.cu:300-301 zeroes both together; culled: slot_base was never written)
	const bool visible = in_range && (a.radii ? radius > 0 : tiles > 0);
	if (!visible) tiles = 0;

	// ---- (C) validity bytes of all the slots a lane sums by itself: its <= GSR_SLOT_COOP bytes sit in at most
	//      GSR_VALID_WORDS consecutive aligned dwords (every lane's loads go to lines of its own, so the cost of this step
	//      is its number of load instructions: 6 instead of 18).  Bytes are 0 or 1 (render_backward.hip / tile_ranges).
	GsrSlotMask vmask = 0;
	if (tiles > 0 && tiles <= GSR_SLOT_COOP) {
		const uint32_t lead = base & 3u;
		const uint32_t* vp = reinterpret_cast<const uint32_t*>(a.slot_valid + (base - lead));  // slot_valid itself is 128-byte aligned
		uint32_t w[GSR_VALID_WORDS];
#pragma unroll
		for (int i = 0; i < GSR_VALID_WORDS; i++) w[i] = (4u * i < lead + tiles) ? vp[i] : 0u;  // the words behind the run belong to the buffer (4 R bytes, R used)
		unsigned long long bits = 0ull;
#pragma unroll
		for (int i = 0; i < GSR_VALID_WORDS; i++) {
			const uint32_t b4 = (w[i] & 1u) | ((w[i] >> 7) & 2u) | ((w[i] >> 14) & 4u) | ((w[i] >> 21) & 8u);
			bits |= (unsigned long long)b4 << (4 * i);
		}
		vmask = (GsrSlotMask)(bits >> lead) & (((GsrSlotMask)1 << tiles) - (GsrSlotMask)1);
	}
	// ---- (E) fixed-order sum of this Gaussian's (Gaussian,tile) slots: its VALID slots in ascending order, six per round;
	//      the records of a round are all requested before the first add, and only by the lanes that have one (a lane's
	//      record is a line of its own here too: masked-off lanes cost nothing)
	float acc[GSR_NACC];
#pragma unroll
	for (int i = 0; i < GSR_NACC; i++) acc[i] = 0.f;
	GsrSlotMask rem = vmask;
	while (rem) {
		float4 s0[GSR_SLOT_ROUND], s1[GSR_SLOT_ROUND];
		float s2[GSR_SLOT_ROUND];
		bool ok[GSR_SLOT_ROUND];
#pragma unroll
		for (int j = 0; j < GSR_SLOT_ROUND; j++) {
			ok[j] = rem != 0;
			const uint32_t k = gsr_slot_mask_first(rem);
			rem &= rem - (GsrSlotMask)1;
			if (ok[j]) {
				const float4* sl = reinterpret_cast<const float4*>(a.slots + (base + k));
				s0[j] = sl[0]; s1[j] = sl[1]; s2[j] = sl[2].x;
			}
		}
#pragma unroll
		for (int j = 0; j < GSR_SLOT_ROUND; j++)
			if (ok[j]) {
				acc[0] += s0[j].x; acc[1] += s0[j].y; acc[2] += s0[j].z; acc[3] += s0[j].w; acc[4] += s1[j].x;
				acc[5] += s1[j].y; acc[6] += s1[j].z; acc[7] += s1[j].w; acc[8] += s2[j];
			}
	}
	unsigned long long big = __builtin_amdgcn_ballot_w64(tiles > GSR_SLOT_COOP);
	while (big) {  // wave-uniform
		const int src = __ffsll((long long)big) - 1;
		big &= big - 1;
		const uint32_t s_tiles = __shfl(tiles, src, 64), s_base = __shfl(base, src, 64);
		float part[GSR_NACC];
#pragma unroll
		for (int i = 0; i < GSR_NACC; i++) part[i] = 0.f;
		for (uint32_t k = lane; k < s_tiles; k += 64) gsr_add_slot(a.slots, a.slot_valid, s_base + k, part);
#pragma unroll
		for (int i = 0; i < GSR_NACC; i++) {
			const float tot = __shfl(gsr_wave_sum_to_lane63(part[i]), 63, 64);
			if (lane == src) acc[i] = tot;
		}
	}
	__builtin_amdgcn_wave_barrier();

	float dmean2D[3] = {acc[0], acc[1], 0.f}, dconic[4] = {acc[2], acc[3], 0.f, acc[4]}, dop = acc[5];
	float dcolor[3] = {acc[6], acc[7], acc[8]};
	float dmean3D[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	float dscale[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
	// gradient outputs are indexed relative to out_row0 (0, or `first` when each part has a buffer of its own)
	const size_t orow = (size_t)(idx - a.out_row0);
	const int out_wave_first = wave_first - a.out_row0;
	float* dsh_global = (!LEAF && a.dL_dsh && in_range) ? a.dL_dsh + orow * M * 3 : nullptr;
	float dsh_local[48];  // LEAF without the LDS path: the row of the feature gradient
	float q_raw[4] = {0.f, 0.f, 0.f, 0.f}, q_den = 1.f;
	float dRGB[3] = {0.f, 0.f, 0.f};  // dL/dcolor with the channels clamped by the forward zeroed
	float basis_keep[16];             // SH basis at the view direction (entries >= (D+1)^2 zero), visible Gaussians
#pragma unroll
	for (int k = 0; k < 16; k++) basis_keep[k] = 0.f;

	if (visible) {
		// ---- computeCov2DCUDA, backward.cu:144-277 ----
		float cov3D[6];
		if (a.cov3D_precomp) {
#pragma unroll
			for (int k = 0; k < 6; k++) cov3D[k] = cov_in[k];
		} else {
			if (LEAF) {
				sc[0] = gsr_act_exp(sc[0]); sc[1] = gsr_act_exp(sc[1]); sc[2] = gsr_act_exp(sc[2]);
				q_raw[0] = q[0]; q_raw[1] = q[1]; q_raw[2] = q[2]; q_raw[3] = q[3];
				q_den = gsr_act_normalize4(q_raw, q);
			}
			gsr_cov3d(sc, a.scale_modifier, q, cov3D);  // recomputed: identical bits to the forward's
		}
		const float dcx = dconic[0], dcy = dconic[1], dcz = dconic[3];
		GsrCov2D c2;
		gsr_cov2d(mean, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, cov3D, a.viewmatrix, c2);
		const float h_x = a.focal_x, h_y = a.focal_y;
		const float x_grad_mul = (c2.txtz < -c2.limx || c2.txtz > c2.limx) ? 0.f : 1.f;
		const float y_grad_mul = (c2.tytz < -c2.limy || c2.tytz > c2.limy) ? 0.f : 1.f;
		const GsrMat3& T = c2.T;
		const GsrMat3& Wm = c2.W;
		const GsrMat3& Vrk = c2.Vrk;
		const GsrVec3 t = c2.t;
		const float ca = c2.a, cb = c2.b, cc = c2.c;
		const float denom = ca * cc - cb * cb;
		float dL_da = 0, dL_db = 0, dL_dc = 0;
		const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
		if (denom2inv != 0) {
			dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
			dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
			dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
			dcov[0] = (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
			dcov[3] = (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
			dcov[5] = (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);
			dcov[1] = 2 * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][1] * dL_dc;
			dcov[2] = 2 * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][2] * dL_dc;
			dcov[4] = 2 * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db + 2 * T.m[1][1] * T.m[1][2] * dL_dc;
		}
#define TV(r_, k_) (T.m[r_][0] * Vrk.m[k_][0] + T.m[r_][1] * Vrk.m[k_][1] + T.m[r_][2] * Vrk.m[k_][2])
		const float dL_dT00 = 2 * TV(0, 0) * dL_da + TV(1, 0) * dL_db;
		const float dL_dT01 = 2 * TV(0, 1) * dL_da + TV(1, 1) * dL_db;
		const float dL_dT02 = 2 * TV(0, 2) * dL_da + TV(1, 2) * dL_db;
		const float dL_dT10 = 2 * TV(1, 0) * dL_dc + TV(0, 0) * dL_db;
		const float dL_dT11 = 2 * TV(1, 1) * dL_dc + TV(0, 1) * dL_db;
		const float dL_dT12 = 2 * TV(1, 2) * dL_dc + TV(0, 2) * dL_db;
#undef TV
		const float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
		const float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
		const float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
		const float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
		const float tz = 1.f / t.z;
		const float tz2 = tz * tz;
		const float tz3 = tz2 * tz;
		const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
		const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
		const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
		const float* vm = a.viewmatrix;  // transformVec4x3Transpose, auxiliary.h:91-99
		dmean3D[0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
		dmean3D[1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
		dmean3D[2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;

		// ---- preprocessCUDA backward, backward.cu:349-399 ----
		const float* proj = a.projmatrix;
		const float m_hom_w = proj[3] * mean.x + proj[7] * mean.y + proj[11] * mean.z + proj[15];
		const float m_w = 1.0f / (m_hom_w + 0.0000001f);
		const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
		const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
		const float gx2 = dmean2D[0], gy2 = dmean2D[1];
		dmean3D[0] += (proj[0] * m_w - proj[3] * mul1) * gx2 + (proj[1] * m_w - proj[3] * mul2) * gy2;
		dmean3D[1] += (proj[4] * m_w - proj[7] * mul1) * gx2 + (proj[5] * m_w - proj[7] * mul2) * gy2;
		dmean3D[2] += (proj[8] * m_w - proj[11] * mul1) * gx2 + (proj[9] * m_w - proj[11] * mul2) * gy2;

		if (a.shs) {
			if (sh_via_lds) {
				// dL_dsh = basis x dL/dRGB: kept as its two factors until the block store below
				gsr_sh_backward(a.D, M, mean, a.cam_pos, ddir9, clamp_bits, dcolor, dmean3D, nullptr, false, dRGB, basis_keep);
				if (!skip_dsh && LEAF) {
					const int used_sh = (a.D + 1) * (a.D + 1);
					float o[48];
#pragma unroll
					for (int e = 0; e < 48; e++) o[e] = (e / 3 < used_sh) ? basis_keep[e / 3] * dRGB[e % 3] : 0.f;
					gsr_sh_lin_row_put(reinterpret_cast<float*>(s_sh[wave]), lane, o);
				}
			} else if (LEAF) {
				const int used = (a.D + 1) * (a.D + 1);
				float basis[16];
				gsr_sh_backward(a.D, used, mean, a.cam_pos, ddir9, clamp_bits, dcolor, dmean3D, nullptr, false, dRGB, basis);
#pragma unroll
				for (int e = 0; e < 48; e++) dsh_local[e] = (e / 3 < used) ? basis[e / 3] * dRGB[e % 3] : 0.f;
			} else {
				gsr_sh_backward(a.D, M, mean, a.cam_pos, ddir9, clamp_bits, dcolor, dmean3D, dsh_global, !skip_dsh, dRGB);
			}
		}
		if (a.scales)
			gsr_cov3d_backward(sc, a.scale_modifier, q, dcov, dscale, drot);
		if (LEAF) {
			// exp backward: grad * result;  sigmoid backward: grad * ((1 - y) * y)
			dscale[0] *= sc[0]; dscale[1] *= sc[1]; dscale[2] *= sc[2];
			const float o = leaf_opacity;
			dop = dop * ((1.0f - o) * o);
			// F.normalize backward: y = x / d, d = clamp_min(||x||, 1e-12)
			float gd = 0.f;
#pragma unroll
			for (int k = 0; k < 4; k++) gd += -drot[k] * q_raw[k] / (q_den * q_den);
			const float r = (q_den > 1e-12f) ? gd / q_den : 0.f;
#pragma unroll
			for (int k = 0; k < 4; k++) drot[k] = drot[k] / q_den + q_raw[k] * r;
		}
	}

	// ---- dL_dsh: zeros for culled Gaussians; coalesced write-out of the wave's block ----
	if (skip_dsh) {
		// view-parallel mode: no SH gradient here; dL_dcolor carries the clamp-masked dL/dRGB instead
		dcolor[0] = dRGB[0]; dcolor[1] = dRGB[1]; dcolor[2] = dRGB[2];
	} else if (sh_via_lds && LEAF) {
		if (!visible) {
			float z[48];
#pragma unroll
			for (int e = 0; e < 48; e++) z[e] = 0.f;
			gsr_sh_lin_row_put(reinterpret_cast<float*>(s_sh[wave]), lane, z);
		}
		__builtin_amdgcn_wave_barrier();
		if (nrows > 0) gsr_sh_lin_store(reinterpret_cast<const float*>(s_sh[wave]), a.dL_dsh, a.dL_dsh_rest, out_wave_first, nrows, lane);
	} else if (sh_via_lds) {
		// packed (P,16,3) output: each half of the wave writes its rows (12 conflict-free float4 per lane; zeros for culled
		// Gaussians, +0 for the rows >= (D+1)^2), then the whole wave streams the half out with coalesced 16-byte stores
		const int used_sh = visible ? (a.D + 1) * (a.D + 1) : 0;
#pragma unroll
		for (int half = 0; half < 2; half++) {
			if ((lane >> 5) == half) {
#pragma unroll
				for (int j = 0; j < 12; j++) {
					float o[4];
#pragma unroll
					for (int t = 0; t < 4; t++)
						o[t] = ((4 * j + t) / 3 < used_sh) ? basis_keep[(4 * j + t) / 3] * dRGB[(4 * j + t) % 3] : 0.f;
					s_sh[wave][(lane & 31) * GSR_SH_ROW4 + j] = make_float4(o[0], o[1], o[2], o[3]);
				}
			}
			__builtin_amdgcn_wave_barrier();
			if (nrows > 32 * half) gsr_sh_rows_store_half(s_sh[wave], a.dL_dsh, out_wave_first, nrows, lane, half);
			__builtin_amdgcn_wave_barrier();
		}
	} else if (LEAF) {
		if (in_range) {
			const int used = visible ? (a.D + 1) * (a.D + 1) : 0;
#pragma unroll
			for (int k = 0; k < 16; k++)
				if (k < M) {
#pragma unroll
					for (int ch = 0; ch < 3; ch++) {
						const float v = k < used ? dsh_local[k * 3 + ch] : 0.f;
						if (k == 0) a.dL_dsh[3 * orow + ch] = v;
						else a.dL_dsh_rest[(orow * (M - 1) + (k - 1)) * 3 + ch] = v;
					}
				}
		}
	} else if ((!visible || !a.shs) && dsh_global) {
		for (int k = 0; k < M * 3; k++) dsh_global[k] = 0.f;
	}

	if (!in_range) return;
#pragma unroll
	for (int k = 0; k < 3; k++) {
		a.dL_dmean2D[3 * orow + k] = dmean2D[k];
		if (a.dL_dcolor) a.dL_dcolor[3 * orow + k] = dcolor[k];
		a.dL_dmean3D[3 * orow + k] = dmean3D[k];
		a.dL_dscale[3 * orow + k] = dscale[k];
	}
#pragma unroll
	for (int k = 0; k < 4; k++) {
		if (a.dL_dconic) a.dL_dconic[4 * orow + k] = dconic[k];
		a.dL_drot[4 * orow + k] = drot[k];
	}
	if (a.dL_dcov3D) {
#pragma unroll
		for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * orow + k] = dcov[k];
	}
	a.dL_dopacity[orow] = dop;
	// ---- densification statistics of this view (train.py:157-159, scene/gaussian_model.py:599-602) ----
	//   max_radii2D[vis] = max(max_radii2D[vis], radii[vis]);  xyz_gradient_accum[vis] += norm(viewspace.grad[vis,:2]);  denom[vis] += 1
	if (visible) {
		if (a.stat_xyz_gradient_accum) a.stat_xyz_gradient_accum[idx] += sqrtf(dmean2D[0] * dmean2D[0] + dmean2D[1] * dmean2D[1]);
		if (a.stat_denom) a.stat_denom[idx] += 1.0f;
		if (a.stat_max_radii2D) {
			a.stat_max_radii2D[idx] = fmaxf(a.stat_max_radii2D[idx], (float)radius);
		}
	}
}

void gsr_launch_gaussian_backward(const GsrGaussianBackwardArgs& a, hipStream_t s)
{
	// LDS-transposed SH path: the flagship layout (16 coefficients) with 16-byte aligned tensors
	const int skip_dsh = (a.shs && !a.dL_dsh) ? 1 : 0;  // view-parallel mode (include/gsr.h)
	int sh_via_lds = (a.shs && a.M == 16 && ((uintptr_t)a.shs & 15u) == 0 && (skip_dsh || ((uintptr_t)a.dL_dsh & 15u) == 0)) ? 1 : 0;
	if (a.leaf) {
		if (((uintptr_t)a.shs_rest & 15u) != 0 || (!skip_dsh && ((uintptr_t)a.dL_dsh_rest & 15u) != 0)) sh_via_lds = 0;
		hipLaunchKernelGGL(gsr_gaussian_backward_kernel<true>, dim3((a.count + GSR_GB_THREADS - 1) / GSR_GB_THREADS), dim3(GSR_GB_THREADS), 0, s, a, sh_via_lds, skip_dsh);
	} else {
		hipLaunchKernelGGL(gsr_gaussian_backward_kernel<false>, dim3((a.count + GSR_GB_THREADS - 1) / GSR_GB_THREADS), dim3(GSR_GB_THREADS), 0, s, a, sh_via_lds, skip_dsh);
	}
}

// ---- view-parallel SH gradient (no reference counterpart) ------------------------------
// dL/dsh of ONE view is the outer product basis(dir) x dL/dRGB per Gaussian (backward.cu:45-96), so the
// SH gradient summed over V views can be rebuilt from V x 3 floats per Gaussian (+ V camera
// positions) instead of exchanging 3*M floats per Gaussian: ranks all-gather the clamp-masked
// dL/dRGB of their views (12 B per Gaussian per view) and every rank runs this kernel.  Views are
// added in index order with the products rounded first -- the sum a fixed-order all-reduce of the
// per-view gradients would give.
__global__ void __launch_bounds__(GSR_GB_THREADS) gsr_sh_grad_from_views_kernel(int P, int D, int M, int V, const float* __restrict__ means3D,
                                                                      const float* __restrict__ cam_pos,
                                                                      const float* __restrict__ dL_dRGB, long long view_stride,
                                                                      float* __restrict__ dL_dsh, int via_lds)
{
	__shared__ float4 s_sh[GSR_GB_THREADS / 64][64 * GSR_SH_ROW4];
	const int idx = blockIdx.x * GSR_GB_THREADS + threadIdx.x;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const bool in_range = idx < P;
	const int used = (D + 1) * (D + 1);
	float acc[48];
#pragma unroll
	for (int i = 0; i < 48; i++) acc[i] = 0.f;
	if (in_range) {
		const float mx = means3D[3 * idx], my = means3D[3 * idx + 1], mz = means3D[3 * idx + 2];
		for (int v = 0; v < V; v++) {
			const float* g = dL_dRGB + (size_t)v * view_stride + (size_t)idx * 3;
			const float g0 = g[0], g1 = g[1], g2 = g[2];
			if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;  // culled in this view (or no gradient): adds exact zeros
			const float dx = mx - cam_pos[3 * v], dy = my - cam_pos[3 * v + 1], dz = mz - cam_pos[3 * v + 2];
			const float len = sqrtf(dx * dx + dy * dy + dz * dz);
			float basis[16];
			gsr_sh_basis(D, dx / len, dy / len, dz / len, basis);
#pragma unroll
			for (int k = 0; k < 16; k++)
				if (k < used) {
					acc[3 * k] += basis[k] * g0;
					acc[3 * k + 1] += basis[k] * g1;
					acc[3 * k + 2] += basis[k] * g2;
				}
		}
	}
	if (via_lds) {  // M == 16: coalesced float4 stream through an LDS transpose
		const int wave_first = blockIdx.x * GSR_GB_THREADS + wave * 64;
		const int nrows = min(64, P - wave_first);
#pragma unroll
		for (int j = 0; j < 12; j++)
			s_sh[wave][lane * GSR_SH_ROW4 + j] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
		__builtin_amdgcn_wave_barrier();
		if (nrows > 0) {
			float4* dst = reinterpret_cast<float4*>(dL_dsh + (size_t)wave_first * 48);
#pragma unroll
			for (int it = 0; it < 12; it++) {
				const int f = it * 64 + lane;
				if (f < nrows * 12) dst[f] = s_sh[wave][(f / 12) * GSR_SH_ROW4 + (f % 12)];
			}
		}
	} else if (in_range) {
		float* out = dL_dsh + (size_t)idx * M * 3;  // M <= 16 (checked by the caller)
#pragma unroll
		for (int i = 0; i < 48; i++)  // constant indices only: a dynamic one would move `acc` to scratch for BOTH paths
			if (i < M * 3) out[i] = acc[i];
	}
}

void gsr_launch_sh_grad_from_views(int P, int D, int M, int V, const float* means3D, const float* cam_pos, const float* dL_dRGB,
                                   int64_t view_stride, float* dL_dsh, hipStream_t s)
{
	const int via_lds = (M == 16 && ((uintptr_t)dL_dsh & 15u) == 0) ? 1 : 0;
	hipLaunchKernelGGL(gsr_sh_grad_from_views_kernel, dim3((P + GSR_GB_THREADS - 1) / GSR_GB_THREADS), dim3(GSR_GB_THREADS), 0, s, P, D, M, V, means3D, cam_pos, dL_dRGB,
	                   (long long)view_stride, dL_dsh, via_lds);
}
