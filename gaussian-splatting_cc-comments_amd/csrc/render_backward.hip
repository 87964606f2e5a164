// render_backward.hip -- per-tile back-to-front gradient of the alpha compositing.
//
// Replaces renderCUDA<3> backward (cuda_rasterizer/backward.cu:408-601).  The reference issues
// nine fp32 atomicAdd per contributing (pixel, Gaussian) pair, all 256 lanes of a tile hitting
// the same addresses.  Here each pair's nine partials are summed across the wave with DPP, across
// the four waves through LDS, and the tile's total for one (Gaussian, tile) instance is written
// once, with plain stores, into that instance's own 48-byte slot (slot = the instance's position
// in the unsorted, per-Gaussian-contiguous order).  The per-Gaussian kernel then adds a Gaussian's
// contiguous run of slots in a fixed order: no atomics, bitwise reproducible.
// Per-pixel arithmetic is that of backward.cu:507-599.
#include <stdlib.h>

#include "render_common.h"

#define GSR_BWD_NV 9

// ---- workgroup-per-tile kernel (the reference's decomposition; kept for A/B, GSR_RENDER_V0=1) --
__global__ void __launch_bounds__(GSR_TILE_PIX) gsr_render_backward_kernel(
	int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const GsrSplat* __restrict__ splat, const float* __restrict__ bg, const float* __restrict__ final_Ts,
	const uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ tile_max_contrib,
	const float* __restrict__ dL_dpixels, GsrGradSlot* __restrict__ slots, uint8_t* __restrict__ slot_valid)
{
	__shared__ float4 s_rec[3][GSR_TILE_PIX];
	__shared__ float s_part[4][GSR_TILE_PIX][GSR_BWD_NV + 1];  // per-wave sums per batch instance (+ hit count)

	const int tile = blockIdx.x;
	const int tx = tile % gx, ty = tile / gx;
	const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
	const int px = tx * GSR_TILE_X + lx, py = ty * GSR_TILE_Y + ly;
	const bool inside = px < W && py < H;
	const float pfx = (float)px, pfy = (float)py;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

	const uint2 range = ranges[tile];
	// instances at positions >= tile_max_contrib were never blended by any pixel of the tile
	const int n = (int)min(range.y - range.x, tile_max_contrib[tile]);
	const int rounds = (n + GSR_TILE_PIX - 1) / GSR_TILE_PIX;

	const uint32_t pix_id = inside ? (uint32_t)(W * py + px) : 0u;
	const size_t plane = (size_t)H * W;
	const float T_final = inside ? final_Ts[pix_id] : 0.f;
	float T = T_final;
	int contributor = n;
	const int last_contributor = inside ? (int)n_contrib[pix_id] : 0;

	float accum0 = 0.f, accum1 = 0.f, accum2 = 0.f;
	float dpx0 = 0.f, dpx1 = 0.f, dpx2 = 0.f;
	if (inside) {
		dpx0 = dL_dpixels[pix_id];
		dpx1 = dL_dpixels[plane + pix_id];
		dpx2 = dL_dpixels[2 * plane + pix_id];
	}
	float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;
	const float bg_dot_dpixel = bg[0] * dpx0 + bg[1] * dpx1 + bg[2] * dpx2;
	const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

	int toDo = n;
	for (int i = 0; i < rounds; i++, toDo -= GSR_TILE_PIX) {
		__syncthreads();
		const int progress = i * GSR_TILE_PIX + threadIdx.x;
		if (progress < n) {
			const uint32_t id = point_list[range.x + (n - progress - 1)];
			const float4* rec = reinterpret_cast<const float4*>(splat + id);
			s_rec[0][threadIdx.x] = rec[0];
			s_rec[1][threadIdx.x] = rec[1];
			s_rec[2][threadIdx.x] = rec[2];
		}
		__syncthreads();
		const int cnt = min(GSR_TILE_PIX, toDo);
		for (int j = 0; j < cnt; j++) {
			contributor--;
			float v[GSR_BWD_NV];
#pragma unroll
			for (int k = 0; k < GSR_BWD_NV; k++) v[k] = 0.f;
			bool hit = false;
			if (inside && contributor < last_contributor) {
				const float4 r0 = s_rec[0][j];
				const float4 r1 = s_rec[1][j];
				const float dx = r0.x - pfx, dy = r0.y - pfy;
				const float power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;
				if (!(power > 0.0f)) {
					const float G = __expf(power);
					const float alpha = fminf(0.99f, r1.y * G);
					if (!(alpha < 1.0f / 255.0f)) {
						hit = true;
						const float inv1ma = 1.f / (1.f - alpha);
						T = T * inv1ma;
						const float dchannel_dcolor = alpha * T;
						const float c0 = r1.z, c1 = r1.w, c2 = s_rec[2][j].x;
						accum0 = last_alpha * lc0 + (1.f - last_alpha) * accum0;
						accum1 = last_alpha * lc1 + (1.f - last_alpha) * accum1;
						accum2 = last_alpha * lc2 + (1.f - last_alpha) * accum2;
						lc0 = c0; lc1 = c1; lc2 = c2;
						float dL_dalpha = (c0 - accum0) * dpx0 + (c1 - accum1) * dpx1 + (c2 - accum2) * dpx2;
						v[6] = dchannel_dcolor * dpx0;
						v[7] = dchannel_dcolor * dpx1;
						v[8] = dchannel_dcolor * dpx2;
						dL_dalpha *= T;
						last_alpha = alpha;
						dL_dalpha += (-T_final * inv1ma) * bg_dot_dpixel;
						const float dL_dG = r1.y * dL_dalpha;
						const float gdx = G * dx, gdy = G * dy;
						const float dG_ddelx = -gdx * r0.z - gdy * r0.w;
						const float dG_ddely = -gdy * r1.x - gdx * r0.w;
						v[0] = dL_dG * dG_ddelx * ddelx_dx;
						v[1] = dL_dG * dG_ddely * ddely_dy;
						v[2] = -0.5f * gdx * dx * dL_dG;
						v[3] = -0.5f * gdx * dy * dL_dG;
						v[4] = -0.5f * gdy * dy * dL_dG;
						v[5] = G * dL_dalpha;
					}
				}
			}
			const unsigned long long hits = __ballot(hit);
			if (hits) {  // wave-uniform
#pragma unroll
				for (int k = 0; k < GSR_BWD_NV; k++) v[k] = gsr_wave_sum_to_lane63(v[k]);
			}
			if (lane == 63) {
#pragma unroll
				for (int k = 0; k < GSR_BWD_NV; k++) s_part[wave][j][k] = v[k];
				s_part[wave][j][GSR_BWD_NV] = (float)__popcll(hits);
			}
		}
		__syncthreads();
		// one thread per batch instance: add the four waves, store the instance's slot
		if (threadIdx.x < cnt) {
			const int j = threadIdx.x;
			float sum[GSR_BWD_NV + 1];
#pragma unroll
			for (int k = 0; k <= GSR_BWD_NV; k++)
				sum[k] = (s_part[0][j][k] + s_part[1][j][k]) + (s_part[2][j][k] + s_part[3][j][k]);
			if (sum[GSR_BWD_NV] > 0.f) {
				const float4 r2 = s_rec[2][j];
				const uint32_t slot_base = __float_as_uint(r2.y), rmin = __float_as_uint(r2.z), rwh = __float_as_uint(r2.w);
				const uint32_t slot = slot_base + ((uint32_t)ty - (rmin >> 16)) * (rwh & 0xffffu) + ((uint32_t)tx - (rmin & 0xffffu));
				float4* out = reinterpret_cast<float4*>(slots + slot);
				out[0] = make_float4(sum[0], sum[1], sum[2], sum[3]);
				out[1] = make_float4(sum[4], sum[5], sum[6], sum[7]);
				out[2] = make_float4(sum[8], 0.f, 0.f, 0.f);
				slot_valid[slot] = 1;
			}
		}
	}
}

// ---- wave-per-tile kernel (default) ------------------------------------------------------------
// One wave64 per tile, four pixels per lane (render_common.h).  Per surviving instance each lane
// first adds its four pixels' nine partials in registers, then ONE wave-wide DPP reduction per
// value yields the tile total, which lane 63 stores straight into the instance's gradient slot:
// no LDS partials, no workgroup barrier, no atomics.  Instances are visited back to front in
// batches of 64, gathered one batch ahead and culled/compacted exactly like the forward kernel.
__global__ void __launch_bounds__(64 * GSR_WAVES_PER_WG) gsr_render_backward_wave_kernel(
	int W, int H, int gx, int ntiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const GsrSplat* __restrict__ splat, const float* __restrict__ bg, const float* __restrict__ final_Ts,
	const uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ tile_max_contrib,
	const float* __restrict__ dL_dpixels, GsrGradSlot* __restrict__ slots, uint8_t* __restrict__ slot_valid)
{
	__shared__ float4 s_rec[GSR_WAVES_PER_WG][3][64];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int tile = blockIdx.x * GSR_WAVES_PER_WG + wave;
	if (tile >= ntiles) return;  // wave-uniform; no barriers below
	float4(*rec)[64] = s_rec[wave];

	const int tx = tile % gx, ty = tile / gx;
	const int px = tx * GSR_TILE_X + (lane & 15);
	const int py0 = ty * GSR_TILE_Y + (lane >> 4);
	const float pfx = (float)px;
	const float x0f = (float)(tx * GSR_TILE_X), y0f = (float)(ty * GSR_TILE_Y);

	const uint2 range = ranges[tile];
	const int n = (int)min(range.y - range.x, tile_max_contrib[tile]);  // the tail was never blended
	const uint32_t* plist = point_list + range.x;
	const size_t plane = (size_t)H * W;
	const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
	const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

	float T[GSR_PIX_PER_LANE], T_final[GSR_PIX_PER_LANE], pfy[GSR_PIX_PER_LANE], bgdot[GSR_PIX_PER_LANE];
	float ac0[GSR_PIX_PER_LANE], ac1[GSR_PIX_PER_LANE], ac2[GSR_PIX_PER_LANE];
	float lc0[GSR_PIX_PER_LANE], lc1[GSR_PIX_PER_LANE], lc2[GSR_PIX_PER_LANE], last_alpha[GSR_PIX_PER_LANE];
	float dp0[GSR_PIX_PER_LANE], dp1[GSR_PIX_PER_LANE], dp2[GSR_PIX_PER_LANE];
	int last_contributor[GSR_PIX_PER_LANE];
#pragma unroll
	for (int k = 0; k < GSR_PIX_PER_LANE; k++) {
		const int py = py0 + 4 * k;
		const bool inside = px < W && py < H;
		const uint32_t pix_id = inside ? (uint32_t)(W * py + px) : 0u;
		pfy[k] = (float)py;
		T_final[k] = inside ? final_Ts[pix_id] : 0.f;
		T[k] = T_final[k];
		last_contributor[k] = inside ? (int)n_contrib[pix_id] : 0;
		dp0[k] = inside ? dL_dpixels[pix_id] : 0.f;
		dp1[k] = inside ? dL_dpixels[plane + pix_id] : 0.f;
		dp2[k] = inside ? dL_dpixels[2 * plane + pix_id] : 0.f;
		bgdot[k] = bg0 * dp0[k] + bg1 * dp1[k] + bg2 * dp2[k];
		ac0[k] = ac1[k] = ac2[k] = 0.f;
		lc0[k] = lc1[k] = lc2[k] = 0.f;
		last_alpha[k] = 0.f;
	}

	// back to front: batch position q = base + lane maps to range position n - 1 - q
	float4 ra = make_float4(0, 0, 0, 0), rb = ra, rc = ra;
	if (lane < n) {
		const float4* p = reinterpret_cast<const float4*>(splat + plist[n - 1 - lane]);
		ra = p[0]; rb = p[1]; rc = p[2];
	}
	uint32_t id_next = (64 + lane < n) ? plist[n - 1 - (64 + lane)] : 0u;

	for (int base = 0; base < n; base += 64) {
		const bool keep = (base + lane < n) && gsr_tile_may_hit(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, x0f, y0f);
		const unsigned long long mask = __ballot(keep);
		const int cnt = __popcll(mask);
		if (keep) {
			const int pos = gsr_mbcnt(mask);
			const uint32_t rmin = __float_as_uint(rc.z), rwh = __float_as_uint(rc.w);
			const uint32_t slot = __float_as_uint(rc.y) + ((uint32_t)ty - (rmin >> 16)) * (rwh & 0xffffu) + ((uint32_t)tx - (rmin & 0xffffu));
			rec[0][pos] = ra;
			rec[1][pos] = rb;
			rec[2][pos] = make_float4(rc.x, __int_as_float(n - 1 - (base + lane)), __uint_as_float(slot), 0.f);
		}
		if (base + 64 + lane < n) {
			const float4* p = reinterpret_cast<const float4*>(splat + id_next);
			ra = p[0]; rb = p[1]; rc = p[2];
		}
		id_next = (base + 128 + lane < n) ? plist[n - 1 - (base + 128 + lane)] : 0u;
		__builtin_amdgcn_wave_barrier();

		for (int j = 0; j < cnt; j++) {
			const float4 A = rec[0][j];
			const float4 B = rec[1][j];
			const float4 Cc = rec[2][j];
			const int contributor = __float_as_int(Cc.y);  // position in the full range (backward.cu:511-515)
			const float dx = A.x - pfx;
			float v[GSR_BWD_NV];
#pragma unroll
			for (int i = 0; i < GSR_BWD_NV; i++) v[i] = 0.f;
			bool any = false;
#pragma unroll
			for (int k = 0; k < GSR_PIX_PER_LANE; k++) {
				const float dy = A.y - pfy[k];
				const float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
				const float G = __expf(power);
				const float alpha = fminf(0.99f, B.y * G);
				const bool hit = contributor < last_contributor[k] && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
				if (hit) {
					any = true;
					const float inv1ma = 1.f / (1.f - alpha);
					T[k] = T[k] * inv1ma;
					const float dchannel_dcolor = alpha * T[k];
					ac0[k] = last_alpha[k] * lc0[k] + (1.f - last_alpha[k]) * ac0[k];
					ac1[k] = last_alpha[k] * lc1[k] + (1.f - last_alpha[k]) * ac1[k];
					ac2[k] = last_alpha[k] * lc2[k] + (1.f - last_alpha[k]) * ac2[k];
					lc0[k] = B.z; lc1[k] = B.w; lc2[k] = Cc.x;
					float dL_dalpha = (B.z - ac0[k]) * dp0[k] + (B.w - ac1[k]) * dp1[k] + (Cc.x - ac2[k]) * dp2[k];
					v[6] += dchannel_dcolor * dp0[k];
					v[7] += dchannel_dcolor * dp1[k];
					v[8] += dchannel_dcolor * dp2[k];
					dL_dalpha *= T[k];
					last_alpha[k] = alpha;
					dL_dalpha += (-T_final[k] * inv1ma) * bgdot[k];
					const float dL_dG = B.y * dL_dalpha;
					const float gdx = G * dx, gdy = G * dy;
					const float dG_ddelx = -gdx * A.z - gdy * A.w;
					const float dG_ddely = -gdy * B.x - gdx * A.w;
					v[0] += dL_dG * dG_ddelx * ddelx_dx;
					v[1] += dL_dG * dG_ddely * ddely_dy;
					v[2] += -0.5f * gdx * dx * dL_dG;
					v[3] += -0.5f * gdx * dy * dL_dG;
					v[4] += -0.5f * gdy * dy * dL_dG;
					v[5] += G * dL_dalpha;
				}
			}
			if (__ballot(any)) {  // wave-uniform
#pragma unroll
				for (int i = 0; i < GSR_BWD_NV; i++) v[i] = gsr_wave_sum_to_lane63(v[i]);
				if (lane == 63) {
					const uint32_t slot = __float_as_uint(Cc.z);
					float4* out = reinterpret_cast<float4*>(slots + slot);
					out[0] = make_float4(v[0], v[1], v[2], v[3]);
					out[1] = make_float4(v[4], v[5], v[6], v[7]);
					out[2] = make_float4(v[8], 0.f, 0.f, 0.f);
					slot_valid[slot] = 1;
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
}

static bool gsr_use_v0()
{
	static int v = -1;
	if (v < 0) {
		const char* e = getenv("GSR_RENDER_V0");
		v = (e && e[0] == '1') ? 1 : 0;
	}
	return v == 1;
}

void gsr_launch_render_backward(int W, int H, GsrImage img, const uint32_t* point_list, const GsrSplat* splat,
                                const float* bg, const float* dL_dpix, GsrGradSlot* slots, uint8_t* slot_valid,
                                hipStream_t s)
{
	const int gx = gsr_grid_x(W), gy = gsr_grid_y(H);
	const int ntiles = gx * gy;
	if (gsr_use_v0()) {
		hipLaunchKernelGGL(gsr_render_backward_kernel, dim3(ntiles), dim3(GSR_TILE_PIX), 0, s, W, H, gx, img.ranges,
		                   point_list, splat, bg, img.final_T, img.n_contrib, img.tile_max_contrib, dL_dpix, slots,
		                   slot_valid);
		return;
	}
	const int nwg = (ntiles + GSR_WAVES_PER_WG - 1) / GSR_WAVES_PER_WG;
	hipLaunchKernelGGL(gsr_render_backward_wave_kernel, dim3(nwg), dim3(64 * GSR_WAVES_PER_WG), 0, s, W, H, gx, ntiles,
	                   img.ranges, point_list, splat, bg, img.final_T, img.n_contrib, img.tile_max_contrib, dL_dpix,
	                   slots, slot_valid);
}
