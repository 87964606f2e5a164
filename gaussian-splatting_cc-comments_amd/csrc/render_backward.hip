// render_backward.hip -- per-tile back-to-front gradient of the alpha compositing.
//
// Replaces renderCUDA<3> backward (cuda_rasterizer/backward.cu:408-601).  The reference issues
// nine fp32 atomicAdd per contributing (pixel, Gaussian) pair, all 256 lanes of a tile hitting the
// same addresses -- the slowest atomic shape on this chip (MI355X_MICROARCH.md "Global float
// atomics").  Here one wave64 owns the tile (render_common.h): per surviving instance each lane
// first adds its four pixels' nine partials in registers, a butterfly of v_permlane32/16_swap + DPP
// folds the wave, and the tile's total for this (Gaussian, tile) instance is written once, with
// plain stores, into that instance's own 48-byte slot (slot = its position in the depth-ordered,
// per-Gaussian-contiguous emission order).  The per-Gaussian kernel then adds each Gaussian's
// contiguous run of slots in a fixed order: no atomics, no LDS partials, no workgroup barrier,
// bitwise reproducible.  Per-pixel arithmetic is that of backward.cu:507-599; G and alpha come
// from the same instruction sequence as in the forward kernel (gsr_pair_power), so the products
// the forward formed are the ones undone here.
#include "render_common.h"

#define GSR_BWD_NV 9

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
	const int out_index = gsr_bfly_index(lane >> 3);  // which of v[0..7] this lane's 8-lane group ends up holding

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
			const float4 A = rec[0][j];   // x, y, conic a, conic b
			const float4 B = rec[1][j];   // conic c, opacity, r, g
			const float4 Cc = rec[2][j];  // b, position in the full range, slot
			const int contributor = __float_as_int(Cc.y);  // backward.cu:511-515
			const float dx = A.x - pfx;
			const float ax2 = __fmul_rn(__fmul_rn(A.z, dx), dx), bdx = __fmul_rn(A.w, dx);
			float v[GSR_BWD_NV];
#pragma unroll
			for (int i = 0; i < GSR_BWD_NV; i++) v[i] = 0.f;
			bool any = false;
#pragma unroll
			for (int k = 0; k < GSR_PIX_PER_LANE; k++) {
				const float dy = A.y - pfy[k];
				const float power = gsr_pair_power(ax2, bdx, B.x, dy);
				const float G = __expf(power);
				const float alpha = fminf(0.99f, B.y * G);
				const bool hit = contributor < last_contributor[k] && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
				if (__ballot(hit) == 0ull) continue;  // wave-uniform: none of the 64 pixels of this row group
				if (hit) {
					any = true;
					const float inv1ma = __builtin_amdgcn_rcpf(1.f - alpha);  // 1 ulp; exact IEEE division changed no parity figure
					T[k] = T[k] * inv1ma;
					const float dchannel_dcolor = alpha * T[k];
					// accum_rec and (c - accum_rec) cancel heavily when neighbouring colours are close, so
					// this recurrence and dot product keep the reference's exact operation order with no FMA
					// contraction (backward.cu:553-559); everything else may contract
					const float la = last_alpha[k], oml = __fsub_rn(1.f, la);
					ac0[k] = __fadd_rn(__fmul_rn(la, lc0[k]), __fmul_rn(oml, ac0[k]));
					ac1[k] = __fadd_rn(__fmul_rn(la, lc1[k]), __fmul_rn(oml, ac1[k]));
					ac2[k] = __fadd_rn(__fmul_rn(la, lc2[k]), __fmul_rn(oml, ac2[k]));
					lc0[k] = B.z; lc1[k] = B.w; lc2[k] = Cc.x;
					float dL_dalpha = __fadd_rn(__fadd_rn(__fmul_rn(__fsub_rn(B.z, ac0[k]), dp0[k]),
					                                      __fmul_rn(__fsub_rn(B.w, ac1[k]), dp1[k])),
					                            __fmul_rn(__fsub_rn(Cc.x, ac2[k]), dp2[k]));
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
				const float t8 = gsr_bfly8(v, lane);           // group g holds the total of v[gsr_bfly_index(g)]
				const float t9 = gsr_wave_sum_to_lane63(v[8]);  // lane 63 holds the total of v[8]
				const uint32_t slot = __float_as_uint(Cc.z);
				float* out = reinterpret_cast<float*>(slots + slot);
				if ((lane & 7) == 0) out[out_index] = t8;
				if (lane == 63) {
					out[8] = t9;
					slot_valid[slot] = 1;
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
}

void gsr_launch_render_backward(int W, int H, GsrImage img, const uint32_t* point_list, const GsrSplat* splat,
                                const float* bg, const float* dL_dpix, GsrGradSlot* slots, uint8_t* slot_valid,
                                hipStream_t s)
{
	const int gx = gsr_grid_x(W), gy = gsr_grid_y(H);
	const int ntiles = gx * gy;
	const int nwg = (ntiles + GSR_WAVES_PER_WG - 1) / GSR_WAVES_PER_WG;
	hipLaunchKernelGGL(gsr_render_backward_wave_kernel, dim3(nwg), dim3(64 * GSR_WAVES_PER_WG), 0, s, W, H, gx, ntiles,
	                   img.ranges, point_list, splat, bg, img.final_T, img.n_contrib, img.tile_max_contrib, dL_dpix,
	                   slots, slot_valid);
}
