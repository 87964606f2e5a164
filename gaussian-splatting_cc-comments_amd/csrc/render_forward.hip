// render_forward.hip -- per-tile front-to-back alpha compositing.
//
// Replaces renderCUDA<3> forward (cuda_rasterizer/forward.cu:331-485).  Per-pixel results follow
// forward.cu:425-468: skip power > 0, alpha = min(0.99, o*G), skip alpha < 1/255, stop (without
// blending this instance) when T(1-alpha) < 1e-4, otherwise C += rgb*alpha*T, T *= 1-alpha.
//
// Decomposition (render_common.h): one wave64 per tile, four pixels per lane, no workgroup
// barrier.  Per batch of 64 instances each lane gathers one 48-byte splat record (prefetched one
// batch ahead, ids two batches ahead), tests it against the tile (gsr_tile_band_mask) and the
// survivors are compacted into the wave's LDS slice; the inner loop broadcasts one record per
// iteration to all lanes.  Skipped instances
// are exactly those that blend into no pixel of the tile, so results are unchanged and
// n_contrib keeps the instance's position in the FULL range (forward.cu:426,466).
//
// The reference's per-thread `done` flag is one wave-uniform lane mask per band here (alive[k], two SGPRs): the
// comparisons of a band produce lane masks anyway, so updating and testing the flag is scalar arithmetic beside the
// vector instructions; a pixel's T is only ever advanced by a passing instance, so Tout is both the running and the
// reported transmittance.
#include "render_common.h"

#ifndef GSR_FWD_DONE_STRIDE
#define GSR_FWD_DONE_STRIDE 16
#endif

GSR_TILE_CLOCK_BUFFER(gsr_forward_tile_clock, gsr_debug_tile_clock_forward)

__global__ void __launch_bounds__(64 * GSR_WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(6, 6))) gsr_render_forward_wave_kernel(
	int W, int H, int gx, int nslots, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const GsrSplat* __restrict__ splat, float4* __restrict__ checkpoints, float* __restrict__ final_C, const float* __restrict__ bg, float* __restrict__ final_T,
	uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_max_contrib, const uint32_t* __restrict__ tile_order,
	float* __restrict__ out_color, int cull)
{
	__shared__ float4 s_rec[GSR_WAVES_PER_WG][3][64];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int slot_id = blockIdx.x * GSR_WAVES_PER_WG + wave;
	if (slot_id >= nslots) return;  // wave-uniform; no barriers below
	GSR_TILE_CLOCK_START();
	// longest ranges first (binning.hip gsr_tile_order_kernel); a heavy tile comes as four entries, one per 16x4-pixel band
	// (bits 28..30 = band + 1): this wave then blends that band alone -- its other pixels start out "done"
	const uint32_t entry = __builtin_amdgcn_readfirstlane(tile_order ? tile_order[slot_id] : (uint32_t)slot_id);  // a scalar: the per-band masks below stay in SGPRs
	if (entry == 0xFFFFFFFFu) return;  // unused split entry
	const int tile = (int)(entry & 0x0FFFFFFFu);
	const int only = (int)(entry >> 28);  // 0: the whole tile
	float4(*rec)[64] = s_rec[wave];

	const int tx = tile % gx, ty = tile / gx;
	const int px = tx * GSR_TILE_X + (lane & 15);
	const int py0 = ty * GSR_TILE_Y + (lane >> 4);
	float pfx = (float)px;
	asm volatile("" : "+v"(pfx));
	const float x0f = (float)(tx * GSR_TILE_X), y0f = (float)(ty * GSR_TILE_Y);

	const uint2 range = ranges[tile];
	const int n = (int)(range.y - range.x);
	const uint32_t* plist = point_list + range.x;
	// a heavy tile leaves depth checkpoints for the backward (gsr_internal.h GSR_CKPT_STRIDE): wave-uniform
	const bool heavy = n >= 2 * GSR_CKPT_STRIDE;
	bool walked_deep = false;   // a checkpoint was stored: only then can the backward cut this tile, and only then is final_C read

	float Tout[GSR_PIX_PER_LANE], C0[GSR_PIX_PER_LANE], C1[GSR_PIX_PER_LANE], C2[GSR_PIX_PER_LANE];
	float pfy[GSR_PIX_PER_LANE];
	uint32_t last[GSR_PIX_PER_LANE];
	unsigned long long alive[GSR_PIX_PER_LANE];  // lanes whose pixel of band k still blends: wave-uniform, lives in SGPRs
#pragma unroll
	for (int k = 0; k < GSR_PIX_PER_LANE; k++) {
		const int py = py0 + 4 * k;
		pfy[k] = (float)py;
		asm volatile("" : "+v"(pfy[k]));  // keep it in its register: the compiler would redo the conversion per instance
		Tout[k] = 1.0f;
		alive[k] = (only == 0 || only == k + 1) ? __builtin_amdgcn_ballot_w64(px < W && py < H) : 0ull;
		C0[k] = C1[k] = C2[k] = 0.f;
		last[k] = 0u;
	}

	GSR_TILE_STAT(unsigned long long st_staged = 0ull; unsigned long long st_bands = 0ull; unsigned long long st_hits = 0ull;)   // (diagnostic twin: wave-uniform work counters)
	// software pipeline: records one batch ahead, ids two batches ahead
	float4 ra = make_float4(0, 0, 0, 0), rb = ra, rc = ra;
	if (lane < n) {
		const uint32_t id = plist[lane];
		const float4* p = reinterpret_cast<const float4*>(splat + id);
		ra = p[0]; rb = p[1]; rc = p[2];
	}
	uint32_t id_next = (64 + lane < n) ? plist[64 + lane] : 0u;

	for (int base = 0; base < n; base += 64) {
		if ((alive[0] | alive[1] | alive[2] | alive[3]) == 0ull) break;
		if (heavy && base != 0 && (base & (GSR_CKPT_STRIDE - 1)) == 0) {
			// every pixel's state BEFORE the instance at position `base` (a finished pixel's state is its final one)
			walked_deep = true;
			float4* ck = checkpoints + ((size_t)(range.x + (uint32_t)base) / GSR_CKPT_STRIDE) * 256 + lane;
#pragma unroll
			for (int k = 0; k < GSR_PIX_PER_LANE; k++)
				if (only == 0 || only == k + 1) ck[64 * k] = make_float4(Tout[k], C0[k], C1[k], C2[k]);
		}
		const uint32_t bands = (base + lane < n) ? (cull ? gsr_tile_band_mask(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, x0f, y0f) : 0xFu) : 0u;
		const bool keep = bands != 0u;
		const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
		const int cnt = __popcll(mask);
		GSR_TILE_STAT(st_staged += (unsigned long long)cnt;)
		if (keep) {
			const int pos = gsr_mbcnt(mask);
			rec[0][pos] = make_float4(ra.x, ra.y, -0.5f * ra.z, ra.w);  // conic a, c pre-multiplied by -0.5 (exact)
			rec[1][pos] = make_float4(-0.5f * rb.x, rb.y, rc.x, rc.y);
			rec[2][pos] = make_float4(rc.z, __uint_as_float((uint32_t)(base + lane + 1)), __uint_as_float(bands), 0.f);
		}
		if (base + 64 + lane < n) {
			const float4* p = reinterpret_cast<const float4*>(splat + id_next);
			ra = p[0]; rb = p[1]; rc = p[2];
		}
		id_next = (base + 128 + lane < n) ? plist[base + 128 + lane] : 0u;
		__builtin_amdgcn_wave_barrier();

		// the all-done test runs once per GSR_FWD_DONE_STRIDE instances, not per instance: its eleven scalar instructions per
		// instance cost more than the few instances a finished tile now reads past its end (their bands are skipped by two
		// scalar instructions each) -- 0.232 -> 0.224 ms at C3
		for (int j0 = 0; j0 < cnt; j0 += GSR_FWD_DONE_STRIDE) {
		// bands that still have a pixel to blend, as of now: a band that finishes inside the stride is evaluated (to no
		// effect: `pass` needs alive) until the next refresh, which is cheaper than a second test per band and instance
		const uint32_t open_bands = __builtin_amdgcn_readfirstlane((alive[0] ? 1u : 0u) | (alive[1] ? 2u : 0u) | (alive[2] ? 4u : 0u) | (alive[3] ? 8u : 0u));
		if (open_bands == 0u) break;
		const int j1 = min(cnt, j0 + GSR_FWD_DONE_STRIDE);
		for (int j = j0; j < j1; j++) {
			const float4 A = rec[0][j];   // x, y, -0.5 conic a, conic b
			const float4 B = rec[1][j];   // -0.5 conic c, opacity, r, g
			const float4 Cc = rec[2][j];  // b, contributor, band mask
			const uint32_t contributor = __float_as_uint(Cc.y);
			uint32_t bands = __builtin_amdgcn_readfirstlane(__float_as_uint(Cc.z)) & open_bands;  // wave-uniform
			asm volatile("" : "+s"(bands));  // one AND per instance (the compiler would distribute it over the four bit tests)
			const float dx = A.x - pfx;
			const float ax2 = __fmul_rn(__fmul_rn(A.z, dx), dx), bdx = __fmul_rn(A.w, dx);
#pragma unroll
			for (int k = 0; k < GSR_PIX_PER_LANE; k++) {
				if (!(bands & (1u << k))) continue;  // this 16x4 band cannot be reached, or all its 64 pixels are done: scalar branch
				const float dy = A.y - pfy[k];
				const float power = gsr_pair_power_halved(ax2, bdx, B.x, dy);
				const float alpha = fminf(0.99f, B.y * __expf(power));
				// (1 - alpha) is rounded BEFORE the product, as in forward.cu:449: the backward pass divides
				// by that rounded value, so contracting this into fma(-T, alpha, T) would break the pairing
				const float test_T = __fmul_rn(Tout[k], __fsub_rn(1.0f, alpha));
				// the three comparisons are SGPR lane masks; what follows them is scalar arithmetic
				const unsigned long long live = __builtin_amdgcn_ballot_w64(!(power > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f));
				const unsigned long long passm = live & alive[k] & __builtin_amdgcn_ballot_w64(!(test_T < 0.0001f));
				GSR_TILE_STAT(st_bands++; st_hits += passm ? 1ull : 0ull;)
				alive[k] = (alive[k] & ~live) | passm;  // a live instance either passes or ends the pixel
				const bool pass = __builtin_amdgcn_inverse_ballot_w64(passm);
				const float w = pass ? alpha * Tout[k] : 0.0f;
				C0[k] = __builtin_fmaf(B.z, w, C0[k]);
				C1[k] = __builtin_fmaf(B.w, w, C1[k]);
				C2[k] = __builtin_fmaf(Cc.x, w, C2[k]);
				Tout[k] = pass ? test_T : Tout[k];
				last[k] = pass ? contributor : last[k];
			}
		}
		}
		__builtin_amdgcn_wave_barrier();
	}

	const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
	const size_t plane = (size_t)H * W;
	uint32_t m = 0u;
#pragma unroll
	for (int k = 0; k < GSR_PIX_PER_LANE; k++) {
		const int py = py0 + 4 * k;
		if (px < W && py < H && (only == 0 || only == k + 1)) {
			const uint32_t pix_id = (uint32_t)(W * py + px);
			final_T[pix_id] = Tout[k];
			n_contrib[pix_id] = last[k];
			out_color[pix_id] = C0[k] + Tout[k] * bg0;
			out_color[plane + pix_id] = C1[k] + Tout[k] * bg1;
			out_color[2 * plane + pix_id] = C2[k] + Tout[k] * bg2;
			if (walked_deep) { final_C[pix_id] = C0[k]; final_C[plane + pix_id] = C1[k]; final_C[2 * plane + pix_id] = C2[k]; }
			m = max(m, last[k]);
		}
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_down(m, off, 64));
	if (lane == 0) {
		if (only) atomicMax(&tile_max_contrib[tile], m);  // zeroed by the order kernel for the tiles it split
		else tile_max_contrib[tile] = m;
	}
	// one record per dispatch entry; work counters: instances staged after the band cull (20 bits), bands evaluated (22), bands in which
	// at least one pixel blended the instance (22)
	GSR_TILE_CLOCK_STOP(gsr_forward_tile_clock, slot_id, lane, (unsigned long long)entry,
	                    (st_staged & 0xFFFFFull) | ((st_bands & 0x3FFFFFull) << 20) | ((st_hits & 0x3FFFFFull) << 42));
}

void gsr_launch_render_forward(int W, int H, GsrImage img, const uint32_t* point_list, const GsrSplat* splat, float4* checkpoints,
                               const float* bg, float* out_color, bool ordered, bool cull, hipStream_t s)
{
	const int gx = gsr_grid_x(W), gy = gsr_grid_y(H);
	const int ntiles = gx * gy;
	const int nslots = ntiles + (ordered ? 3 * (int)gsr_tile_order_max_split(ntiles) : 0);
	const int nwg = (nslots + GSR_WAVES_PER_WG - 1) / GSR_WAVES_PER_WG;
	hipLaunchKernelGGL(gsr_render_forward_wave_kernel, dim3(nwg), dim3(64 * GSR_WAVES_PER_WG), 0, s, W, H, gx, nslots,
	                   img.ranges, point_list, splat, checkpoints, img.final_C, bg, img.final_T, img.n_contrib, img.tile_max_contrib,
	                   ordered ? img.tile_order : nullptr, out_color, cull ? 1 : 0);
}
