// render_forward.hip -- per-tile front-to-back alpha compositing.
//
// Replaces renderCUDA<3> forward (cuda_rasterizer/forward.cu:331-485).  One 16x16 tile per
// 256-thread workgroup = 4 wave64; the tile's sorted instance range is staged through LDS in
// batches of 256 whole 48-byte splat records (colour included, so the inner loop never touches
// global memory).  Per-pixel arithmetic and thresholds are those of forward.cu:425-468.
#include <stdlib.h>

#include "render_common.h"

// ---- workgroup-per-tile kernel (the reference's decomposition; kept for A/B, GSR_RENDER_V0=1) --
__global__ void __launch_bounds__(GSR_TILE_PIX) gsr_render_forward_kernel(
	int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const GsrSplat* __restrict__ splat, const float* __restrict__ bg, float* __restrict__ final_T,
	uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_max_contrib, float* __restrict__ out_color)
{
	__shared__ float4 s_rec[3][GSR_TILE_PIX];  // [xy,ca,cb] [cc,op,r,g] [b,..]
	__shared__ uint32_t s_max[GSR_TILE_PIX / 64];

	const int tile = blockIdx.x;
	const int tx = tile % gx, ty = tile / gx;
	const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
	const int px = tx * GSR_TILE_X + lx, py = ty * GSR_TILE_Y + ly;
	const bool inside = px < W && py < H;
	const float pfx = (float)px, pfy = (float)py;

	const uint2 range = ranges[tile];
	const int n = (int)(range.y - range.x);
	const int rounds = (n + GSR_TILE_PIX - 1) / GSR_TILE_PIX;

	bool done = !inside;
	int toDo = n;
	float T = 1.0f;
	uint32_t contributor = 0, last_contributor = 0;
	float C0 = 0.f, C1 = 0.f, C2 = 0.f;

	for (int i = 0; i < rounds; i++, toDo -= GSR_TILE_PIX) {
		// block-wide early exit (forward.cu:414-416); only skips work, never changes a pixel
		if (__syncthreads_count(done) == GSR_TILE_PIX) break;
		const int progress = i * GSR_TILE_PIX + threadIdx.x;
		if (progress < n) {
			const uint32_t id = point_list[range.x + progress];
			const float4* rec = reinterpret_cast<const float4*>(splat + id);
			s_rec[0][threadIdx.x] = rec[0];
			s_rec[1][threadIdx.x] = rec[1];
			s_rec[2][threadIdx.x] = rec[2];
		}
		__syncthreads();
		const int cnt = min(GSR_TILE_PIX, toDo);
		for (int j = 0; !done && j < cnt; j++) {
			contributor++;
			const float4 r0 = s_rec[0][j];
			const float4 r1 = s_rec[1][j];
			const float dx = r0.x - pfx, dy = r0.y - pfy;
			const float power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;
			if (power > 0.0f) continue;
			const float alpha = fminf(0.99f, r1.y * __expf(power));
			if (alpha < 1.0f / 255.0f) continue;
			const float test_T = T * (1 - alpha);
			if (test_T < 0.0001f) {
				done = true;
				continue;
			}
			const float w = alpha * T;
			C0 += r1.z * w;
			C1 += r1.w * w;
			C2 += s_rec[2][j].x * w;
			T = test_T;
			last_contributor = contributor;
		}
	}

	if (inside) {
		const uint32_t pix_id = (uint32_t)(W * py + px);
		const size_t plane = (size_t)H * W;
		final_T[pix_id] = T;
		n_contrib[pix_id] = last_contributor;
		out_color[pix_id] = C0 + T * bg[0];
		out_color[plane + pix_id] = C1 + T * bg[1];
		out_color[2 * plane + pix_id] = C2 + T * bg[2];
	}

	// tile-wide max of n_contrib: lets the backward blend skip the never-reached tail of the range
	uint32_t m = inside ? last_contributor : 0u;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_down(m, off, 64));
	if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
	__syncthreads();
	if (threadIdx.x == 0) tile_max_contrib[tile] = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
}

// ---- wave-per-tile kernel (default) ------------------------------------------------------------
// One wave64 per tile, four pixels per lane, no workgroup barrier.  Per batch of 64 instances each
// lane gathers one 48-byte splat record (prefetched one batch ahead, ids two batches ahead), tests
// it against the tile with gsr_tile_may_hit() and the survivors are compacted into the wave's LDS
// slice; the inner loop then broadcasts one record per iteration to all lanes.  Skipped instances
// are exactly those that blend into no pixel of the tile, so results are unchanged;
// `last_contributor` keeps the instance's position in the FULL range (forward.cu:426,466).
__global__ void __launch_bounds__(64 * GSR_WAVES_PER_WG) gsr_render_forward_wave_kernel(
	int W, int H, int gx, int ntiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const GsrSplat* __restrict__ splat, const float* __restrict__ bg, float* __restrict__ final_T,
	uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_max_contrib, float* __restrict__ out_color)
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
	const int n = (int)(range.y - range.x);
	const uint32_t* plist = point_list + range.x;

	float T[GSR_PIX_PER_LANE], C0[GSR_PIX_PER_LANE], C1[GSR_PIX_PER_LANE], C2[GSR_PIX_PER_LANE], pfy[GSR_PIX_PER_LANE];
	uint32_t last[GSR_PIX_PER_LANE];
	bool done[GSR_PIX_PER_LANE], inside[GSR_PIX_PER_LANE];
#pragma unroll
	for (int k = 0; k < GSR_PIX_PER_LANE; k++) {
		const int py = py0 + 4 * k;
		inside[k] = px < W && py < H;
		done[k] = !inside[k];
		pfy[k] = (float)py;
		T[k] = 1.0f; C0[k] = C1[k] = C2[k] = 0.f; last[k] = 0u;
	}

	// software pipeline: records one batch ahead, ids two batches ahead
	float4 ra = make_float4(0, 0, 0, 0), rb = ra, rc = ra;
	if (lane < n) {
		const float4* p = reinterpret_cast<const float4*>(splat + plist[lane]);
		ra = p[0]; rb = p[1]; rc = p[2];
	}
	uint32_t id_next = (64 + lane < n) ? plist[64 + lane] : 0u;

	for (int base = 0; base < n; base += 64) {
		if (__ballot(!(done[0] && done[1] && done[2] && done[3])) == 0ull) break;
		const bool keep = (base + lane < n) && gsr_tile_may_hit(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, x0f, y0f);
		const unsigned long long mask = __ballot(keep);
		const int cnt = __popcll(mask);
		if (keep) {
			const int pos = gsr_mbcnt(mask);
			rec[0][pos] = ra;
			rec[1][pos] = rb;
			rec[2][pos] = make_float4(rc.x, __uint_as_float((uint32_t)(base + lane + 1)), 0.f, 0.f);
		}
		if (base + 64 + lane < n) {
			const float4* p = reinterpret_cast<const float4*>(splat + id_next);
			ra = p[0]; rb = p[1]; rc = p[2];
		}
		id_next = (base + 128 + lane < n) ? plist[base + 128 + lane] : 0u;
		__builtin_amdgcn_wave_barrier();

		for (int j = 0; j < cnt; j++) {
			const float4 A = rec[0][j];
			const float4 B = rec[1][j];
			const float4 Cc = rec[2][j];
			const uint32_t contributor = __float_as_uint(Cc.y);
			const float dx = A.x - pfx;
#pragma unroll
			for (int k = 0; k < GSR_PIX_PER_LANE; k++) {
				const float dy = A.y - pfy[k];
				const float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
				const float alpha = fminf(0.99f, B.y * __expf(power));
				const float test_T = T[k] * (1 - alpha);
				const bool live = !done[k] && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
				const bool stop = live && test_T < 0.0001f;
				if (stop) done[k] = true;
				if (live && !stop) {
					const float w = alpha * T[k];
					C0[k] += B.z * w;
					C1[k] += B.w * w;
					C2[k] += Cc.x * w;
					T[k] = test_T;
					last[k] = contributor;
				}
			}
			if (__ballot(!(done[0] && done[1] && done[2] && done[3])) == 0ull) break;
		}
		__builtin_amdgcn_wave_barrier();
	}

	const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
	const size_t plane = (size_t)H * W;
	uint32_t m = 0u;
#pragma unroll
	for (int k = 0; k < GSR_PIX_PER_LANE; k++) {
		if (inside[k]) {
			const uint32_t pix_id = (uint32_t)(W * (py0 + 4 * k) + px);
			final_T[pix_id] = T[k];
			n_contrib[pix_id] = last[k];
			out_color[pix_id] = C0[k] + T[k] * bg0;
			out_color[plane + pix_id] = C1[k] + T[k] * bg1;
			out_color[2 * plane + pix_id] = C2[k] + T[k] * bg2;
			m = max(m, last[k]);
		}
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_down(m, off, 64));
	if (lane == 0) tile_max_contrib[tile] = m;
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

void gsr_launch_render_forward(int W, int H, GsrImage img, const uint32_t* point_list, const GsrSplat* splat,
                               const float* bg, float* out_color, hipStream_t s)
{
	const int gx = gsr_grid_x(W), gy = gsr_grid_y(H);
	const int ntiles = gx * gy;
	if (gsr_use_v0()) {
		hipLaunchKernelGGL(gsr_render_forward_kernel, dim3(ntiles), dim3(GSR_TILE_PIX), 0, s, W, H, gx, img.ranges,
		                   point_list, splat, bg, img.final_T, img.n_contrib, img.tile_max_contrib, out_color);
		return;
	}
	const int nwg = (ntiles + GSR_WAVES_PER_WG - 1) / GSR_WAVES_PER_WG;
	hipLaunchKernelGGL(gsr_render_forward_wave_kernel, dim3(nwg), dim3(64 * GSR_WAVES_PER_WG), 0, s, W, H, gx, ntiles,
	                   img.ranges, point_list, splat, bg, img.final_T, img.n_contrib, img.tile_max_contrib, out_color);
}
