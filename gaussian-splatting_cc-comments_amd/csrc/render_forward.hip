// render_forward.hip -- per-tile front-to-back alpha compositing.
//
// Replaces renderCUDA<3> forward (cuda_rasterizer/forward.cu:331-485).  One 16x16 tile per
// 256-thread workgroup = 4 wave64; the tile's sorted instance range is staged through LDS in
// batches of 256 whole 48-byte splat records (colour included, so the inner loop never touches
// global memory).  Per-pixel arithmetic and thresholds are those of forward.cu:425-468.
#include "gsr_internal.h"

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

void gsr_launch_render_forward(int W, int H, GsrImage img, const uint32_t* point_list, const GsrSplat* splat,
                               const float* bg, float* out_color, hipStream_t s)
{
	const int gx = gsr_grid_x(W), gy = gsr_grid_y(H);
	hipLaunchKernelGGL(gsr_render_forward_kernel, dim3(gx * gy), dim3(GSR_TILE_PIX), 0, s, W, H, gx, img.ranges,
	                   point_list, splat, bg, img.final_T, img.n_contrib, img.tile_max_contrib, out_color);
}
