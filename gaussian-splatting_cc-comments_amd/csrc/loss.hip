// loss.hip -- fused training loss next to the rasterizer path.
//
// Replaces, for a (C,H,W) rendered image and its ground truth, the stock-PyTorch sequence of
// train.py:126-128 with utils/loss_utils.py:16-63:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - SSIM(x, y)),   loss.backward()
// SSIM: 11x11 Gaussian window (sigma 1.5, fp32 taps as create_window builds them), zero padding,
// per channel, C1 = 0.01^2, C2 = 0.03^2, mean over C*H*W.  The reference runs 5 grouped conv2d
// forward and their autograd backward (~10 image-sized passes plus elementwise ops); here:
//   kernel A: per 64x32 tile (GSR_LOSS_TY), x and y with a 5-pixel halo go to LDS, the five window means
//             (x, y, x^2, y^2, xy) are built separably (register-tiled: 4 outputs per thread and
//             pass, 14 inputs held in VGPRs), SSIM and its three image-dependent
//             partial derivatives (d/dmu1, d/dE[x^2], d/dE[xy]) are evaluated per pixel, the partials
//             written out (12 B/pixel/channel) and the L1 / SSIM sums reduced per workgroup;
//   kernel B: the adjoint of a symmetric window is the same convolution: the three maps are
//             convolved (tile + halo through LDS) and combined with x, y and sign(x - y) into
//             dloss/dx -- exactly the dL/dpix the backward blend consumes;
//   kernel C: one workgroup folds the per-tile sums into {loss, l1, ssim} in a fixed order.
// All arithmetic fp32 like the reference; HBM-bound (about 60 B per pixel-channel).
#include "gsr_internal.h"

#define GSR_LOSS_TX 64
// tile height: 16 / 32 rows -> 0.151 / 0.137 ms for the 1980x1080 loss (fwd + bwd), 0.59 / 0.52 ms at 3840x2160: the taller tile
// re-reads less halo (42 / 32 rows instead of 26 / 16) and its 512-thread workgroups fill the SIMDs better
// 32-row tiles need ~79 KB (forward) / ~70 KB (backward) of static LDS per workgroup: more than the 64 KB of every gfx9
// part except gfx950 (160 KB per CU, two workgroups resident).  This library is built for gfx950 only (csrc/Makefile);
// -DGSR_LOSS_TY=16 is the variant that fits 64 KB, should the ARCH override of the Makefile ever be used.
#ifndef GSR_LOSS_TY
#define GSR_LOSS_TY 32
#endif
#define GSR_LOSS_THREADS (64 * (GSR_LOSS_TY / 4))  // one column and four rows of the tile per thread in the vertical pass
#define GSR_LOSS_R 5
#define GSR_LOSS_HX (GSR_LOSS_TX + 2 * GSR_LOSS_R)  // 74 columns with halo
#define GSR_LOSS_HXS 76                               // row stride in LDS: multiple of 4 floats so b128 reads stay aligned
#define GSR_LOSS_HY (GSR_LOSS_TY + 2 * GSR_LOSS_R)  // 42 rows with halo
#define GSR_LOSS_NQ (GSR_LOSS_TX / 4)                 // quads of outputs per row

struct GsrLossTaps { float g[11]; };

// 14 consecutive floats starting at a 16-byte aligned LDS address
__device__ __forceinline__ void gsr_lds_load14(const float* __restrict__ p, float (&v)[14])
{
	const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1], c = ((const float4*)p)[2];
	const float2 d = ((const float2*)p)[6];
	v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
	v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w; v[12] = d.x; v[13] = d.y;
}

// four adjacent outputs of the 11-tap window over 14 inputs held in registers (taps in ascending order)
__device__ __forceinline__ float4 gsr_conv4(const float (&v)[14], const GsrLossTaps& t)
{
	float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int k = 0; k < 11; k++) {
#pragma unroll
		for (int j = 0; j < 4; j++) o[j] += t.g[k] * v[j + k];
	}
	return make_float4(o[0], o[1], o[2], o[3]);
}

// zero-padded (F.conv2d padding=5) tile + halo of NP planes -> LDS [HY][HXS] each.  All global loads
// of a thread are issued before the first LDS store, so one round of memory latency covers them.
template <int NP>
__device__ __forceinline__ void gsr_loss_stage(const float* const (&plane)[NP], int H, int W, int x0, int y0, float* const (&dst)[NP])
{
	constexpr int N = GSR_LOSS_HY * GSR_LOSS_HX, IT = (N + GSR_LOSS_THREADS - 1) / GSR_LOSS_THREADS;
	float v[NP][IT];
#pragma unroll
	for (int k = 0; k < IT; k++) {
		const int i = threadIdx.x + k * GSR_LOSS_THREADS;
		const int r = i / GSR_LOSS_HX, q = i % GSR_LOSS_HX;
		const int gy = y0 + r, gx = x0 + q;
		const bool in = i < N && gy >= 0 && gy < H && gx >= 0 && gx < W;
		const size_t o = in ? (size_t)gy * W + gx : 0;
#pragma unroll
		for (int p = 0; p < NP; p++) { const float t = plane[p][o]; v[p][k] = in ? t : 0.f; }
	}
#pragma unroll
	for (int k = 0; k < IT; k++) {
		const int i = threadIdx.x + k * GSR_LOSS_THREADS;
		const int r = i / GSR_LOSS_HX, q = i % GSR_LOSS_HX;
		if (i < N) {
#pragma unroll
			for (int p = 0; p < NP; p++) dst[p][r * GSR_LOSS_HXS + q] = v[p][k];
		}
	}
}

// vertical pass: this thread's column lx, output rows ly0..ly0+3, from a row-filtered plane [HY][TX]
__device__ __forceinline__ float4 gsr_conv_col4(const float* __restrict__ tmp, int ly0, int lx, const GsrLossTaps& t)
{
	float v[14];
#pragma unroll
	for (int k = 0; k < 14; k++) v[k] = tmp[(ly0 + k) * GSR_LOSS_TX + lx];
	return gsr_conv4(v, t);
}

__global__ void __launch_bounds__(GSR_LOSS_THREADS) gsr_ssim_forward_kernel(int H, int W, const float* __restrict__ img,
                                                               const float* __restrict__ gt, GsrLossTaps taps,
                                                               float* __restrict__ dm, float* __restrict__ d11,
                                                               float* __restrict__ d12, float2* __restrict__ partial)
{
	__shared__ __attribute__((aligned(16))) float sx[GSR_LOSS_HY * GSR_LOSS_HXS], sy[GSR_LOSS_HY * GSR_LOSS_HXS];
	__shared__ __attribute__((aligned(16))) float tmp[5][GSR_LOSS_HY * GSR_LOSS_TX];
	__shared__ float2 wsum[GSR_LOSS_THREADS / 64];
	static_assert(sizeof(sx) + sizeof(sy) + sizeof(tmp) <= 80 * 1024, "two workgroups per CU must fit gfx950's 160 KB of LDS");
	const int c = blockIdx.z;
	const size_t plane = (size_t)H * W;
	const int x0 = blockIdx.x * GSR_LOSS_TX - GSR_LOSS_R, y0 = blockIdx.y * GSR_LOSS_TY - GSR_LOSS_R;
	{
		const float* const src[2] = {img + c * plane, gt + c * plane};
		float* const dst[2] = {sx, sy};
		gsr_loss_stage<2>(src, H, W, x0, y0, dst);
	}
	__syncthreads();
	// horizontal pass: one row, four adjacent outputs of all five window means per work item;
	// x^2, y^2 and xy are formed in registers, never stored
	for (int it = threadIdx.x; it < GSR_LOSS_HY * GSR_LOSS_NQ; it += GSR_LOSS_THREADS) {
		const int r = it / GSR_LOSS_NQ, c4 = (it % GSR_LOSS_NQ) * 4;
		float xv[14], yv[14], pv[14];
		gsr_lds_load14(sx + r * GSR_LOSS_HXS + c4, xv);
		gsr_lds_load14(sy + r * GSR_LOSS_HXS + c4, yv);
		const int o = r * GSR_LOSS_TX + c4;
		*(float4*)(tmp[0] + o) = gsr_conv4(xv, taps);
		*(float4*)(tmp[1] + o) = gsr_conv4(yv, taps);
#pragma unroll
		for (int k = 0; k < 14; k++) pv[k] = xv[k] * xv[k];
		*(float4*)(tmp[2] + o) = gsr_conv4(pv, taps);
#pragma unroll
		for (int k = 0; k < 14; k++) pv[k] = yv[k] * yv[k];
		*(float4*)(tmp[3] + o) = gsr_conv4(pv, taps);
#pragma unroll
		for (int k = 0; k < 14; k++) pv[k] = xv[k] * yv[k];
		*(float4*)(tmp[4] + o) = gsr_conv4(pv, taps);
	}
	__syncthreads();

	// vertical pass + SSIM: column lx, four rows per thread
	const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
	const int lx = threadIdx.x & 63, ly0 = (threadIdx.x >> 6) * 4;
	const int gx = blockIdx.x * GSR_LOSS_TX + lx;
	const float4 m1 = gsr_conv_col4(tmp[0], ly0, lx, taps), m2 = gsr_conv_col4(tmp[1], ly0, lx, taps);
	const float4 q11 = gsr_conv_col4(tmp[2], ly0, lx, taps), q22 = gsr_conv_col4(tmp[3], ly0, lx, taps);
	const float4 q12 = gsr_conv_col4(tmp[4], ly0, lx, taps);
	const float mu1v[4] = {m1.x, m1.y, m1.z, m1.w}, mu2v[4] = {m2.x, m2.y, m2.z, m2.w};
	const float e11v[4] = {q11.x, q11.y, q11.z, q11.w}, e22v[4] = {q22.x, q22.y, q22.z, q22.w}, e12v[4] = {q12.x, q12.y, q12.z, q12.w};
	float l1 = 0.f, ss = 0.f;
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const int ly = ly0 + j, gy = blockIdx.y * GSR_LOSS_TY + ly;
		if (gy >= H || gx >= W) continue;
		const float mu1 = mu1v[j], mu2 = mu2v[j], e11 = e11v[j], e22 = e22v[j], e12 = e12v[j];
		const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
		const float s11 = e11 - mu1_sq, s22 = e22 - mu2_sq, s12 = e12 - mu1_mu2;  // loss_utils.py:50-52
		const float A1 = 2.f * mu1_mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s11 + s22 + C2;
		const float inv = 1.f / (B1 * B2);
		const float S = (A1 * A2) * inv;                                            // loss_utils.py:57
		const size_t o = c * plane + (size_t)gy * W + gx;
		dm[o] = (2.f * mu2 * (A2 - A1)) * inv - S * (2.f * mu1 * (B2 - B1)) * inv;
		d11[o] = -S / B2;
		d12[o] = 2.f * A1 * inv;
		ss += S;
		const int li = (ly + GSR_LOSS_R) * GSR_LOSS_HXS + lx + GSR_LOSS_R;
		l1 += fabsf(sx[li] - sy[li]);
	}
	// fixed-order workgroup reduction -> one partial per tile (deterministic loss value)
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) { l1 += __shfl_down(l1, off, 64); ss += __shfl_down(ss, off, 64); }
	if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = make_float2(l1, ss);
	__syncthreads();
	if (threadIdx.x == 0) {
		float2 t = make_float2(0.f, 0.f);
#pragma unroll
		for (int w = 0; w < GSR_LOSS_THREADS / 64; w++) { t.x += wsum[w].x; t.y += wsum[w].y; }
		partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
	}
}

__global__ void __launch_bounds__(GSR_LOSS_THREADS) gsr_ssim_backward_kernel(int C, int H, int W, const float* __restrict__ img,
                                                                const float* __restrict__ gt, GsrLossTaps taps,
                                                                const float* __restrict__ dm, const float* __restrict__ d11,
                                                                const float* __restrict__ d12, float lambda,
                                                                float* __restrict__ dL_dimg)
{
	__shared__ __attribute__((aligned(16))) float s0[GSR_LOSS_HY * GSR_LOSS_HXS], s1[GSR_LOSS_HY * GSR_LOSS_HXS], s2[GSR_LOSS_HY * GSR_LOSS_HXS];
	__shared__ __attribute__((aligned(16))) float tmp[3][GSR_LOSS_HY * GSR_LOSS_TX];
	const int c = blockIdx.z;
	const size_t plane = (size_t)H * W;
	const int x0 = blockIdx.x * GSR_LOSS_TX - GSR_LOSS_R, y0 = blockIdx.y * GSR_LOSS_TY - GSR_LOSS_R;
	{
		const float* const src[3] = {dm + c * plane, d11 + c * plane, d12 + c * plane};
		float* const dst[3] = {s0, s1, s2};
		gsr_loss_stage<3>(src, H, W, x0, y0, dst);
	}
	// this thread's own pixels (column lx, rows ly0..ly0+3) are fetched now, used after both passes
	const int lx = threadIdx.x & 63, ly0 = (threadIdx.x >> 6) * 4;
	const int gx = blockIdx.x * GSR_LOSS_TX + lx;
	float xs[4], ys[4];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const int gy = blockIdx.y * GSR_LOSS_TY + ly0 + j;
		const bool in = gy < H && gx < W;
		const size_t o = in ? c * plane + (size_t)gy * W + gx : 0;
		xs[j] = img[o];
		ys[j] = gt[o];
	}
	__syncthreads();
	for (int it = threadIdx.x; it < GSR_LOSS_HY * GSR_LOSS_NQ; it += GSR_LOSS_THREADS) {
		const int r = it / GSR_LOSS_NQ, c4 = (it % GSR_LOSS_NQ) * 4;
		const int o = r * GSR_LOSS_TX + c4;
		float v[14];
		gsr_lds_load14(s0 + r * GSR_LOSS_HXS + c4, v);
		*(float4*)(tmp[0] + o) = gsr_conv4(v, taps);
		gsr_lds_load14(s1 + r * GSR_LOSS_HXS + c4, v);
		*(float4*)(tmp[1] + o) = gsr_conv4(v, taps);
		gsr_lds_load14(s2 + r * GSR_LOSS_HXS + c4, v);
		*(float4*)(tmp[2] + o) = gsr_conv4(v, taps);
	}
	__syncthreads();
	const float inv_total = 1.0f / ((float)C * (float)H * (float)W);
	const float4 a4 = gsr_conv_col4(tmp[0], ly0, lx, taps), b4 = gsr_conv_col4(tmp[1], ly0, lx, taps), c4v = gsr_conv_col4(tmp[2], ly0, lx, taps);
	const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w}, cv[4] = {c4v.x, c4v.y, c4v.z, c4v.w};
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const int gy = blockIdx.y * GSR_LOSS_TY + ly0 + j;
		if (gy >= H || gx >= W) continue;
		const size_t o = c * plane + (size_t)gy * W + gx;
		const float xv = xs[j], yv = ys[j];
		const float dssim = av[j] + 2.f * xv * bv[j] + yv * cv[j];
		const float sgn = (xv > yv) ? 1.f : ((xv < yv) ? -1.f : 0.f);
		dL_dimg[o] = ((1.f - lambda) * sgn - lambda * dssim) * inv_total;
	}
}

__global__ void __launch_bounds__(256) gsr_loss_finalize_kernel(const float2* __restrict__ partial, int n, int C, int H, int W,
                                                                float lambda, float* __restrict__ loss_out)
{
	__shared__ double sl[256], ssum[256];
	double l1 = 0, s = 0;
	for (int i = threadIdx.x; i < n; i += 256) { l1 += partial[i].x; s += partial[i].y; }
	sl[threadIdx.x] = l1; ssum[threadIdx.x] = s;
	__syncthreads();
	for (int off = 128; off > 0; off >>= 1) {
		if ((int)threadIdx.x < off) { sl[threadIdx.x] += sl[threadIdx.x + off]; ssum[threadIdx.x] += ssum[threadIdx.x + off]; }
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		const double total = (double)C * H * W;
		const double ml1 = sl[0] / total, mss = ssum[0] / total;
		loss_out[0] = (float)((1.0 - lambda) * ml1 + lambda * (1.0 - mss));
		loss_out[1] = (float)ml1;
		loss_out[2] = (float)mss;
	}
}

static GsrLossTaps gsr_loss_taps()
{
	// loss_utils.py:21-24: float32 tensor of exp(-(x-5)^2 / (2 sigma^2)), divided by its float32 sum
	GsrLossTaps t;
	float raw[11], sum = 0.f;
	for (int i = 0; i < 11; i++) raw[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
	for (int i = 0; i < 11; i++) sum += raw[i];
	for (int i = 0; i < 11; i++) t.g[i] = raw[i] / sum;
	return t;
}

size_t gsr_loss_scratch_layout(int C, int H, int W, size_t* maps_off, size_t* partial_off, int* ntiles)
{
	const size_t n = (size_t)C * H * W;
	const int gx = (W + GSR_LOSS_TX - 1) / GSR_LOSS_TX, gy = (H + GSR_LOSS_TY - 1) / GSR_LOSS_TY;
	*ntiles = gx * gy * C;
	*maps_off = 0;
	*partial_off = gsr_align_up(3 * n * sizeof(float));
	return *partial_off + gsr_align_up((size_t)(*ntiles) * sizeof(float2));
}

void gsr_launch_l1_ssim(int C, int H, int W, const float* img, const float* gt, float lambda, float* loss_out, float* dL_dimg,
                        void* scratch, hipStream_t s)
{
	size_t maps_off, partial_off;
	int ntiles;
	gsr_loss_scratch_layout(C, H, W, &maps_off, &partial_off, &ntiles);
	const size_t n = (size_t)C * H * W;
	float* dm = (float*)((char*)scratch + maps_off);
	float* d11 = dm + n;
	float* d12 = d11 + n;
	float2* partial = (float2*)((char*)scratch + partial_off);
	const GsrLossTaps taps = gsr_loss_taps();
	const dim3 grid((W + GSR_LOSS_TX - 1) / GSR_LOSS_TX, (H + GSR_LOSS_TY - 1) / GSR_LOSS_TY, C);
	{
		GsrProfScope p(s, "ssim_forward");
		hipLaunchKernelGGL(gsr_ssim_forward_kernel, grid, dim3(GSR_LOSS_THREADS), 0, s, H, W, img, gt, taps, dm, d11, d12, partial);
	}
	if (dL_dimg) {
		GsrProfScope p(s, "ssim_backward");
		hipLaunchKernelGGL(gsr_ssim_backward_kernel, grid, dim3(GSR_LOSS_THREADS), 0, s, C, H, W, img, gt, taps, dm, d11, d12, lambda, dL_dimg);
	}
	hipLaunchKernelGGL(gsr_loss_finalize_kernel, dim3(1), dim3(256), 0, s, partial, ntiles, C, H, W, lambda, loss_out);
}
