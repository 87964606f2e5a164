// loss.hip -- fused training loss next to the rasterizer path.
//
// Replaces, for a (C,H,W) rendered image and its ground truth, the stock-PyTorch sequence of
// train.py:126-128 with utils/loss_utils.py:16-63:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - SSIM(x, y)),   loss.backward()
// SSIM: 11x11 Gaussian window (sigma 1.5, fp32 taps as create_window builds them), zero padding,
// per channel, C1 = 0.01^2, C2 = 0.03^2, mean over C*H*W.  The reference runs 5 grouped conv2d
// forward and their autograd backward (~10 image-sized passes plus elementwise ops); here:
//   kernel A: per 32x16 tile, x and y with a 5-pixel halo go to LDS, the five window means
//             (x, y, x^2, y^2, xy) are built separably in LDS, SSIM and its three image-dependent
//             partial derivatives (d/dmu1, d/dE[x^2], d/dE[xy]) are evaluated per pixel, the partials
//             written out (12 B/pixel/channel) and the L1 / SSIM sums reduced per workgroup;
//   kernel B: the adjoint of a symmetric window is the same convolution: the three maps are
//             convolved (tile + halo through LDS) and combined with x, y and sign(x - y) into
//             dloss/dx -- exactly the dL/dpix the backward blend consumes;
//   kernel C: one workgroup folds the per-tile sums into {loss, l1, ssim} in a fixed order.
// All arithmetic fp32 like the reference; HBM-bound (about 60 B per pixel-channel).
#include "gsr_internal.h"

#define GSR_LOSS_TX 32
#define GSR_LOSS_TY 16
#define GSR_LOSS_R 5
#define GSR_LOSS_HX (GSR_LOSS_TX + 2 * GSR_LOSS_R)  // 42
#define GSR_LOSS_HY (GSR_LOSS_TY + 2 * GSR_LOSS_R)  // 26

struct GsrLossTaps { float g[11]; };

// horizontal then vertical 11-tap pass over an LDS tile with halo; `src` is [HY][HX], `tmp` is [HY][TX]
__device__ __forceinline__ void gsr_conv_rows(const float* __restrict__ src, float* __restrict__ tmp, const GsrLossTaps& t)
{
	for (int i = threadIdx.x; i < GSR_LOSS_HY * GSR_LOSS_TX; i += 256) {
		const int r = i / GSR_LOSS_TX, c = i % GSR_LOSS_TX;
		const float* p = src + r * GSR_LOSS_HX + c;
		float a = 0.f;
#pragma unroll
		for (int k = 0; k < 11; k++) a += t.g[k] * p[k];
		tmp[i] = a;
	}
}

__device__ __forceinline__ float gsr_conv_col(const float* __restrict__ tmp, int ly, int lx, const GsrLossTaps& t)
{
	const float* p = tmp + ly * GSR_LOSS_TX + lx;
	float a = 0.f;
#pragma unroll
	for (int k = 0; k < 11; k++) a += t.g[k] * p[k * GSR_LOSS_TX];
	return a;
}

__global__ void __launch_bounds__(256) gsr_ssim_forward_kernel(int H, int W, const float* __restrict__ img,
                                                               const float* __restrict__ gt, GsrLossTaps taps,
                                                               float* __restrict__ dm, float* __restrict__ d11,
                                                               float* __restrict__ d12, float2* __restrict__ partial)
{
	__shared__ float sx[GSR_LOSS_HY * GSR_LOSS_HX], sy[GSR_LOSS_HY * GSR_LOSS_HX], sq[GSR_LOSS_HY * GSR_LOSS_HX];
	__shared__ float tmp[5][GSR_LOSS_HY * GSR_LOSS_TX];
	__shared__ float2 wsum[4];
	const int c = blockIdx.z;
	const size_t plane = (size_t)H * W;
	const float* x = img + c * plane;
	const float* y = gt + c * plane;
	const int x0 = blockIdx.x * GSR_LOSS_TX - GSR_LOSS_R, y0 = blockIdx.y * GSR_LOSS_TY - GSR_LOSS_R;
	for (int i = threadIdx.x; i < GSR_LOSS_HY * GSR_LOSS_HX; i += 256) {
		const int r = i / GSR_LOSS_HX, q = i % GSR_LOSS_HX;
		const int gy = y0 + r, gx = x0 + q;
		const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;  // zero padding (F.conv2d padding=5)
		sx[i] = in ? x[(size_t)gy * W + gx] : 0.f;
		sy[i] = in ? y[(size_t)gy * W + gx] : 0.f;
	}
	__syncthreads();
	gsr_conv_rows(sx, tmp[0], taps);
	gsr_conv_rows(sy, tmp[1], taps);
	for (int i = threadIdx.x; i < GSR_LOSS_HY * GSR_LOSS_HX; i += 256) sq[i] = sx[i] * sx[i];
	__syncthreads();
	gsr_conv_rows(sq, tmp[2], taps);
	__syncthreads();
	for (int i = threadIdx.x; i < GSR_LOSS_HY * GSR_LOSS_HX; i += 256) sq[i] = sy[i] * sy[i];
	__syncthreads();
	gsr_conv_rows(sq, tmp[3], taps);
	__syncthreads();
	for (int i = threadIdx.x; i < GSR_LOSS_HY * GSR_LOSS_HX; i += 256) sq[i] = sx[i] * sy[i];
	__syncthreads();
	gsr_conv_rows(sq, tmp[4], taps);
	__syncthreads();

	const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
	float l1 = 0.f, ss = 0.f;
	for (int i = threadIdx.x; i < GSR_LOSS_TY * GSR_LOSS_TX; i += 256) {
		const int ly = i / GSR_LOSS_TX, lx = i % GSR_LOSS_TX;
		const int gy = blockIdx.y * GSR_LOSS_TY + ly, gx = blockIdx.x * GSR_LOSS_TX + lx;
		if (gy >= H || gx >= W) continue;
		const float mu1 = gsr_conv_col(tmp[0], ly, lx, taps), mu2 = gsr_conv_col(tmp[1], ly, lx, taps);
		const float e11 = gsr_conv_col(tmp[2], ly, lx, taps), e22 = gsr_conv_col(tmp[3], ly, lx, taps);
		const float e12 = gsr_conv_col(tmp[4], ly, lx, taps);
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
		const float xv = sx[(ly + GSR_LOSS_R) * GSR_LOSS_HX + lx + GSR_LOSS_R], yv = sy[(ly + GSR_LOSS_R) * GSR_LOSS_HX + lx + GSR_LOSS_R];
		l1 += fabsf(xv - yv);
	}
	// fixed-order workgroup reduction -> one partial per tile (deterministic loss value)
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) { l1 += __shfl_down(l1, off, 64); ss += __shfl_down(ss, off, 64); }
	if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = make_float2(l1, ss);
	__syncthreads();
	if (threadIdx.x == 0) {
		const float2 a = wsum[0], b = wsum[1], cc = wsum[2], d = wsum[3];
		partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = make_float2((a.x + b.x) + (cc.x + d.x), (a.y + b.y) + (cc.y + d.y));
	}
}

__global__ void __launch_bounds__(256) gsr_ssim_backward_kernel(int C, int H, int W, const float* __restrict__ img,
                                                                const float* __restrict__ gt, GsrLossTaps taps,
                                                                const float* __restrict__ dm, const float* __restrict__ d11,
                                                                const float* __restrict__ d12, float lambda,
                                                                float* __restrict__ dL_dimg)
{
	__shared__ float s0[GSR_LOSS_HY * GSR_LOSS_HX], s1[GSR_LOSS_HY * GSR_LOSS_HX], s2[GSR_LOSS_HY * GSR_LOSS_HX];
	__shared__ float tmp[3][GSR_LOSS_HY * GSR_LOSS_TX];
	const int c = blockIdx.z;
	const size_t plane = (size_t)H * W;
	const int x0 = blockIdx.x * GSR_LOSS_TX - GSR_LOSS_R, y0 = blockIdx.y * GSR_LOSS_TY - GSR_LOSS_R;
	for (int i = threadIdx.x; i < GSR_LOSS_HY * GSR_LOSS_HX; i += 256) {
		const int r = i / GSR_LOSS_HX, q = i % GSR_LOSS_HX;
		const int gy = y0 + r, gx = x0 + q;
		const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
		const size_t o = c * plane + (size_t)gy * W + gx;
		s0[i] = in ? dm[o] : 0.f;
		s1[i] = in ? d11[o] : 0.f;
		s2[i] = in ? d12[o] : 0.f;
	}
	__syncthreads();
	gsr_conv_rows(s0, tmp[0], taps);
	gsr_conv_rows(s1, tmp[1], taps);
	gsr_conv_rows(s2, tmp[2], taps);
	__syncthreads();
	const float inv_total = 1.0f / ((float)C * (float)H * (float)W);
	for (int i = threadIdx.x; i < GSR_LOSS_TY * GSR_LOSS_TX; i += 256) {
		const int ly = i / GSR_LOSS_TX, lx = i % GSR_LOSS_TX;
		const int gy = blockIdx.y * GSR_LOSS_TY + ly, gx = blockIdx.x * GSR_LOSS_TX + lx;
		if (gy >= H || gx >= W) continue;
		const size_t o = c * plane + (size_t)gy * W + gx;
		const float xv = img[o], yv = gt[o];
		const float dssim = gsr_conv_col(tmp[0], ly, lx, taps) + 2.f * xv * gsr_conv_col(tmp[1], ly, lx, taps) +
		                    yv * gsr_conv_col(tmp[2], ly, lx, taps);
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
		hipLaunchKernelGGL(gsr_ssim_forward_kernel, grid, dim3(256), 0, s, H, W, img, gt, taps, dm, d11, d12, partial);
	}
	if (dL_dimg) {
		GsrProfScope p(s, "ssim_backward");
		hipLaunchKernelGGL(gsr_ssim_backward_kernel, grid, dim3(256), 0, s, C, H, W, img, gt, taps, dm, d11, d12, lambda, dL_dimg);
	}
	hipLaunchKernelGGL(gsr_loss_finalize_kernel, dim3(1), dim3(256), 0, s, partial, ntiles, C, H, W, lambda, loss_out);
}
