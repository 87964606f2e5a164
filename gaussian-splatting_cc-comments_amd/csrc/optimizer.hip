// optimizer.hip -- one-launch Adam over the six leaf-parameter groups.
//
// The reference trains with torch.optim.Adam(l, lr=0.0, eps=1e-15) over the groups xyz, f_dc, f_rest,
// opacity, scaling, rotation (scene/gaussian_model.py:243-252): betas (0.9, 0.999), no weight decay,
// no amsgrad.  Stock PyTorch runs that as ~8 elementwise passes per group; here every element of
// every group is updated by one kernel that reads (param, grad, exp_avg, exp_avg_sq) once and writes
// (param, exp_avg, exp_avg_sq) once: 28 B per float, the streaming minimum.  HBM-bound.
//
// Arithmetic, fp32, in the order of torch/optim/adam.py (_single_tensor_adam / _multi_tensor_adam):
//     exp_avg    = exp_avg + (1 - beta1) * (grad - exp_avg)                 (lerp_)
//     exp_avg_sq = exp_avg_sq * beta2 + (1 - beta2) * (grad * grad)         (mul_, addcmul_)
//     denom      = sqrt(exp_avg_sq) * (1 / sqrt(1 - beta2^step)) + eps      (sqrt, div by scalar, add_)
//     param      = param + (-lr / (1 - beta1^step)) * (exp_avg / denom)     (addcdiv_)
// The bias corrections are evaluated in double on the host like the Python scalars they replace.
//
// Optional (NOT the reference's semantics, off unless `radii` is passed): skip Gaussians that were
// not visible in the view(s) of this step (radii <= 0) -- their moments and parameters stay untouched,
// and their 28 B/float are never moved.
#include <math.h>

#include "gsr_internal.h"

#define GSR_ADAM_ELEMS_PER_BLOCK 1024  // 256 threads x float4

struct GsrAdamTable {
	int n;
	float* param[GSR_ADAM_MAX_GROUPS];
	const float* grad[GSR_ADAM_MAX_GROUPS];
	float* m[GSR_ADAM_MAX_GROUPS];
	float* v[GSR_ADAM_MAX_GROUPS];
	long long numel[GSR_ADAM_MAX_GROUPS];
	int row[GSR_ADAM_MAX_GROUPS];
	int vec_ok[GSR_ADAM_MAX_GROUPS];
	float neg_step_size[GSR_ADAM_MAX_GROUPS];
	float inv_bc2_sqrt[GSR_ADAM_MAX_GROUPS];
	unsigned first_block[GSR_ADAM_MAX_GROUPS + 1];
};

__device__ __forceinline__ void gsr_adam_one(float& p, float g, float& m, float& v, float w1, float beta2, float w2, float eps,
                                             float neg_step, float inv_bc2_sqrt)
{
	m = m + w1 * (g - m);
	v = v * beta2 + w2 * (g * g);
	const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
	p = p + neg_step * (m / denom);
}

__global__ void __launch_bounds__(256) gsr_adam_kernel(GsrAdamTable t, float w1, float beta2, float w2, float eps,
                                                       const int* __restrict__ radii)
{
	int gi = 0;
#pragma unroll
	for (int k = 1; k < GSR_ADAM_MAX_GROUPS; k++)
		if (k < t.n && blockIdx.x >= t.first_block[k]) gi = k;
	const long long n = t.numel[gi];
	const long long e0 = ((long long)(blockIdx.x - t.first_block[gi]) * 256 + threadIdx.x) * 4;
	if (e0 >= n) return;
	float* __restrict__ P = t.param[gi];
	const float* __restrict__ G = t.grad[gi];
	float* __restrict__ M = t.m[gi];
	float* __restrict__ V = t.v[gi];
	const float neg_step = t.neg_step_size[gi], ibc2 = t.inv_bc2_sqrt[gi];
	const int row = t.row[gi];
	if (e0 + 3 < n && t.vec_ok[gi]) {
		bool upd[4] = {true, true, true, true};
		if (radii) {
			bool any = false;
#pragma unroll
			for (int j = 0; j < 4; j++) { upd[j] = radii[(e0 + j) / row] > 0; any |= upd[j]; }
			if (!any) return;
		}
		float4 p = *reinterpret_cast<float4*>(P + e0), m = *reinterpret_cast<float4*>(M + e0), v = *reinterpret_cast<float4*>(V + e0);
		const float4 g = *reinterpret_cast<const float4*>(G + e0);
		if (upd[0]) gsr_adam_one(p.x, g.x, m.x, v.x, w1, beta2, w2, eps, neg_step, ibc2);
		if (upd[1]) gsr_adam_one(p.y, g.y, m.y, v.y, w1, beta2, w2, eps, neg_step, ibc2);
		if (upd[2]) gsr_adam_one(p.z, g.z, m.z, v.z, w1, beta2, w2, eps, neg_step, ibc2);
		if (upd[3]) gsr_adam_one(p.w, g.w, m.w, v.w, w1, beta2, w2, eps, neg_step, ibc2);
		*reinterpret_cast<float4*>(P + e0) = p;
		*reinterpret_cast<float4*>(M + e0) = m;
		*reinterpret_cast<float4*>(V + e0) = v;
	} else {
		for (int j = 0; j < 4; j++) {
			const long long e = e0 + j;
			if (e >= n) break;
			if (radii && !(radii[e / row] > 0)) continue;
			float p = P[e], m = M[e], v = V[e];
			gsr_adam_one(p, G[e], m, v, w1, beta2, w2, eps, neg_step, ibc2);
			P[e] = p; M[e] = m; V[e] = v;
		}
	}
}

int gsr_launch_adam(int ngroups, const gsr_adam_group* groups, double beta1, double beta2, double eps, const int* radii, hipStream_t s)
{
	GsrAdamTable t = {};
	t.n = ngroups;
	unsigned blocks = 0;
	for (int k = 0; k < ngroups; k++) {
		const gsr_adam_group& g = groups[k];
		t.param[k] = g.param; t.grad[k] = g.grad; t.m[k] = g.exp_avg; t.v[k] = g.exp_avg_sq;
		t.numel[k] = g.numel;
		t.row[k] = g.row > 0 ? g.row : 1;
		t.vec_ok[k] = ((((uintptr_t)g.param) | ((uintptr_t)g.grad) | ((uintptr_t)g.exp_avg) | ((uintptr_t)g.exp_avg_sq)) & 15u) == 0;
		// torch/optim/adam.py: bias_correction1 = 1 - beta1 ** step (Python floats = double)
		const double bc1 = 1.0 - pow(beta1, (double)g.step);
		const double bc2 = 1.0 - pow(beta2, (double)g.step);
		t.neg_step_size[k] = (float)(-(g.lr / bc1));
		t.inv_bc2_sqrt[k] = 1.0f / (float)sqrt(bc2);
		t.first_block[k] = blocks;
		blocks += (unsigned)((g.numel + GSR_ADAM_ELEMS_PER_BLOCK - 1) / GSR_ADAM_ELEMS_PER_BLOCK);
	}
	t.first_block[ngroups] = blocks;
	if (blocks == 0) return 0;
	// Python scalars are doubles: 1 - beta is formed in double, then narrowed like a kernel argument
	const float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2);
	hipLaunchKernelGGL(gsr_adam_kernel, dim3(blocks), dim3(256), 0, s, t, w1, (float)beta2, w2, (float)eps, radii);
	return 0;
}
