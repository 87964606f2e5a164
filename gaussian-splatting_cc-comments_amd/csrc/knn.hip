// knn.hip -- mean squared distance to the 3 nearest other points (compiled with -ffp-contract=off).
//
// Replaces simple_knn._C.distCUDA2 (submodules/simple-knn/spatial.cu:14-25 -> SimpleKNN::knn,
// simple_knn.cu:175-220), which initialises the Gaussian scales from the input cloud
// (scene/gaussian_model.py:215).  What the reference computes is EXACT: its Morton ordering and
// 1024-point boxes only prune (a box is skipped when it lies farther than the 3rd-best distance found
// among the +-3 neighbours in Morton order, simple_knn.cu:150-168), so any exact search returns the
// same three squared distances and the same (d0 + d1 + d2) / 3.0f.  The point itself is excluded by
// position, not by value: coincident points count with distance 0.
//
// MI355X design (the reference gathers `points[indices[i]]` from HBM for every candidate of every
// thread):
//   * points are gathered ONCE into Morton order as float4 {x, y, z, original index};
//   * two levels of bounding boxes over the sorted array: 256-point boxes and 4096-point super boxes;
//   * one wave = 64 consecutive sorted queries (spatially coherent).  Box pruning is done on the wave
//     level with the wave's query bounds and its largest rejection radius -- uniform control flow,
//     operands in SGPRs -- then per lane; the candidates of an accepted box are read with scalar loads
//     (uniform address: one s_load serves 64 lanes), so the scan is pure VALU work with no LDS and no
//     barrier;
//   * the wave's radius shrinks as the scan proceeds (own box first), pruning most boxes.
#include <float.h>

#include "gsr_internal.h"

#define GSR_KNN_BOX 256
#define GSR_KNN_SUPER 16  // boxes per super box

struct GsrKnnBox { float mnx, mny, mnz, mxx, mxy, mxz; };

__device__ __forceinline__ float gsr_wave_min(float v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
	return v;
}
__device__ __forceinline__ float gsr_wave_max(float v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
	return v;
}

// ---- bounds of the cloud (simple_knn.cu:182-190: both reductions start from {0,0,0}) ------------------
__global__ void __launch_bounds__(256) gsr_knn_bounds_partial_kernel(int P, const float* __restrict__ pts, GsrKnnBox* __restrict__ partial)
{
	GsrKnnBox me = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
		const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
		me.mnx = fminf(me.mnx, x); me.mny = fminf(me.mny, y); me.mnz = fminf(me.mnz, z);
		me.mxx = fmaxf(me.mxx, x); me.mxy = fmaxf(me.mxy, y); me.mxz = fmaxf(me.mxz, z);
	}
	__shared__ GsrKnnBox w[4];
	me.mnx = gsr_wave_min(me.mnx); me.mny = gsr_wave_min(me.mny); me.mnz = gsr_wave_min(me.mnz);
	me.mxx = gsr_wave_max(me.mxx); me.mxy = gsr_wave_max(me.mxy); me.mxz = gsr_wave_max(me.mxz);
	if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = me;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int k = 1; k < 4; k++) {
			me.mnx = fminf(me.mnx, w[k].mnx); me.mny = fminf(me.mny, w[k].mny); me.mnz = fminf(me.mnz, w[k].mnz);
			me.mxx = fmaxf(me.mxx, w[k].mxx); me.mxy = fmaxf(me.mxy, w[k].mxy); me.mxz = fmaxf(me.mxz, w[k].mxz);
		}
		partial[blockIdx.x] = me;
	}
}

__global__ void __launch_bounds__(64) gsr_knn_bounds_final_kernel(int n, const GsrKnnBox* __restrict__ partial, GsrKnnBox* __restrict__ bounds)
{
	GsrKnnBox me = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	for (int i = threadIdx.x; i < n; i += 64) {
		const GsrKnnBox o = partial[i];
		me.mnx = fminf(me.mnx, o.mnx); me.mny = fminf(me.mny, o.mny); me.mnz = fminf(me.mnz, o.mnz);
		me.mxx = fmaxf(me.mxx, o.mxx); me.mxy = fmaxf(me.mxy, o.mxy); me.mxz = fmaxf(me.mxz, o.mxz);
	}
	me.mnx = gsr_wave_min(me.mnx); me.mny = gsr_wave_min(me.mny); me.mnz = gsr_wave_min(me.mnz);
	me.mxx = gsr_wave_max(me.mxx); me.mxy = gsr_wave_max(me.mxy); me.mxz = gsr_wave_max(me.mxz);
	if (threadIdx.x == 0) *bounds = me;
}

// ---- 30-bit Morton code of the position inside the bounds (simple_knn.cu:45-71) ------------------------
__device__ __forceinline__ uint32_t gsr_spread10(uint32_t x)  // bit i of the low 10 bits -> bit 3i
{
	x &= 0x3FFu;
	x = (x | (x << 16)) & 0x030000FFu;
	x = (x | (x << 8)) & 0x0300F00Fu;
	x = (x | (x << 4)) & 0x030C30C3u;
	x = (x | (x << 2)) & 0x09249249u;
	return x;
}
__device__ __forceinline__ uint32_t gsr_grid10(float v, float lo, float hi)
{
	const float t = ((v - lo) / (hi - lo)) * 1023.0f;   // 0/0 = NaN for a degenerate axis: any cell is fine (ordering only prunes)
	return (t >= 0.f) ? (uint32_t)fminf(t, 1023.0f) : 0u;
}
__global__ void __launch_bounds__(256) gsr_knn_morton_kernel(int P, const float* __restrict__ pts, const GsrKnnBox* __restrict__ bounds,
                                                             uint32_t* __restrict__ codes, uint32_t* __restrict__ idx)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= P) return;
	const GsrKnnBox b = *bounds;
	const uint32_t x = gsr_spread10(gsr_grid10(pts[3 * (size_t)i], b.mnx, b.mxx));
	const uint32_t y = gsr_spread10(gsr_grid10(pts[3 * (size_t)i + 1], b.mny, b.mxy));
	const uint32_t z = gsr_spread10(gsr_grid10(pts[3 * (size_t)i + 2], b.mnz, b.mxz));
	codes[i] = x | (y << 1) | (z << 2);
	idx[i] = (uint32_t)i;
}

// ---- gather into Morton order + per-box bounds (one workgroup = one box) --------------------------------
__global__ void __launch_bounds__(GSR_KNN_BOX) gsr_knn_gather_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ idx,
                                                                     float4* __restrict__ sorted, GsrKnnBox* __restrict__ boxes)
{
	const int i = blockIdx.x * GSR_KNN_BOX + threadIdx.x;
	GsrKnnBox me = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
	if (i < P) {
		const uint32_t src = idx[i];
		const float x = pts[3 * (size_t)src], y = pts[3 * (size_t)src + 1], z = pts[3 * (size_t)src + 2];
		sorted[i] = make_float4(x, y, z, __uint_as_float(src));
		me.mnx = me.mxx = x; me.mny = me.mxy = y; me.mnz = me.mxz = z;
	}
	__shared__ GsrKnnBox w[GSR_KNN_BOX / 64];
	me.mnx = gsr_wave_min(me.mnx); me.mny = gsr_wave_min(me.mny); me.mnz = gsr_wave_min(me.mnz);
	me.mxx = gsr_wave_max(me.mxx); me.mxy = gsr_wave_max(me.mxy); me.mxz = gsr_wave_max(me.mxz);
	if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = me;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int k = 1; k < GSR_KNN_BOX / 64; k++) {
			me.mnx = fminf(me.mnx, w[k].mnx); me.mny = fminf(me.mny, w[k].mny); me.mnz = fminf(me.mnz, w[k].mnz);
			me.mxx = fmaxf(me.mxx, w[k].mxx); me.mxy = fmaxf(me.mxy, w[k].mxy); me.mxz = fmaxf(me.mxz, w[k].mxz);
		}
		boxes[blockIdx.x] = me;
	}
}

__global__ void __launch_bounds__(256) gsr_knn_super_kernel(int nb, int ns, const GsrKnnBox* __restrict__ boxes, GsrKnnBox* __restrict__ supers)
{
	const int s = blockIdx.x * 256 + threadIdx.x;
	if (s >= ns) return;
	GsrKnnBox me = boxes[s * GSR_KNN_SUPER];
	for (int b = s * GSR_KNN_SUPER + 1; b < min(nb, (s + 1) * GSR_KNN_SUPER); b++) {
		const GsrKnnBox o = boxes[b];
		me.mnx = fminf(me.mnx, o.mnx); me.mny = fminf(me.mny, o.mny); me.mnz = fminf(me.mnz, o.mnz);
		me.mxx = fmaxf(me.mxx, o.mxx); me.mxy = fmaxf(me.mxy, o.mxy); me.mxz = fmaxf(me.mxz, o.mxz);
	}
	supers[s] = me;
}

// ---- the search ---------------------------------------------------------------------------------------------
// squared distance from p to the box (0 inside), simple_knn.cu:116-126
__device__ __forceinline__ float gsr_point_box_dist2(const GsrKnnBox& b, float x, float y, float z)
{
	float dx = 0.f, dy = 0.f, dz = 0.f;
	if (x < b.mnx || x > b.mxx) dx = fminf(fabsf(x - b.mnx), fabsf(x - b.mxx));
	if (y < b.mny || y > b.mxy) dy = fminf(fabsf(y - b.mny), fabsf(y - b.mxy));
	if (z < b.mnz || z > b.mxz) dz = fminf(fabsf(z - b.mnz), fabsf(z - b.mxz));
	return dx * dx + dy * dy + dz * dz;
}
// lower bound of the squared distance between any point of q and any point of b; never above the fp32
// point-box distance of a point inside q (same expression shape, monotone rounding)
__device__ __forceinline__ float gsr_box_box_dist2(const GsrKnnBox& b, const GsrKnnBox& q)
{
	const float dx = fmaxf(0.f, fmaxf(b.mnx - q.mxx, q.mnx - b.mxx));
	const float dy = fmaxf(0.f, fmaxf(b.mny - q.mxy, q.mny - b.mxy));
	const float dz = fmaxf(0.f, fmaxf(b.mnz - q.mxz, q.mnz - b.mxz));
	return dx * dx + dy * dy + dz * dz;
}
// updateKBest<3>, simple_knn.cu:128-142, as a branch-free insertion
__device__ __forceinline__ void gsr_knn_insert(float4 me, float4 c, float& k0, float& k1, float& k2)
{
	const float dx = c.x - me.x, dy = c.y - me.y, dz = c.z - me.z;
	float d = dx * dx + dy * dy + dz * dz;
	float t = fmaxf(k0, d); k0 = fminf(k0, d); d = t;
	t = fmaxf(k1, d); k1 = fminf(k1, d); d = t;
	k2 = fminf(k2, d);
}

__device__ __forceinline__ void gsr_knn_scan_box(int P, int b, int self, float4 me, const float4* __restrict__ sorted, float& k0,
                                                 float& k1, float& k2)
{
	const int lo = b * GSR_KNN_BOX, hi = min(P, lo + GSR_KNN_BOX);
	for (int j = lo; j < hi; j++) {   // uniform index: candidates arrive through scalar loads
		const float4 c = sorted[j];
		if (j != self) gsr_knn_insert(me, c, k0, k1, k2);
	}
}

__global__ void __launch_bounds__(256) gsr_knn_search_kernel(int P, const float4* __restrict__ sorted, const GsrKnnBox* __restrict__ boxes,
                                                             const GsrKnnBox* __restrict__ supers, int nb, int ns, float* __restrict__ out)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	const bool active = i < P;
	const float4 me = active ? sorted[i] : make_float4(0.f, 0.f, 0.f, 0.f);
	float k0 = FLT_MAX, k1 = FLT_MAX, k2 = FLT_MAX;
	if (active)  // the +-3 neighbours in Morton order give the rejection radius (simple_knn.cu:152-158)
		for (int j = max(0, i - 3); j <= min(P - 1, i + 3); j++)
			if (j != i) gsr_knn_insert(me, sorted[j], k0, k1, k2);
	const float reject = k2;
	k0 = k1 = k2 = FLT_MAX;

	// the wave's query bounds
	GsrKnnBox q;
	q.mnx = gsr_wave_min(active ? me.x : FLT_MAX); q.mny = gsr_wave_min(active ? me.y : FLT_MAX); q.mnz = gsr_wave_min(active ? me.z : FLT_MAX);
	q.mxx = gsr_wave_max(active ? me.x : -FLT_MAX); q.mxy = gsr_wave_max(active ? me.y : -FLT_MAX); q.mxz = gsr_wave_max(active ? me.z : -FLT_MAX);
	const int own = blockIdx.x;  // GSR_KNN_BOX == workgroup size: the wave's own box
	if (active) gsr_knn_scan_box(P, own, i, me, sorted, k0, k1, k2);
	float wave_r = gsr_wave_max(active ? fminf(reject, k2) : -1.f);  // largest radius any lane still accepts

	for (int s = 0; s < ns; s++) {
		const GsrKnnBox S = supers[s];
		if (gsr_box_box_dist2(S, q) > wave_r) continue;                        // wave-uniform
		for (int b = s * GSR_KNN_SUPER; b < min(nb, (s + 1) * GSR_KNN_SUPER); b++) {
			if (b == own) continue;
			const GsrKnnBox B = boxes[b];
			if (gsr_box_box_dist2(B, q) > wave_r) continue;                    // wave-uniform
			const float d = gsr_point_box_dist2(B, me.x, me.y, me.z);
			const bool need = active && !(d > reject || d > k2);               // simple_knn.cu:164
			if (need) gsr_knn_scan_box(P, b, i, me, sorted, k0, k1, k2);
			wave_r = gsr_wave_max(active ? fminf(reject, k2) : -1.f);
		}
	}
	if (active) out[__float_as_uint(me.w)] = (k0 + k1 + k2) / 3.0f;          // simple_knn.cu:170
}

// ---- host side ----------------------------------------------------------------------------------------------
#define GSR_KNN_PARTIALS 1024

struct GsrKnnScratch {
	GsrKnnBox* bounds;
	GsrKnnBox* partial;
	uint32_t *codes, *codes_alt, *idx, *idx_alt;
	float4* sorted;
	GsrKnnBox* boxes;
	GsrKnnBox* supers;
	void* table;
	size_t total;
};

static GsrKnnScratch gsr_knn_carve(void* base, int P)
{
	const int nb = (P + GSR_KNN_BOX - 1) / GSR_KNN_BOX, ns = (nb + GSR_KNN_SUPER - 1) / GSR_KNN_SUPER;
	size_t off = 0;
	auto take = [&](size_t bytes) { void* p = (char*)base + off; off += gsr_align_up(bytes); return p; };
	GsrKnnScratch s;
	s.bounds = (GsrKnnBox*)take(sizeof(GsrKnnBox));
	s.partial = (GsrKnnBox*)take(sizeof(GsrKnnBox) * GSR_KNN_PARTIALS);
	s.codes = (uint32_t*)take(4 * (size_t)P); s.codes_alt = (uint32_t*)take(4 * (size_t)P);
	s.idx = (uint32_t*)take(4 * (size_t)P); s.idx_alt = (uint32_t*)take(4 * (size_t)P);
	s.sorted = (float4*)take(16 * (size_t)P);
	s.boxes = (GsrKnnBox*)take(sizeof(GsrKnnBox) * (size_t)nb);
	s.supers = (GsrKnnBox*)take(sizeof(GsrKnnBox) * (size_t)ns);
	s.table = take(gsr_radix_table_bytes((size_t)P));
	s.total = off;
	return s;
}

size_t gsr_knn_scratch_size(int P) { return P > 0 ? gsr_knn_carve(nullptr, P).total : 0; }

void gsr_launch_knn(int P, const float* points, float* mean_dist2, void* scratch, hipStream_t s)
{
	GsrKnnScratch k = gsr_knn_carve(scratch, P);
	const int nb = (P + GSR_KNN_BOX - 1) / GSR_KNN_BOX, ns = (nb + GSR_KNN_SUPER - 1) / GSR_KNN_SUPER;
	const int np = min(GSR_KNN_PARTIALS, (P + 255) / 256);
	GsrProfScope p(s, "knn");
	hipLaunchKernelGGL(gsr_knn_bounds_partial_kernel, dim3(np), dim3(256), 0, s, P, points, k.partial);
	hipLaunchKernelGGL(gsr_knn_bounds_final_kernel, dim3(1), dim3(64), 0, s, np, k.partial, k.bounds);
	hipLaunchKernelGGL(gsr_knn_morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, k.bounds, k.codes, k.idx);
	int in_first = 1;
	gsr_radix_sort_u32(k.codes, k.idx, k.codes_alt, k.idx_alt, (size_t)P, 30, k.table, &in_first, 1, 4, s);
	const uint32_t* order = in_first ? k.idx : k.idx_alt;
	hipLaunchKernelGGL(gsr_knn_gather_kernel, dim3(nb), dim3(GSR_KNN_BOX), 0, s, P, points, order, k.sorted, k.boxes);
	hipLaunchKernelGGL(gsr_knn_super_kernel, dim3((ns + 255) / 256), dim3(256), 0, s, nb, ns, k.boxes, k.supers);
	hipLaunchKernelGGL(gsr_knn_search_kernel, dim3(nb), dim3(256), 0, s, P, k.sorted, k.boxes, k.supers, nb, ns, mean_dist2);
}
