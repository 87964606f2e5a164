// binning.hip -- tile-count prefix sum, (tile|depth) key emission, sort, tile ranges.
//
// Replaces cub::DeviceScan::InclusiveSum (rasterizer_impl.cu:323), duplicateWithKeys (:78-126),
// cub::DeviceRadixSort::SortPairs on bits [0,32+bit) (:357-374) and identifyTileRanges (:133-159).
#include "gsr_internal.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

// ---- prefix sum over tiles_touched ---------------------------------------------------------
// Stage A (in the preprocess kernel): block_sums[b] = sum of tiles_touched over workgroup b.
// Stage B (here, one workgroup): block_sums -> exclusive prefix, grand total -> status[1].
// Stage C (gsr_finalize_offsets_kernel): in-workgroup inclusive scan + block prefix.
__device__ __forceinline__ uint32_t gsr_wave_incl_scan(uint32_t v)
{
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		uint32_t n = __shfl_up(v, off, 64);
		if (lane >= off) v += n;
	}
	return v;
}

// inclusive scan across a 1024- or 256-thread workgroup; returns inclusive value, *total = sum
template <int BLOCK>
__device__ __forceinline__ uint32_t gsr_block_incl_scan(uint32_t v, uint32_t* total, uint32_t* lds /* BLOCK/64+1 */)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t incl = gsr_wave_incl_scan(v);
	if (lane == 63) lds[wave] = incl;
	__syncthreads();
	uint32_t base = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < BLOCK / 64; w++) {
		uint32_t s = lds[w];
		if (w < wave) base += s;
		tot += s;
	}
	__syncthreads();
	*total = tot;
	return incl + base;
}

__global__ void __launch_bounds__(1024) gsr_scan_block_sums_kernel(uint32_t* block_sums, int nb, uint32_t* status)
{
	__shared__ uint32_t lds[1024 / 64 + 1];
	uint32_t carry = 0;
	for (int base = 0; base < nb; base += 1024) {
		const int i = base + threadIdx.x;
		uint32_t v = (i < nb) ? block_sums[i] : 0u;
		uint32_t total;
		uint32_t incl = gsr_block_incl_scan<1024>(v, &total, lds);
		if (i < nb) block_sums[i] = carry + incl - v;  // exclusive
		carry += total;
	}
	if (threadIdx.x == 0) status[1] = carry;
}

void gsr_launch_scan_block_sums(GsrGeometry g, int P, hipStream_t s)
{
	const int nb = (P + GSR_PREPROCESS_BLOCK - 1) / GSR_PREPROCESS_BLOCK;
	hipLaunchKernelGGL(gsr_scan_block_sums_kernel, dim3(1), dim3(1024), 0, s, g.block_sums, nb, g.status);
}

__global__ void __launch_bounds__(GSR_PREPROCESS_BLOCK) gsr_finalize_offsets_kernel(GsrGeometry g, int P)
{
	__shared__ uint32_t lds[GSR_PREPROCESS_BLOCK / 64 + 1];
	const int idx = blockIdx.x * GSR_PREPROCESS_BLOCK + threadIdx.x;
	const uint32_t t = (idx < P) ? g.tiles_touched[idx] : 0u;
	uint32_t total;
	const uint32_t incl = gsr_block_incl_scan<GSR_PREPROCESS_BLOCK>(t, &total, lds) + g.block_sums[blockIdx.x];
	if (idx < P) {
		g.point_offsets[idx] = incl;
		if (t) g.splat[idx].slot_base = incl - t;
	}
}

void gsr_launch_finalize_offsets(GsrGeometry g, int P, hipStream_t s)
{
	const int nb = (P + GSR_PREPROCESS_BLOCK - 1) / GSR_PREPROCESS_BLOCK;
	hipLaunchKernelGGL(gsr_finalize_offsets_kernel, dim3(nb), dim3(GSR_PREPROCESS_BLOCK), 0, s, g, P);
}

// ---- key emission --------------------------------------------------------------------------
// One Gaussian per lane for small rectangles; rectangles above GSR_DUP_COOP tiles are expanded
// by the whole wave (lanes stride over the rectangle, so stores are contiguous runs).
#define GSR_DUP_COOP 24

__device__ __forceinline__ void gsr_emit(uint64_t* keys, uint32_t* vals, uint32_t off, uint32_t k, uint32_t minx,
                                         uint32_t miny, uint32_t w, int gx, uint32_t dbits, uint32_t idx)
{
	const uint32_t y = miny + k / w, x = minx + k % w;  // y outer, x inner (rasterizer_impl.cu:107-118)
	uint64_t key = (uint64_t)(y * (uint32_t)gx + x);
	key <<= 32;
	key |= dbits;
	keys[off + k] = key;
	vals[off + k] = idx;
}

__global__ void __launch_bounds__(256) gsr_duplicate_keys_kernel(GsrGeometry g, int P, int gx, uint64_t* keys, uint32_t* vals)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	uint32_t tiles = 0, off = 0, minx = 0, miny = 0, w = 1, dbits = 0;
	if (idx < P) {
		tiles = g.tiles_touched[idx];
		if (tiles) {
			const GsrSplat& sp = g.splat[idx];
			off = sp.slot_base;
			minx = sp.rect_min & 0xffffu; miny = sp.rect_min >> 16;
			w = sp.rect_wh & 0xffffu;
			dbits = __float_as_uint(g.depths[idx]);
		}
	}
	if (tiles && tiles <= GSR_DUP_COOP)
		for (uint32_t k = 0; k < tiles; k++) gsr_emit(keys, vals, off, k, minx, miny, w, gx, dbits, (uint32_t)idx);

	unsigned long long big = __ballot(tiles > GSR_DUP_COOP);
	const int lane = threadIdx.x & 63;
	while (big) {
		const int src = __ffsll((long long)big) - 1;
		big &= big - 1;
		const uint32_t s_tiles = __shfl(tiles, src, 64), s_off = __shfl(off, src, 64);
		const uint32_t s_minx = __shfl(minx, src, 64), s_miny = __shfl(miny, src, 64), s_w = __shfl(w, src, 64);
		const uint32_t s_d = __shfl(dbits, src, 64);
		const uint32_t s_idx = (uint32_t)(idx - lane + src);
		for (uint32_t k = lane; k < s_tiles; k += 64) gsr_emit(keys, vals, s_off, k, s_minx, s_miny, s_w, gx, s_d, s_idx);
	}
}

void gsr_launch_duplicate_keys(GsrGeometry g, const int* radii, int P, int W, int H, GsrBinning b, hipStream_t s)
{
	(void)radii; (void)H;
	hipLaunchKernelGGL(gsr_duplicate_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, s, g, P, gsr_grid_x(W),
	                   b.keys_unsorted, b.point_list_unsorted);
}

// ---- sort ----------------------------------------------------------------------------------
size_t gsr_sort_temp_bytes(int64_t R)
{
	if (R <= 0) return 0;
	size_t bytes = 0;
	hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
	                                         (uint32_t*)nullptr, (size_t)R, 0u, 64u, (hipStream_t)0, false);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		return 0;
	}
	return bytes;
}

int gsr_sort_pairs(GsrBinning b, int64_t R, int end_bit, hipStream_t s)
{
	size_t bytes = b.sort_temp_bytes;
	hipError_t e = rocprim::radix_sort_pairs(b.sort_temp, bytes, b.keys_unsorted, b.keys, b.point_list_unsorted,
	                                         b.point_list, (size_t)R, 0u, (unsigned)end_bit, s, false);
	return gsr_check_hip(e, "rocprim::radix_sort_pairs");
}

// ---- tile ranges (rasterizer_impl.cu:133-159; ranges zeroed by the caller, :377) --------------
__global__ void __launch_bounds__(256) gsr_tile_ranges_kernel(const uint64_t* keys, int64_t L, uint2* ranges)
{
	const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (idx >= L) return;
	const uint32_t currtile = (uint32_t)(keys[idx] >> 32);
	if (idx == 0) ranges[currtile].x = 0;
	else {
		const uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
		if (currtile != prevtile) {
			ranges[prevtile].y = (uint32_t)idx;
			ranges[currtile].x = (uint32_t)idx;
		}
	}
	if (idx == L - 1) ranges[currtile].y = (uint32_t)L;
}

void gsr_launch_tile_ranges(const uint64_t* keys, int64_t R, uint2* ranges, int ntiles, hipStream_t s)
{
	(void)hipMemsetAsync(ranges, 0, (size_t)ntiles * sizeof(uint2), s);
	if (R > 0)
		hipLaunchKernelGGL(gsr_tile_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, keys, R, ranges);
}
