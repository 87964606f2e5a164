// sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs, hand-written for wave64.
//
// Replaces cub::DeviceRadixSort::SortPairs on the 64-bit (tile | depth) keys
// (cuda_rasterizer/rasterizer_impl.cu:357-374).  The reference sorts R (Gaussian, tile) instances
// on 32 + bit key bits (46 at 1080p: six 8-bit passes over 12-byte pairs).  Here the same total
// order is produced in two cheaper steps (binning.hip):
//   1. the P Gaussians are sorted by their depth bits (stable, ties keep ascending index),
//   2. instances are emitted in that order and sorted, stably, by the tile id alone.
// Within a tile the instances are then in (depth, index) order -- exactly the order of the
// reference's stable 64-bit sort -- but step 2 needs only ceil(bit / 8) passes over 8-byte pairs.
//
// One pass = three launches: per-block digit histogram -> per-digit scan over blocks -> scatter
// with an in-block stable ranking.  The ranking uses wave64 ballots ("which lanes hold my digit")
// instead of per-thread counters, so a lane's rank is one popcount.
#include "gsr_internal.h"

#define GSR_SORT_THREADS 256
#define GSR_SORT_RADIX 256
// Elements per thread: 16 (4096 per workgroup: long, well-coalesced digit runs) for instance-sized
// sorts; 4 for the Gaussian-sized depth sort, which would otherwise run on fewer workgroups than
// there are CUs with 16 serial ranking rounds each (measured 18 us per pass at P = 1M).
// Swept on MI355X for the instance sort: 8 / 16 / 32 items -> 0.168 / 0.166 / 0.216 ms at R = 9.2M (C3)
// and 1.52 / 1.43 / 1.72 ms at R = 71M (C5): 16 it is.
#define GSR_SORT_ITEMS_LARGE 16
#define GSR_SORT_ITEMS_SMALL 4
#define GSR_SORT_SMALL_N (4u << 20)

// element index of item `it` of this lane: each wave owns a contiguous run of 64 * ITEMS elements,
// visited 64 at a time, so element order == (wave, it, lane) order and loads are coalesced
template <int ITEMS>
__device__ __forceinline__ size_t gsr_sort_index(int block, int wave, int it, int lane)
{
	return (size_t)block * (GSR_SORT_THREADS * ITEMS) + (size_t)wave * (64 * ITEMS) + (size_t)it * 64 + lane;
}

template <int ITEMS>
__global__ void __launch_bounds__(GSR_SORT_THREADS) gsr_radix_hist_kernel(const uint32_t* __restrict__ keys, size_t n,
                                                                          int shift, uint32_t mask,
                                                                          uint32_t* __restrict__ table, int nblocks)
{
	__shared__ uint32_t hist[GSR_SORT_RADIX];
	hist[threadIdx.x] = 0;
	__syncthreads();
	// the histogram does not care which thread counts which element of the block's tile: 16-byte loads
	constexpr int TILE = GSR_SORT_THREADS * ITEMS;
	const size_t first = (size_t)blockIdx.x * TILE;
	if (first + TILE <= n) {
		const uint4* src = reinterpret_cast<const uint4*>(keys + first);  // tile starts are multiples of 1024 elements
		uint4 v[ITEMS / 4];
#pragma unroll
		for (int it = 0; it < ITEMS / 4; it++) v[it] = src[it * GSR_SORT_THREADS + threadIdx.x];
#pragma unroll
		for (int it = 0; it < ITEMS / 4; it++) {
			atomicAdd(&hist[(v[it].x >> shift) & mask], 1u);
			atomicAdd(&hist[(v[it].y >> shift) & mask], 1u);
			atomicAdd(&hist[(v[it].z >> shift) & mask], 1u);
			atomicAdd(&hist[(v[it].w >> shift) & mask], 1u);
		}
	} else {
		const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
		for (int it = 0; it < ITEMS; it++) {
			const size_t i = gsr_sort_index<ITEMS>(blockIdx.x, wave, it, lane);
			if (i < n) atomicAdd(&hist[(keys[i] >> shift) & mask], 1u);
		}
	}
	__syncthreads();
	table[(size_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];  // [digit][block]
}

// one workgroup per digit: exclusive scan of that digit's counts over the blocks, digit total out
__global__ void __launch_bounds__(GSR_SORT_THREADS) gsr_radix_rowscan_kernel(uint32_t* __restrict__ table, int nblocks,
                                                                             uint32_t* __restrict__ digit_total)
{
	__shared__ uint32_t wsum[GSR_SORT_THREADS / 64];
	uint32_t* row = table + (size_t)blockIdx.x * nblocks;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t carry = 0;
	for (int base = 0; base < nblocks; base += GSR_SORT_THREADS) {
		const int i = base + threadIdx.x;
		const uint32_t v = (i < nblocks) ? row[i] : 0u;
		uint32_t incl = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const uint32_t t = __shfl_up(incl, off, 64);
			if (lane >= off) incl += t;
		}
		if (lane == 63) wsum[wave] = incl;
		__syncthreads();
		uint32_t wbase = 0, tot = 0;
#pragma unroll
		for (int w = 0; w < GSR_SORT_THREADS / 64; w++) {
			const uint32_t s = wsum[w];
			if (w < wave) wbase += s;
			tot += s;
		}
		__syncthreads();
		if (i < nblocks) row[i] = carry + wbase + incl - v;
		carry += tot;
	}
	if (threadIdx.x == 0) digit_total[blockIdx.x] = carry;
}

template <int ITEMS>
__global__ void __launch_bounds__(GSR_SORT_THREADS) gsr_radix_scatter_kernel(
	const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
	uint32_t* __restrict__ vals_out, size_t n, int shift, int nbits, const uint32_t* __restrict__ table, int nblocks,
	const uint32_t* __restrict__ digit_total)
{
	__shared__ uint32_t wcount[GSR_SORT_THREADS / 64][GSR_SORT_RADIX];  // per-wave digit counts, then local bases
	__shared__ uint32_t gofs[GSR_SORT_RADIX];                           // global base of a digit minus its local base
	__shared__ uint32_t wsum[GSR_SORT_THREADS / 64];
	constexpr int TILE = GSR_SORT_THREADS * ITEMS;
	__shared__ uint32_t skey[TILE], sval[TILE];       // the block's elements in digit order
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t mask = (1u << nbits) - 1u;

#pragma unroll
	for (int w = 0; w < GSR_SORT_THREADS / 64; w++) wcount[w][threadIdx.x] = 0;

	// global exclusive base of digit d = (sum of totals of smaller digits) + this block's offset in d
	uint32_t my_gbase;
	{
		const uint32_t v = digit_total[threadIdx.x];
		uint32_t incl = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const uint32_t t = __shfl_up(incl, off, 64);
			if (lane >= off) incl += t;
		}
		if (lane == 63) wsum[wave] = incl;
		__syncthreads();
		uint32_t wb = 0;
#pragma unroll
		for (int w = 0; w < GSR_SORT_THREADS / 64; w++)
			if (w < wave) wb += wsum[w];
		my_gbase = wb + incl - v + table[(size_t)threadIdx.x * nblocks + blockIdx.x];
	}
	__syncthreads();

	uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
	uint32_t* mycount = wcount[wave];
#pragma unroll
	for (int it = 0; it < ITEMS; it++) {
		const size_t i = gsr_sort_index<ITEMS>(blockIdx.x, wave, it, lane);
		const bool valid = i < n;
		key[it] = valid ? keys_in[i] : 0u;
		val[it] = valid ? vals_in[i] : 0u;
		const uint32_t d = (key[it] >> shift) & mask;
		unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
		for (int b = 0; b < nbits; b++) {
			const bool bit = (d >> b) & 1u;
			const unsigned long long m = __builtin_amdgcn_ballot_w64(bit);
			peers &= bit ? m : ~m;
		}
		// peers = valid lanes of this wave holding digit d in this round; lowest one is the leader
		const int leader = __ffsll((long long)peers) - 1;
		uint32_t old = 0;
		if (valid && lane == leader) {
			old = mycount[d];
			mycount[d] = old + (uint32_t)__popcll(peers);
		}
		old = __shfl(old, leader < 0 ? 0 : leader, 64);
		rank[it] = old + (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
	}
	__syncthreads();
	// per digit (thread d): block total -> exclusive scan over digits = local base of the digit inside
	// the block; per-wave counts -> per-wave local bases
	{
		uint32_t c[GSR_SORT_THREADS / 64], tot = 0;
#pragma unroll
		for (int w = 0; w < GSR_SORT_THREADS / 64; w++) { c[w] = wcount[w][threadIdx.x]; tot += c[w]; }
		uint32_t incl = tot;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const uint32_t t = __shfl_up(incl, off, 64);
			if (lane >= off) incl += t;
		}
		if (lane == 63) wsum[wave] = incl;
		__syncthreads();
		uint32_t wb = 0;
#pragma unroll
		for (int w = 0; w < GSR_SORT_THREADS / 64; w++)
			if (w < wave) wb += wsum[w];
		uint32_t run = wb + incl - tot;  // local exclusive base of digit threadIdx.x
		gofs[threadIdx.x] = my_gbase - run;
#pragma unroll
		for (int w = 0; w < GSR_SORT_THREADS / 64; w++) { wcount[w][threadIdx.x] = run; run += c[w]; }
	}
	__syncthreads();
	// stage the elements in digit order, then write them out linearly: consecutive threads store to
	// consecutive addresses inside each digit's run (a direct scatter makes every lane of a store
	// instruction hit a different cache line)
#pragma unroll
	for (int it = 0; it < ITEMS; it++) {
		const size_t i = gsr_sort_index<ITEMS>(blockIdx.x, wave, it, lane);
		if (i < n) {
			const uint32_t lp = mycount[(key[it] >> shift) & mask] + rank[it];
			skey[lp] = key[it];
			sval[lp] = val[it];
		}
	}
	__syncthreads();
	const size_t first = (size_t)blockIdx.x * TILE;
	const uint32_t count = (uint32_t)((n - first < (size_t)TILE) ? (n - first) : (size_t)TILE);
	for (uint32_t i = threadIdx.x; i < count; i += GSR_SORT_THREADS) {
		const uint32_t k = skey[i];
		const uint32_t dst = gofs[(k >> shift) & mask] + i;
		keys_out[dst] = k;
		vals_out[dst] = sval[i];
	}
}

int gsr_radix_num_passes(int nbits_total) { return (nbits_total + 7) / 8; }

static inline int gsr_sort_items(size_t n) { return n <= GSR_SORT_SMALL_N ? GSR_SORT_ITEMS_SMALL : GSR_SORT_ITEMS_LARGE; }

size_t gsr_radix_table_bytes(size_t n)
{
	const size_t tile = (size_t)GSR_SORT_THREADS * gsr_sort_items(n);
	const size_t nblocks = (n + tile - 1) / tile;
	return gsr_align_up((nblocks * GSR_SORT_RADIX + GSR_SORT_RADIX) * sizeof(uint32_t));
}

template <int ITEMS>
static void gsr_radix_pass(const uint32_t* ki, const uint32_t* vi, uint32_t* ko, uint32_t* vo, size_t n, int shift, int bits,
                           uint32_t* table, uint32_t* digit_total, hipStream_t s)
{
	const int nblocks = (int)((n + (size_t)GSR_SORT_THREADS * ITEMS - 1) / ((size_t)GSR_SORT_THREADS * ITEMS));
	const uint32_t mask = (1u << bits) - 1u;
	hipLaunchKernelGGL(gsr_radix_hist_kernel<ITEMS>, dim3(nblocks), dim3(GSR_SORT_THREADS), 0, s, ki, n, shift, mask, table, nblocks);
	hipLaunchKernelGGL(gsr_radix_rowscan_kernel, dim3(GSR_SORT_RADIX), dim3(GSR_SORT_THREADS), 0, s, table, nblocks, digit_total);
	hipLaunchKernelGGL(gsr_radix_scatter_kernel<ITEMS>, dim3(nblocks), dim3(GSR_SORT_THREADS), 0, s, ki, vi, ko, vo, n, shift, bits,
	                   table, nblocks, digit_total);
}

// Sorts on key bits [0, nbits_total).  Ping-pongs between (k0,v0) and (k1,v1); the sorted result
// ends in (k0,v0) when the pass count is even and in (k1,v1) when odd -- returned through *in_first.
void gsr_radix_sort_u32(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, size_t n, int nbits_total, void* table_mem,
                        int* result_in_first, hipStream_t s)
{
	const int npass = gsr_radix_num_passes(nbits_total);
	*result_in_first = (npass % 2 == 0) ? 1 : 0;
	if (n == 0 || npass == 0) { *result_in_first = 1; return; }
	const int items = gsr_sort_items(n);
	const size_t tile = (size_t)GSR_SORT_THREADS * items;
	const size_t nblocks = (n + tile - 1) / tile;
	uint32_t* table = (uint32_t*)table_mem;
	uint32_t* digit_total = table + nblocks * GSR_SORT_RADIX;
	int shift = 0;
	for (int p = 0; p < npass; p++) {
		const int bits = (nbits_total - shift + (npass - p) - 1) / (npass - p);  // spread bits evenly over passes
		uint32_t *ki = (p % 2 == 0) ? k0 : k1, *vi = (p % 2 == 0) ? v0 : v1;
		uint32_t *ko = (p % 2 == 0) ? k1 : k0, *vo = (p % 2 == 0) ? v1 : v0;
		if (items == GSR_SORT_ITEMS_SMALL) gsr_radix_pass<GSR_SORT_ITEMS_SMALL>(ki, vi, ko, vo, n, shift, bits, table, digit_total, s);
		else gsr_radix_pass<GSR_SORT_ITEMS_LARGE>(ki, vi, ko, vo, n, shift, bits, table, digit_total, s);
		shift += bits;
	}
}
