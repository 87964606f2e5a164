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
// One pass = two launches: per-block digit histogram -> scatter with an in-block stable ranking.  The histogram
// kernel also adds each block's counts into per-CHUNK and per-SUPER-CHUNK sums (chunk = 64 consecutive blocks, super-chunk =
// 64 chunks; two atomics per digit per block), so the scatter kernel finds its block's offset inside a digit with
// <= #super-chunks + 63 + 63 loads per thread and no scan kernel runs in between (it was 6 us x 6 passes per step).
// The in-wave ranking asks "which lanes hold my digit" through LDS (ds_or_b64 into the digit's 64-bit word, see the scatter
// kernel), so a lane's rank among its peers is one mbcnt.
// Keys may be biased: with a `bias` pointer the sort orders key' = key - min (culled keys 0xFFFFFFFF -> range + 1),
// min / max taken from 64-way partial maxima left by the preprocess kernel; the depth sort then needs only the
// passes that cover the bits of max - min (3 instead of 4 for any view whose depth range is below 2^24 float steps).
#include "gsr_internal.h"
#include "gsr_radix_walk.h"
#include "gsr_depth_key.h"

#define GSR_SORT_THREADS 256
#define GSR_SORT_RADIX 256
// Elements per thread: 16 (4096 per workgroup: long, well-coalesced digit runs) for instance-sized
// sorts; 4 for the Gaussian-sized depth sort, which would otherwise run on fewer workgroups than
// there are CUs with 16 serial ranking rounds each (measured 18 us per pass at P = 1M).
// Swept on MI355X for the instance sort (with the ballot ranking of round 1): 8 / 16 / 32 items -> 0.168 / 0.166 / 0.216 ms
// at R = 9.2M (C3) and 1.52 / 1.43 / 1.72 ms at R = 71M (C5).
#ifndef GSR_SORT_ITEMS_LARGE
#define GSR_SORT_ITEMS_LARGE 16
#endif
#ifndef GSR_SORT_ITEMS_SMALL
#define GSR_SORT_ITEMS_SMALL 4
#endif
#define GSR_SORT_SMALL_N (4u << 20)
// ... and 32 for the largest sorts: re-swept with the LDS ranking, 12 / 16 / 24 / 32 items -> 0.124 / 0.126 / 0.136 / 0.150 ms at
// R = 9.2M (C3) but 1.30 / 1.17 / 1.12 / 1.07 ms at R = 71M (C5), where the longer digit runs of an 8 192-element block pay
#define GSR_SORT_ITEMS_HUGE 32
#ifndef GSR_SORT_HUGE_N
#define GSR_SORT_HUGE_N (32u << 20)
#endif

// element index of item `it` of this lane: each wave owns a contiguous run of 64 * ITEMS elements,
// visited 64 at a time, so element order == (wave, it, lane) order and loads are coalesced
template <int ITEMS>
__device__ __forceinline__ size_t gsr_sort_index(int block, int wave, int it, int lane)
{
	return (size_t)block * (GSR_SORT_THREADS * ITEMS) + (size_t)wave * (64 * ITEMS) + (size_t)it * 64 + lane;
}

// blocks_per_wg (a power of two <= GSR_SORT_CHUNK): a workgroup counts that many CONSECUTIVE blocks -- all in one chunk -- and adds
// to the chunk (super-chunk) sums once, from registers: every address of those rows otherwise takes 64 (4 096) adders, which
// serialise in the L2 (C5, the depth sort of 6 M Gaussians in 5 860 blocks: three-level tables).  1 for small sorts, whose
// histogram kernels are a few microseconds of latency.
// The Gaussians' first gradient slots become global: the preprocess kernel left slot_base[i] = exclusive scan of tiles_touched inside its
// workgroup of GSR_PREPROCESS_BLOCK Gaussians and the workgroups' totals in block_tiles; the TILE_ Gaussians of block blockIdx.x add the
// totals of every workgroup in front of theirs -- every workgroup here sums those up itself (<= P / 256 words, 16 loads in flight per
// thread), as cheap as the walk of the offset tables and without a scan kernel of its own.  Index order: see preprocess.hip.  Called by all
// GSR_SORT_THREADS threads of a workgroup that owns block blockIdx.x.
template <int TILE_>
__device__ __forceinline__ void gsr_slot_base_finish(uint32_t* __restrict__ slot_base, const uint32_t* __restrict__ block_tiles, uint32_t* __restrict__ status, size_t n)
{
	constexpr int SUB = TILE_ / GSR_PREPROCESS_BLOCK;
	static_assert(TILE_ % GSR_PREPROCESS_BLOCK == 0, "a block must hold whole preprocess workgroups");
	__shared__ uint32_t s_part[GSR_SORT_THREADS / 64], s_sub[SUB > 0 ? SUB : 1];
	const int nb_all = (int)((n + GSR_PREPROCESS_BLOCK - 1) / GSR_PREPROCESS_BLOCK);
	const int nb_before = (int)blockIdx.x * SUB;
	uint32_t part = 0;
	for (int j0 = 0; j0 < nb_before; j0 += 16 * GSR_SORT_THREADS) {
		uint32_t t[16];
#pragma unroll
		for (int u = 0; u < 16; u++) {
			const int j = j0 + u * GSR_SORT_THREADS + (int)threadIdx.x;
			t[u] = j < nb_before ? block_tiles[j] : 0u;
		}
#pragma unroll
		for (int u = 0; u < 16; u++) part += t[u];
	}
	const uint32_t own = ((int)threadIdx.x < SUB && nb_before + (int)threadIdx.x < nb_all) ? block_tiles[nb_before + threadIdx.x] : 0u;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) part += (uint32_t)__shfl_xor(part, off, 64);
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
	__syncthreads();
	if ((int)threadIdx.x < SUB) {   // (SUB <= 64: one wave) first slot of this block's sub-block `threadIdx.x`
		uint32_t before = 0;
#pragma unroll
		for (int w = 0; w < GSR_SORT_THREADS / 64; w++) before += s_part[w];
		uint32_t incl = own;
#pragma unroll
		for (int off = 1; off < SUB; off <<= 1) {
			const uint32_t t = __shfl_up(incl, off, 64);
			if ((int)threadIdx.x >= off) incl += t;
		}
		s_sub[threadIdx.x] = before + incl - own;
	}
	__syncthreads();
	const size_t first = (size_t)blockIdx.x * TILE_;
	for (int e0 = 4 * (int)threadIdx.x; e0 < TILE_; e0 += 4 * GSR_SORT_THREADS) {   // four consecutive Gaussians per thread: one sub-block
		const uint32_t add = s_sub[e0 / GSR_PREPROCESS_BLOCK];
		if (first + e0 + 3 < n) {
			uint4* q = reinterpret_cast<uint4*>(slot_base + first + e0);
			uint4 v = *q;
			v.x += add; v.y += add; v.z += add; v.w += add;
			*q = v;
		} else {
			for (int c = 0; c < 4; c++)
				if (first + e0 + c < n) slot_base[first + e0 + c] += add;
		}
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) status[3] = 1u;   // (gsr_internal.h: slot_base is final, index order)
}

template <int ITEMS, typename KeyT>
__global__ void __launch_bounds__(GSR_SORT_THREADS) gsr_radix_hist_kernel(const KeyT* __restrict__ keys, size_t n,
                                                                          int shift, uint32_t mask,
                                                                          uint32_t* __restrict__ table, int nblocks,
                                                                          uint32_t* __restrict__ chunk_sums, int nchunks,
                                                                          const uint32_t* __restrict__ bias, int blocks_per_wg,
                                                                          uint32_t* __restrict__ slot_base = nullptr,
                                                                          const uint32_t* __restrict__ block_tiles = nullptr,
                                                                          uint32_t* __restrict__ status = nullptr)
{
	__shared__ uint32_t hist[GSR_SORT_RADIX];
	__shared__ uint32_t s_bias[2];
	// First kernel of the bucket depth sort only (slot_base != NULL; one block per workgroup): the gradient slots' numbering becomes global
	if (slot_base) {
		if constexpr (sizeof(KeyT) == 4 && (GSR_SORT_THREADS * ITEMS) % GSR_PREPROCESS_BLOCK == 0) gsr_slot_base_finish<GSR_SORT_THREADS * ITEMS>(slot_base, block_tiles, status, n);
	}
	const GsrKeyBias kb = gsr_sort_bias(bias, s_bias);
	const bool biased = bias != nullptr;
	// the histogram does not care which thread counts which element of the block's tile: 16-byte loads (4 keys of 32 bits or
	// 8 keys of 16 bits each)
	constexpr int TILE = GSR_SORT_THREADS * ITEMS;
	constexpr int KPV = 16 / (int)sizeof(KeyT), NV = ITEMS / KPV;
	static_assert(NV >= 1 && NV * KPV == ITEMS, "a thread's keys must fill whole 16-byte loads");
	uint32_t acc = 0u;
	const int block0 = (int)blockIdx.x * blocks_per_wg;
	for (int block = block0; block < block0 + blocks_per_wg && block < nblocks; block++) {
		__syncthreads();  // (the previous block's counts have been read)
		hist[threadIdx.x] = 0;
		__syncthreads();
		const size_t first = (size_t)block * TILE;
		if (first + TILE <= n) {
			const uint4* src = reinterpret_cast<const uint4*>(keys + first);  // tile starts are multiples of 1024 elements
			uint4 v[NV];
#pragma unroll
			for (int it = 0; it < NV; it++) v[it] = src[it * GSR_SORT_THREADS + threadIdx.x];
#pragma unroll
			for (int it = 0; it < NV; it++) {
				const uint32_t w[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
				for (int c = 0; c < 4; c++) {
					if constexpr (sizeof(KeyT) == 4) {
						atomicAdd(&hist[gsr_sort_digit(w[c], kb, biased, shift, mask)], 1u);
					} else {
						atomicAdd(&hist[((w[c] & 0xffffu) >> shift) & mask], 1u);
						atomicAdd(&hist[((w[c] >> 16) >> shift) & mask], 1u);
					}
				}
			}
		} else {
			const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
			for (int it = 0; it < ITEMS; it++) {
				const size_t i = gsr_sort_index<ITEMS>(block, wave, it, lane);
				if (i < n) atomicAdd(&hist[gsr_sort_digit((uint32_t)keys[i], kb, biased, shift, mask)], 1u);
			}
		}
		__syncthreads();
		const uint32_t c = hist[threadIdx.x];
		table[(size_t)block * GSR_SORT_RADIX + threadIdx.x] = c;  // [block][digit]: a block's row is one coalesced kilobyte
		acc += c;
	}
	if (acc) {  // [chunk][digit] and, behind them, [super-chunk][digit]: zeroed beforehand
		atomicAdd(&chunk_sums[(size_t)(block0 / GSR_SORT_CHUNK) * GSR_SORT_RADIX + threadIdx.x], acc);
		// the third level only where the second alone would be long (> 64 chunks = 262 144 blocks' worth of elements / 64):
		// its rows take 4 096 / blocks_per_wg adders each (measured: +35 us per pass at 2 236 blocks when every block added to ONE row)
		if (nchunks > GSR_SORT_CHUNK)
			atomicAdd(&chunk_sums[(size_t)(nchunks + block0 / (GSR_SORT_CHUNK * GSR_SORT_CHUNK)) * GSR_SORT_RADIX + threadIdx.x], acc);
	}
}

// REC (first level of the bucket depth sort): every element also carries the 8 bytes rec_in[its index] along, and what is written
// beside the key is the 16-byte record {rec_in[i].x, rec_in[i].y, value, 0} into rec_out, not the value into vals_out.
template <int ITEMS, typename KeyT, bool REC = false>
__global__ void __launch_bounds__(GSR_SORT_THREADS) gsr_radix_scatter_kernel(
	const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, KeyT* __restrict__ keys_out,
	uint32_t* __restrict__ vals_out, size_t n, int shift, int nbits, const uint32_t* __restrict__ table, int nblocks,
	const uint32_t* __restrict__ chunk_sums, int nchunks, const uint32_t* __restrict__ bias,
	const uint2* __restrict__ rec_in = nullptr, uint4* __restrict__ rec_out = nullptr)
{
	__shared__ uint32_t wcount[GSR_SORT_THREADS / 64][GSR_SORT_RADIX];  // per-wave digit counts, then local bases
	__shared__ uint32_t gofs[GSR_SORT_RADIX];                           // global base of a digit minus its local base
	__shared__ uint32_t wsum[GSR_SORT_THREADS / 64];
	constexpr int TILE = GSR_SORT_THREADS * ITEMS;
	constexpr int KEY_WORDS = TILE * (int)sizeof(KeyT) / 4;
	__shared__ __attribute__((aligned(16))) uint32_t sstage[KEY_WORDS + TILE];  // the block's elements in digit order; before that, the offset walk's partial sums
	static_assert(KEY_WORDS + TILE >= 2 * (GSR_SORT_THREADS / 64) * GSR_SORT_RADIX, "the walk's partial sums and the ranking's peer masks must fit the staging area");
	KeyT* const skey = reinterpret_cast<KeyT*>(sstage);
	uint32_t* const sval = sstage + KEY_WORDS;
	__shared__ uint2 srec[REC ? TILE : 1];
	__shared__ uint32_t s_bias[2];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t mask = (1u << nbits) - 1u;
	const bool biased = bias != nullptr;

#pragma unroll
	for (int w = 0; w < GSR_SORT_THREADS / 64; w++) wcount[w][threadIdx.x] = 0;

	// the block's elements first: these loads need nothing from the table walk below and travel beside it
	uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
#pragma unroll
	for (int it = 0; it < ITEMS; it++) {
		const size_t i = gsr_sort_index<ITEMS>(blockIdx.x, wave, it, lane);
		const bool valid = i < n;
		key[it] = valid ? (uint32_t)keys_in[i] : 0u;
		val[it] = valid ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;   // vals_in == NULL: the identity (first level of the bucket depth sort)
	}
	uint2 rec[REC ? ITEMS : 1];
	if constexpr (REC) {
#pragma unroll
		for (int it = 0; it < ITEMS; it++) {
			const size_t i = gsr_sort_index<ITEMS>(blockIdx.x, wave, it, lane);
			rec[it] = i < n ? rec_in[i] : make_uint2(0u, 0u);
		}
	}

	// global exclusive base of digit d = (sum of totals of smaller digits) + this block's offset in d.  The rows that
	// enter it -- [super][digit] (three levels only), [chunk][digit], [block][digit] of this chunk's earlier blocks; <= nsuper
	// + 63 + 63 of them whatever the size of the sort -- are ONE list, dealt round-robin to the four waves: a lane reads
	// digits 4l .. 4l+3 of a row (one coalesced kilobyte per wave instruction), GSR_WALK_ROWS rows in flight per wave (32
	// rows per workgroup and round trip; the per-thread walk below does 16, and restarts for each of the three tables).
	// The waves' partial sums meet in LDS (in the staging area, which is not in use yet).  Instance-sized sorts only:
	// tile sort 1.33 -> 1.22 ms at R = 71M (C5), 0.147 -> 0.146 ms at R = 9.2M (C3); 16 rows in flight cost 150 VGPRs and
	// were slower.
	uint32_t my_gbase;
	{
		uint32_t v = 0, before = 0;
		if constexpr (ITEMS >= GSR_SORT_ITEMS_LARGE) {
			gsr_radix_walk_256(table, chunk_sums, nchunks, nchunks, (int)blockIdx.x, sstage, v, before);  // (gsr_radix_walk.h; the staging area is not in use yet)
		} else {
			// Gaussian-sized sorts (1024-element blocks): a block lives too briefly for the extra barrier of the
			// cooperative walk to pay (measured: 71 -> 81 us for the three depth passes at P = 1M); thread d walks word d of the
			// rows itself, 16 loads in flight
			const int my_chunk = blockIdx.x / GSR_SORT_CHUNK, my_super = my_chunk / GSR_SORT_CHUNK;
			const int nsuper = (nchunks + GSR_SORT_CHUNK - 1) / GSR_SORT_CHUNK;
			// [super][digit], [chunk][digit] and [block][digit] rows: thread d reads word d of every row, so each load instruction
			// of the workgroup fetches one contiguous kilobyte; 16 loads are in flight per group.  Total of the digit = all
			// super-chunks; in front of this block = earlier super-chunks + earlier chunks of its super-chunk + earlier blocks of
			// its chunk: <= nsuper + 63 + 63 rows whatever the size of the sort
			const uint32_t* super_sums = chunk_sums + (size_t)nchunks * GSR_SORT_RADIX;
			const bool three_level = nchunks > GSR_SORT_CHUNK;  // uniform
			if (three_level) {
				for (int c0 = 0; c0 < nsuper; c0 += 16) {
					uint32_t t[16];
#pragma unroll
					for (int j = 0; j < 16; j++) t[j] = (c0 + j < nsuper) ? super_sums[(size_t)(c0 + j) * GSR_SORT_RADIX + threadIdx.x] : 0u;
#pragma unroll
					for (int j = 0; j < 16; j++) { v += t[j]; before += (c0 + j < my_super) ? t[j] : 0u; }
				}
			}
			// two levels: every chunk (total and the part in front); three levels: only the chunks of this super-chunk in front
			const int c_first = three_level ? my_super * GSR_SORT_CHUNK : 0, c_end = three_level ? my_chunk : nchunks;
			for (int c0 = c_first; c0 < c_end; c0 += 16) {
				uint32_t t[16];
#pragma unroll
				for (int j = 0; j < 16; j++) t[j] = (c0 + j < c_end) ? chunk_sums[(size_t)(c0 + j) * GSR_SORT_RADIX + threadIdx.x] : 0u;
#pragma unroll
				for (int j = 0; j < 16; j++) { if (!three_level) v += t[j]; before += (c0 + j < my_chunk) ? t[j] : 0u; }
			}
			const int b0 = my_chunk * GSR_SORT_CHUNK, nb = (int)blockIdx.x - b0;  // blocks of this chunk in front of this one
			for (int j0 = 0; j0 < nb; j0 += 16) {
				uint32_t t[16];
#pragma unroll
				for (int j = 0; j < 16; j++) t[j] = (j0 + j < nb) ? table[(size_t)(b0 + j0 + j) * GSR_SORT_RADIX + threadIdx.x] : 0u;
#pragma unroll
				for (int j = 0; j < 16; j++) before += t[j];
			}
		}
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
		my_gbase = wb + incl - v + before;
	}
	const GsrKeyBias kb = gsr_sort_bias(bias, s_bias);  // (contains the barrier that also publishes wsum's readers' results)
	__syncthreads();

	// In-wave ranking.  The lanes that hold digit d in a round find each other through LDS: every lane ORs its bit into the
	// wave's 64-bit word of d (ds_or_b64) and reads the word back -- the set of its peers, for ~10 vector instructions per element
	// where seven ballots with their 64-bit selects cost ~60 (the kernel ran at 73 % of the VALU issue rate).  A wave's LDS
	// operations execute in program order, so the read sees every lane's OR, and the first peer's updates (digit count, mask
	// back to zero) come after every peer's reads.  The words live in the staging area, which is not in use yet.
	uint32_t* mycount = wcount[wave];
	unsigned long long* mymask = reinterpret_cast<unsigned long long*>(sstage) + wave * GSR_SORT_RADIX;
	{
		uint4* z = reinterpret_cast<uint4*>(mymask);  // 256 words of 8 bytes = 128 uint4 per wave
		z[lane] = make_uint4(0u, 0u, 0u, 0u);
		z[64 + lane] = make_uint4(0u, 0u, 0u, 0u);
	}
	__builtin_amdgcn_wave_barrier();
	const unsigned long long lanebit = 1ull << lane;
#pragma unroll
	for (int it = 0; it < ITEMS; it++) {
		const size_t i = gsr_sort_index<ITEMS>(blockIdx.x, wave, it, lane);
		const bool valid = i < n;
		const uint32_t d = gsr_sort_digit(key[it], kb, biased, shift, mask);
		unsigned long long peers = lanebit;
		uint32_t old = 0;
		if (valid) {
			atomicOr(&mymask[d], lanebit);
			__builtin_amdgcn_wave_barrier();
			peers = __atomic_load_n(&mymask[d], __ATOMIC_RELAXED);
			old = __atomic_load_n(&mycount[d], __ATOMIC_RELAXED);
		}
		__builtin_amdgcn_wave_barrier();
		const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));  // peers in lower lanes
		if (valid && below == 0u) {  // the first peer
			__atomic_store_n(&mycount[d], old + (uint32_t)__popcll(peers), __ATOMIC_RELAXED);
			__atomic_store_n(&mymask[d], 0ull, __ATOMIC_RELAXED);
		}
		__builtin_amdgcn_wave_barrier();
		rank[it] = old + below;
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
			const uint32_t lp = mycount[gsr_sort_digit(key[it], kb, biased, shift, mask)] + rank[it];
			skey[lp] = key[it];
			sval[lp] = val[it];
			if constexpr (REC) srec[lp] = rec[it];
		}
	}
	__syncthreads();
	const size_t first = (size_t)blockIdx.x * TILE;
	const uint32_t count = (uint32_t)((n - first < (size_t)TILE) ? (n - first) : (size_t)TILE);
	for (uint32_t i = threadIdx.x; i < count; i += GSR_SORT_THREADS) {
		const KeyT k = skey[i];
		const uint32_t dst = gofs[gsr_sort_digit((uint32_t)k, kb, biased, shift, mask)] + i;
		if (dst < n) {  // always true for consistent tables; a corrupted table must not turn into a wild store
			keys_out[dst] = k;
			if constexpr (REC) rec_out[dst] = make_uint4(srec[i].x, srec[i].y, sval[i], 0u);
			else vals_out[dst] = sval[i];
		}
	}
}

int gsr_radix_num_passes(int nbits_total) { return (nbits_total + 7) / 8; }

int gsr_tile_key_bytes(int ntiles, size_t num_rendered)
{
	(void)num_rendered;
	return ntiles <= 65536 ? 2 : 4;
}

static inline int gsr_sort_items(size_t n) { return n <= GSR_SORT_SMALL_N ? GSR_SORT_ITEMS_SMALL : (n < GSR_SORT_HUGE_N ? GSR_SORT_ITEMS_LARGE : GSR_SORT_ITEMS_HUGE); }
static inline size_t gsr_sort_nblocks(size_t n) { const size_t tile = (size_t)GSR_SORT_THREADS * gsr_sort_items(n); return (n + tile - 1) / tile; }
static inline size_t gsr_sort_nchunks(size_t n) { return (gsr_sort_nblocks(n) + GSR_SORT_CHUNK - 1) / GSR_SORT_CHUNK; }
static inline size_t gsr_sort_nsuper(size_t n) { return (gsr_sort_nchunks(n) + GSR_SORT_CHUNK - 1) / GSR_SORT_CHUNK; }
#define GSR_SORT_MAX_PASSES 4

// table memory: [chunk + super-chunk sums of up to 4 passes: 4 x (nchunks + nsuper) x RADIX u32, zero before the first pass]
// [nblocks x RADIX u32]
size_t gsr_radix_clear_words(size_t n) { return (size_t)GSR_SORT_MAX_PASSES * GSR_SORT_RADIX * (gsr_sort_nchunks(n) + gsr_sort_nsuper(n)); }

size_t gsr_radix_table_bytes(size_t n)
{
	return gsr_align_up((gsr_radix_clear_words(n) + gsr_sort_nblocks(n) * GSR_SORT_RADIX) * sizeof(uint32_t));
}

template <int ITEMS, typename KeyT>
static void gsr_radix_pass(const KeyT* ki, const uint32_t* vi, KeyT* ko, uint32_t* vo, size_t n, int shift, int bits,
                           uint32_t* table, uint32_t* chunk_sums, int nchunks, const uint32_t* bias, hipStream_t s)
{
	const int nblocks = (int)((n + (size_t)GSR_SORT_THREADS * ITEMS - 1) / ((size_t)GSR_SORT_THREADS * ITEMS));
	const uint32_t mask = (1u << bits) - 1u;
	// consecutive blocks per histogram workgroup: 1 up to 2 048 blocks, then as many as keep >= 1 024 workgroups (at most a chunk)
	int per = 1;
	while (per < GSR_SORT_CHUNK && nblocks / (2 * per) >= 1024) per *= 2;
	hipLaunchKernelGGL((gsr_radix_hist_kernel<ITEMS, KeyT>), dim3((nblocks + per - 1) / per), dim3(GSR_SORT_THREADS), 0, s, ki, n, shift, mask, table,
	                   nblocks, chunk_sums, nchunks, bias, per);
	hipLaunchKernelGGL((gsr_radix_scatter_kernel<ITEMS, KeyT>), dim3(nblocks), dim3(GSR_SORT_THREADS), 0, s, ki, vi, ko, vo, n, shift, bits,
	                   table, nblocks, chunk_sums, nchunks, bias);
}

// Runs passes [pass_first, pass_first + pass_count) of the LSD sort on key bits [0, nbits_total) (spread evenly over
// npass_total passes).  Pass p reads (k0,v0) when p is even and (k1,v1) when odd and writes the other pair; the chunk sums
// in table_mem must be zero (each pass uses its own slice).  bias: NULL, or the 128 partial maxima of gsr_sort_bias().
// key_bytes: 4, or 2 = the keys are uint16_t (instance-sized sorts of tile ids below 65 536: a quarter less traffic per pass).
void gsr_radix_sort_passes(void* k0, uint32_t* v0, void* k1, uint32_t* v1, size_t n, int nbits_total, int npass_total,
                           int pass_first, int pass_count, void* table_mem, const uint32_t* bias, int key_bytes, hipStream_t s)
{
	if (n == 0) return;
	// 16-bit keys: 8 per 16-byte load, so the small sorts take 8 items per thread instead of 4 (fewer blocks than the table
	// and the cleared chunk sums were sized for: both stay inside their areas)
	const int items = (key_bytes == 2 && gsr_sort_items(n) == GSR_SORT_ITEMS_SMALL) ? 8 : gsr_sort_items(n);
	const size_t nblocks_ = (n + (size_t)GSR_SORT_THREADS * items - 1) / ((size_t)GSR_SORT_THREADS * items);
	const int nchunks = (int)((nblocks_ + GSR_SORT_CHUNK - 1) / GSR_SORT_CHUNK);
	const int nsuper = (nchunks + GSR_SORT_CHUNK - 1) / GSR_SORT_CHUNK;
	uint32_t* chunk_base = (uint32_t*)table_mem;
	uint32_t* table = chunk_base + gsr_radix_clear_words(n);
	int shift = 0;
	for (int p = 0; p < npass_total && p < pass_first + pass_count; p++) {
		const int bits = (nbits_total - shift + (npass_total - p) - 1) / (npass_total - p);  // spread bits evenly over passes
		if (p >= pass_first) {
			void *ki = (p % 2 == 0) ? k0 : k1, *ko = (p % 2 == 0) ? k1 : k0;
			uint32_t *vi = (p % 2 == 0) ? v0 : v1, *vo = (p % 2 == 0) ? v1 : v0;
			uint32_t* cs = chunk_base + (size_t)p * GSR_SORT_RADIX * (nchunks + nsuper);
			if (key_bytes == 2) {
				if (items == 8) gsr_radix_pass<8, uint16_t>((const uint16_t*)ki, vi, (uint16_t*)ko, vo, n, shift, bits, table, cs, nchunks, bias, s);
				else if (items == GSR_SORT_ITEMS_LARGE) gsr_radix_pass<GSR_SORT_ITEMS_LARGE, uint16_t>((const uint16_t*)ki, vi, (uint16_t*)ko, vo, n, shift, bits, table, cs, nchunks, bias, s);
				else gsr_radix_pass<GSR_SORT_ITEMS_HUGE, uint16_t>((const uint16_t*)ki, vi, (uint16_t*)ko, vo, n, shift, bits, table, cs, nchunks, bias, s);
			} else if (items == GSR_SORT_ITEMS_SMALL) gsr_radix_pass<GSR_SORT_ITEMS_SMALL, uint32_t>((const uint32_t*)ki, vi, (uint32_t*)ko, vo, n, shift, bits, table, cs, nchunks, bias, s);
			else if (items == GSR_SORT_ITEMS_LARGE) gsr_radix_pass<GSR_SORT_ITEMS_LARGE, uint32_t>((const uint32_t*)ki, vi, (uint32_t*)ko, vo, n, shift, bits, table, cs, nchunks, bias, s);
			else gsr_radix_pass<GSR_SORT_ITEMS_HUGE, uint32_t>((const uint32_t*)ki, vi, (uint32_t*)ko, vo, n, shift, bits, table, cs, nchunks, bias, s);
		}
		shift += bits;
	}
}

// First level of the bucket depth sort (depthsort.hip): ONE stable pass of the Gaussian-sized kernels on the top-digit bucket
// of the biased depth keys (gsr_depth_key.h; shift = -1), values = the identity; the elements take their tile rectangles along
// (rec_in = GsrGeometry::rect, read in index order: coalesced) and arrive as 16-byte records {rectangle, id, 0}.  The histogram
// kernel also makes the Gaussians' first gradient slots global (slot_base / block_tiles / status word 3: see the kernel).  Uses pass 0's slice of the chunk sums, which
// must be zero; afterwards [chunk][digit] sums to the bucket sizes.
void gsr_radix_top_pass(const uint32_t* k0, uint32_t* k1, const uint2* rec_in, uint4* rec_out, size_t n, void* table_mem, const uint32_t* bias, hipStream_t s,
                        uint32_t* slot_base, const uint32_t* block_tiles, uint32_t* status)
{
	if (n == 0) return;
	uint32_t* chunk_sums = (uint32_t*)table_mem;
	uint32_t* table = chunk_sums + gsr_radix_clear_words(n);
	constexpr int ITEMS = GSR_SORT_ITEMS_SMALL;
	const int nblocks = (int)((n + (size_t)GSR_SORT_THREADS * ITEMS - 1) / ((size_t)GSR_SORT_THREADS * ITEMS));
	const int nchunks = gsr_radix_top_chunks(n);
	hipLaunchKernelGGL((gsr_radix_hist_kernel<ITEMS, uint32_t>), dim3(nblocks), dim3(GSR_SORT_THREADS), 0, s, k0, n, -1, 255u, table, nblocks, chunk_sums, nchunks, bias, 1,
	                   slot_base, block_tiles, status);
	hipLaunchKernelGGL((gsr_radix_scatter_kernel<ITEMS, uint32_t, true>), dim3(nblocks), dim3(GSR_SORT_THREADS), 0, s, k0, (const uint32_t*)nullptr, k1,
	                   (uint32_t*)nullptr, n, -1, 8, table, nblocks, chunk_sums, nchunks, bias, rec_in, rec_out);
}
// rows of the top pass's [chunk][digit] sums (two levels only: the caller keeps n <= 64 * 64 * 1024)
int gsr_radix_top_chunks(size_t n)
{
	const size_t nblocks_ = (n + (size_t)GSR_SORT_THREADS * GSR_SORT_ITEMS_SMALL - 1) / ((size_t)GSR_SORT_THREADS * GSR_SORT_ITEMS_SMALL);
	return (int)((nblocks_ + GSR_SORT_CHUNK - 1) / GSR_SORT_CHUNK);
}

// Sorts on key bits [0, nbits_total).  Ping-pongs between (k0,v0) and (k1,v1); the sorted result
// ends in (k0,v0) when the pass count is even and in (k1,v1) when odd -- returned through *in_first.
// clear_table: zero the chunk sums here (a memset on the stream); 0 when the kernel that produced the keys already did.
void gsr_radix_sort_u32(void* k0, uint32_t* v0, void* k1, uint32_t* v1, size_t n, int nbits_total, void* table_mem,
                        int* result_in_first, int clear_table, int key_bytes, hipStream_t s)
{
	const int npass = gsr_radix_num_passes(nbits_total);
	*result_in_first = (npass % 2 == 0) ? 1 : 0;
	if (n == 0 || npass == 0) { *result_in_first = 1; return; }
	if (clear_table) (void)hipMemsetAsync(table_mem, 0, gsr_radix_clear_words(n) * sizeof(uint32_t), s);
	gsr_radix_sort_passes(k0, v0, k1, v1, n, nbits_total, npass, 0, npass, table_mem, nullptr, key_bytes, s);
}

// The same numbering for lists that take the global radix passes (beyond GSR_BUCKET_SORT_MAX_P Gaussians, GSR_DEBUG_RADIX_DEPTH): a kernel
// of its own in front of them (the passes' first histogram kernel owns several blocks per workgroup).  Every workgroup adds up the totals
// in front of it, so the blocks are large here -- 64 preprocess workgroups, 16 384 Gaussians: at 6 M Gaussians 366 workgroups read 17 MB
// of totals between them; with blocks of 1 024 it was 5 860 workgroups and 274 MB, 0.10 ms on the path of the depth sort.
#define GSR_SLOT_FINISH_TILE (64 * GSR_PREPROCESS_BLOCK)
__global__ void __launch_bounds__(GSR_SORT_THREADS) gsr_slot_base_finish_kernel(uint32_t* __restrict__ slot_base, const uint32_t* __restrict__ block_tiles,
                                                                                uint32_t* __restrict__ status, size_t n)
{
	gsr_slot_base_finish<GSR_SLOT_FINISH_TILE>(slot_base, block_tiles, status, n);
}

void gsr_launch_slot_base_finish(uint32_t* slot_base, const uint32_t* block_tiles, uint32_t* status, size_t n, hipStream_t s)
{
	if (n == 0) return;
	const size_t tile = (size_t)GSR_SLOT_FINISH_TILE;
	hipLaunchKernelGGL(gsr_slot_base_finish_kernel, dim3((unsigned)((n + tile - 1) / tile)), dim3(GSR_SORT_THREADS), 0, s, slot_base, block_tiles, status, n);
}
