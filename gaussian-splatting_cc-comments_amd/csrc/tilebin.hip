// tilebin.hip -- the sorted instance list by COLUMN PAIRS: two counting passes, no per-instance keys.
//
// Replaces duplicateWithKeys (cuda_rasterizer/rasterizer_impl.cu:78-126), cub::DeviceRadixSort::SortPairs (:357-374) and
// identifyTileRanges (:133-159) for images of at most 256 x 256 tiles (4096 x 4096 pixels); larger images take the
// emission + tile sort of binning.hip / sort.hip.  Result: the same `point_list` and `ranges`, bit for bit.
//
// The Gaussians are already in (depth, index) order (sort.hip, stage 1).  A Gaussian's tile rectangle [x0, x0 + w) x [y0, y0 + h)
// is w COLUMN PAIRS (Gaussian, x) of h tiles each.  Two stable counting passes give the reference's order (tile row-major,
// then depth, then index):
//   pass 1 (by x, 8 bits):  the column pairs of the depth-ordered Gaussians are GENERATED -- nothing is read but the
//       rectangles -- ranked by tile column and written as (y0 | h - 1 << 8, Gaussian) : R / h_mean elements (2.7 M of 9.2 M at 1M
//       Gaussians / 1080p).  Afterwards column x holds its pairs in depth order.
//   pass 2 (by y, 8 bits):  the instances of the column-sorted pairs are generated the same way (pair -> h instances, y0 .. y0 + h - 1)
//       and ranked by tile row: row y then holds (x ascending, depth ascending) = the tiles (y, 0), (y, 1), ... one after the
//       other.  Only the 4-byte Gaussian id of each instance is ever written, once, at its final place; the tile ranges fall
//       out of the pass's offsets (the first workgroup of column x knows where tile (y, x) starts for every y).
// Against the instance-sized pipeline (6-byte pairs emitted, read and written by two radix passes, keys re-read for the
// ranges: ~52 B per instance) this moves ~4 B per instance + ~24 B per column pair.
//
// Both passes are "histogram kernel + scatter kernel" like sort.hip and share its three-level offset tables
// (gsr_radix_walk.h).  What is different is where the elements come from: a workgroup owns 1024 SEGMENTS (pass 1: Gaussians,
// pass 2: column pairs), a segment is a run of consecutive digits (d0, len), and
//   * the histogram of a workgroup is a difference array: +1 at d0, -1 at d0 + len, prefix sum over the 256 digits;
//   * the scatter kernel lets each wave expand its 4 x 64 segments 64 elements at a time (owner of an element: head flags
//     in LDS + a running maximum over the lanes, as the key emission of binning.hip did), ranks the 64 elements of a round
//     by digit through LDS peer masks (ds_or_b64, as sort.hip), and stores each element at its final global position.
// Pass 2's workgroups are column-aligned (a workgroup never spans two tile columns; the map from workgroup to (column, offset)
// is a 256-entry prefix sum every workgroup takes from pass 1's digit totals) and its kernels are persistent -- a fixed grid,
// every workgroup a run of consecutive blocks -- because only the device knows how many column pairs there are.  Consecutive
// blocks also let a workgroup add to the tables' chunk sums once per run (histograms) and walk the tables once (scatters).
//
// Measured and not kept (C3, MI355X; the scatter kernels are bound by their LDS instruction stream -- the LDS unit of a CU is
// busy 35-45 % of their duration -- and by dependent global round trips, not by bytes): elements staged in LDS in digit
// order and written out as contiguous runs (the stores coalesce, but the staging adds an LDS write and read per element and
// costs occupancy: pass 2 0.055 -> 0.086 ms); every lane ranking its own segment's elements in a loop (three LDS round
// trips per group instead of ten per round, but the loops run to the longest segment of the wave: 0.055 -> 0.084 ms).
#include "gsr_internal.h"
#include "gsr_radix_walk.h"
#include "gsr_rect_trim.h"

#define TB_THREADS 256
#define TB_WAVES (TB_THREADS / 64)
#ifndef TB_GROUPS
#define TB_GROUPS 4                          // groups of 64 segments per wave
#endif
#define TB_BLOCK (TB_THREADS * TB_GROUPS)    // segments per workgroup
#define TB_RADIX 256
#ifndef TB_GRID_PER_CU
#define TB_GRID_PER_CU 6                     // workgroups per CU of the persistent pass-2 kernels
#endif
#ifndef TB_WALK_ROWS
#define TB_WALK_ROWS 8                       // rows of the offset tables a wave keeps in flight (gsr_radix_walk.h)
#endif

static inline size_t tb_col_blocks(size_t P) { return (P + TB_BLOCK - 1) / TB_BLOCK; }
static inline size_t tb_chunks(size_t blocks) { return (blocks + GSR_SORT_CHUNK - 1) / GSR_SORT_CHUNK; }
static inline size_t tb_supers(size_t blocks) { return (tb_chunks(blocks) + GSR_SORT_CHUNK - 1) / GSR_SORT_CHUNK; }
// pass 2: column-aligned workgroups over at most R column pairs: sum over columns of ceil(n_x / 1024) <= R / 1024 + 256
static inline size_t tb_row_blocks_max(size_t R) { return R / TB_BLOCK + TB_RADIX; }

bool gsr_tilebin_applies(int W, int H) { return gsr_grid_x(W) <= TB_RADIX && gsr_grid_y(H) <= TB_RADIX; }

// geometry blob: [chunk + super-chunk sums of pass 1, zeroed by the preprocess kernel][block rows][digit totals of pass 1]
// [the Gaussians' rectangles and ids in depth order, 16 bytes each: written by pass 1's histogram, read by its scatter]
size_t gsr_tilebin_col_clear_words(size_t P) { return (tb_chunks(tb_col_blocks(P)) + tb_supers(tb_col_blocks(P))) * TB_RADIX; }
static inline size_t tb_col_table_words(size_t P) { return gsr_tilebin_col_clear_words(P) + tb_col_blocks(P) * TB_RADIX + TB_RADIX; }
// ... and, behind them, the records {rectangle, id, -} that the bucket depth sort's first level moves along with its keys (depthsort.hip)
size_t gsr_tilebin_col_table_bytes(size_t P) { return gsr_align_up(tb_col_table_words(P) * sizeof(uint32_t) + 2 * P * sizeof(uint4)); }
// binning blob: [chunk + super-chunk sums of pass 2, zeroed by pass 1's scatter kernel][block rows]
size_t gsr_tilebin_row_clear_words(size_t R) { return (tb_chunks(tb_row_blocks_max(R)) + tb_supers(tb_row_blocks_max(R))) * TB_RADIX; }
size_t gsr_tilebin_row_table_bytes(size_t R)
{
	return gsr_align_up((gsr_tilebin_row_clear_words(R) + tb_row_blocks_max(R) * TB_RADIX) * sizeof(uint32_t));
}

// ---- device helpers -----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t tb_wave_incl_scan(uint32_t v)
{
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t n = __shfl_up(v, off, 64);
		if (lane >= off) v += n;
	}
	return v;
}

// Per-wave digit counts of the wave's 4 x 64 segments (difference array in LDS, prefix sum over the digits by the wave itself):
// wcount[d] = elements of digit d the wave will generate.  wdiff: TB_RADIX + 4 ints of this wave.
__device__ __forceinline__ void tb_wave_counts(const uint32_t (&d0)[TB_GROUPS], const uint32_t (&len)[TB_GROUPS], int32_t* wdiff, uint32_t* wcount)
{
	const int lane = threadIdx.x & 63;
	reinterpret_cast<int4*>(wdiff)[lane] = make_int4(0, 0, 0, 0);
	if (lane == 0) reinterpret_cast<int4*>(wdiff)[64] = make_int4(0, 0, 0, 0);
	__builtin_amdgcn_wave_barrier();
#pragma unroll
	for (int q = 0; q < TB_GROUPS; q++)
		if (len[q]) {
			atomicAdd(&wdiff[d0[q]], 1);
			atomicAdd(&wdiff[d0[q] + len[q]], -1);
		}
	__builtin_amdgcn_wave_barrier();
	// a wave's LDS operations execute in program order: the loads below see every lane's atomics
	const uint32_t a = (uint32_t)__atomic_load_n(&wdiff[4 * lane], __ATOMIC_RELAXED), b = (uint32_t)__atomic_load_n(&wdiff[4 * lane + 1], __ATOMIC_RELAXED);
	const uint32_t c = (uint32_t)__atomic_load_n(&wdiff[4 * lane + 2], __ATOMIC_RELAXED), d = (uint32_t)__atomic_load_n(&wdiff[4 * lane + 3], __ATOMIC_RELAXED);
	const uint32_t p0 = a, p1 = p0 + b, p2 = p1 + c, p3 = p2 + d;
	const uint32_t base = tb_wave_incl_scan(p3) - p3;
	reinterpret_cast<uint4*>(wcount)[lane] = make_uint4(base + p0, base + p1, base + p2, base + p3);
}

// turns per-wave digit counts into the waves' first global positions: thread d, gbase = first position of the workgroup's digit d
__device__ __forceinline__ void tb_wave_bases(uint32_t (*wcount)[TB_RADIX], uint32_t gbase)
{
	uint32_t run = gbase;
#pragma unroll
	for (int w = 0; w < TB_WAVES; w++) {
		const uint32_t c = wcount[w][threadIdx.x];
		wcount[w][threadIdx.x] = run;
		run += c;
	}
}

// what an element needs of its segment: where the segment starts in the group's element sequence (< 2^15), its first digit,
// its payloads.  Pass 2 packs it into 8 bytes (one ds_read_b64 per element), pass 1 carries a 16-bit key as well.
struct TbOwnRow {
	uint2 v;
	static __device__ TbOwnRow make(uint32_t start, uint32_t d0, uint32_t, uint32_t b, uint32_t = 0u) { TbOwnRow o; o.v = make_uint2(start | (d0 << 16), b); return o; }
	__device__ uint32_t start() const { return v.x & 0xffffu; }
	__device__ uint32_t d0() const { return v.x >> 16; }
	__device__ uint32_t a() const { return 0u; }
	__device__ uint32_t b() const { return v.y; }
	__device__ uint32_t c() const { return 0u; }
};
// pass 1: a = the column's rows before trimming (y0 | (h - 1) << 8), the number of the rectangle's first kept column in bits 16..23 and
// log2 of the columns per nibble / rows per unit of the trim word in bits 24..26 / 27..29; b = Gaussian id, c = the rectangle's trim word (gsr_rect_trim.h)
struct TbOwnCol {
	uint4 v;
	static __device__ TbOwnCol make(uint32_t start, uint32_t d0, uint32_t a, uint32_t b, uint32_t c = 0u) { TbOwnCol o; o.v = make_uint4(start | (d0 << 16), a, b, c); return o; }
	__device__ uint32_t start() const { return v.x & 0xffffu; }
	__device__ uint32_t d0() const { return v.x >> 16; }
	__device__ uint32_t a() const { return v.y; }
	__device__ uint32_t b() const { return v.z; }
	__device__ uint32_t c() const { return v.w; }
};

// One group of 64 segments of a wave, expanded 64 elements per round and ranked by digit.  Lane l holds segment l: first
// digit d0 (< 256), length len (0 = none, <= 256), payloads a, b, c.  For every generated element: store(position, its segment's record, its
// number inside the segment) with
// position = the wave's running count of the element's digit (mycount[d], initialised to the global position of the wave's first
// element of digit d) -- stable: elements of one digit leave in generation order.
//   s_own[64], s_flag[64]: this wave's (s_flag zero before the wave's first call); round: the wave's round counter -- it tags
//   the head flags, so that s_flag needs no clearing between rounds; mymask: this wave's TB_RADIX 64-bit words, all zero on
//   entry and on exit.
template <typename Own, typename Store>
__device__ __forceinline__ void tb_expand_group(uint32_t d0, uint32_t len, uint32_t a, uint32_t b, uint32_t c, Own* s_own, uint32_t* s_flag,
                                                uint32_t& round, unsigned long long* mymask, uint32_t* mycount, Store&& store)
{
	const int lane = threadIdx.x & 63;
	const uint32_t incl = tb_wave_incl_scan(len);
	const uint32_t start = incl - len;   // < 64 * 256
	const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
	s_own[lane] = Own::make(start, d0, a, b, c);
	__builtin_amdgcn_wave_barrier();
	const unsigned long long lanebit = 1ull << lane;
	uint32_t carry = 0u;  // lane + 1 of the segment that owns the element in front of the round
	for (uint32_t j0 = 0; j0 < total; j0 += 64) {
		// owner of every element of the round: the segments that start inside the round flag their first element with their
		// lane number (two non-empty segments never share a start), tagged with the round; the lanes -- now as elements -- take
		// the running maximum of the flags in front of them, and elements in front of the round's first flag continue the
		// previous round's last owner
		round++;
		const uint32_t rel = start - j0;
		if (len && rel < 64u) s_flag[rel] = (round << 8) | ((uint32_t)lane + 1u);
		__builtin_amdgcn_wave_barrier();
		const uint32_t f = __atomic_load_n(&s_flag[lane], __ATOMIC_RELAXED);
		uint32_t o = (f >> 8) == round ? (f & 0xffu) : 0u;
		o = max(o, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x111, 0xF, 0xF, false));  // row_shr:1
		o = max(o, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x112, 0xF, 0xF, false));  // row_shr:2
		o = max(o, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x114, 0xF, 0xF, false));  // row_shr:4
		o = max(o, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x118, 0xF, 0xF, false));  // row_shr:8  -> running maximum inside each row of 16
		o = max(o, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1, 3
		o = max(o, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2, 3
		o = max(o, carry);
		carry = (uint32_t)__builtin_amdgcn_readlane((int)o, 63);
		const uint32_t j = j0 + lane;
		const bool valid = j < total && o != 0u;   // (o != 0 always holds for j < total: guards the LDS index)
		Own own = Own::make(0u, 0u, 0u, 0u);
		if (valid) own = s_own[o - 1u];
		const uint32_t d = (own.d0() + (j - own.start())) & (TB_RADIX - 1u);
		// rank among the round's elements of the same digit: every lane ORs its bit into the wave's word of its digit and
		// reads the word back (sort.hip: a wave's LDS operations execute in program order)
		unsigned long long peers = lanebit;
		uint32_t old = 0;
		if (valid) {
			atomicOr(&mymask[d], lanebit);
			__builtin_amdgcn_wave_barrier();
			peers = __atomic_load_n(&mymask[d], __ATOMIC_RELAXED);
			old = __atomic_load_n(&mycount[d], __ATOMIC_RELAXED);
		}
		__builtin_amdgcn_wave_barrier();
		const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
		if (valid && below == 0u) {  // the first peer
			__atomic_store_n(&mycount[d], old + (uint32_t)__popcll(peers), __ATOMIC_RELAXED);
			__atomic_store_n(&mymask[d], 0ull, __ATOMIC_RELAXED);
		}
		__builtin_amdgcn_wave_barrier();
		if (valid) store(old + below, own, j - own.start());
	}
	__builtin_amdgcn_wave_barrier();  // s_own is rewritten by the next group
}

// ---- pass 1, histogram: per 1024 depth-ordered Gaussians --------------------------------------------
// Also leaves the Gaussians' rectangles and ids in depth order (seg: the scatter kernel then starts from one coalesced
// 16-byte load per Gaussian instead of the chain status word -> permutation -> 8-byte gather; perm == NULL: the bucket depth
// sort has written seg already, and this kernel starts from that coalesced load too), and records where the depth sort left its
// result.  Runs in forward stage 1, behind the depth sort, while the host waits for the count.
__global__ void __launch_bounds__(TB_THREADS) gsr_tb_col_hist_kernel(const uint32_t* __restrict__ perm, const uint2* __restrict__ rect, int P,
                                                                     uint32_t* __restrict__ table, int nblocks,
                                                                     uint32_t* __restrict__ chunk_sums, int nchunks, uint4* __restrict__ seg,
                                                                     uint32_t* __restrict__ status, uint32_t result_in_alt, int blocks_per_wg)
{
	__shared__ int32_t diff[TB_RADIX + 4];
	__shared__ uint32_t wsum[TB_WAVES];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (blockIdx.x == 0 && threadIdx.x == 0) status[2] = result_in_alt;
	// blocks_per_wg consecutive blocks (a power of two <= 64: one chunk) per workgroup, their counts added to the chunk and
	// super-chunk sums once, from registers (sort.hip gsr_radix_hist_kernel: the adders per address of those rows)
	uint32_t acc = 0u;
	const int block0 = (int)blockIdx.x * blocks_per_wg;
	for (int block = block0; block < block0 + blocks_per_wg && block < nblocks; block++) {
		diff[threadIdx.x] = 0;
		if (threadIdx.x < 4) diff[TB_RADIX + threadIdx.x] = 0;
		uint32_t id[TB_GROUPS];
		uint2 rc[TB_GROUPS];
		if (perm) {
#pragma unroll
			for (int q = 0; q < TB_GROUPS; q++) {
				const int i = block * TB_BLOCK + wave * (64 * TB_GROUPS) + q * 64 + lane;
				id[q] = i < P ? perm[i] : 0xFFFFFFFFu;
			}
#pragma unroll
			for (int q = 0; q < TB_GROUPS; q++) rc[q] = id[q] != 0xFFFFFFFFu ? rect[id[q]] : make_uint2(GSR_RECT_NONE, 0u);
		} else {   // the bucket depth sort (depthsort.hip) left the rectangles in depth order already: one coalesced load
#pragma unroll
			for (int q = 0; q < TB_GROUPS; q++) {
				const int i = block * TB_BLOCK + wave * (64 * TB_GROUPS) + q * 64 + lane;
				const uint4 sg = i < P ? seg[i] : make_uint4(GSR_RECT_NONE, 0u, 0xFFFFFFFFu, 0u);
				rc[q] = make_uint2(sg.x, sg.y);
				id[q] = sg.z;
			}
		}
		__syncthreads();
#pragma unroll
		for (int q = 0; q < TB_GROUPS; q++) {
			const int i = block * TB_BLOCK + wave * (64 * TB_GROUPS) + q * 64 + lane;
			uint32_t x0, y0, w, h, lead, wt;
			gsr_rect_unpack(rc[q].x, x0, y0, w, h);
			gsr_trim_columns(rc[q].y, w, h, lead, wt);   // the columns that keep a row: [x0 + lead, x0 + lead + wt)
			if (perm && i < P) seg[i] = make_uint4(rc[q].x, rc[q].y, id[q], 0u);
			if (wt) {
				atomicAdd(&diff[x0 + lead], 1);
				atomicAdd(&diff[x0 + lead + wt], -1);
			}
		}
		__syncthreads();
		// difference array (+1 at the first column, -1 behind the last) -> prefix sum over the digits = pairs per column
		const uint32_t c = (uint32_t)diff[threadIdx.x];
		const uint32_t cnt = gsr_excl_scan_256(c, wsum) + c;   // (wrapping arithmetic: the prefix sums themselves are >= 0; ends with a barrier: diff may be rewritten)
		table[(size_t)block * TB_RADIX + threadIdx.x] = cnt;
		acc += cnt;
	}
	if (acc) {
		atomicAdd(&chunk_sums[(size_t)(block0 / GSR_SORT_CHUNK) * TB_RADIX + threadIdx.x], acc);
		if (nchunks > GSR_SORT_CHUNK)
			atomicAdd(&chunk_sums[(size_t)(nchunks + block0 / (GSR_SORT_CHUNK * GSR_SORT_CHUNK)) * TB_RADIX + threadIdx.x], acc);
	}
}

// ---- pass 1, scatter ------------------------------------------------------------------------------
// Output: column pairs sorted by tile column, inside a column in depth order: (y0 | (h - 1) << 8, Gaussian id) as one 8-byte word.
// Also: pass 1's digit totals for pass 2's workgroup map (workgroup 0), and zeroes for pass 2's chunk sums.  (Until round 4 this
// kernel also numbered the gradient slots, in depth order; they are numbered in index order now, before the depth sort: sort.hip.)
__global__ void __launch_bounds__(TB_THREADS) gsr_tb_col_scatter_kernel(const uint4* __restrict__ seg, int P, int nblocks,
                                                                        const uint32_t* __restrict__ table, const uint32_t* __restrict__ chunk_sums,
                                                                        int nchunks,
                                                                        uint32_t* __restrict__ col_totals, uint2* __restrict__ cpair, uint32_t capacity,
                                                                        uint32_t* __restrict__ clear, size_t clear_words)
{
	__shared__ __attribute__((aligned(16))) unsigned long long s_mask[TB_WAVES][TB_RADIX];  // the walk's partial sums first, then the ranking's peer masks
	__shared__ __attribute__((aligned(16))) uint32_t wcount[TB_WAVES][TB_RADIX];
	__shared__ __attribute__((aligned(16))) int32_t wdiff[TB_WAVES][TB_RADIX + 4];
	__shared__ TbOwnCol s_own[TB_WAVES][64];
	__shared__ uint32_t s_flag[TB_WAVES][64];
	__shared__ uint32_t wsum[TB_WAVES];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (size_t w = (size_t)blockIdx.x * TB_THREADS + threadIdx.x; w < clear_words; w += (size_t)gridDim.x * TB_THREADS) clear[w] = 0u;
	s_flag[wave][lane] = 0u;
	// Workgroup w takes the consecutive blocks [w per, (w + 1) per) -- one block while every block of the launch is resident at once
	// (up to ~1 M Gaussians), several beyond -- and walks the offset tables for the first of them only (see pass 2's scatter)
	const int per = (nblocks + (int)gridDim.x - 1) / (int)gridDim.x;
	const int blk_first = (int)blockIdx.x * per, blk_end = min(blk_first + per, nblocks);
	if (blk_first >= blk_end) return;   // (uniform)

	uint32_t v, before;
	gsr_radix_walk_256<TB_WALK_ROWS>(table, chunk_sums, nchunks, nchunks, blk_first, reinterpret_cast<uint32_t*>(&s_mask[0][0]), v, before);
	if (blockIdx.x == 0) col_totals[threadIdx.x] = v;
	const uint32_t dbase = gsr_excl_scan_256(v, wsum);
	uint32_t round = 0u;

	for (int blk = blk_first; blk < blk_end; blk++) {
		if (blk != blk_first) before += table[(size_t)(blk - 1) * TB_RADIX + threadIdx.x];   // the predecessor's own counts move the offsets on
		const uint32_t gbase = dbase + before;
		uint4 sg[TB_GROUPS];
#pragma unroll
		for (int q = 0; q < TB_GROUPS; q++) {
			const int i = blk * TB_BLOCK + wave * (64 * TB_GROUPS) + q * 64 + lane;
			sg[q] = i < P ? seg[i] : make_uint4(GSR_RECT_NONE, 0u, 0u, 0u);
		}
		uint32_t id[TB_GROUPS], d0[TB_GROUPS], len[TB_GROUPS], key[TB_GROUPS], trim[TB_GROUPS];
#pragma unroll
		for (int q = 0; q < TB_GROUPS; q++) {
			uint32_t x0, y0, w, h, lead, wt;
			gsr_rect_unpack(sg[q].x, x0, y0, w, h);
			gsr_trim_columns(sg[q].y, w, h, lead, wt);
			id[q] = sg[q].z;
			len[q] = wt;              // one element per column that keeps a row
			d0[q] = x0 + lead;
			trim[q] = sg[q].y;
			key[q] = y0 | ((h - 1u) << 8) | (lead << 16) | (gsr_trim_col_shift(w) << 24) | (gsr_trim_row_shift(h) << 27);
		}
		tb_wave_counts(d0, len, wdiff[wave], wcount[wave]);
		__syncthreads();
		// the peer masks start at zero (the walk's partial sums lay there)
		reinterpret_cast<uint4*>(s_mask[wave])[lane] = make_uint4(0u, 0u, 0u, 0u);
		reinterpret_cast<uint4*>(s_mask[wave])[64 + lane] = make_uint4(0u, 0u, 0u, 0u);
		tb_wave_bases(wcount, gbase);
		__syncthreads();
#pragma unroll 1
		for (int q = 0; q < TB_GROUPS; q++)
			tb_expand_group(d0[q], len[q], key[q], id[q], trim[q], s_own[wave], s_flag[wave], round, s_mask[wave], wcount[wave],
			                [&](uint32_t pos, const TbOwnCol& own, uint32_t e) {
				                // the rows this column keeps: its rectangle's, less what the trim word takes off its top and bottom
				                uint32_t t, b;
				                const uint32_t h1 = (own.a() >> 8) & 0xffu;   // h - 1
				                gsr_trim_of(own.c(), ((own.a() >> 16) & 0xffu) + e, (own.a() >> 24) & 7u, own.a() >> 27, t, b);
				                if (t + b > h1) t = b = 0u;   // (an empty column between kept ones: the producer leaves none; whole if it did)
				                const uint32_t rows = (own.a() & 0xffffu) + t - ((t + b) << 8);   // (y0 + t) | (h - t - b - 1) << 8
				                if (pos < capacity) cpair[pos] = make_uint2(rows, own.b());   // always true for consistent tables; a corrupted table must not turn into a wild store
			                });
		__syncthreads();   // every LDS array is rewritten by the next block
	}
}

// ---- pass 2: the map from workgroup to (tile column, first pair) ----------------------------------------
// s_bstart[x] = first workgroup of column x, s_cstart[x] = first column pair of column x ([256] = totals).
__device__ __forceinline__ void tb_row_map(const uint32_t* __restrict__ col_totals, uint32_t* s_bstart, uint32_t* s_cstart, uint32_t* wsum)
{
	const uint32_t n = col_totals[threadIdx.x];
	const uint32_t nb = (n + TB_BLOCK - 1) / TB_BLOCK;
	uint32_t tb, tc;
	const uint32_t eb = gsr_excl_scan_256(nb, wsum, &tb);
	const uint32_t ec = gsr_excl_scan_256(n, wsum, &tc);
	s_bstart[threadIdx.x] = eb;
	s_cstart[threadIdx.x] = ec;
	if (threadIdx.x == 0) { s_bstart[TB_RADIX] = tb; s_cstart[TB_RADIX] = tc; }
	__syncthreads();
}

// Persistent: workgroup w takes the CONSECUTIVE blocks [w per, (w + 1) per) and adds to the chunk (and super-chunk) sums once per
// run of blocks inside one chunk (super-chunk), from registers, instead of once per block: every atomic on those rows has up to
// 64 (4 096) adders per address, which serialise in the L2 (C5, 20 600 blocks: the kernel took 0.15 ms for 168 MB).  (A ticket
// per chunk -- the last block of a chunk adds the finished chunk row -- needs device-scope fences, which write back and
// invalidate an XCD's whole L2 on this chip: 0.15 -> 1.4 ms, measured.)
__global__ void __launch_bounds__(TB_THREADS) gsr_tb_row_hist_kernel(const uint32_t* __restrict__ col_totals, const uint2* __restrict__ cpair,
                                                                     uint32_t* __restrict__ table, uint32_t* __restrict__ chunk_sums, int chunk_rows)
{
	__shared__ int32_t diff[TB_RADIX + 4];
	__shared__ uint32_t s_bstart[TB_RADIX + 1], s_cstart[TB_RADIX + 1], wsum[TB_WAVES], s_x;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	tb_row_map(col_totals, s_bstart, s_cstart, wsum);
	const uint32_t nblocks = s_bstart[TB_RADIX];
	const bool three_level = (nblocks + GSR_SORT_CHUNK - 1) / GSR_SORT_CHUNK > GSR_SORT_CHUNK;
	const uint32_t per = (nblocks + gridDim.x - 1) / gridDim.x;
	const uint32_t eb_first = blockIdx.x * per, eb_end = min(eb_first + per, nblocks);
	uint32_t acc_chunk = 0u, acc_super = 0u;   // thread d: digit d's count over the blocks of the current chunk / super-chunk
	for (uint32_t eb = eb_first; eb < eb_end; eb++) {
		if (s_bstart[threadIdx.x] <= eb && eb < s_bstart[threadIdx.x + 1]) s_x = threadIdx.x;
		diff[threadIdx.x] = 0;
		if (threadIdx.x < 4) diff[TB_RADIX + threadIdx.x] = 0;
		__syncthreads();
		const uint32_t x = s_x;
		const uint32_t c0 = s_cstart[x] + (eb - s_bstart[x]) * TB_BLOCK, c1 = min(c0 + TB_BLOCK, s_cstart[x + 1]);
		uint32_t k[TB_GROUPS];
#pragma unroll
		for (int q = 0; q < TB_GROUPS; q++) {
			const uint32_t c = c0 + wave * (64 * TB_GROUPS) + q * 64 + lane;
			k[q] = c < c1 ? cpair[c].x : 0xFFFFFFFFu;
		}
#pragma unroll
		for (int q = 0; q < TB_GROUPS; q++)
			if (k[q] != 0xFFFFFFFFu) {
				const uint32_t y0 = k[q] & 0xffu, h = (k[q] >> 8) + 1u;
				atomicAdd(&diff[y0], 1);
				atomicAdd(&diff[y0 + h], -1);
			}
		__syncthreads();
		const uint32_t c = (uint32_t)diff[threadIdx.x];
		const uint32_t cnt = gsr_excl_scan_256(c, wsum) + c;   // (ends with a barrier: s_x and diff may be rewritten)
		table[(size_t)eb * TB_RADIX + threadIdx.x] = cnt;
		acc_chunk += cnt;
		if (eb + 1 == eb_end || (eb + 1) % GSR_SORT_CHUNK == 0) {   // the run inside this chunk ends
			if (acc_chunk) atomicAdd(&chunk_sums[(size_t)(eb / GSR_SORT_CHUNK) * TB_RADIX + threadIdx.x], acc_chunk);
			acc_super += acc_chunk;
			acc_chunk = 0u;
			if (three_level && (eb + 1 == eb_end || (eb + 1) % (GSR_SORT_CHUNK * GSR_SORT_CHUNK) == 0)) {
				if (acc_super) atomicAdd(&chunk_sums[(size_t)(chunk_rows + eb / (GSR_SORT_CHUNK * GSR_SORT_CHUNK)) * TB_RADIX + threadIdx.x], acc_super);
				acc_super = 0u;
			}
			if (!three_level) acc_super = 0u;
		}
	}
}

// ---- pass 2, scatter: point_list and the tile ranges -------------------------------------------------
// ranges: the first workgroup of column x writes, for every tile row y, where tile (y, x) starts -- which is also where the
// tile of the previous non-empty column ends -- and (0, 0) for the tiles of the empty columns in between; the last non-empty
// column's closes its tiles with the rows' ends.  A tile without instances inside a non-empty column is left as (p, p); the
// forward's tile-order kernel, which reads every range anyway, turns those into the (0, 0) the reference has.
// Also clears the validity bytes of the backward's gradient slots (`valid`, R bytes).
__global__ void __launch_bounds__(TB_THREADS) gsr_tb_row_scatter_kernel(const uint32_t* __restrict__ col_totals, const uint2* __restrict__ cpair,
                                                                        const uint32_t* __restrict__ table,
                                                                        const uint32_t* __restrict__ chunk_sums, int chunk_rows,
                                                                        uint32_t* __restrict__ point_list, uint32_t R, uint2* __restrict__ ranges,
                                                                        uint32_t gx, uint32_t gy, uint32_t* __restrict__ valid)
{
	__shared__ __attribute__((aligned(16))) unsigned long long s_mask[TB_WAVES][TB_RADIX];
	__shared__ __attribute__((aligned(16))) uint32_t wcount[TB_WAVES][TB_RADIX];
	__shared__ __attribute__((aligned(16))) int32_t wdiff[TB_WAVES][TB_RADIX + 4];
	__shared__ TbOwnRow s_own[TB_WAVES][64];
	__shared__ uint32_t s_flag[TB_WAVES][64];
	__shared__ uint32_t s_bstart[TB_RADIX + 1], s_cstart[TB_RADIX + 1], wsum[TB_WAVES], s_x;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (size_t w = (size_t)blockIdx.x * TB_THREADS + threadIdx.x; w < ((size_t)R + 3) / 4; w += (size_t)gridDim.x * TB_THREADS) valid[w] = 0u;
	s_flag[wave][lane] = 0u;
	uint32_t round = 0u;
	tb_row_map(col_totals, s_bstart, s_cstart, wsum);
	const uint32_t nblocks = s_bstart[TB_RADIX];
	const int nchunks = (int)((nblocks + GSR_SORT_CHUNK - 1) / GSR_SORT_CHUNK);
	uint32_t* const r32 = reinterpret_cast<uint32_t*>(ranges);
	// Workgroup w takes the CONSECUTIVE blocks [w per, (w + 1) per): the offset walk (digit totals + what lies in front: up to
	// 130 table rows) is made for the first of them only -- the totals are the same for every block, and a block's offsets are
	// its predecessor's plus the predecessor's own table row.  (With strided blocks every block walked: 77 rows on average at
	// C3, 223 MB of L2 reads for 22 MB of pairs.)
	const uint32_t per = (nblocks + gridDim.x - 1) / gridDim.x;
	const uint32_t eb_first = blockIdx.x * per, eb_end = min(eb_first + per, nblocks);
	uint32_t v = 0u, before = 0u, dbase = 0u;   // thread d: total of digit d, its count in the blocks in front, first position of the digit
	if (eb_first < eb_end) {
		gsr_radix_walk_256<TB_WALK_ROWS>(table, chunk_sums, nchunks, chunk_rows, (int)eb_first, reinterpret_cast<uint32_t*>(&s_mask[0][0]), v, before);
		dbase = gsr_excl_scan_256(v, wsum);
	}
	for (uint32_t eb = eb_first; eb < eb_end; eb++) {
		if (s_bstart[threadIdx.x] <= eb && eb < s_bstart[threadIdx.x + 1]) s_x = threadIdx.x;
		__syncthreads();
		const uint32_t x = s_x;
		const uint32_t c0 = s_cstart[x] + (eb - s_bstart[x]) * TB_BLOCK, c1 = min(c0 + TB_BLOCK, s_cstart[x + 1]);
		uint32_t d0[TB_GROUPS], len[TB_GROUPS], id[TB_GROUPS];
#pragma unroll
		for (int q = 0; q < TB_GROUPS; q++) {
			const uint32_t c = c0 + wave * (64 * TB_GROUPS) + q * 64 + lane;
			const uint2 kv = c < c1 ? cpair[c] : make_uint2(0u, 0u);
			id[q] = kv.y;
			d0[q] = kv.x & 0xffu;
			len[q] = c < c1 ? (kv.x >> 8) + 1u : 0u;
		}
		if (eb != eb_first) before += table[(size_t)(eb - 1u) * TB_RADIX + threadIdx.x];
		const uint32_t gbase = dbase + before;
		if (eb == s_bstart[x] && threadIdx.x < gy) {  // first workgroup of its column: thread = tile row
			const uint32_t row = threadIdx.x * gx;
			int xp = (int)x - 1;
			while (xp >= 0 && s_bstart[xp + 1] == s_bstart[xp]) xp--;   // previous non-empty column
			r32[2 * (row + x)] = gbase;
			if (xp >= 0) r32[2 * (row + (uint32_t)xp) + 1] = gbase;
			for (uint32_t xe = (uint32_t)(xp + 1); xe < x; xe++) ranges[row + xe] = make_uint2(0u, 0u);
			if (s_bstart[x + 1] == nblocks) {  // last non-empty column: its tiles end where the rows end
				r32[2 * (row + x) + 1] = dbase + v;
				for (uint32_t xe = x + 1; xe < gx; xe++) ranges[row + xe] = make_uint2(0u, 0u);
			}
		}
		tb_wave_counts(d0, len, wdiff[wave], wcount[wave]);
		__syncthreads();
		reinterpret_cast<uint4*>(s_mask[wave])[lane] = make_uint4(0u, 0u, 0u, 0u);
		reinterpret_cast<uint4*>(s_mask[wave])[64 + lane] = make_uint4(0u, 0u, 0u, 0u);
		tb_wave_bases(wcount, gbase);
		__syncthreads();
#pragma unroll 1
		for (int q = 0; q < TB_GROUPS; q++)
			tb_expand_group(d0[q], len[q], 0u, id[q], 0u, s_own[wave], s_flag[wave], round, s_mask[wave], wcount[wave],
			                [&](uint32_t pos, const TbOwnRow& own, uint32_t) {
				                if (pos < R) point_list[pos] = own.b();
			                });
		__syncthreads();   // every LDS array is rewritten by the next round
	}
}

// ---- launchers -----------------------------------------------------------------------------------
struct TbColTable { uint32_t *chunk_sums, *table, *totals; uint4* seg; int nblocks, nchunks; };
static TbColTable tb_col_table(void* mem, int P)
{
	TbColTable t;
	t.nblocks = (int)tb_col_blocks((size_t)P);
	t.nchunks = (int)tb_chunks((size_t)t.nblocks);
	t.chunk_sums = (uint32_t*)mem;
	t.table = t.chunk_sums + gsr_tilebin_col_clear_words((size_t)P);
	t.totals = t.table + (size_t)t.nblocks * TB_RADIX;
	t.seg = reinterpret_cast<uint4*>(t.totals + TB_RADIX);   // (every part in front is a multiple of 1 KB)
	return t;
}

// where pass 1 keeps the Gaussians' rectangles and ids in depth order (the bucket depth sort writes them itself)
uint4* gsr_tilebin_seg(GsrGeometry g, int P) { return tb_col_table(g.col_table, P).seg; }
uint4* gsr_tilebin_recs(GsrGeometry g, int P) { return tb_col_table(g.col_table, P).seg + (size_t)P; }

// seg_ready: the depth sort left `seg` (and its order in (depth_keys, perm): result_in_alt = 0)
void gsr_launch_tilebin_col_hist(GsrGeometry g, int P, int result_in_alt, hipStream_t s, bool seg_ready)
{
	const TbColTable t = tb_col_table(g.col_table, P);
	int per = 1;   // consecutive blocks per workgroup: 1 up to 2 048 blocks, then as many as keep >= 1 024 workgroups
	while (per < GSR_SORT_CHUNK && t.nblocks / (2 * per) >= 1024) per *= 2;
	const uint32_t* perm = seg_ready ? nullptr : (result_in_alt ? g.perm_alt : g.perm);
	hipLaunchKernelGGL(gsr_tb_col_hist_kernel, dim3((t.nblocks + per - 1) / per), dim3(TB_THREADS), 0, s, perm, g.rshape, P,
	                   t.table, t.nblocks, t.chunk_sums, t.nchunks, t.seg, g.status, (uint32_t)result_in_alt, per);
}

// the sorted column pairs: 8 bytes each, in the two arrays of the binning blob that only the tile sort uses (point_list_alt and,
// behind it, tile_keys: 8 R bytes together, and there are at most R pairs)
static uint2* tb_pairs(const GsrBinning& b) { return reinterpret_cast<uint2*>(b.point_list_alt); }
static uint32_t tb_pair_capacity(const GsrBinning& b) { return (uint32_t)(((const char*)b.tile_keys_alt - (const char*)b.point_list_alt) / sizeof(uint2)); }

// workgroups of the persistent pass-2 kernels: six per CU (what a CU holds of the scatter kernel: 78 VGPRs, 21 KB of LDS), three
// when the pass is long -- each workgroup then owns more consecutive blocks and walks the offset tables once for all of them
// (measured, scatter kernel: C3, 2 900 blocks: 0.055 ms with six, 0.059 with three; C5, 20 600 blocks: 0.48 with six, 0.43 with three)
static int tb_persistent_grid(size_t blocks_max)
{
	const size_t cap = 256 * (size_t)(blocks_max >= 32768 ? TB_GRID_PER_CU / 2 : TB_GRID_PER_CU);
	return (int)(blocks_max < cap ? blocks_max : cap);
}

void gsr_launch_tilebin_col_scatter(GsrGeometry g, int P, GsrBinning b, int64_t R, hipStream_t s)
{
	const TbColTable t = tb_col_table(g.col_table, P);
	const int grid = t.nblocks < 256 * 4 ? t.nblocks : 256 * 4;   // what the chip holds at once (116 VGPRs: four workgroups per CU)
	hipLaunchKernelGGL(gsr_tb_col_scatter_kernel, dim3(grid), dim3(TB_THREADS), 0, s, t.seg, P, t.nblocks, t.table, t.chunk_sums,
	                   t.nchunks, t.totals, tb_pairs(b), tb_pair_capacity(b), (uint32_t*)b.sort_table,
	                   gsr_tilebin_row_clear_words((size_t)R));
}

void gsr_launch_tilebin_row_hist(GsrGeometry g, int P, GsrBinning b, int64_t R, hipStream_t s)
{
	const TbColTable t = tb_col_table(g.col_table, P);
	uint32_t* chunk_sums = (uint32_t*)b.sort_table;
	uint32_t* table = chunk_sums + gsr_tilebin_row_clear_words((size_t)R);
	hipLaunchKernelGGL(gsr_tb_row_hist_kernel, dim3(tb_persistent_grid(tb_row_blocks_max((size_t)R))), dim3(TB_THREADS), 0, s, t.totals,
	                   (const uint2*)tb_pairs(b), table, chunk_sums, (int)tb_chunks(tb_row_blocks_max((size_t)R)));
}

void gsr_launch_tilebin_row_scatter(GsrGeometry g, int P, GsrBinning b, int64_t R, uint2* ranges, int W, int H, hipStream_t s)
{
	const TbColTable t = tb_col_table(g.col_table, P);
	uint32_t* chunk_sums = (uint32_t*)b.sort_table;
	uint32_t* table = chunk_sums + gsr_tilebin_row_clear_words((size_t)R);
	hipLaunchKernelGGL(gsr_tb_row_scatter_kernel, dim3(tb_persistent_grid(tb_row_blocks_max((size_t)R))), dim3(TB_THREADS), 0, s, t.totals,
	                   (const uint2*)tb_pairs(b), table, chunk_sums, (int)tb_chunks(tb_row_blocks_max((size_t)R)),
	                   b.point_list, (uint32_t)R, ranges, (uint32_t)gsr_grid_x(W), (uint32_t)gsr_grid_y(H), b.tile_keys_alt);
}
