// depthsort.hip -- depth order of the Gaussians in three launches: bucket by the top digit, then sort the buckets inside LDS.
//
// Replaces the depth half of cub::DeviceRadixSort::SortPairs (cuda_rasterizer/rasterizer_impl.cu:357-374; the tile half is
// tilebin.hip / sort.hip) for up to GSR_BUCKET_SORT_MAX_P Gaussians; longer lists keep the global LSD passes of sort.hip.
// Same result, bit for bit: Gaussian ids in (depth bits, id) order, culled Gaussians last.
//
// The global LSD sort needed three passes = six launches for the 24 bits of (depth bits - smallest depth bits), and a launch of
// these latency-bound kernels costs 5-13 us whatever it moves (8 MB at a million Gaussians).  Here:
//   level 1 (sort.hip, gsr_radix_top_pass: the histogram + scatter kernels of one LSD pass): a stable counting pass on the TOP
//       digit of the biased key -- bucket = key' >> top_shift, at most 255 buckets over the view's depth range, and a bucket of
//       their own for the culled Gaussians (gsr_depth_key.h).  Afterwards the buckets are in order and, inside a bucket, the
//       Gaussians in index order.
//   level 2 (this file, one launch; 256-thread workgroups that take work items in turn, three per CU):
//       * a bucket of at most DS_CAP (2048) elements is ONE item: one coalesced load, a stable LSD sort of (key', position in
//         the bucket) on the remaining top_shift bits inside LDS, one pass over the result that gathers the elements' records
//         (level 1 moved {tile rectangle, id} along with every key) and writes everything out;
//       * a larger bucket (up to DS_REG_CAP = 12 288 elements: 5 600 on average at a million Gaussians) is cut into PARTS of about
//         DS_PART_TARGET elements by equal ranges of its next digit, one item each.  Every part's workgroup loads the whole bucket's keys into registers
//         (one round trip; the bucket lies in L2), counts what lies in front of its part and what belongs to it, compacts
//         its own elements into LDS in order, sorts them and writes them where the counts say they start.  No workgroup waits
//         for another one, and no histogram pass is needed;
//       * beyond that, or when a part turns out larger than DS_CAP (keys bunched inside the bucket), the bucket is SHARED by
//         next-digit histogram: workgroup j takes a run of next-digit values of about DS_SHARE_QUOTA elements (every workgroup
//         of the bucket counts the histogram itself), and a value that alone exceeds DS_CAP is taken apart by ITS next digit,
//         and so on down to identical keys, which are already in order (stable level 1) and are copied;
//       * the culled Gaussians' bucket (and every bucket when no bits remain) is copied in pieces.
//   The kernel writes the sorted keys, the ids and -- so that the column-pair binning's first histogram starts from one coalesced
//   load -- the Gaussians' tile rectangles in depth order (`seg`, 16 bytes each; tilebin.hip).
//
// Nothing here depends on the keys being spread out: the histogram levels re-read the bucket once per group of DS_CAP elements,
// by one workgroup -- correct for any input, fast only for the inputs that occur (tests/test_depth_sort_gpu.py holds the others).
#include "gsr_internal.h"
#include "gsr_depth_key.h"

#define DS_THREADS 256
#define DS_WAVES (DS_THREADS / 64)
#define DS_CAP 2048                        // elements a workgroup sorts in LDS at a time
#define DS_ROUNDS (DS_CAP / DS_THREADS)    // ranking rounds of a wave per pass, at most
#define DS_REG_ROUNDS 48                   // keys a lane holds when a bucket is cut into parts: buckets of up to ...
#define DS_REG_CAP (DS_REG_ROUNDS * DS_THREADS)   // ... 12 288 elements
#define DS_PART_TARGET 1600                // elements a part aims for (DS_CAP leaves room for 1.28 x that)
#define DS_SHARE_CAP DS_CAP
#define DS_SHARE_QUOTA (DS_CAP / 2)        // elements a workgroup of a histogram-shared bucket aims for
#define DS_LEVELS 4                        // next-digit levels below the top digit: 25 remaining bits at most = 8 + 8 + 8 + 1
#define DS_BATCH 8                         // loads a lane keeps in flight in the traversals of a bucket (round trips to L2, not bytes)
#define DS_GRID_PER_CU 3                   // workgroups a CU holds (LDS)

// a value every lane of the wave holds alike, as a scalar: the branches and loops it steers are then scalar branches, not
// exec-masked regions with workgroup barriers inside
#define DS_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))

struct DsLds {
	uint32_t key[2][DS_CAP];                  // key' of the elements being sorted
	uint16_t idx[2][DS_CAP];                  // their positions in the bucket (one item, parts) or in the collected sequence (shared)
	uint32_t val[DS_CAP];                     // shared buckets: the collected elements' positions in the bucket
	unsigned long long mask[DS_WAVES][256];   // peer masks of the ranking
	uint32_t wcount[DS_WAVES][256];           // per-wave digit counts (ranking), per-wave next-digit histograms (shared buckets)
	uint32_t hist[DS_LEVELS][256];
	uint32_t cum[DS_LEVELS][257];
	uint32_t wsum[4];
	uint32_t wcnt[DS_WAVES];
	uint32_t tab[5];                          // the current item's bucket: offset, size, first item, bucket, parts (0 = shared by histogram)
	uint32_t vlo, vhi;
	uint32_t bias[2];
	// state of the next-digit levels of a shared bucket (index = level 1 .. DS_LEVELS)
	uint32_t st_mask[DS_LEVELS + 1], st_value[DS_LEVELS + 1], st_sh[DS_LEVELS + 1], st_wbits[DS_LEVELS + 1];
	uint32_t st_lo[DS_LEVELS + 1], st_v[DS_LEVELS + 1], st_hi[DS_LEVELS + 1], st_base[DS_LEVELS + 1];
};

// exclusive scan of v over threads 0 .. 255 (thread d -> sum of v of threads < d); every thread of the workgroup calls it, the
// result is meaningful for threads < 256.  Contains two gsr_sync().
__device__ __forceinline__ uint32_t ds_scan_256(uint32_t v, uint32_t* wsum4, uint32_t* total = nullptr)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t incl = v;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t t = __shfl_up(incl, off, 64);
		if (lane >= off) incl += t;
	}
	if (wave < 4 && lane == 63) wsum4[wave] = incl;
	gsr_sync();
	uint32_t wb = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < 4; w++) {
		const uint32_t s = wsum4[w];
		if (w < wave) wb += s;
		tot += s;
	}
	if (total) *total = tot;
	gsr_sync();
	return wb + incl - v;
}

// rec = {tile rectangle (2 words), Gaussian id, -}: what level 1 moved along with the key
__device__ __forceinline__ void ds_emit(uint32_t* __restrict__ keys_out, uint32_t* __restrict__ perm_out, uint4* __restrict__ seg, uint32_t P,
                                        uint32_t pos, uint32_t raw_key, uint4 rec)
{
	if (pos >= P) return;   // always false for consistent tables; a corrupted table must not turn into a wild store
	keys_out[pos] = raw_key;
	perm_out[pos] = rec.z;
	if (seg) seg[pos] = make_uint4(rec.x, rec.y, rec.z, 0u);
}

// 16 bytes at a 4-byte-aligned address (a bucket starts anywhere)
struct __attribute__((packed, aligned(4))) DsWords4 { uint32_t x, y, z, w; };

// a wave's part of a bucket of n elements: [first, last), visited 64 elements at a time
__device__ __forceinline__ void ds_wave_part(uint32_t n, uint32_t& first, uint32_t& last)
{
	const uint32_t wave = threadIdx.x >> 6;
	const uint32_t q = (n + DS_THREADS - 1) / DS_THREADS * 64;
	first = min(n, wave * q);
	last = min(n, first + q);
}

// selection of elements inside a bucket: (key' & mask) == value and digit (key' >> sh) & wmask in [lo, hi)
__device__ __forceinline__ bool ds_selected(uint32_t kp, uint32_t mask, uint32_t value, uint32_t sh, uint32_t wmask, uint32_t lo, uint32_t hi)
{
	const uint32_t d = (kp >> sh) & wmask;
	return (kp & mask) == value && d >= lo && d < hi;
}

// Stable LSD radix sort of L.key[0] / L.idx[0] [0, cnt) on the low `bits` bits of the keys, 8 bits per pass, ping-pong between
// the two buffers; returns the buffer that holds the result.  Ranking as in sort.hip: a wave owns a contiguous run of the
// elements, visits it 64 at a time, and the lanes that hold the same digit find each other through a 64-bit word in LDS.
__device__ __forceinline__ int ds_sort_lds(DsLds& L, uint32_t cnt, uint32_t bits)
{
	cnt = DS_UNIFORM(cnt);
	bits = DS_UNIFORM(bits);
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t run64 = (cnt + DS_THREADS - 1) / DS_THREADS * 64;   // a wave's run
	const uint32_t rounds = run64 / 64;                                 // <= DS_ROUNDS (cnt <= DS_CAP)
	const unsigned long long lanebit = 1ull << lane;
	uint32_t* const mycount = L.wcount[wave];
	unsigned long long* const mymask = L.mask[wave];
	int src = 0;
	for (uint32_t shift = 0; shift < bits; shift += 8, src ^= 1) {
		const uint32_t dmask = (1u << min(8u, bits - shift)) - 1u;
		reinterpret_cast<uint4*>(mycount)[lane] = make_uint4(0u, 0u, 0u, 0u);
		reinterpret_cast<uint4*>(mymask)[lane] = make_uint4(0u, 0u, 0u, 0u);
		reinterpret_cast<uint4*>(mymask)[64 + lane] = make_uint4(0u, 0u, 0u, 0u);
		__builtin_amdgcn_wave_barrier();
		uint32_t rank[DS_ROUNDS];
#pragma unroll
		for (uint32_t it = 0; it < DS_ROUNDS; it++) {
			rank[it] = 0u;
			if (it < rounds) {   // (uniform)
				const uint32_t i = wave * run64 + it * 64 + lane;
				const bool valid = i < cnt;
				const uint32_t d = valid ? (L.key[src][i] >> shift) & dmask : 0u;
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
				rank[it] = old + below;
			}
		}
		gsr_sync();
		{   // thread d < 256: first position of digit d, then of each wave's part of it
			uint32_t tot = 0;
			if (threadIdx.x < 256) {
#pragma unroll
				for (int w = 0; w < DS_WAVES; w++) tot += L.wcount[w][threadIdx.x];
			}
			uint32_t at = ds_scan_256(tot, L.wsum);
			if (threadIdx.x < 256) {
#pragma unroll
				for (int w = 0; w < DS_WAVES; w++) {
					const uint32_t c = L.wcount[w][threadIdx.x];
					L.wcount[w][threadIdx.x] = at;
					at += c;
				}
			}
		}
		gsr_sync();
#pragma unroll
		for (uint32_t it = 0; it < DS_ROUNDS; it++)
			if (it < rounds) {
				const uint32_t i = wave * run64 + it * 64 + lane;
				if (i < cnt) {
					const uint32_t k = L.key[src][i];
					const uint32_t pos = mycount[(k >> shift) & dmask] + rank[it];
					if (pos < DS_CAP) { L.key[src ^ 1][pos] = k; L.idx[src ^ 1][pos] = L.idx[src][i]; }
				}
			}
		gsr_sync();
	}
	return src;
}

// A bucket of n elements that is neither copied nor one item: cut into `parts` parts of about DS_PART_TARGET elements by its
// next digit d (`wbits` bits wide) -- part of an element = (d * parts) >> wbits -- when it fits the registers; else 0: shared by
// histogram.
__device__ __forceinline__ uint32_t ds_parts_of(uint32_t n, uint32_t wbits)
{
	if (n > DS_REG_CAP) return 0u;
	const uint32_t parts = (n + DS_PART_TARGET - 1) / DS_PART_TARGET;
	return parts > (1u << wbits) ? 0u : parts;   // (too few next-digit values for parts that small: by histogram)
}

// work items of a bucket of n elements: copy-type buckets (the culled Gaussians; every bucket when no bits remain) are cut into
// pieces of DS_CAP, a bucket that fits LDS is one item, a larger one takes one item per part or, shared by histogram,
// ceil(n / DS_SHARE_QUOTA) of them
__device__ __forceinline__ uint32_t ds_items_of(uint32_t n, bool copy_type, uint32_t wbits)
{
	if (n == 0u) return 0u;
	if (copy_type) return (n + DS_CAP - 1) / DS_CAP;
	if (n <= DS_CAP) return 1u;
	const uint32_t parts = ds_parts_of(n, wbits);
	return parts ? parts : (n + DS_SHARE_QUOTA - 1) / DS_SHARE_QUOTA;
}

__global__ void __launch_bounds__(DS_THREADS) gsr_ds_bucket_kernel(const uint32_t* __restrict__ keys_in, const uint4* __restrict__ recs_in,
                                                                  uint32_t* __restrict__ keys_out, uint32_t* __restrict__ perm_out,
                                                                  uint4* __restrict__ seg, uint32_t P,
                                                                  const uint32_t* __restrict__ chunk_sums, int nchunks,
                                                                  const uint32_t* __restrict__ bias)
{
	__shared__ __attribute__((aligned(16))) DsLds L;
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	// bucket sizes = level 1's [chunk][digit] sums added up (thread d < 256: bucket d), their offsets, their work items
	uint32_t n_d = 0;
	if (threadIdx.x < 256)
		for (int c0 = 0; c0 < nchunks; c0 += 16) {
			uint32_t t[16];
#pragma unroll
			for (int j = 0; j < 16; j++) t[j] = (c0 + j < nchunks) ? chunk_sums[(size_t)(c0 + j) * 256 + threadIdx.x] : 0u;   // (rows of the depth sort's table: nchunks is the host's)
#pragma unroll
			for (int j = 0; j < 16; j++) n_d += t[j];
		}
	const GsrKeyBias kb = gsr_sort_bias(bias, L.bias);
	const uint32_t s = kb.top_shift;   // bits below the top digit: 0 .. 25
	const bool copy_d = threadIdx.x == kb.culled_digit || s == 0u;
	const uint32_t sh1 = s > 8u ? s - 8u : 0u, wbits1 = s - sh1;   // the next digit: bits [sh1, s) of key'
	const uint32_t wmask1 = (1u << wbits1) - 1u;
	const uint32_t m_d = threadIdx.x < 256 ? ds_items_of(n_d, copy_d, wbits1) : 0u;
	const uint32_t off_d = ds_scan_256(n_d, L.wsum);
	uint32_t items;
	const uint32_t first_d = ds_scan_256(m_d, L.wsum, &items);
	items = DS_UNIFORM(items);

	for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {   // (uniform)
		gsr_sync();   // (the previous item is done with LDS)
		if (threadIdx.x < 256 && item >= first_d && item < first_d + m_d) { L.tab[0] = off_d; L.tab[1] = n_d; L.tab[2] = first_d; L.tab[3] = threadIdx.x; L.tab[4] = (copy_d || n_d <= DS_CAP) ? 1u : ds_parts_of(n_d, wbits1); }
		gsr_sync();
		const uint32_t off = DS_UNIFORM(L.tab[0]), n = DS_UNIFORM(L.tab[1]), j = DS_UNIFORM(item - L.tab[2]), bucket = DS_UNIFORM(L.tab[3]), parts = DS_UNIFORM(L.tab[4]);
		if (off > P || n > P - off) continue;   // inconsistent tables

		if (bucket == kb.culled_digit || s == 0u) {   // nothing to sort: piece j of the bucket as it is
			const uint32_t i0 = j * DS_CAP;
			uint32_t k[DS_ROUNDS];
			uint4 rec[DS_ROUNDS];
#pragma unroll
			for (int u = 0; u < DS_ROUNDS; u++) {
				const uint32_t i = i0 + u * DS_THREADS + threadIdx.x;
				k[u] = i < n ? keys_in[off + i] : 0xFFFFFFFFu;
				rec[u] = i < n ? recs_in[off + i] : make_uint4(0u, 0u, 0u, 0u);
			}
#pragma unroll
			for (int u = 0; u < DS_ROUNDS; u++) {
				const uint32_t i = i0 + u * DS_THREADS + threadIdx.x;
				if (i < n) ds_emit(keys_out, perm_out, seg, P, off + i, k[u], rec[u]);
			}
			continue;
		}

		if (n <= DS_CAP) {   // the whole bucket: DS_ROUNDS elements per thread, every load in flight at once
			{
				uint32_t k[DS_ROUNDS];
#pragma unroll
				for (int u = 0; u < DS_ROUNDS; u++) {
					const uint32_t i = u * DS_THREADS + threadIdx.x;
					k[u] = i < n ? keys_in[off + i] : 0u;
				}
#pragma unroll
				for (int u = 0; u < DS_ROUNDS; u++) {
					const uint32_t i = u * DS_THREADS + threadIdx.x;
					if (i < n) { L.key[0][i] = k[u] - kb.min; L.idx[0][i] = (uint16_t)i; }
				}
			}
			gsr_sync();
			const int buf = ds_sort_lds(L, n, s);
			// the records through the positions: one gather inside the bucket
			uint32_t kp[DS_ROUNDS];
			uint4 rec[DS_ROUNDS];
#pragma unroll
			for (int u = 0; u < DS_ROUNDS; u++) {
				const uint32_t r = u * DS_THREADS + threadIdx.x;
				kp[u] = r < n ? L.key[buf][r] : 0u;
				rec[u] = r < n ? recs_in[off + L.idx[buf][r]] : make_uint4(0u, 0u, 0u, 0u);
			}
#pragma unroll
			for (int u = 0; u < DS_ROUNDS; u++) {
				const uint32_t r = u * DS_THREADS + threadIdx.x;
				if (r < n) ds_emit(keys_out, perm_out, seg, P, off + r, kp[u] + kb.min, rec[u]);
			}
			continue;
		}

		// ---- shared bucket ---------------------------------------------------------------------------------------------
		uint32_t first, last;
		ds_wave_part(n, first, last);
		first = DS_UNIFORM(first);   // (the same in every lane of the wave: the traversals below are scalar loops)
		last = DS_UNIFORM(last);
		uint32_t* const sval = L.val;   // positions in the bucket of the collected elements

		// histogram of a level's digit over the elements (key' & mask) == value, counted per wave over the wave's part of the
		// bucket (L.wcount[wave][digit], which the ranking does not need yet), and its prefix sums: L.hist[lv], L.cum[lv][0 .. 256]
		auto histogram = [&](int lv, uint32_t mask, uint32_t value, uint32_t sh, uint32_t wmask) {
			gsr_sync();
			reinterpret_cast<uint4*>(L.wcount[wave])[lane] = make_uint4(0u, 0u, 0u, 0u);
			__builtin_amdgcn_wave_barrier();
			for (uint32_t i0 = first; i0 < last; i0 += 64 * DS_BATCH) {
				uint32_t k[DS_BATCH];
#pragma unroll
				for (int u = 0; u < DS_BATCH; u++) { const uint32_t i = i0 + u * 64 + lane; k[u] = i < last ? keys_in[off + i] : 0u; }
#pragma unroll
				for (int u = 0; u < DS_BATCH; u++) {
					const uint32_t i = i0 + u * 64 + lane;
					const uint32_t kp = k[u] - kb.min;
					if (i < last && (kp & mask) == value) atomicAdd(&L.wcount[wave][(kp >> sh) & wmask], 1u);
				}
			}
			gsr_sync();
			uint32_t h = 0;
			if (threadIdx.x < 256) {
#pragma unroll
				for (int w = 0; w < DS_WAVES; w++) h += L.wcount[w][threadIdx.x];
				L.hist[lv][threadIdx.x] = h;
			}
			const uint32_t e = ds_scan_256(h, L.wsum);
			if (threadIdx.x < 256) L.cum[lv][threadIdx.x] = e;
			if (threadIdx.x == 255) L.cum[lv][256] = e + h;
			gsr_sync();
		};
		// Stable compaction of the selected elements: into LDS (key', collected position, id; at most DS_SHARE_CAP of them -- the
		// caller knows their number from a histogram) or, to_global, straight to their output positions (identical keys:
		// already in order).  have_counts: L.wcnt holds the waves' counts already; else a counting traversal comes first.
		auto collect = [&](bool to_global, bool have_counts, uint32_t mask, uint32_t value, uint32_t sh, uint32_t wmask, uint32_t lo, uint32_t hi,
		                   uint32_t out_pos) -> uint32_t {
			if (!have_counts) {
				uint32_t c = 0;
				for (uint32_t i0 = first; i0 < last; i0 += 64 * DS_BATCH) {
					uint32_t k[DS_BATCH];
#pragma unroll
					for (int u = 0; u < DS_BATCH; u++) { const uint32_t i = i0 + u * 64 + lane; k[u] = i < last ? keys_in[off + i] : 0u; }
#pragma unroll
					for (int u = 0; u < DS_BATCH; u++) {
						const uint32_t i = i0 + u * 64 + lane;
						c += (uint32_t)__popcll(__ballot(i < last && ds_selected(k[u] - kb.min, mask, value, sh, wmask, lo, hi)));
					}
				}
				gsr_sync();   // (the previous users of wcnt are done)
				if (lane == 0) L.wcnt[wave] = c;
			}
			gsr_sync();   // (... and those of the key / idx buffers)
			uint32_t run = 0, total = 0;
#pragma unroll
			for (uint32_t w = 0; w < DS_WAVES; w++) {
				const uint32_t c = L.wcnt[w];
				if (w < wave) run += c;
				total += c;
			}
			for (uint32_t i0 = first; i0 < last; i0 += 64 * DS_BATCH) {
				uint32_t k[DS_BATCH];
#pragma unroll
				for (int u = 0; u < DS_BATCH; u++) { const uint32_t i = i0 + u * 64 + lane; k[u] = i < last ? keys_in[off + i] : 0u; }
#pragma unroll
				for (int u = 0; u < DS_BATCH; u++) {
					const uint32_t i = i0 + u * 64 + lane;
					const uint32_t kp = k[u] - kb.min;
					const bool p = i < last && ds_selected(kp, mask, value, sh, wmask, lo, hi);
					const unsigned long long m = __ballot(p);
					const uint32_t pos = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
					if (p) {
						if (to_global) ds_emit(keys_out, perm_out, seg, P, out_pos + pos, k[u], recs_in[off + i]);
						else if (pos < DS_SHARE_CAP) { L.key[0][pos] = kp; L.idx[0][pos] = (uint16_t)pos; sval[pos] = i; }
					}
					run += (uint32_t)__popcll(m);
				}
			}
			gsr_sync();
			return DS_UNIFORM(total);
		};
		// sort the collected elements on their low `bits` bits and write them to the output positions from pos0 on
		auto sort_and_write = [&](uint32_t cnt, uint32_t bits, uint32_t pos0) {
			cnt = min(cnt, (uint32_t)DS_SHARE_CAP);
			const int buf = ds_sort_lds(L, cnt, bits);
			uint32_t kp[DS_SHARE_CAP / DS_THREADS];
			uint4 rec[DS_SHARE_CAP / DS_THREADS];
#pragma unroll
			for (int u = 0; u < DS_SHARE_CAP / DS_THREADS; u++) {
				const uint32_t r = u * DS_THREADS + threadIdx.x;
				kp[u] = r < cnt ? L.key[buf][r] : 0u;
				rec[u] = r < cnt ? recs_in[off + sval[L.idx[buf][r]]] : make_uint4(0u, 0u, 0u, 0u);
			}
#pragma unroll
			for (int u = 0; u < DS_SHARE_CAP / DS_THREADS; u++) {
				const uint32_t r = u * DS_THREADS + threadIdx.x;
				if (r < cnt) ds_emit(keys_out, perm_out, seg, P, pos0 + r, kp[u] + kb.min, rec[u]);
			}
		};

		uint32_t lo1, hi1;
		if (parts) {
			// ---- one part: the whole bucket's keys in registers.  LANE-major: lane l of wave w holds the `rounds` consecutive
			// elements from first + l * rounds on -- 16-byte loads, and an element's place among the part's elements is
			// (elements of the waves in front) + (of the lanes in front) + (of this lane's earlier registers): counts per lane
			// and one scan over the lanes, no ballot per register
			const uint32_t rounds = (n + DS_THREADS - 1) / DS_THREADS;   // <= DS_REG_ROUNDS
			const uint32_t lane_first = first + lane * rounds;
			uint32_t k[DS_REG_ROUNDS];
#pragma unroll
			for (int u = 0; u < DS_REG_ROUNDS; u += 4) {
				k[u] = k[u + 1] = k[u + 2] = k[u + 3] = 0xFFFFFFFFu;   // (not a key of a sorted bucket: marks what lies beyond the lane's run)
				if ((uint32_t)u < rounds) {   // (uniform)
					const uint32_t i = lane_first + u;
					if ((uint32_t)u + 3 < rounds && i + 3 < last) {
						const DsWords4 v = *reinterpret_cast<const DsWords4*>(keys_in + off + i);
						k[u] = v.x; k[u + 1] = v.y; k[u + 2] = v.z; k[u + 3] = v.w;
					} else {
#pragma unroll
						for (int c = 0; c < 4; c++)
							if ((uint32_t)(u + c) < rounds && i + c < last) k[u + c] = keys_in[off + i + c];
					}
				}
			}
			auto part_of = [&](uint32_t key) { return ((((key - kb.min) >> sh1) & wmask1) * parts) >> wbits1; };
			uint32_t c_below = 0, c_mine = 0;
#pragma unroll
			for (int u = 0; u < DS_REG_ROUNDS; u++) {
				const bool valid = k[u] != 0xFFFFFFFFu;
				const uint32_t q = part_of(k[u]);
				c_below += (valid && q < j) ? 1u : 0u;
				c_mine += (valid && q == j) ? 1u : 0u;
			}
			uint32_t incl = c_mine;   // inclusive scan of the lanes' counts
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const uint32_t t = __shfl_up(incl, o, 64);
				if (lane >= (uint32_t)o) incl += t;
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) c_below += (uint32_t)__shfl_xor(c_below, o, 64);
			if (lane == 63) { L.wcnt[wave] = incl; L.wsum[wave] = c_below; }
			// (the placing loop below recomputes each element's part: kept live from the counting loop, 48 more registers cost a
			// workgroup per CU)
#pragma unroll
			for (int u = 0; u < DS_REG_ROUNDS; u++) asm volatile("" : "+v"(k[u]));
			gsr_sync();
			uint32_t run = incl - c_mine, mine = 0, below = 0;
#pragma unroll
			for (uint32_t w = 0; w < DS_WAVES; w++) {
				if (w < wave) run += L.wcnt[w];
				mine += L.wcnt[w];
				below += L.wsum[w];
			}
			mine = DS_UNIFORM(mine);
			below = DS_UNIFORM(below);
			if (mine == 0u) continue;   // (uniform)
			if (mine <= DS_CAP) {
#pragma unroll
				for (int u = 0; u < DS_REG_ROUNDS; u++) {
					const bool p = k[u] != 0xFFFFFFFFu && part_of(k[u]) == j;
					if (p && run < DS_CAP) { L.key[0][run] = k[u] - kb.min; L.idx[0][run] = (uint16_t)(lane_first + u); }
					run += p ? 1u : 0u;
				}
				gsr_sync();
				const int buf = ds_sort_lds(L, mine, s);
				uint32_t kp[DS_ROUNDS];
				uint4 rec[DS_ROUNDS];
#pragma unroll
				for (int u = 0; u < DS_ROUNDS; u++) {
					const uint32_t r = u * DS_THREADS + threadIdx.x;
					kp[u] = r < mine ? L.key[buf][r] : 0u;
					rec[u] = r < mine ? recs_in[off + L.idx[buf][r]] : make_uint4(0u, 0u, 0u, 0u);
				}
#pragma unroll
				for (int u = 0; u < DS_ROUNDS; u++) {
					const uint32_t r = u * DS_THREADS + threadIdx.x;
					if (r < mine) ds_emit(keys_out, perm_out, seg, P, off + below + r, kp[u] + kb.min, rec[u]);
				}
				continue;
			}
			// the part is larger than LDS (the bucket's keys are bunched): its next-digit values by histogram, level by level
			histogram(0, 0u, 0u, sh1, wmask1);
			lo1 = ((j << wbits1) + parts - 1u) / parts;           // the values d with (d * parts) >> wbits1 == j
			hi1 = (((j + 1u) << wbits1) + parts - 1u) / parts;
		} else {
			// ---- shared by histogram of the next digit over the whole bucket: workgroup j takes the values whose first element's
			// rank lies in [j, j + 1) * DS_SHARE_QUOTA -- a run of consecutive values (possibly none)
			if (threadIdx.x == 0) { L.vlo = 0xFFFFFFFFu; L.vhi = 0u; }
			histogram(0, 0u, 0u, sh1, wmask1);
			if (L.hist[0][threadIdx.x] != 0u && L.cum[0][threadIdx.x] / DS_SHARE_QUOTA == j) {
				atomicMin(&L.vlo, threadIdx.x);
				atomicMax(&L.vhi, threadIdx.x + 1u);
			}
			gsr_sync();
			lo1 = DS_UNIFORM(L.vlo);
			hi1 = DS_UNIFORM(L.vhi);
			if (lo1 >= hi1) continue;   // (uniform)
			if (L.cum[0][hi1] - L.cum[0][lo1] <= DS_SHARE_CAP) {
				// (almost always) one group: the waves' counts follow from the per-wave histogram, no counting traversal
				if (threadIdx.x < DS_WAVES) L.wcnt[threadIdx.x] = 0u;
				gsr_sync();
				if (threadIdx.x >= lo1 && threadIdx.x < hi1) {
#pragma unroll
					for (int w = 0; w < DS_WAVES; w++)
						if (L.wcount[w][threadIdx.x]) atomicAdd(&L.wcnt[w], L.wcount[w][threadIdx.x]);
				}
				const uint32_t cnt = collect(false, true, 0u, 0u, sh1, wmask1, lo1, hi1, 0u);
				sort_and_write(cnt, s, off + L.cum[0][lo1]);
				continue;
			}
		}
		const uint32_t base1 = off + L.cum[0][lo1];
		// The values [lo, hi) of a level's digit inside the selection (mask, value) go out in groups of consecutive values of at
		// most DS_SHARE_CAP elements (collect, sort on every bit below the selection, write); a value that alone exceeds
		// DS_SHARE_CAP is taken apart by the next level's digit or, when no bits are left, copied (identical keys are in order
		// already).  The levels' state lives in LDS (L.st_*[level]; L.hist / L.cum[level - 1] = the level's histogram).
		if (threadIdx.x == 0) {
			L.st_mask[1] = 0u; L.st_value[1] = 0u; L.st_sh[1] = sh1; L.st_wbits[1] = wbits1;
			L.st_lo[1] = lo1; L.st_v[1] = lo1; L.st_hi[1] = hi1; L.st_base[1] = base1;
		}
		int level = 1;
		while (level >= 1) {   // (uniform: everything it tests lives in LDS)
			gsr_sync();
			const uint32_t v = DS_UNIFORM(L.st_v[level]), hi = DS_UNIFORM(L.st_hi[level]);
			if (v >= hi) { level--; continue; }
			const uint32_t mask = DS_UNIFORM(L.st_mask[level]), value = DS_UNIFORM(L.st_value[level]), sh = DS_UNIFORM(L.st_sh[level]), wbits = DS_UNIFORM(L.st_wbits[level]);
			const uint32_t wmask = (1u << wbits) - 1u;
			const uint32_t* cum = L.cum[level - 1];
			const uint32_t c0 = DS_UNIFORM(cum[v]);
			uint32_t e = hi;
			if (DS_UNIFORM(cum[hi]) - c0 > DS_SHARE_CAP) {   // the largest e with cum[e] - cum[v] <= DS_SHARE_CAP
				uint32_t a = v, z = hi;                       // cum[a] - c0 <= DS_SHARE_CAP < cum[z] - c0
				while (z - a > 1) {
					const uint32_t m = (a + z) / 2;
					if (DS_UNIFORM(cum[m]) - c0 <= DS_SHARE_CAP) a = m; else z = m;
				}
				e = a;
			}
			const uint32_t pos0 = DS_UNIFORM(L.st_base[level] + (c0 - cum[L.st_lo[level]]));
			gsr_sync();   // (every thread has read the level's state)
			if (e == v) {      // value v alone exceeds DS_SHARE_CAP
				const uint32_t m2 = mask | (wmask << sh), v2 = value | (v << sh);
				if (threadIdx.x == 0) L.st_v[level] = v + 1u;
				if (sh == 0u || level == DS_LEVELS) {   // (level == DS_LEVELS implies sh == 0: 25 bits at most)
					(void)collect(true, false, m2, v2, 0u, 0u, 0u, 1u, pos0);
				} else {
					const uint32_t sh2 = sh > 8u ? sh - 8u : 0u, w2 = sh - sh2;
					histogram(level, m2, v2, sh2, (1u << w2) - 1u);
					if (threadIdx.x == 0) {
						L.st_mask[level + 1] = m2; L.st_value[level + 1] = v2; L.st_sh[level + 1] = sh2; L.st_wbits[level + 1] = w2;
						L.st_lo[level + 1] = 0u; L.st_v[level + 1] = 0u; L.st_hi[level + 1] = 1u << w2; L.st_base[level + 1] = pos0;
					}
					level++;
				}
			} else {
				if (threadIdx.x == 0) L.st_v[level] = e;
				const uint32_t cnt = collect(false, false, mask, value, sh, wmask, v, e, 0u);
				sort_and_write(cnt, sh + wbits, pos0);
			}
		}
	}
}

// ---- launcher -------------------------------------------------------------------------------------
bool gsr_bucket_sort_applies(int P) { return P > 0 && P <= GSR_BUCKET_SORT_MAX_P; }

// Depth order of the P Gaussians into (depth_keys, perm) -- status word 2 = 0 -- and, seg != NULL, their rectangles in that order.
// The depth sort's table must have its chunk sums at zero (the preprocess kernel clears them).
void gsr_launch_depth_bucket_sort(GsrGeometry g, int P, uint4* seg, hipStream_t s)
{
	const uint32_t* bias = g.status + GSR_STATUS_NEGMIN;
	uint4* recs = gsr_tilebin_recs(g, P);   // level 1's records {tile rectangle, id, -} in bucket order
	gsr_radix_top_pass(g.depth_keys, g.depth_keys_alt, g.rshape, recs, (size_t)P, g.sort_table, bias, s, g.slot_base, g.block_tiles, g.status);
	// what the chip holds at once (three workgroups per CU: LDS), each taking items in turn; fewer when there cannot be that many items
	const size_t items_max = 2 * ((size_t)P / DS_PART_TARGET) + 256;
	const unsigned grid = (unsigned)(items_max < 256 * DS_GRID_PER_CU ? items_max : 256 * DS_GRID_PER_CU);
	hipLaunchKernelGGL(gsr_ds_bucket_kernel, dim3(grid), dim3(DS_THREADS), 0, s, g.depth_keys_alt, (const uint4*)recs, g.depth_keys, g.perm, seg, (uint32_t)P,
	                   (const uint32_t*)g.sort_table, gsr_radix_top_chunks((size_t)P), bias);
}
