// binning.hip -- instance counting, depth-ordered (tile) key emission, tile ranges.
//
// Replaces cub::DeviceScan::InclusiveSum (rasterizer_impl.cu:323), duplicateWithKeys (:78-126)
// and identifyTileRanges (:133-159).  Together with sort.hip it produces the same sorted instance
// list and tile ranges as the reference's 64-bit (tile | depth) sort:
//   stage 1 (per Gaussian, before the host learns num_rendered):
//     preprocess writes depth bits as sort keys -> stable radix sort of the P Gaussians by depth
//     -> per-workgroup sums of tiles_touched in depth order -> their exclusive scan
//   stage 2 (per instance): emit (tile id, Gaussian id) pairs in depth order -> stable radix sort
//     by tile id -> tile ranges.
#include "gsr_internal.h"

// ---- workgroup scan helpers ------------------------------------------------------------------
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

// inclusive scan across a BLOCK-thread workgroup; returns the inclusive value, *total = sum
template <int BLOCK>
__device__ __forceinline__ uint32_t gsr_block_incl_scan(uint32_t v, uint32_t* total, uint32_t* lds /* BLOCK/64 */)
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

// per-workgroup sums of tiles_touched taken in depth order (perm = Gaussian ids sorted by depth)
__global__ void __launch_bounds__(GSR_PREPROCESS_BLOCK) gsr_sorted_block_sums_kernel(const uint32_t* __restrict__ perm,
                                                                                    const uint32_t* __restrict__ tiles_touched,
                                                                                    int P, uint32_t* __restrict__ sums,
                                                                                    uint32_t* __restrict__ status, uint32_t result_in_alt)
{
	__shared__ uint32_t lds[GSR_PREPROCESS_BLOCK / 64];
	const int i = blockIdx.x * GSR_PREPROCESS_BLOCK + threadIdx.x;
	if (i == 0) status[2] = result_in_alt;  // which ping-pong pair holds the depth order: read by the second forward stage and by tests
	const uint32_t t = (i < P) ? tiles_touched[perm[i]] : 0u;
	uint32_t total;
	(void)gsr_block_incl_scan<GSR_PREPROCESS_BLOCK>(t, &total, lds);
	if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

void gsr_launch_sorted_block_sums(GsrGeometry g, int P, int result_in_alt, hipStream_t s)
{
	const int nb = (P + GSR_PREPROCESS_BLOCK - 1) / GSR_PREPROCESS_BLOCK;
	hipLaunchKernelGGL(gsr_sorted_block_sums_kernel, dim3(nb), dim3(GSR_PREPROCESS_BLOCK), 0, s, result_in_alt ? g.perm_alt : g.perm,
	                   g.tiles_touched, P, g.sorted_block_sums, g.status, (uint32_t)result_in_alt);
}

// ---- key emission ----------------------------------------------------------------------------
// Thread i handles the i-th Gaussian in depth order.  Its first slot is the workgroup prefix plus
// an in-workgroup scan, kept per Gaussian in GsrGeometry::slot_base.  The 64 Gaussians of a wave own ONE contiguous run
// of output positions, so the wave emits cooperatively: lane l writes positions l, l+64, ... of the
// run; the owning Gaussian of a position comes from head flags in LDS and a running maximum over the lanes
// (below).  Every store instruction then writes 64 consecutive elements, whatever the rectangle
// sizes (1 ... >2000 tiles) -- a per-Gaussian loop writes 64 scattered words per instruction
// (measured 2.5x write amplification).  Order inside a Gaussian: y outer, x inner
// (rasterizer_impl.cu:107-118).
__global__ void __launch_bounds__(GSR_PREPROCESS_BLOCK) gsr_duplicate_keys_kernel(GsrGeometry g, int P, uint32_t gx,
                                                                                 void* __restrict__ keys_, int key_bytes,
                                                                                 uint32_t* __restrict__ vals,
                                                                                 uint32_t* __restrict__ clear, size_t clear_words)
{
	// zero the chunk sums of the tile sort's passes (sort.hip)
	for (size_t w = (size_t)blockIdx.x * GSR_PREPROCESS_BLOCK + threadIdx.x; w < clear_words; w += (size_t)gridDim.x * GSR_PREPROCESS_BLOCK)
		clear[w] = 0u;
	const uint32_t* __restrict__ perm = g.status[2] ? g.perm_alt : g.perm;  // where the depth sort left its result
	uint32_t* __restrict__ const keys32 = static_cast<uint32_t*>(keys_);      // tile ids as 32-bit words, or (every id < 65 536)
	uint16_t* __restrict__ const keys16 = static_cast<uint16_t*>(keys_);      // as 16-bit ones: the tile sort moves a quarter less
	__shared__ uint32_t lds[GSR_PREPROCESS_BLOCK / 64];
	__shared__ uint4 s_own[GSR_PREPROCESS_BLOCK / 64][64];       // per Gaussian: start, minx | miny << 16, rectangle width, id
	__shared__ uint32_t s_flag[GSR_PREPROCESS_BLOCK / 64][64];   // per position of the current row: lane + 1 of the Gaussian that starts there
	const int i = blockIdx.x * GSR_PREPROCESS_BLOCK + threadIdx.x;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t idx = 0, tiles = 0, rmin = 0, w = 1;
	if (i < P) {
		idx = perm[i];
		const uint2 rc = g.rect[idx];  // 8-byte gather from a dense array (L2 / Infinity-Cache resident)
		rmin = rc.x;
		w = rc.y & 0xffffu;
		tiles = w * (rc.y >> 16);      // = tiles_touched (0 for culled Gaussians)
		if (!tiles) w = 1;
	}
	// offset of this workgroup's first instance = sum of the tile counts of all workgroups in front of it (depth order):
	// every workgroup adds those up itself (<= nb/256 coalesced loads per thread, all in flight together) -- cheaper than
	// a scan kernel of its own between the depth sort and this one
	uint32_t before = 0;
	{
		const uint32_t* __restrict__ sums = g.sorted_block_sums;
		for (uint32_t b0 = 0; b0 < blockIdx.x; b0 += 8 * GSR_PREPROCESS_BLOCK) {
			uint32_t t[8];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const uint32_t b = b0 + j * GSR_PREPROCESS_BLOCK + threadIdx.x;
				t[j] = b < blockIdx.x ? sums[b] : 0u;
			}
#pragma unroll
			for (int j = 0; j < 8; j++) before += t[j];
		}
	}
	uint32_t total, before_total;
	(void)gsr_block_incl_scan<GSR_PREPROCESS_BLOCK>(before, &before_total, lds);
	const uint32_t incl = gsr_block_incl_scan<GSR_PREPROCESS_BLOCK>(tiles, &total, lds) + before_total;
	const uint32_t off = incl - tiles;
	// the backward blend addresses its per-(Gaussian,tile) gradient slots with this; a 4-byte scatter into a dense
	// 4*P-byte array that the caches absorb (into the 48-byte splat records it cost 2x write amplification)
	// (the gradient slots are numbered in index order before the depth sort, sort.hip: `off` only places this kernel's keys)
	const uint32_t wave_first = __shfl(off, 0, 64);
	const uint32_t wave_total = __shfl(incl, 63, 64) - wave_first;
	s_own[wave][lane] = make_uint4(off - wave_first, rmin, w, idx);
	__builtin_amdgcn_wave_barrier();
	// Owner of every position of a row of 64: the Gaussians whose run starts inside the row flag their first position with
	// their lane number (two Gaussians with tiles never share a start), the lanes -- now as positions -- take the running
	// maximum of the flags in front of them (six DPP steps), and positions in front of the row's first flag continue the
	// Gaussian the previous row ended in.  (A binary search over the 64 starts cost six dependent LDS reads per row.)
	const uint32_t mystart = off - wave_first;
	uint32_t carry = 0u;  // lane + 1 of the Gaussian that owns the position in front of the row
	for (uint32_t j0 = 0; j0 < wave_total; j0 += 64) {
		s_flag[wave][lane] = 0u;
		__builtin_amdgcn_wave_barrier();
		const uint32_t rel = mystart - j0;
		if (tiles && rel < 64u) s_flag[wave][rel] = (uint32_t)lane + 1u;
		__builtin_amdgcn_wave_barrier();
		uint32_t o = s_flag[wave][lane];
		o = max(o, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x111, 0xF, 0xF, false));  // row_shr:1
		o = max(o, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x112, 0xF, 0xF, false));  // row_shr:2
		o = max(o, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x114, 0xF, 0xF, false));  // row_shr:4
		o = max(o, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x118, 0xF, 0xF, false));  // row_shr:8  -> running maximum inside each row of 16
		o = max(o, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1, 3
		o = max(o, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2, 3
		o = max(o, carry);
		carry = (uint32_t)__builtin_amdgcn_readlane((int)o, 63);
		__builtin_amdgcn_wave_barrier();
		const uint32_t j = j0 + lane;
		if (j >= wave_total) continue;
		const uint32_t lo = o - 1u;
		const uint4 own = s_own[wave][lo];  // one 16-byte LDS read
		const uint32_t k = j - own.x;
		const uint32_t rm = own.y, ww = own.z;
		// row of the rectangle = k / ww: k < tiles and ww <= 65535.  Below 2^22 the float quotient is off by at most one, which
		// the remainder test repairs (a 32-bit integer division is ~25 instructions); larger rectangles take the division
		uint32_t q;
		if (k < (1u << 22)) {
			q = (uint32_t)((float)k * __builtin_amdgcn_rcpf((float)ww));
			const int32_t r = (int32_t)(k - q * ww);
			q = r < 0 ? q - 1u : (r >= (int32_t)ww ? q + 1u : q);
		} else {
			q = k / ww;
		}
		const uint32_t y = (rm >> 16) + q, x = (rm & 0xffffu) + (k - q * ww);
		if (key_bytes == 2) keys16[wave_first + j] = (uint16_t)(y * gx + x);
		else keys32[wave_first + j] = y * gx + x;
		vals[wave_first + j] = own.w;
	}
}

void gsr_launch_duplicate_keys(GsrGeometry g, int P, int W, void* keys, int key_bytes, uint32_t* vals, uint32_t* clear, size_t clear_words, hipStream_t s)
{
	const int nb = (P + GSR_PREPROCESS_BLOCK - 1) / GSR_PREPROCESS_BLOCK;
	hipLaunchKernelGGL(gsr_duplicate_keys_kernel, dim3(nb), dim3(GSR_PREPROCESS_BLOCK), 0, s, g, P, (uint32_t)gsr_grid_x(W),
	                   keys, key_bytes, vals, clear, clear_words);
}

// ---- tile ranges (rasterizer_impl.cu:133-159; ranges zeroed first, :377) -------------------------
// four consecutive instances per thread: one 16-byte load of the keys plus the key in front of them.
// Tiles without instances get (0,0) from the thread that sees the gap in the key sequence (the reference
// zeroes the whole array first, rasterizer_impl.cu:377): every element of `ranges` is written exactly once.
// Also clears the per-instance validity bytes of the backward pass (`valid`, one byte per instance = one
// dword per thread here): they live in the sort's ping-pong buffer, which is dead once the sort has finished.
template <typename KeyT>
__global__ void __launch_bounds__(256) gsr_tile_ranges_kernel(const KeyT* __restrict__ tile_keys, int64_t L, uint2* ranges,
                                                              uint32_t ntiles, uint32_t* __restrict__ valid)
{
	const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
	if (i0 >= L) return;
	valid[i0 >> 2] = 0u;
	uint32_t k[4];
	if (i0 + 3 < L) {  // the array is 256-byte aligned inside the blob
		if constexpr (sizeof(KeyT) == 4) {
			const uint4 v = *reinterpret_cast<const uint4*>(tile_keys + i0);
			k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
		} else {
			const uint2 v = *reinterpret_cast<const uint2*>(tile_keys + i0);
			k[0] = v.x & 0xffffu; k[1] = v.x >> 16; k[2] = v.y & 0xffffu; k[3] = v.y >> 16;
		}
	} else {
#pragma unroll
		for (int j = 0; j < 4; j++) k[j] = (i0 + j < L) ? tile_keys[i0 + j] : 0u;
	}
	uint32_t prevtile = (i0 > 0) ? tile_keys[i0 - 1] : 0u;
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const int64_t idx = i0 + j;
		if (idx >= L) break;
		const uint32_t currtile = k[j];
		if (idx == 0) {
			for (uint32_t t = 0; t < currtile; t++) ranges[t] = make_uint2(0u, 0u);
			ranges[currtile].x = 0;
		} else if (currtile != prevtile) {
			ranges[prevtile].y = (uint32_t)idx;
			for (uint32_t t = prevtile + 1; t < currtile; t++) ranges[t] = make_uint2(0u, 0u);
			ranges[currtile].x = (uint32_t)idx;
		}
		if (idx == L - 1) {
			ranges[currtile].y = (uint32_t)L;
			for (uint32_t t = currtile + 1; t < ntiles; t++) ranges[t] = make_uint2(0u, 0u);
		}
		prevtile = currtile;
	}
}

// ---- dispatch order of the backward blend ---------------------------------------------------------
// One wave per tile, 4 waves per SIMD: ~2 waves of tiles per launch, so the last tiles to start decide when the
// kernel ends.  Tiles are handed out in descending order of their work (backward: instances it will stage =
// min(range length, largest n_contrib of the tile); forward: range length): longest-processing-time-first.  A counting sort on
// work / 16 (1024 bins, saturating) by ONE workgroup; the order inside a bin is arbitrary (it only affects
// scheduling: every tile's result is independent of when it runs).
#define GSR_ORDER_BINS 1024
#define GSR_ORDER_PER_THREAD 9  // 9 216 tiles per pass: 1920x1080 (8 160) and 1980x1080 (8 432) in one
// work estimate of a tile: instances the backward will stage (tile_max_contrib given) or the length of its range (the
// forward's upper bound, before anything is known about where its pixels saturate)
__device__ __forceinline__ uint32_t gsr_tile_work(const uint2* ranges, const uint32_t* tile_max_contrib, uint32_t t)
{
#ifdef GSR_TILE_CLOCK
	if (!ranges) return tile_max_contrib[t];  // diagnostic twin: a key supplied by the tool
#endif
	const uint2 r = ranges[t];
	uint32_t work = r.y - r.x;
	if (tile_max_contrib) work = min(work, tile_max_contrib[t]);
	return work;
}
__device__ __forceinline__ uint32_t gsr_tile_work_bin(uint32_t work)
{
	return GSR_ORDER_BINS - 1 - min(work >> 4, (uint32_t)GSR_ORDER_BINS - 1);  // bin 0 = most work
}

// number of depth segments the backward cuts a tile's walk of `work` instances into (1 = not cut), and their length in
// checkpoint strides: at most GSR_CKPT_MAX_SEGMENTS segments of m strides each (render_backward.hip computes the same)
// (`coarse` = 1, 2, 4 or 8: the order kernel doubles it until the extra entries fit its budget)
__device__ __forceinline__ uint32_t gsr_tile_segments(uint32_t work, uint32_t coarse)
{
	if (work < (coarse + 1u) * GSR_CKPT_STRIDE) return 1u;
	const uint32_t nblk = (work + GSR_CKPT_STRIDE - 1) / GSR_CKPT_STRIDE;
	const uint32_t m = (nblk + GSR_CKPT_MAX_SEGMENTS - 1) / GSR_CKPT_MAX_SEGMENTS * coarse;
	return (nblk + m - 1) / m;
}

#ifdef GSR_TILE_CLOCK
// diagnostic twin only (tools/tile_clock.py --measured-key): dispatch the backward by a key the tool supplies (e.g. the
// durations it measured in the previous step), to bound what a better work estimate could be worth
static uint32_t* g_debug_backward_key = nullptr;
extern "C" int gsr_debug_backward_key(uint32_t* key) { g_debug_backward_key = key; return 0; }
// ... and the forward (tools/tile_clock.py --forward-key: e.g. the work the previous forward of the same view counted per tile)
static uint32_t* g_debug_forward_key = nullptr;
extern "C" int gsr_debug_forward_key(uint32_t* key) { g_debug_forward_key = key; return 0; }
#endif

// Forward only (split_min > 0): HEAVY tiles are handed out as four entries, one per 16x4-pixel band (entry = tile |
// (band + 1) << 28; the forward kernel then blends that band alone, one pixel per lane).  A tile's pixels are independent, so
// nothing changes in any result; what changes is the longest job of the launch: on a scene whose heaviest tiles carry 20-30x
// the mean list and are walked to the end (low opacities), the one wave of the heaviest tile WAS the launch (342 of 342 us,
// profiles/r3_tile_clock_c3_lowop.txt).  Heavy = range length >= max(split_min, 2 x the mean listed range), the heaviest
// first, at most max_split tiles; the entries [ntiles + 3 nsplit, ntiles + 3 max_split) are marked empty.
// (ranges and ranges_fix are the same array when the forward normalises empty tiles: neither may be __restrict__; what is written
// there -- (0, 0) over (p, p) -- leaves every tile's work at 0, so the later reads through `ranges` see the same work either way)
__global__ void __launch_bounds__(1024) gsr_tile_order_kernel(const uint2* ranges, const uint32_t* __restrict__ tile_max_contrib,
                                                              uint32_t ntiles, uint32_t* __restrict__ order, uint32_t split_min, uint32_t max_split,
                                                              uint32_t* tile_max_contrib_out, uint32_t seg_budget, int allow_cut,
                                                              uint2* ranges_fix)
{
	__shared__ uint32_t bin[GSR_ORDER_BINS];
	__shared__ uint32_t wsum[1024 / 64];
	__shared__ uint32_t s_nsplit, s_seg_items[4], s_seg_cursor, s_incl[GSR_ORDER_BINS], s_total;
	if (threadIdx.x == 0) { s_nsplit = 0u; s_seg_cursor = 0u; s_total = 0u; }
	if (threadIdx.x < 4) s_seg_items[threadIdx.x] = 0u;
	bin[threadIdx.x] = 0;
	// The kernel is one workgroup of dependent round trips: every pass reads GSR_ORDER_PER_THREAD tiles per thread with all
	// loads issued before the first use, and the bins of the first 1024 * GSR_ORDER_PER_THREAD tiles (all of them up to
	// 1080p) stay in registers between the counting and the placing pass.
	uint32_t b0[GSR_ORDER_PER_THREAD], w0[GSR_ORDER_PER_THREAD];
#pragma unroll
	for (int j = 0; j < GSR_ORDER_PER_THREAD; j++) {
		const uint32_t t = j * 1024 + threadIdx.x;
		w0[j] = t < ntiles ? gsr_tile_work(ranges, tile_max_contrib, t) : 0u;
		b0[j] = t < ntiles ? gsr_tile_work_bin(w0[j]) : 0xffffffffu;
		// the column-pair binning (tilebin.hip) leaves a tile without instances as (p, p): the reference has (0, 0) there
		if (ranges_fix && t < ntiles && w0[j] == 0u) ranges_fix[t] = make_uint2(0u, 0u);
	}
	if (ranges_fix)
		for (uint32_t t = 1024 * GSR_ORDER_PER_THREAD + threadIdx.x; t < ntiles; t += 1024) {
			const uint2 r = ranges_fix[t];
			if (r.x == r.y) ranges_fix[t] = make_uint2(0u, 0u);
		}
	__syncthreads();
	// Backward only (seg_budget > 0): a HEAVY tile -- one whose walk min(range, deepest n_contrib) is at least two checkpoint
	// strides -- is handed out as one entry per DEPTH SEGMENT (entry = tile | (segment + 1) << 28): the forward left per-pixel
	// (T, C) checkpoints every GSR_CKPT_STRIDE instances, so a wave can start in the middle of the list, and every segment
	// writes the gradient slots of its own instances only.  All segment entries go to the front of the list (they belong to the
	// heaviest tiles); if they do not fit the budget nothing is cut.
	if (seg_budget && allow_cut) {   // entries the cut tiles would take, for the four coarseness levels (work values: the registers above)
		uint32_t my_items[4] = {0u, 0u, 0u, 0u};
		auto count = [&](uint32_t w) {
			if (w < 2 * GSR_CKPT_STRIDE) return;   // (almost every tile)
#pragma unroll
			for (int f = 0; f < 4; f++)
				if (w >= ((1u << f) + 1u) * GSR_CKPT_STRIDE) my_items[f] += gsr_tile_segments(w, 1u << f);
		};
#pragma unroll
		for (int j = 0; j < GSR_ORDER_PER_THREAD; j++) count(w0[j]);
		for (uint32_t t = 1024 * GSR_ORDER_PER_THREAD + threadIdx.x; t < ntiles; t += 1024) count(gsr_tile_work(ranges, tile_max_contrib, t));
#pragma unroll
		for (int f = 0; f < 4; f++)
			if (my_items[f]) atomicAdd(&s_seg_items[f], my_items[f]);
	}
	uint32_t my_work = 0u;   // (forward: the listed instances of this thread's tiles -- the mean list decides what is heavy)
#pragma unroll
	for (int j = 0; j < GSR_ORDER_PER_THREAD; j++)
		if (b0[j] != 0xffffffffu) { atomicAdd(&bin[b0[j]], 1u); my_work += w0[j]; }
	for (uint32_t base = 1024 * GSR_ORDER_PER_THREAD; base < ntiles; base += 1024 * GSR_ORDER_PER_THREAD) {
		uint32_t b[GSR_ORDER_PER_THREAD];
#pragma unroll
		for (int j = 0; j < GSR_ORDER_PER_THREAD; j++) {
			const uint32_t t = base + j * 1024 + threadIdx.x;
			const uint32_t w = t < ntiles ? gsr_tile_work(ranges, tile_max_contrib, t) : 0u;
			b[j] = t < ntiles ? gsr_tile_work_bin(w) : 0xffffffffu;
			my_work += w;
		}
#pragma unroll
		for (int j = 0; j < GSR_ORDER_PER_THREAD; j++)
			if (b[j] != 0xffffffffu) atomicAdd(&bin[b[j]], 1u);
	}
	if (split_min) {   // one LDS atomic per wave, not per thread (1 024 adders on one address are microseconds in a one-workgroup kernel)
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) my_work += (uint32_t)__shfl_xor((int)my_work, off, 64);
		if ((threadIdx.x & 63) == 0 && my_work) atomicAdd(&s_total, my_work);
	}
	__syncthreads();
	// heavy (forward only, split_min > 0): a list of at least split_min instances that is also at least twice the MEAN list -- the mean
	// of what is listed (the binning leaves out the tiles a splat provably misses: only the device knows how many are left)
	int split_bin_max = -1;
	if (split_min) {
		const uint32_t heavy = max(split_min, 2u * (s_total / max(ntiles, 1u)));
		split_bin_max = GSR_ORDER_BINS - 1 - (int)min((heavy + 15u) / 16u, (uint32_t)GSR_ORDER_BINS - 1u);
	}
	const uint32_t c = bin[threadIdx.x];
	uint32_t total;
	const uint32_t incl = gsr_block_incl_scan<1024>(c, &total, wsum);
	bin[threadIdx.x] = incl - c;  // first position of the bin
	// tiles of the bins 0 .. split_bin_max are split, as long as they are at most max_split: the largest such bin wins
	if ((int)threadIdx.x <= split_bin_max && c != 0u && incl <= max_split) atomicMax(&s_nsplit, incl);
	// depth segments: at coarseness c the tiles with work >= (c + 1) strides are cut -- exactly the bins 0 .. 1023 - 64 (c + 1) (a
	// stride is 64 bins), a prefix of the descending order; the finest c whose extra entries (entries - tiles) fit the budget wins
	s_incl[threadIdx.x] = incl;
	__syncthreads();
	uint32_t coarse = 0u, nheavy = 0u, front = 0u;
	if (seg_budget && allow_cut)
		for (int f = 0; f < 4 && !coarse; f++) {
			const uint32_t tiles_f = s_incl[GSR_ORDER_BINS - 1 - ((1u << f) + 1u) * (GSR_CKPT_STRIDE / 16)];
			if (s_seg_items[f] != 0u && s_seg_items[f] - tiles_f <= seg_budget) { coarse = 1u << f; nheavy = tiles_f; front = s_seg_items[f]; }
		}
	if (seg_budget && threadIdx.x == 0) order[ntiles + seg_budget] = coarse;   // read by the backward blend (0: nothing is cut)
	const uint32_t nsplit = s_nsplit;  // the nsplit heaviest tiles sit at positions [0, nsplit) of the order
	// (Measured and not kept, round 3: the LAST 512 / 1024 / 2048 tiles of the order as four band entries each, for finer jobs at
	// the end of the launch: forward blend 0.214 - 0.216 ms against 0.214 at C3, the order kernel 1 - 2 us slower.)
	// position q of the descending order -> where the entry goes; split tiles take four entries at the front
	auto place = [&](uint32_t q, uint32_t t) {
		if (q < nheavy) {   // backward: the tile's depth segments, anywhere in the front block
			const uint32_t ns = gsr_tile_segments(gsr_tile_work(ranges, tile_max_contrib, t), coarse);
			const uint32_t at = atomicAdd(&s_seg_cursor, ns);
			for (uint32_t k = 0; k < ns; k++) order[at + k] = t | ((k + 1u) << 28);
		} else if (nheavy) {
			order[front + (q - nheavy)] = t;
		} else if (q < nsplit) {
#pragma unroll
			for (uint32_t k = 0; k < 4; k++) order[4 * q + k] = t | ((k + 1u) << 28);
			tile_max_contrib_out[t] = 0u;  // the four band waves of the tile meet there with atomicMax
		} else {
			order[3 * nsplit + q] = t;
		}
	};
	// the blend kernels are launched over the whole list, split or not: what is not used is marked empty
	if (max_split)
		for (uint32_t e = ntiles + 3 * nsplit + threadIdx.x; e < ntiles + 3 * max_split; e += 1024) order[e] = 0xFFFFFFFFu;
	if (seg_budget)
		for (uint32_t e = ntiles + (front - nheavy) + threadIdx.x; e < ntiles + seg_budget; e += 1024) order[e] = 0xFFFFFFFFu;
#pragma unroll
	for (int j = 0; j < GSR_ORDER_PER_THREAD; j++)
		if (b0[j] != 0xffffffffu) place(atomicAdd(&bin[b0[j]], 1u), j * 1024 + threadIdx.x);
	for (uint32_t base = 1024 * GSR_ORDER_PER_THREAD; base < ntiles; base += 1024 * GSR_ORDER_PER_THREAD) {
		uint32_t b[GSR_ORDER_PER_THREAD];
#pragma unroll
		for (int j = 0; j < GSR_ORDER_PER_THREAD; j++) {
			const uint32_t t = base + j * 1024 + threadIdx.x;
			b[j] = t < ntiles ? gsr_tile_work_bin(gsr_tile_work(ranges, tile_max_contrib, t)) : 0xffffffffu;
		}
#pragma unroll
		for (int j = 0; j < GSR_ORDER_PER_THREAD; j++)
			if (b[j] != 0xffffffffu) place(atomicAdd(&bin[b[j]], 1u), base + j * 1024 + threadIdx.x);
	}
}

// entries of the forward's dispatch list beyond the ntiles whole-tile ones: room for max_split tiles as four band entries each
uint32_t gsr_tile_order_max_segments(int ntiles) { return (uint32_t)(ntiles < 64 ? 0 : (ntiles / 2 < 4096 ? ntiles / 2 : 4096)); }
uint32_t gsr_tile_order_max_split(int ntiles) { return (uint32_t)(ntiles < 64 ? 0 : (ntiles / 4 < 2048 ? ntiles / 4 : 2048)); }

void gsr_launch_tile_order(GsrImage img, int ntiles, bool backward, int64_t num_rendered, bool split, hipStream_t s, bool normalise_empty)
{
	const uint32_t* key = backward ? img.tile_max_contrib : (const uint32_t*)nullptr;
	const uint2* ranges = img.ranges;
#ifdef GSR_TILE_CLOCK
	if (backward && g_debug_backward_key) { key = g_debug_backward_key; ranges = nullptr; }
	if (!backward && g_debug_forward_key) { key = g_debug_forward_key; ranges = nullptr; split = false; }
#endif
	// heavy (forward only): a list of at least GSR_SPLIT_MIN_LIST instances that is also at least twice the mean list, which the kernel
	// takes from the ranges (how deep a list is walked is not known before the forward has run; on the low-opacity blob scene the
	// tiles that set the span were walked deep whatever their length, 2 - 25 x the mean)
	(void)num_rendered;
	const uint32_t max_split = backward ? 0u : gsr_tile_order_max_split(ntiles);   // the forward's list always has this room
	const uint32_t split_min = (!backward && split && max_split) ? (uint32_t)GSR_SPLIT_MIN_LIST : 0u;
	const uint32_t seg_budget = backward ? gsr_tile_order_max_segments(ntiles) : 0u;   // the backward's list always has this room
	int allow_cut = split ? 1 : 0;
#ifdef GSR_TILE_CLOCK
	if (!ranges) allow_cut = 0;
#endif
	hipLaunchKernelGGL(gsr_tile_order_kernel, dim3(1), dim3(1024), 0, s, ranges, key, (uint32_t)ntiles, img.tile_order, split_min, max_split,
	                   img.tile_max_contrib, seg_budget, allow_cut, (normalise_empty && !backward) ? img.ranges : (uint2*)nullptr);
}

void gsr_launch_tile_ranges(const void* tile_keys, int key_bytes, int64_t R, uint2* ranges, int ntiles, uint32_t* valid, hipStream_t s)
{
	if (R > 0 && key_bytes == 2)
		hipLaunchKernelGGL(gsr_tile_ranges_kernel<uint16_t>, dim3((unsigned)((R + 1023) / 1024)), dim3(256), 0, s, (const uint16_t*)tile_keys, R,
		                   ranges, (uint32_t)ntiles, valid);
	else if (R > 0)
		hipLaunchKernelGGL(gsr_tile_ranges_kernel<uint32_t>, dim3((unsigned)((R + 1023) / 1024)), dim3(256), 0, s, (const uint32_t*)tile_keys, R,
		                   ranges, (uint32_t)ntiles, valid);
	else
		(void)hipMemsetAsync(ranges, 0, (size_t)ntiles * sizeof(uint2), s);
}
