// gsr_radix_walk.h -- the offset walk shared by the scatter kernels of sort.hip and tilebin.hip.
//
// A radix pass keeps three levels of digit counts: [block][digit] rows, [chunk][digit] sums (chunk = GSR_SORT_CHUNK consecutive
// blocks) and, for long sorts, [super-chunk][digit] sums (super-chunk = GSR_SORT_CHUNK chunks), the last two accumulated by
// the histogram kernels with atomics.  A scatter workgroup needs, per digit, the total over all blocks and the part in
// front of its own block: <= #super-chunks + 63 + 63 rows whatever the size of the sort.  Those rows are ONE list, dealt
// round-robin to the workgroup's waves: a lane reads digits 4l .. 4l+3 of a row (one coalesced kilobyte per wave
// instruction), ROWS rows in flight per wave; the waves' partial sums meet in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef GSR_WALK_ROWS
#define GSR_WALK_ROWS 8    // rows of the offset tables a wave keeps in flight (one uint4 per lane each)
#endif
#ifndef GSR_SORT_CHUNK
#define GSR_SORT_CHUNK 64   // blocks per chunk, chunks per super-chunk of the three-level offset table
#endif
#define GSR_WALK_RADIX 256

// For thread d = threadIdx.x of a 256-thread workgroup: total = count of digit d over all blocks, before = count of digit d
// in the blocks in front of `block`.  nchunks = chunks in use (three levels iff nchunks > GSR_SORT_CHUNK); chunk_rows = rows
// the chunk level was laid out for (the super-chunk rows follow them; == nchunks when the host knows the block count).
// part: 2 * 4 * 256 words of LDS, 16-byte aligned, free for other use after the call.  Contains one __syncthreads().
template <int ROWS = GSR_WALK_ROWS>
__device__ __forceinline__ void gsr_radix_walk_256(const uint32_t* __restrict__ table, const uint32_t* __restrict__ chunk_sums,
                                                   int nchunks, int chunk_rows, int block, uint32_t* part,
                                                   uint32_t& total, uint32_t& before)
{
	constexpr int WAVES = 4;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int my_chunk = block / GSR_SORT_CHUNK, my_super = my_chunk / GSR_SORT_CHUNK;
	const bool three_level = nchunks > GSR_SORT_CHUNK;  // uniform
	const int nS = three_level ? (nchunks + GSR_SORT_CHUNK - 1) / GSR_SORT_CHUNK : 0;
	// two levels: every chunk (total, and the part in front); three levels: only the chunks of this super-chunk in front
	const int c_first = three_level ? my_super * GSR_SORT_CHUNK : 0, c_end = three_level ? my_chunk : nchunks;
	const int nC = c_end - c_first;
	const int b0 = my_chunk * GSR_SORT_CHUNK, nB = block - b0;  // blocks of this chunk in front of this one
	const int Q = nS + nC + nB;
	const uint32_t* super_sums = chunk_sums + (size_t)chunk_rows * GSR_WALK_RADIX;
	uint4 v4 = make_uint4(0u, 0u, 0u, 0u), bf4 = v4;
	for (int q0 = wave; q0 < Q; q0 += ROWS * WAVES) {
		uint4 t[ROWS];
#pragma unroll
		for (int j = 0; j < ROWS; j++) {
			const int q = q0 + j * WAVES;
			const uint32_t* row = q < nS ? super_sums + (size_t)q * GSR_WALK_RADIX
			                    : q < nS + nC ? chunk_sums + (size_t)(c_first + q - nS) * GSR_WALK_RADIX
			                                  : table + (size_t)(b0 + q - nS - nC) * GSR_WALK_RADIX;
			t[j] = q < Q ? reinterpret_cast<const uint4*>(row)[lane] : make_uint4(0u, 0u, 0u, 0u);
		}
#pragma unroll
		for (int j = 0; j < ROWS; j++) {
			const int q = q0 + j * WAVES;
			const bool to_total = three_level ? q < nS : (q >= nS && q < nS + nC);
			const bool to_before = q < nS ? q < my_super : (q < nS + nC ? c_first + q - nS < my_chunk : true);
			if (to_total) { v4.x += t[j].x; v4.y += t[j].y; v4.z += t[j].z; v4.w += t[j].w; }
			if (to_before) { bf4.x += t[j].x; bf4.y += t[j].y; bf4.z += t[j].z; bf4.w += t[j].w; }
		}
	}
	uint4* p4 = reinterpret_cast<uint4*>(part);  // [total | before][wave][lane] uint4 = [total | before][wave][digit] words
	p4[wave * 64 + lane] = v4;
	p4[(WAVES + wave) * 64 + lane] = bf4;
	__syncthreads();
	uint32_t v = 0, bf = 0;
#pragma unroll
	for (int w = 0; w < WAVES; w++) {
		v += part[w * GSR_WALK_RADIX + threadIdx.x];
		bf += part[(WAVES + w) * GSR_WALK_RADIX + threadIdx.x];
	}
	total = v;
	before = bf;
}

// exclusive scan of `v` over the 256 threads of the workgroup (thread d -> sum of v of threads < d); wsum: 4 words of LDS.
// Contains two __syncthreads().
__device__ __forceinline__ uint32_t gsr_excl_scan_256(uint32_t v, uint32_t* wsum, uint32_t* total = nullptr)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t incl = v;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t t = __shfl_up(incl, off, 64);
		if (lane >= off) incl += t;
	}
	if (lane == 63) wsum[wave] = incl;
	__syncthreads();
	uint32_t wb = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < 4; w++) {
		const uint32_t s = wsum[w];
		if (w < wave) wb += s;
		tot += s;
	}
	if (total) *total = tot;
	__syncthreads();  // wsum may be reused at once
	return wb + incl - v;
}
