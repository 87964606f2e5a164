// gsr_depth_key.h -- how the depth sort reads its keys: shared by sort.hip (global radix passes) and depthsort.hip
// (bucket sort in LDS).
//
// Keys are the float bits of the view-space depth (> 0.2, so unsigned order == float order); culled Gaussians carry
// 0xFFFFFFFF.  The preprocess kernel leaves 64-way partial maxima of ~key and key over the VISIBLE Gaussians in the status
// words (GsrGeometry::status + GSR_STATUS_NEGMIN); every sort workgroup reduces them itself.  The sort then orders
//   key' = key - min        (visible)        0 .. max - min
//   key' = max - min + 1    (culled)
// so only the bits of max - min need work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// __syncthreads() with the release fence's wait spelled out.  __syncthreads() is fence(release) + s_barrier + fence(acquire),
// and the release fence has to become `s_waitcnt lgkmcnt(0)` whenever one of the wave's own LDS writes may still be in flight:
// the other waves of the workgroup sit on other SIMDs with their own path into the LDS and can otherwise read the old value
// after the barrier.  hipcc 7.2 drops that wait on one loop back edge of depthsort.hip (thread 0 stores the next level's state,
// the loop head's barrier follows with nothing in between): once in ten runs of a scene with 45 000 equal depths per bucket
// the other three waves read the previous item's state, left the level loop early, and from then on the workgroup's barriers
// paired up wrongly -- missing output, then wild indices ("The stale level state").  tools/barrier_audit.py
// (`make audit`) checks every s_barrier of the library's ISA for this; the depth sort uses gsr_sync() throughout.
__device__ __forceinline__ void gsr_sync()
{
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__syncthreads();
}

struct GsrKeyBias {
	uint32_t min, culled;      // culled = value that stands for 0xFFFFFFFF keys = (max - min) + 1
	// top-digit mode (depthsort.hip): visible keys fall into buckets key' >> top_shift = 0 .. culled_digit - 1 (at most 255 of
	// them), the culled ones into bucket culled_digit (<= 255) of their own
	uint32_t top_shift, culled_digit;
};

__device__ __forceinline__ GsrKeyBias gsr_key_bias_of(uint32_t mn, uint32_t mx)
{
	GsrKeyBias kb;
	const bool any = mx >= mn;  // (no visible Gaussian at all: min = ~0, max = 0)
	const uint32_t vr = any ? mx - mn : 0u;
	kb.min = mn;
	kb.culled = any ? vr + 1u : 0u;  // no visible Gaussian: every key is the culled value
	const uint32_t nb = vr ? 32u - (uint32_t)__builtin_clz(vr) : 0u;
	uint32_t s = nb > 8u ? nb - 8u : 0u;
	if ((vr >> s) >= 255u) s++;      // the largest visible bucket is at most 254: 255 stays free for the culled ones
	kb.top_shift = s;
	kb.culled_digit = any ? (vr >> s) + 1u : 0u;
	return kb;
}

// min / range of the biased keys from the 64 + 64 partial maxima {max(~key)}, {max(key)} (GsrGeometry::status); lds2: two
// words of LDS.  Contains one __syncthreads().
__device__ __forceinline__ GsrKeyBias gsr_sort_bias(const uint32_t* __restrict__ bias, uint32_t* lds2)
{
	GsrKeyBias kb = {0u, 0xFFFFFFFFu, 0u, 0u};
	if (!bias) return kb;  // uniform
	if (threadIdx.x < 64) {
		uint32_t nmin = bias[threadIdx.x], mx = bias[64 + threadIdx.x];
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) {
			nmin = max(nmin, (uint32_t)__shfl_xor(nmin, off, 64));
			mx = max(mx, (uint32_t)__shfl_xor(mx, off, 64));
		}
		if (threadIdx.x == 0) { lds2[0] = ~nmin; lds2[1] = mx; }
	}
	__syncthreads();
	return gsr_key_bias_of(lds2[0], lds2[1]);
}

__device__ __forceinline__ uint32_t gsr_sort_key(uint32_t k, const GsrKeyBias& kb, bool biased)
{
	return biased ? (k == 0xFFFFFFFFu ? kb.culled : k - kb.min) : k;
}

// digit of a key in a radix pass: bits [shift, shift + log2(mask + 1)) of the (biased) key, or, shift < 0, its top-digit bucket
__device__ __forceinline__ uint32_t gsr_sort_digit(uint32_t k, const GsrKeyBias& kb, bool biased, int shift, uint32_t mask)
{
	if (shift < 0) return k == 0xFFFFFFFFu ? kb.culled_digit : (k - kb.min) >> kb.top_shift;  // (uniform branch)
	return (gsr_sort_key(k, kb, biased) >> shift) & mask;
}
