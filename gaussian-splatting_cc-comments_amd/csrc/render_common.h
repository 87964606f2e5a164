// render_common.h -- pieces shared by the forward and backward blend kernels.
//
// Work decomposition (both directions): ONE wave64 owns one 16x16 tile; lane l owns the four
// pixels (x = l & 15, y = (l >> 4) + 4k), k = 0..3.  A wave needs no workgroup barrier, the
// per-instance LDS broadcast is amortised over four pixels, and in the backward pass three of the
// four cross-lane reduction levels of the reference's 256-thread block become plain register adds.
// A 256-thread workgroup simply carries four independent tiles.
#pragma once
#include "gsr_internal.h"

// The blend kernels take `cull` as a launch argument: 0 (GSR_DEBUG_NO_CULL in the `debug` mask of the C ABI, include/gsr.h)
// disables the exact-result-preserving culling below -- every instance of the reference's tile list is evaluated on every
// band -- to bisect a suspected culling error.  Nothing is read from the environment.

#define GSR_WAVES_PER_WG 1
#define GSR_PIX_PER_LANE 4

// Diagnostic build only (-DGSR_TILE_CLOCK, `make tile_clock`; tools/tile_clock.py): each blend kernel gets a device
// pointer and every wave records when (100 MHz s_memrealtime) and where (HW_ID, XCC_ID) it ran, four words per tile.
// The product library has none of it.
#ifdef GSR_TILE_CLOCK
#include <hip/hip_runtime.h>
#define GSR_TILE_CLOCK_BUFFER(sym, setter)                                                           \
	__device__ unsigned long long* sym = nullptr;                                                    \
	extern "C" int setter(unsigned long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(sym), &buf, sizeof(buf)); }
// eight words per tile: start / end on the 100 MHz constant clock, HW_ID, XCC_ID, start / end on the shader clock
// (s_memtime: with the pair above, the clock the wave actually ran at), two words of work counters (sa, sb)
#define GSR_TILE_CLOCK_START() const unsigned long long gsr_tc0 = __builtin_amdgcn_s_memrealtime(), gsr_tcc0 = __builtin_amdgcn_s_memtime()
#define GSR_TILE_CLOCK_STOP(sym, tile, lane, sa, sb)                                                 \
	do {                                                                                             \
		unsigned long long* gsr_tc = sym;                                                            \
		if (gsr_tc && (lane) == 0) {                                                                 \
			gsr_tc[8 * (size_t)(tile)] = gsr_tc0;                                                    \
			gsr_tc[8 * (size_t)(tile) + 1] = __builtin_amdgcn_s_memrealtime();                       \
			gsr_tc[8 * (size_t)(tile) + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  /* HW_REG_HW_ID */  \
			gsr_tc[8 * (size_t)(tile) + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20); /* HW_REG_XCC_ID */ \
			gsr_tc[8 * (size_t)(tile) + 4] = gsr_tcc0;                                               \
			gsr_tc[8 * (size_t)(tile) + 5] = __builtin_amdgcn_s_memtime();                           \
			gsr_tc[8 * (size_t)(tile) + 6] = (sa);                                                   \
			gsr_tc[8 * (size_t)(tile) + 7] = (sb);                                                   \
		}                                                                                            \
	} while (0)
#define GSR_TILE_STAT(x) x
#else
#define GSR_TILE_CLOCK_BUFFER(sym, setter)
#define GSR_TILE_CLOCK_START()
#define GSR_TILE_CLOCK_STOP(sym, tile, lane, sa, sb)
#define GSR_TILE_STAT(x)
#endif

// Conservative, exact-result-preserving culling.  alpha = o*exp(-q/2) with
// q = a dx^2 + 2 b dx dy + c dy^2 can reach 1/255 inside a pixel rectangle only if the minimum of
// q over the rectangle is <= 2 ln(255 o).  The minimum of the convex quadratic over a rectangle is
// 0 when the centre lies inside it and otherwise sits on one of the four edges, where it has a
// closed form.  The test is kept conservative by a slack of 1e-3 in q (5e-4 relative in alpha) plus
// 1e-5 of the magnitude of the terms that cancel in q (for a giant elongated splat far from its
// centre they are O(1e6) and their fp32 sum is only good to ~1): an instance that the reference
// would blend into any pixel of the rectangle is never dropped, and dropped instances contribute
// exactly nothing in the reference either (forward.cu:438-447, backward.cu:521-531).
// Returns a 4-bit mask: bit k set = the instance may reach the band of pixel rows 4k..4k+3 of the
// 16x16 tile at (x0, y0) (the rows owned by pixel slot k of every lane); 0 = drop the instance.
__device__ __forceinline__ uint32_t gsr_tile_band_mask(float mx, float my, float a, float b, float c, float op, float x0,
                                                       float y0)
{
	if (!(a > 0.f) || !(c > 0.f) || !(a * c - b * b > 0.f)) return 0xFu;  // not positive definite: keep
	const float thr = 2.0f * __logf(255.0f * op) + 1e-3f;
	if (thr < 0.f) return 0u;  // alpha < 1/255 even at the centre
	const float dxl = mx - (x0 + 15.0f), dxh = mx - x0;  // dx = mean.x - pixel.x over the tile's columns
	const float nbc = -b / c, nba = -b / a;
	const bool x_in = dxl <= 0.f && dxh >= 0.f;
	uint32_t mask = 0u;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const float dyl = my - (y0 + 4.0f * k + 3.0f), dyh = my - (y0 + 4.0f * k);
		float q = 0.f;
		if (!(x_in && dyl <= 0.f && dyh >= 0.f)) {
			// q(X, Y) minus a bound on its own rounding error
#define GSR_Q(X, Y) ((a * (X) * (X) + 2.f * b * (X) * (Y) + c * (Y) * (Y)) - 1e-5f * (a * (X) * (X) + fabsf(2.f * b * (X) * (Y)) + c * (Y) * (Y)))
			float X = dxl, Y = fminf(dyh, fmaxf(dyl, nbc * X));
			q = GSR_Q(X, Y);
			X = dxh; Y = fminf(dyh, fmaxf(dyl, nbc * X));
			q = fminf(q, GSR_Q(X, Y));
			Y = dyl; X = fminf(dxh, fmaxf(dxl, nba * Y));
			q = fminf(q, GSR_Q(X, Y));
			Y = dyh; X = fminf(dxh, fmaxf(dxl, nba * Y));
			q = fminf(q, GSR_Q(X, Y));
#undef GSR_Q
		}
		if (!(q > thr)) mask |= 1u << k;  // NaN -> keep
	}
	return mask;
}

// number of set bits of `mask` below this lane
__device__ __forceinline__ int gsr_mbcnt(unsigned long long mask)
{
	return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// ---- the (pixel, Gaussian) pair, evaluated identically by the forward and the backward kernel ----
// power = -0.5f * (a*dx*dx + c*dy*dy) - b*dx*dy (forward.cu:437, backward.cu:518) is evaluated in
// EXACTLY that operation order with no FMA contraction.  For elongated splats the three terms are
// O(1000) and cancel to O(1), so G = exp(power) carries a relative rounding error of up to ~1e-4
// whose value depends on the operation order; only the reference's own order reproduces the
// reference's numbers (measured: a contracted / base-2 form moved whole-Gaussian gradients by
// 1.2e-5 relative).  ax2 = (a*dx)*dx and bdx = b*dx are shared by the lane's four pixels.
// Forward and backward run the same instruction sequence on the same staged numbers, so their
// accept/reject decisions agree bit for bit (the backward undoes exactly the products the
// forward formed).
__device__ __forceinline__ float gsr_pair_power(float ax2, float bdx, float c, float dy)
{
	const float cy2 = __fmul_rn(__fmul_rn(c, dy), dy);
	return __fsub_rn(__fmul_rn(-0.5f, __fadd_rn(ax2, cy2)), __fmul_rn(bdx, dy));
}

// The same value from conic terms pre-multiplied by -0.5 (ah = -0.5 a, ch = -0.5 c; axh2 = (ah dx) dx): scaling by a
// power of two commutes with every rounding above, so the bits are those of gsr_pair_power -- one multiply less per pixel.
__device__ __forceinline__ float gsr_pair_power_halved(float axh2, float bdx, float ch, float dy)
{
	const float cyh2 = __fmul_rn(__fmul_rn(ch, dy), dy);
	return __fsub_rn(__fadd_rn(axh2, cyh2), __fmul_rn(bdx, dy));
}

// sum over the eight lanes of each 8-lane group (three DPP steps); every lane of the group ends with the group's total
__device__ __forceinline__ float gsr_sum8(float v)
{
	v += gsr_dpp_mov<0xB1, 0xF, 0xF, true>(v);   // quad_perm [1,0,3,2]
	v += gsr_dpp_mov<0x4E, 0xF, 0xF, true>(v);   // quad_perm [2,3,0,1]
	v += gsr_dpp_mov<0x141, 0xF, 0xF, true>(v);  // row_half_mirror
	return v;
}
