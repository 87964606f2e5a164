// render_common.h -- pieces shared by the forward and backward blend kernels.
//
// Work decomposition (both directions): ONE wave64 owns one 16x16 tile; lane l owns the four
// pixels (x = l & 15, y = (l >> 4) + 4k), k = 0..3.  A wave needs no workgroup barrier, the
// per-instance LDS broadcast is amortised over four pixels, and in the backward pass three of the
// four cross-lane reduction levels of the reference's 256-thread block become plain register adds.
// A 256-thread workgroup simply carries four independent tiles.
#pragma once
#include "gsr_internal.h"

#define GSR_WAVES_PER_WG 4
#define GSR_PIX_PER_LANE 4

// Conservative, exact-result-preserving tile culling.  alpha = o*exp(-q/2) with
// q = a dx^2 + 2 b dx dy + c dy^2 can reach 1/255 inside the tile only if
// min over the tile's pixel rectangle of q <= 2 ln(255 o).  The minimum of the convex quadratic
// over the rectangle is 0 when the centre lies inside it and otherwise sits on one of the four
// edges, where it has a closed form.  A slack of 1e-3 in q (5e-4 relative in alpha, four orders
// of magnitude above fp32 rounding of `power`) keeps the test conservative, so an instance that
// the reference would blend into any pixel of this tile is never dropped; dropped instances
// contribute exactly nothing in the reference either (forward.cu:438-447, backward.cu:521-531).
__device__ __forceinline__ bool gsr_tile_may_hit(float mx, float my, float a, float b, float c, float op, float x0,
                                                 float y0)
{
	const float thr = 2.0f * __logf(255.0f * op) + 1e-3f;
	if (!(a > 0.f) || !(c > 0.f) || !(a * c - b * b > 0.f)) return true;  // not positive definite: keep
	const float dxl = mx - (x0 + 15.0f), dxh = mx - x0;  // dx = mean.x - pixel.x over the tile
	const float dyl = my - (y0 + 15.0f), dyh = my - y0;
	if (dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f) return !(thr < 0.f);
	float q = 3.0e38f;
	const float nbc = -b / c, nba = -b / a;
	{
		float X = dxl, Y = fminf(dyh, fmaxf(dyl, nbc * X));
		q = fminf(q, a * X * X + 2.f * b * X * Y + c * Y * Y);
		X = dxh; Y = fminf(dyh, fmaxf(dyl, nbc * X));
		q = fminf(q, a * X * X + 2.f * b * X * Y + c * Y * Y);
	}
	{
		float Y = dyl, X = fminf(dxh, fmaxf(dxl, nba * Y));
		q = fminf(q, a * X * X + 2.f * b * X * Y + c * Y * Y);
		Y = dyh; X = fminf(dxh, fmaxf(dxl, nba * Y));
		q = fminf(q, a * X * X + 2.f * b * X * Y + c * Y * Y);
	}
	return !(q > thr);  // NaN -> keep
}

// number of set bits of `mask` below this lane
__device__ __forceinline__ int gsr_mbcnt(unsigned long long mask)
{
	return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
