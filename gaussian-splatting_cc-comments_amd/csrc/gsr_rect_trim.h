// gsr_rect_trim.h -- which tiles of its rectangle a Gaussian can reach at all, and how the binning is told.
//
// The reference bins a Gaussian into every tile of the square of 3 sigma (radius rounded up) around its centre
// (getRect, auxiliary.h:42-58; duplicateWithKeys, rasterizer_impl.cu:78-126), and the blend loops then find, pixel by pixel, that
// most of those tiles hold no pixel with alpha >= 1 / 255 (forward.cu:449-456, backward.cu:527-534): the splat is an ellipse, its
// opacity lowers the threshold, and the square is rounded up to tiles.  At 1 M Gaussians / 1080p 9.2 M instances are binned and 53 %
// of them can contribute somewhere in their tile.  An instance that contributes to no pixel of its tile changes nothing: not the
// pixels' T or colour, not n_contrib's meaning for the backward (the position of the last contributor in the list the backward walks
// -- the same list), no gradient.  So the column-pair binning (tilebin.hip) leaves out tiles that the ellipse
//     conic_a dx^2 + 2 conic_b dx dy + conic_c dy^2 <= 2 ln(255 opacity)            (alpha >= 1 / 255  <=>  power >= -ln(255 opacity))
// provably misses, per tile COLUMN of the rectangle: up to three tile rows off the top and up to three off the bottom of every
// column (two bits each: a nibble per column, eight columns in one word), empty columns off the left and right edge.  A rectangle
// wider than eight tiles shares a nibble among 2, 4, ... 32 neighbouring columns, one taller than 16 tiles counts the rows in units
// of 2, 4, ... 16: coarser, never wrong.  What is kept is a superset of what can contribute (the bound is taken over the column's whole
// strip of pixel centres, with margins far above the fp32 error of `power` and of exp), so every output -- image, radii, gradients --
// has the same bits with and without it (GSR_DEBUG_NO_TRIM; tests/test_trim_gpu.py, tools/determinism_stress.py), and the slots of
// the left-out instances are exactly slots the backward never marks valid.  num_rendered, tiles_touched and the gradient slots keep
// the reference's numbering over the WHOLE rectangle; point_list and the tile ranges are the reference's with the left-out instances
// removed (the reference's own, bit for bit, with GSR_DEBUG_NO_TRIM and on the tile-sort path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// the rectangle in one word (tile coordinates < 256, sizes 1 .. 256); no tiles: GSR_RECT_NONE
#define GSR_RECT_NONE 0xFFFFFFFFu
__host__ __device__ __forceinline__ uint32_t gsr_rect_pack(uint32_t x0, uint32_t y0, uint32_t w, uint32_t h)
{
	return (w == 0u || h == 0u) ? GSR_RECT_NONE : (x0 | (y0 << 8) | ((w - 1u) << 16) | ((h - 1u) << 24));
}
__host__ __device__ __forceinline__ void gsr_rect_unpack(uint32_t r, uint32_t& x0, uint32_t& y0, uint32_t& w, uint32_t& h)
{
	if (r == GSR_RECT_NONE) { x0 = y0 = w = h = 0u; return; }
	x0 = r & 0xffu; y0 = (r >> 8) & 0xffu; w = ((r >> 16) & 0xffu) + 1u; h = (r >> 24) + 1u;
}

#define GSR_TRIM_GROUPS 8        // nibbles of a trim word: column groups of a rectangle
#define GSR_TRIM_MAX_UNITS 3     // row units it can take off either end of a column group

// log2 of the columns that share a nibble (w <= 8: 0) and of the rows per unit (h <= 16: 0): the smallest powers of two with
// at most 8 groups / at most 16 units
// (w <= 8 * 2^s  <=>  s >= log2(w) - 3: no loop -- pass 1's scatter decodes a word per Gaussian, and a data-dependent loop there cost it 10 us)
__host__ __device__ __forceinline__ uint32_t gsr_trim_col_shift(uint32_t w) { return w <= 8u ? 0u : 29u - (uint32_t)__builtin_clz(w - 1u); }
__host__ __device__ __forceinline__ uint32_t gsr_trim_row_shift(uint32_t h) { return h <= 16u ? 0u : 28u - (uint32_t)__builtin_clz(h - 1u); }

// rows taken off column c (< w) of a w x h rectangle: top (small y), bottom.  col_shift / row_shift: gsr_trim_col_shift(w), gsr_trim_row_shift(h)
__host__ __device__ __forceinline__ void gsr_trim_of(uint32_t trim, uint32_t c, uint32_t col_shift, uint32_t row_shift, uint32_t& top, uint32_t& bottom)
{
	const uint32_t nib = (trim >> (4u * ((c >> col_shift) & 7u))) & 15u;
	top = (nib & 3u) << row_shift;
	bottom = (nib >> 2) << row_shift;
}

// the columns of a w x h rectangle that keep at least one row: [lead, lead + wt).  (The producer leaves no empty column group between
// two kept ones -- gsr_rect_trim -- so that the kept columns are one run: the binning's histograms are difference arrays.)
__host__ __device__ __forceinline__ void gsr_trim_columns(uint32_t trim, uint32_t w, uint32_t h, uint32_t& lead, uint32_t& wt)
{
	lead = 0u; wt = w;
	if (trim == 0u || w == 0u) return;
	const uint32_t cs = gsr_trim_col_shift(w), rs = gsr_trim_row_shift(h);
	const uint32_t groups = (w + (1u << cs) - 1u) >> cs;
	uint32_t first = groups, last = 0u;
#pragma unroll
	for (uint32_t g = 0; g < (uint32_t)GSR_TRIM_GROUPS; g++) {
		uint32_t t, b;
		gsr_trim_of(trim, g << cs, cs, rs, t, b);
		if (g < groups && t + b < h) { first = first < g ? first : g; last = g + 1u; }
	}
	if (first >= last) { lead = 0u; wt = 0u; return; }
	lead = first << cs;
	wt = ((last << cs) < w ? (last << cs) : w) - lead;
}

#ifdef __HIPCC__
// The trim word of one Gaussian: centre (px, py) in pixel coordinates, conic (ca, cb, cc) and opacity as the blend kernels will read
// them from its record, rectangle [x0, x0 + w) x [y0, y0 + h) in tiles.  0 = nothing is taken off.
__device__ __forceinline__ uint32_t gsr_rect_trim(float px, float py, float ca, float cb, float cc, float opacity, int x0, int y0, int w, int h)
{
	if (w <= 0 || h <= 0) return 0u;
	const uint32_t cs = gsr_trim_col_shift((uint32_t)w), rs = gsr_trim_row_shift((uint32_t)h);
	const int groups = (int)(((uint32_t)w + (1u << cs) - 1u) >> cs);
	// (v_sqrt_f32 / v_rcp_f32 / v_log_f32 as they are, 1 ulp: the margins below are a thousand times that; the IEEE expansions of
	// sqrtf and of the division cost the geometry kernel, which the whole binning chain waits for, 7 us at 1 M Gaussians)
	const float det = ca * cc - cb * cb;
	// q = ca dx^2 + 2 cb dx dy + cc dy^2 <= tau is necessary for alpha >= 1 / 255; the margin (0.1 % + 0.01) is thousands of times the
	// error of the kernels' fp32 `power` and exp at these magnitudes (tau <= 11.1)
	const float tau = 2.0f * __logf(255.0f * opacity) * 1.001f + 0.01f;
	if (!(det > 0.0f) || !(ca > 0.0f) || !(cc > 0.0f) || !(tau == tau)) return 0u;   // not an ellipse this bound understands: keep everything
	uint32_t trim = 0u;
	uint32_t first = (uint32_t)groups, last = 0u;
	if (tau > 0.0f) {
		const float inv_det = __builtin_amdgcn_rcpf(det);
		const float ex = __builtin_amdgcn_sqrtf(tau * cc * inv_det) * 1.0001f + 0.01f;      // half width of the ellipse
		const float eyy = __builtin_amdgcn_sqrtf(tau * ca * inv_det) * 1.0001f + 0.01f;     // half height
		const float xs = -cb * __builtin_amdgcn_sqrtf(tau * inv_det * __builtin_amdgcn_rcpf(ca));                // dx of the ellipse's bottom point (largest dy); its top point: -xs
		const float inv_cc = __builtin_amdgcn_rcpf(cc);
#pragma unroll
		for (int c = 0; c < GSR_TRIM_GROUPS; c++) {
			if (c >= groups) break;
			// the strip of pixel centres of the tile columns of group c, relative to the centre, widened by the margin
			const int c0 = x0 + (c << cs), c1 = min(x0 + w, x0 + ((c + 1) << cs));   // tile columns [c0, c1)
			float lo = (float)(c0 * GSR_TILE_X) - px - 0.01f, hi = (float)(c1 * GSR_TILE_X - 1) - px + 0.01f;
			uint32_t t = GSR_TRIM_MAX_UNITS, b = GSR_TRIM_MAX_UNITS;   // the ellipse misses the strip: as much off as the word can say
			if (!(lo > ex) && !(hi < -ex)) {   // (written so that a NaN anywhere keeps the column whole: every comparison below is false then)
				lo = fmaxf(lo, -ex); hi = fminf(hi, ex);
				// largest dy over the strip: the ellipse's bottom point if its dx lies inside, else at the nearer end (the boundary is concave)
				const float rlo = __builtin_amdgcn_sqrtf(fmaxf(0.0f, tau * cc - det * lo * lo)), rhi = __builtin_amdgcn_sqrtf(fmaxf(0.0f, tau * cc - det * hi * hi));
				float ymax = fmaxf((-cb * lo + rlo) * inv_cc, (-cb * hi + rhi) * inv_cc);
				float ymin = fminf((-cb * lo - rlo) * inv_cc, (-cb * hi - rhi) * inv_cc);
				if (lo <= xs && xs <= hi) ymax = eyy;
				if (lo <= -xs && -xs <= hi) ymin = -eyy;
				ymax = ymax * 1.0001f + 0.01f;
				ymin = ymin * 1.0001f - 0.01f;
				// tile rows that hold a pixel centre in [py + ymin, py + ymax] -- or fewer: rows of the bounds themselves
				const float r0 = floorf((py + ymin) * (1.0f / GSR_TILE_Y)), r1 = floorf((py + ymax) * (1.0f / GSR_TILE_Y));
				// (fmaxf / fminf return their other operand for a NaN: 0 rows then)
				const float unit = 1.0f / (float)(1u << rs);   // whole units only: rounded down
				const float tf = fminf(floorf(fmaxf(r0 - (float)y0, 0.0f) * unit), (float)GSR_TRIM_MAX_UNITS);
				const float bf = fminf(floorf(fmaxf((float)(y0 + h - 1) - r1, 0.0f) * unit), (float)GSR_TRIM_MAX_UNITS);
				t = (uint32_t)tf; b = (uint32_t)bf;
			}
			trim |= (t | (b << 2)) << (4 * c);
			if (((t + b) << rs) < (uint32_t)h) { first = first < (uint32_t)c ? first : (uint32_t)c; last = (uint32_t)c + 1u; }
		}
	} else {
		// opacity below 1 / 255 (with the margin): alpha >= 1 / 255 nowhere
#pragma unroll
		for (int c = 0; c < GSR_TRIM_GROUPS; c++)
			if (c < groups) trim |= 15u << (4 * c);
	}
	// no empty column group between two kept ones (cannot happen for a convex shape; the decoder relies on it): give such a group back whole
#pragma unroll
	for (int c = 0; c < GSR_TRIM_GROUPS; c++)
		if ((uint32_t)c > first && (uint32_t)c + 1u < last) {
			const uint32_t nib = (trim >> (4 * c)) & 15u;
			if ((((nib & 3u) + (nib >> 2)) << rs) >= (uint32_t)h) trim &= ~(15u << (4 * c));
		}
	return trim;
}
#endif
