// trim_decode.cpp -- the binning's reading of a trim word (csrc/gsr_rect_trim.h: gsr_rect_unpack, gsr_trim_columns, gsr_trim_of), run on the
// host: for every "packed_rect trim" pair on stdin prints the rectangle's kept tiles as one line of w * h characters ('1' kept, '0' left
// out; row-major).  tests/test_library_cpu.py compares it with the numpy restatement the GPU parity tests use as their expectation.
#include <cstdio>
#include <cstdint>
#define GSR_TILE_X 16
#define GSR_TILE_Y 16
#include "../../gaussian-splatting_cc-comments_amd/csrc/gsr_rect_trim.h"

int main()
{
	unsigned long long p, t;
	while (scanf("%llu %llu", &p, &t) == 2) {
		uint32_t x0, y0, w, h, lead, wt;
		gsr_rect_unpack((uint32_t)p, x0, y0, w, h);
		gsr_trim_columns((uint32_t)t, w, h, lead, wt);
		const uint32_t cs = gsr_trim_col_shift(w), rs = gsr_trim_row_shift(h);
		for (uint32_t r = 0; r < h; r++)
			for (uint32_t c = 0; c < w; c++) {
				bool kept = c >= lead && c < lead + wt;
				if (kept) {   // what pass 1's scatter writes for the column: rows [y0 + top, y0 + h - bottom), the whole column if that is empty
					uint32_t top, bottom;
					gsr_trim_of((uint32_t)t, c, cs, rs, top, bottom);
					if (top + bottom > h - 1u) top = bottom = 0u;
					kept = r >= top && r < h - bottom;
				}
				putchar(kept ? '1' : '0');
			}
		putchar('\n');
	}
	return 0;
}
