// render_backward.hip -- per-tile back-to-front gradient of the alpha compositing.
//
// Replaces renderCUDA<3> backward (cuda_rasterizer/backward.cu:408-601).  The reference issues
// nine fp32 atomicAdd per contributing (pixel, Gaussian) pair, all 256 lanes of a tile hitting the
// same addresses -- the slowest atomic shape on this chip (MI355X_MICROARCH.md "Global float
// atomics").  Here one wave64 owns the tile (render_common.h): per surviving instance each lane
// first adds its four pixels' nine partials in registers; the wave then folds them through LDS --
// lane l stores eight of them as row l of a 64 x 9-word area (odd row stride: 64 rows, 64 banks), the
// eight lanes of group c add column c (eight rows each) and finish with three DPP steps, the ninth
// value takes a DPP chain under that round trip (round 3; rounds 1-2 used a v_permlane32/16_swap
// butterfly on the vector ALU, the port this kernel is bound by) -- and the tile's total for this
// (Gaussian, tile) instance is written once, with plain stores, into that instance's own 48-byte
// slot (slot = its position in the depth-ordered, per-Gaussian-contiguous emission order).  The
// per-Gaussian kernel then adds each Gaussian's contiguous run of slots in a fixed order: no
// atomics, no workgroup barrier (a wave's LDS operations execute in program order), bitwise
// reproducible.  Per-pixel arithmetic is that of backward.cu:507-599; G and alpha come
// from the same roundings as in the forward kernel (gsr_pair_power; the forward's pre-halved conic terms give the same
// bits), so the products the forward formed are the ones undone here.
#include "render_common.h"
#include <hip/hip_ext.h>

#define GSR_BWD_NV 9
typedef float v2f __attribute__((ext_vector_type(2)));

GSR_TILE_CLOCK_BUFFER(gsr_backward_tile_clock, gsr_debug_tile_clock_backward)

// x + y of a pixel pair as ONE v_add_f32 on the pair's two registers.  Left to itself the compiler packs two such sums into a
// v_pk_add_f32 and pays three v_mov_b32 to line the operands up (16 instructions for eight sums instead of 8).
__device__ __forceinline__ float gsr_add_halves(v2f a)
{
	float r;
	asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a.x), "v"(a.y));
	return r;
}

__global__ void __launch_bounds__(64 * GSR_WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(4, 4))) gsr_render_backward_wave_kernel(
	int W, int H, int gx, int nslots, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const GsrSplat* __restrict__ splat, const float4* __restrict__ checkpoints, const float* __restrict__ final_C, const uint32_t* __restrict__ slot_base, const float* __restrict__ bg,
	const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ tile_max_contrib,
	const uint32_t* __restrict__ tile_order, const float* __restrict__ dL_dpixels, GsrGradSlot* __restrict__ slots,
	uint8_t* __restrict__ slot_valid, int cull)
{
	// per-wave staging of the surviving instances of a batch.  Every per-instance scalar that meets the
	// float2 pixel pairs is stored TWICE, so a ds_read_b128 delivers it as an aligned register pair ready
	// for v_pk_*_f32 (the compiler otherwise spends one v_mov per scalar per instance on the duplication).
	// (ROCm 7.2's compiler does fold a splat into op_sel / op_sel_hi for most packed operands by now; re-measured in round 3
	// with every scalar stored once -- three ds_read_b128 per instance instead of five and a dword, 119 VGPRs: 0.467 ->
	// 0.476 ms at C3, 1.616 -> 1.633 at C5, results identical.  The duplicated layout stays.)
	__shared__ float4 s_rec[GSR_WAVES_PER_WG][5][64];
	__shared__ uint32_t s_bands[GSR_WAVES_PER_WG][64];
	// transposition area of the per-instance wave reduction: lane l stores its eight partials at row l (row stride 9 words:
	// 9 is odd, so the 64 rows of one store instruction fall into 64 different banks), then the eight lanes of group c read
	// column c, eight rows each
	__shared__ float s_red[GSR_WAVES_PER_WG][64 * 9 + 8];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int slot_id = blockIdx.x * GSR_WAVES_PER_WG + wave;
	if (slot_id >= nslots) return;  // wave-uniform; no barriers below
	GSR_TILE_CLOCK_START();
	GSR_TILE_STAT(unsigned long long st_staged = 0; unsigned long long st_pairs = 0; unsigned long long st_pairs_hit = 0; unsigned long long st_reductions = 0; unsigned long long st_lanes_hit = 0;)
	// workgroups are dispatched in index order: tile_order lists the tiles by descending work, so the long tiles
	// start first and the short ones fill the end of the launch (binning.hip gsr_tile_order_kernel)
	// A heavy tile (walk of at least two checkpoint strides) comes as one entry per DEPTH SEGMENT (bits 28..31 = segment + 1):
	// this wave then walks the positions [a, b) of the list only, starting from the per-pixel (T, C) the forward left at b.
	const uint32_t entry = __builtin_amdgcn_readfirstlane(tile_order[slot_id]);
	if (entry == 0xFFFFFFFFu) return;  // unused segment entry
	const int tile = (int)(entry & 0x0FFFFFFFu);
	const int seg1 = (int)(entry >> 28);  // 0: the whole tile
	float4(*rec)[64] = s_rec[wave];
	uint32_t* recb = s_bands[wave];

	const int tx = tile % gx, ty = tile / gx;
	const int px = tx * GSR_TILE_X + (lane & 15);
	const int py0 = ty * GSR_TILE_Y + (lane >> 4);
	const v2f pfx2 = {(float)px, (float)px};
	const float x0f = (float)(tx * GSR_TILE_X), y0f = (float)(ty * GSR_TILE_Y);

	const uint2 range = ranges[tile];
	const int n = (int)min(range.y - range.x, tile_max_contrib[tile]);  // the tail was never blended
	int seg_a = 0, top = n;   // positions [seg_a, top) of the walk are this wave's
	if (seg1) {               // (binning.hip gsr_tile_segments: the same cut; its coarseness sits behind the list's last entry)
		const int coarse = (int)__builtin_amdgcn_readfirstlane(tile_order[nslots]);
		const int nblk = (n + GSR_CKPT_STRIDE - 1) / GSR_CKPT_STRIDE, m = (nblk + GSR_CKPT_MAX_SEGMENTS - 1) / GSR_CKPT_MAX_SEGMENTS * coarse;
		seg_a = (seg1 - 1) * m * GSR_CKPT_STRIDE;
		top = min(n, seg1 * m * GSR_CKPT_STRIDE);
	}
	const int nw = top - seg_a;
	const uint32_t* plist = point_list + range.x;
	const size_t plane = (size_t)H * W;
	const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
	const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

	// Per-pixel state as two float2 "pixel pairs" per lane: pair p holds pixel slots k = 2p, 2p+1
	// (rows py0 + 8p and py0 + 8p + 4).  All add/mul/fma on pairs compile to v_pk_*_f32 -- two pixels
	// per VALU issue slot, which is what bounds this kernel (PMC: VALU active ~100 % of the time).
	// 36 registers of state for 4 pixels, so that 4 waves fit per SIMD.
	v2f T[2], tfb[2];                 // running T; -T_final * (bg . dL/dpix)
	v2f ac0[2], ac1[2], ac2[2];       // accum_rec as the NEXT hit will see it
	v2f dp0[2], dp1[2], dp2[2];
	int last_contributor[GSR_PIX_PER_LANE];
	v2f pfy[2];
#pragma unroll
	for (int k = 0; k < GSR_PIX_PER_LANE; k++) {
		const int py = py0 + 4 * k;
		const bool inside = px < W && py < H;
		const uint32_t pix_id = inside ? (uint32_t)(W * py + px) : 0u;
		const float Tf = inside ? final_Ts[pix_id] : 0.f;
		const float d0 = inside ? dL_dpixels[pix_id] : 0.f;
		const float d1 = inside ? dL_dpixels[plane + pix_id] : 0.f;
		const float d2 = inside ? dL_dpixels[2 * plane + pix_id] : 0.f;
		last_contributor[k] = inside ? (int)n_contrib[pix_id] : 0;
		T[k >> 1][k & 1] = Tf;
		dp0[k >> 1][k & 1] = d0; dp1[k >> 1][k & 1] = d1; dp2[k >> 1][k & 1] = d2;
		tfb[k >> 1][k & 1] = -Tf * (bg0 * d0 + bg1 * d1 + bg2 * d2);
		pfy[k >> 1][k & 1] = (float)py;
	}
#pragma unroll
	for (int p = 0; p < 2; p++) ac0[p] = ac1[p] = ac2[p] = v2f{0.f, 0.f};
	if (top < n) {
		// The walk starts in the middle of the list.  The forward stored every pixel's (T, C) before the instance at position
		// `top`; what lies behind it blended to C_final - C, seen through T: the running T is the checkpoint's, and accum_rec --
		// the colour accumulated behind the current instance, backward.cu:553 -- is (C_final - C) / T.  A pixel whose last
		// contributor lies in front of `top` (n_contrib <= top) has nothing behind the segment: it starts from its final state,
		// T_final and 0, like a whole-tile walk -- and its checkpoint is not read (a band wave of the forward that finished early
		// wrote none).
		const float4* ck = checkpoints + ((size_t)(range.x + (uint32_t)top) / GSR_CKPT_STRIDE) * 256 + lane;
#pragma unroll
		for (int k = 0; k < GSR_PIX_PER_LANE; k++) {
			const int py = py0 + 4 * k;
			const bool inside = px < W && py < H;
			const uint32_t pix_id = inside ? (uint32_t)(W * py + px) : 0u;
			if (inside && last_contributor[k] > top) {
				const float4 c = ck[64 * k];
				const float inv = 1.0f / c.x;   // T before an instance that still contributed: > 1e-4
				T[k >> 1][k & 1] = c.x;
				ac0[k >> 1][k & 1] = (final_C[pix_id] - c.y) * inv;
				ac1[k >> 1][k & 1] = (final_C[plane + pix_id] - c.z) * inv;
				ac2[k >> 1][k & 1] = (final_C[2 * plane + pix_id] - c.w) * inv;
			}
		}
	}
	// wave-uniform: the largest n_contrib among the 128 pixels of each pair; instances at or beyond it
	// were blended into none of them, so the pair is skipped without evaluating anything
	int pair_last[2];
#pragma unroll
	for (int p = 0; p < 2; p++) {
		int m = max(last_contributor[2 * p], last_contributor[2 * p + 1]);
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
		pair_last[p] = __builtin_amdgcn_readfirstlane(m);
	}

	// back to front: batch position q = base + lane maps to range position top - 1 - q (top = n for a whole tile, the end of the segment otherwise)
	float4 ra = make_float4(0, 0, 0, 0), rb = ra, rc = ra;
	uint32_t sbase = 0u;  // first gradient slot of the staged Gaussian (dense per-Gaussian array, cache resident)
	if (lane < nw) {
		const uint32_t id = plist[top - 1 - lane];
		const float4* p = reinterpret_cast<const float4*>(splat + id);
		ra = p[0]; rb = p[1]; rc = p[2];
		sbase = slot_base[id];
	}
	uint32_t id_next = (64 + lane < nw) ? plist[top - 1 - (64 + lane)] : 0u;
	const int out_index = lane >> 3;                  // 8-lane group c ends up holding the wave total of v[c]
	float* const red_w = s_red[wave] + lane * 9;                       // this lane's row
	// column lane / 8, rows (lane % 8) + 8 k.  (Two of the eight 8-lane groups share seven banks in each of these reads: 2 conflict
	// cycles per read, SQ_LDS_BANK_CONFLICT = 25 M per launch at C3.  Rows 8 (lane % 8) + k are conflict-free and were measured:
	// 0.4665 -> 0.464 ms, and the other summation order put one needle-splat fuzz case 2 % over its end-to-end bar on
	// dL/drotations (blend sums unchanged at 1e-6): not kept.)
	const float* const red_r = s_red[wave] + (lane & 7) * 9 + (lane >> 3);
	constexpr int red_step = 72;
	const float out_scale = (out_index >= 2 && out_index <= 4) ? -0.5f : 1.0f;
	const float k01 = out_index == 0 ? -ddelx_dx : -ddely_dy;   // backward.cu:574-575: dL/dmean2D is scaled by 0.5 W / 0.5 H

	for (int base = 0; base < nw; base += 64) {
		const uint32_t bands = (base + lane < nw) ? (cull ? gsr_tile_band_mask(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, x0f, y0f) : 0xFu) : 0u;
		const bool keep = bands != 0u;
		const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
		const int cnt = __popcll(mask);
		GSR_TILE_STAT(st_staged += (unsigned)cnt;)
		if (keep) {
			const int pos = gsr_mbcnt(mask);
			const uint32_t rmin = __float_as_uint(rb.z), rwh = __float_as_uint(rb.w);
			const uint32_t slot = sbase + ((uint32_t)ty - (rmin >> 16)) * (rwh & 0xffffu) + ((uint32_t)tx - (rmin & 0xffffu));
			rec[0][pos] = make_float4(ra.x, ra.x, ra.y, ra.y);  // mean x, y
			// conic a and c pre-multiplied by -0.5: a power of two commutes with every rounding of `power`, so its bits are the
			// forward's (gsr_pair_power_halved) with one packed multiply less per pair; the epilogue undoes it inside an FMA
			rec[1][pos] = make_float4(-0.5f * ra.z, -0.5f * ra.z, ra.w, ra.w);  // -0.5 conic a, conic b
			rec[2][pos] = make_float4(-0.5f * rb.x, -0.5f * rb.x, rb.y, rb.y);  // -0.5 conic c, opacity
			rec[3][pos] = make_float4(rc.x, rc.x, rc.y, rc.y);  // r, g
			rec[4][pos] = make_float4(rc.z, rc.z, __int_as_float(top - 1 - (base + lane)), __uint_as_float(slot));  // b, position in the full range, slot
			recb[pos] = bands;
		}
		if (base + 64 + lane < nw) {
			const float4* p = reinterpret_cast<const float4*>(splat + id_next);
			ra = p[0]; rb = p[1]; rc = p[2];
			sbase = slot_base[id_next];
		}
		id_next = (base + 128 + lane < nw) ? plist[top - 1 - (base + 128 + lane)] : 0u;
		__builtin_amdgcn_wave_barrier();

		for (int j = 0; j < cnt; j++) {
			const float4 R0 = rec[0][j], R1 = rec[1][j], R2 = rec[2][j], R3 = rec[3][j], R4 = rec[4][j];
			const int contributor = __builtin_amdgcn_readfirstlane(__float_as_int(R4.z));  // backward.cu:511-515; wave-uniform
			const uint32_t bands = __builtin_amdgcn_readfirstlane(recb[j]);                 // wave-uniform
			const v2f X = {R0.x, R0.y}, Y = {R0.z, R0.w}, CA = {R1.x, R1.y}, CB = {R1.z, R1.w}, CC = {R2.x, R2.y}, OP = {R2.z, R2.w};
			const v2f C0 = {R3.x, R3.y}, C1 = {R3.z, R3.w}, C2 = {R4.x, R4.y};
			const v2f dx = X - pfx2;
			const v2f ax2 = (CA * dx) * dx, bdx = CB * dx;  // this file is compiled with -ffp-contract=off
			// per-lane partial sums over its pixels (one float2 = two pixels, added at the end).  The
			// geometric terms are kept as raw moments of f = G * dL/dG (sum f dx, f dy, f dx^2, f dx dy,
			// f dy^2); the conic and the 0.5*W / 0.5*H / -0.5 factors of backward.cu:574-594 are applied
			// once per instance after the reduction.
			v2f acc[GSR_BWD_NV];
#pragma unroll
			for (int i = 0; i < GSR_BWD_NV; i++) acc[i] = v2f{-0.f, -0.f};  // x + (-0) is x for every x: the first pair's sums need no add
			unsigned long long any = 0ull;  // lanes with a hit, kept as a scalar mask: the loop's branches test masks, not ballots of bools
#pragma unroll
			for (int p = 0; p < 2; p++) {
				if (!(bands & (3u << (2 * p))) || contributor >= pair_last[p]) continue;  // scalar branch: no band of this pair can be reached
				// power = -0.5f * (a*dx*dx + c*dy*dy) - b*dx*dy in the reference's operation order (CA, CC carry the -0.5)
				const v2f dy = Y - pfy[p];
				const v2f power = (ax2 + (CC * dy) * dy) - bdx * dy;
				// __expf(x) = v_exp_f32(x * log2(e)) (the forward's form, same constant, same IEEE product): the two products as one
				// packed multiply
				const v2f pl = power * 1.44269504088896340736f;
				const v2f G = {__builtin_amdgcn_exp2f(pl.x), __builtin_amdgcn_exp2f(pl.y)};
				const v2f og = OP * G;
				const v2f araw = {fminf(0.99f, og.x), fminf(0.99f, og.y)};
				const bool c0 = contributor < last_contributor[2 * p], p0 = !(power.x > 0.0f), a0 = !(araw.x < 1.0f / 255.0f);
				const bool c1 = contributor < last_contributor[2 * p + 1], p1 = !(power.y > 0.0f), a1 = !(araw.y < 1.0f / 255.0f);
				const bool hit0 = c0 && p0 && a0, hit1 = c1 && p1 && a1;
				// each ballot of ONE comparison is that comparison's own lane mask (no instruction); a ballot of the combined
				// bool costs a v_cndmask + v_cmp round trip through a VGPR
				const unsigned long long hits = (__builtin_amdgcn_ballot_w64(c0) & __builtin_amdgcn_ballot_w64(p0) & __builtin_amdgcn_ballot_w64(a0)) |
				                                (__builtin_amdgcn_ballot_w64(c1) & __builtin_amdgcn_ballot_w64(p1) & __builtin_amdgcn_ballot_w64(a1));
				GSR_TILE_STAT(st_pairs++;)
				if (hits == 0ull) continue;  // wave-uniform
				GSR_TILE_STAT(st_pairs_hit++; st_lanes_hit += (unsigned)__popcll(hits);)
				any |= hits;  // (lanes without a hit add exact zeros below)
				// A pixel that did not hit runs the same update with alpha = 0, which is the identity on its state
				// bit for bit (1 - 0 = 1, rcp(1) = 1, T * 1 = T, 0 * c + 1 * acc = acc): no per-state selects
				const v2f alpha = {hit0 ? araw.x : 0.f, hit1 ? araw.y : 0.f};
				const v2f oma = 1.f - alpha;
				const v2f inv1ma = {__builtin_amdgcn_rcpf(oma.x), __builtin_amdgcn_rcpf(oma.y)};  // 1 ulp; IEEE division changed no parity figure
				const v2f Tn = T[p] * inv1ma;
				// accum_rec and (c - accum_rec) cancel heavily when neighbouring colours are close: reference
				// operation order, no contraction (backward.cu:553-559)
				// dL/dalpha = sum_c (colour_c - accum_rec_c) * dL/dpix_c (backward.cu:553-559), as an FMA chain on the three
				// differences.  The differences cancel heavily when neighbouring colours are close, so they are formed first and
				// exactly as the reference forms them; what follows only accumulates.
				const v2f d0 = C0 - ac0[p], d1 = C1 - ac1[p], d2 = C2 - ac2[p];
				v2f dL_dalpha = __builtin_elementwise_fma(d2, dp2[p], __builtin_elementwise_fma(d1, dp1[p], d0 * dp0[p]));
				// accum_rec' = last_alpha * last_color + (1 - last_alpha) * accum_rec (backward.cu:553; the reference applies it
				// lazily at the NEXT hit, from the same operands), written as accum_rec + alpha * (colour - accum_rec) on the
				// difference above: one FMA per channel instead of two products and a sum, and the smaller rounding error of the
				// two forms.  (Measured: 0.577 -> 0.534 ms for the kernel, all parity bars kept -- the blend sums stay at
				// 4e-7 ... 1.4e-6 of the CPU oracle's double-precision sums on C2 / C3.  Two further shortcuts were measured and
				// rejected: f = (o G) dL/dalpha instead of (o dL/dalpha) G saves nothing and moves the needle-splat stress case to
				// 1.03e-5.)
				const v2f n0 = __builtin_elementwise_fma(alpha, d0, ac0[p]);
				const v2f n1 = __builtin_elementwise_fma(alpha, d1, ac1[p]);
				const v2f n2 = __builtin_elementwise_fma(alpha, d2, ac2[p]);
				dL_dalpha = __builtin_elementwise_fma(dL_dalpha, Tn, tfb[p] * inv1ma);
				// zero the partials of the pixel that did not hit (dL_dalpha of such a pixel is not zero by itself)
				const v2f dla = {hit0 ? dL_dalpha.x : 0.f, hit1 ? dL_dalpha.y : 0.f};
				const v2f dch = alpha * Tn;             // dchannel_dcolor; 0 without a hit
				T[p] = Tn;
				ac0[p] = n0;
				ac1[p] = n1;
				ac2[p] = n2;
				acc[6] = __builtin_elementwise_fma(dch, dp0[p], acc[6]);
				acc[7] = __builtin_elementwise_fma(dch, dp1[p], acc[7]);
				acc[8] = __builtin_elementwise_fma(dch, dp2[p], acc[8]);
				acc[5] = __builtin_elementwise_fma(G, dla, acc[5]);  // dL/dopacity
				const v2f f = (OP * dla) * G;                         // dL/dG * G
				const v2f fdx = f * dx, fdy = f * dy;
				acc[0] += fdx;
				acc[1] += fdy;
				acc[2] = __builtin_elementwise_fma(fdx, dx, acc[2]);
				acc[3] = __builtin_elementwise_fma(fdx, dy, acc[3]);
				acc[4] = __builtin_elementwise_fma(fdy, dy, acc[4]);
			}
			if (any) {  // wave-uniform
				GSR_TILE_STAT(st_reductions++;)
				float v[GSR_BWD_NV];
#pragma unroll
				for (int i = 0; i < GSR_BWD_NV; i++) v[i] = gsr_add_halves(acc[i]);
				// the wave reduction of the first eight values through LDS: 64 x 8 partials in, transposed out -- the cross-lane work
				// is LDS traffic (its own issue port) plus 7 adds and the 3 DPP steps inside an 8-lane group, instead of 6 lane swaps,
				// 6 adds, 2 selects and 4 DPP steps on the vector ALU, which is what bounds this kernel.  A wave's LDS operations
				// execute in program order: the loads below see all 64 rows, and the next instance's stores come after them.
#pragma unroll
				for (int i = 0; i < 8; i++) red_w[i] = v[i];
				float col[8];
#pragma unroll
				for (int k = 0; k < 8; k++) col[k] = red_r[red_step * k];
				// the ninth value's DPP chain runs while the LDS round trip is under way
				__builtin_amdgcn_sched_barrier(0);
				const float t9 = gsr_wave_sum_to_lane63(v[8]);  // lane 63 holds the total of v[8]
				__builtin_amdgcn_sched_barrier(0);
				const float tcol = ((col[0] + col[1]) + (col[2] + col[3])) + ((col[4] + col[5]) + (col[6] + col[7]));
				const float t8 = gsr_sum8(tcol);                // group c holds the total of v[c]
				// dL/dmean2D: group 0 (sum f dx = sx) needs sy, group 1 (sy) needs sx -- the partner half row, one DPP move; the
				// same products and the same FMA as the scalar form below (a sx + b sy with a = -2 (-0.5 a)), so the same bits
				const float other = gsr_dpp_mov<0x128, 0xF, 0xF, true>(t8);  // row_ror:8
				const float r01 = k01 * __builtin_fmaf(-2.0f, (out_index == 0 ? CA.x : CC.x) * t8, CB.x * other);
				const uint32_t slot = __builtin_amdgcn_readfirstlane(__float_as_uint(R4.w));
				float* out = reinterpret_cast<float*>(slots + slot);
				const float r = out_index < 2 ? r01 : out_scale * t8;   // dL/dconic .x .y .w (x -0.5); opacity and colour as they are
				if ((lane & 7) == 0) out[out_index] = r;
				if (lane == 63) {
					out[8] = t9;
					slot_valid[slot] = 1;
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
	GSR_TILE_CLOCK_STOP(gsr_backward_tile_clock, slot_id, lane, st_staged | (st_pairs << 20) | (st_pairs_hit << 42), st_reductions | (st_lanes_hit << 24));   // one record per dispatch entry
}

void gsr_launch_render_backward(int W, int H, GsrImage img, const uint32_t* point_list, const GsrSplat* splat, const float4* checkpoints,
                                const uint32_t* slot_base, const float* bg, const float* dL_dpix, GsrGradSlot* slots,
                                uint8_t* slot_valid, bool cull, hipStream_t s, hipEvent_t t_start, hipEvent_t t_stop)
{
	const int gx = gsr_grid_x(W), gy = gsr_grid_y(H);
	const int ntiles = gx * gy;
	const int nslots = ntiles + (int)gsr_tile_order_max_segments(ntiles);   // whole tiles + the extra entries of heavy tiles' depth segments
	const int nwg = (nslots + GSR_WAVES_PER_WG - 1) / GSR_WAVES_PER_WG;
	if (t_start || t_stop) {
		hipExtLaunchKernelGGL(gsr_render_backward_wave_kernel, dim3(nwg), dim3(64 * GSR_WAVES_PER_WG), 0, s, t_start, t_stop, 0, W, H, gx, nslots,
		                      img.ranges, point_list, splat, checkpoints, img.final_C, slot_base, bg, img.final_T, img.n_contrib, img.tile_max_contrib,
		                      img.tile_order, dL_dpix, slots, slot_valid, cull ? 1 : 0);
		return;
	}
	hipLaunchKernelGGL(gsr_render_backward_wave_kernel, dim3(nwg), dim3(64 * GSR_WAVES_PER_WG), 0, s, W, H, gx, nslots,
	                   img.ranges, point_list, splat, checkpoints, img.final_C, slot_base, bg, img.final_T, img.n_contrib, img.tile_max_contrib,
	                   img.tile_order, dL_dpix, slots, slot_valid, cull ? 1 : 0);
}
