// Probe: what does one gfx950 SIMD sustain per clock for the VALU instruction kinds the blend kernels are
// made of, at 1 / 2 / 4 / 8 waves per SIMD (one 256*w-thread workgroup per CU -- two of 1024 threads for w = 8 --
// so that every SIMD holds exactly w waves for the whole run)?  Calibrates bench.py's issue-fraction metric
// (profiles/r2_valu_probe.txt).  Every wave runs ITERS x 64 independent instructions of one kind (8
// accumulators, so dependent-issue latency is not what is measured); the clock is measured in the kernel
// (s_memtime ticks per s_memrealtime tick, 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_probe tools/valu_probe.hip && tools/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 2048

enum { K_FMA, K_PKFMA, K_EXP, K_RCP, K_MIX, K_CNDMASK, K_DPP, K_CNDMASK_S, K_CMP_CND, K_PKMUL, K_CMP, K_CMP_S, K_MOV, K_MAX, K_PERM32, K_MUL2, K_FMAC2, K_BLEND, K_ADD2, K_PKADD, K_FWDMIX, K_BWDMIX, K_NKINDS };
static const char* KNAME[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "7 v_fma + 1 v_exp", "v_cndmask_b32 vcc", "v_add_f32 dpp", "v_cndmask_b32 sgpr", "v_cmp + v_cndmask", "v_pk_mul_f32", "v_cmp_gt_f32 vcc", "v_cmp_gt_f32 sgpr", "v_mov_b32", "v_max_f32", "v_permlane32_swap", "v_mul_f32_e32", "v_fmac_f32_e32", "forward-blend mix", "v_add_f32_e32", "v_pk_add_f32", "forward blend class mix", "backward blend class mix"};

template <int KIND>
__global__ void __launch_bounds__(1024) probe(float* out, unsigned long long* stamps, unsigned long long* sched)
{
	extern __shared__ float pad[];  // sized by the host so that exactly w single-wave workgroups fit per SIMD
	float a[8];
	v2f p[8];
#pragma unroll
	for (int i = 0; i < 8; i++) { a[i] = 1.0f + threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] + 0.5f}; }
	const float b = 0.999f, c = 1e-3f, binv = 1.0f / 0.999f;
	const v2f b2 = {b, b}, c2 = {c, c}, binv2 = {binv, binv};
	const unsigned long long smask = __builtin_amdgcn_ballot_w64(threadIdx.x & 1);
	unsigned long long sm[4] = {0, 0, 0, 0};
	const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	for (int it = 0; it < ITERS; it++) {
#pragma unroll
		for (int u = 0; u < 8; u++) {
#pragma unroll
			for (int i = 0; i < 8; i++) {
				if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
				else if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(b2), "v"(c2));
				else if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
				else if (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
				else if (KIND == K_MIX) {
					if (i == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
					else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
				} else if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
				else if (KIND == K_CNDMASK_S) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(smask));
				else if (KIND == K_CMP_CND) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
				else if (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(b2));
				else if (KIND == K_CMP) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
				else if (KIND == K_CMP_S) asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(sm[i & 3]) : "v"(a[i]), "v"(b));
				else if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
				else if (KIND == K_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				else if (KIND == K_PERM32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
				else if (KIND == K_MUL2) { if (u & 1) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b)); else asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(binv)); }
				else if (KIND == K_FWDMIX) {
					// 22 instructions in the class proportions the counters report for gsr_render_forward_wave_kernel (profiles/r3_pmc_summary.json):
					// 4 add / sub, 7 mul, 3 fmac, 1 exp, 7 "other" (1 min, 3 compares into SGPR pairs, 3 selects by SGPR masks)
					asm volatile("v_sub_f32_e32 %0, %0, %2\n\tv_mul_f32_e32 %0, %1, %0\n\tv_mul_f32_e32 %0, %3, %0\n\tv_add_f32_e32 %0, %2, %0\n\t"
					             "v_mul_f32_e32 %0, %1, %0\n\tv_sub_f32_e32 %0, %0, %2\n\tv_mul_f32_e32 %0, %3, %0\n\tv_exp_f32_e32 %0, %0\n\t"
					             "v_mul_f32_e32 %0, %1, %0\n\tv_min_f32_e32 %0, %1, %0\n\tv_sub_f32_e32 %0, %0, %2\n\tv_mul_f32_e32 %0, %3, %0"
					             : "+v"(a[i]) : "v"(b), "v"(c), "v"(binv));
					asm volatile("v_cmp_gt_f32 %0, %1, %2\n\tv_cmp_lt_f32 %0, %1, %3\n\tv_cmp_gt_f32 %0, %2, %1\n\tv_mul_f32_e32 %1, %2, %1\n\t"
					             "v_cndmask_b32 %1, %1, %2, %4\n\tv_fmac_f32_e32 %1, %2, %3\n\tv_fmac_f32_e32 %1, %3, %2\n\tv_fmac_f32_e32 %1, %2, %3\n\t"
					             "v_cndmask_b32 %1, %1, %3, %4\n\tv_cndmask_b32 %1, %2, %1, %4"
					             : "=&s"(sm[i & 3]), "+v"(a[i]) : "v"(b), "v"(c), "s"(smask));
				} else if (KIND == K_BWDMIX) {
					// 25 instructions in the class proportions of gsr_render_backward_wave_kernel: 7 adds (3 packed, 2 plain, 2 DPP), 6 multiplies
					// (5 packed), 3 packed FMAs, 1 transcendental, 8 "other" (3 compares, 3 selects, 1 min, 1 move)
					asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_mul_f32 %0, %0, %3\n\tv_pk_add_f32 %0, %0, %2\n\t"
					             "v_pk_mul_f32 %0, %0, %1\n\tv_pk_fma_f32 %0, %0, %3, %2\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_fma_f32 %0, %0, %3, %2\n\t"
					             "v_pk_add_f32 %0, %0, %2\n\tv_pk_mul_f32 %0, %0, %3\n\tv_pk_fma_f32 %0, %0, %1, %2"
					             : "+v"(p[i]) : "v"(b2), "v"(c2), "v"(binv2));
					asm volatile("v_add_f32_e32 %1, %3, %1\n\tv_rcp_f32_e32 %1, %1\n\tv_mul_f32_e32 %1, %2, %1\n\tv_add_f32_e32 %1, %3, %1\n\t"
					             "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_min_f32_e32 %1, %2, %1\n\t"
					             "v_cmp_gt_f32 %0, %1, %2\n\tv_cmp_lt_f32 %0, %1, %3\n\tv_cmp_gt_f32 %0, %2, %1\n\t"
					             "v_cndmask_b32 %1, %1, %2, %4\n\tv_cndmask_b32 %1, %1, %3, %4\n\tv_cndmask_b32 %1, %2, %1, %4\n\t"
					             "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_e32 %1, %1"
					             : "=&s"(sm[i & 3]), "+v"(a[i]) : "v"(b), "v"(c), "s"(smask));
				}
				else if (KIND == K_ADD2) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
				else if (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
				else if (KIND == K_FMAC2) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
				else if (KIND == K_BLEND) {
					// the instruction classes of one band of the forward blend (render_forward.hip), 8 instructions per accumulator
					// per round: 5 VOP2 arithmetic, 1 compare into an SGPR pair, 1 select by an SGPR mask, and every other accumulator
					// a v_exp_f32 in place of one multiply (1 transcendental per 16, the kernel has 1 per 22)
					asm volatile("v_sub_f32_e32 %0, %0, %1\n\tv_mul_f32_e32 %0, %1, %0\n\tv_fmac_f32_e32 %0, %1, %2\n\tv_mul_f32_e32 %0, %1, %0"
					             : "+v"(a[i]) : "v"(b), "v"(c));
					if (i & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
					else asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
					asm volatile("v_cmp_gt_f32 %0, %1, %2\n\tv_cndmask_b32 %1, %1, %2, %3\n\tv_fmac_f32_e32 %1, %2, %4"
					             : "=&s"(sm[i & 3]), "+v"(a[i]) : "v"(b), "s"(smask), "v"(c));
				}
				else if (KIND == K_DPP) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
			}
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	float s = 0.f;
#pragma unroll
	for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
	s += (float)((sm[0] ^ sm[1] ^ sm[2] ^ sm[3]) & 1ull);
	out[blockIdx.x * blockDim.x + threadIdx.x] = s + pad[0] * 0.f;
	if ((threadIdx.x & 63) == 0) {
		const int wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
		stamps[2 * wv] = t1 - t0; stamps[2 * wv + 1] = r1 - r0;
		// where and when the wave ran: absolute 100 MHz times and the hardware ids
		sched[4 * wv] = r0; sched[4 * wv + 1] = r1;
		sched[4 * wv + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
		sched[4 * wv + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
	}
}

template <int KIND>
static void run(int w, float* out, unsigned long long* stamps, unsigned long long* h, unsigned long long* sched)
{
	// w workgroups of 256 threads (one wave per SIMD each) per CU: a workgroup's LDS is sized so that exactly w fit a CU's 160 KB
	const int per_cu = w, threads = 256;
	const int nwg = 256 * per_cu, nwaves = nwg * threads / 64;
	const size_t lds = (size_t)(160 * 1024) / (w + 1) + 1024;
	const double per_wave_instr = (KIND == K_BLEND) ? 8.0 : (KIND == K_FWDMIX ? 22.0 : (KIND == K_BWDMIX ? 25.0 : (KIND == K_CMP_CND ? 2.0 : 1.0)));
	hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(probe<KIND>, dim3(nwg), dim3(threads), lds, 0, out, stamps, sched);  // warm-up
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(probe<KIND>, dim3(nwg), dim3(threads), lds, 0, out, stamps, sched);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0.f;
	hipEventElapsedTime(&ms, e0, e1);
	hipMemcpy(h, stamps, (size_t)nwaves * 16, hipMemcpyDeviceToHost);
	double cyc = 0, real = 0;
	for (int i = 0; i < nwaves; i++) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
	const double clock_ghz = cyc / real * 0.1;
	const double instr_per_wave = (double)ITERS * 64.0 * per_wave_instr;
	// in-kernel: cycles one wave needed per instruction; per SIMD: w waves share it
	const double cyc_per_instr_wave = (cyc / nwaves) / instr_per_wave;
	const double cyc_per_instr_simd = cyc_per_instr_wave / w;
	const double wall_cyc_per_instr_simd = ms * 1e-3 * clock_ghz * 1e9 / (instr_per_wave * w);
	if (KIND == K_FMA) {  // the schedule the hardware gave this launch
		unsigned long long* hs = (unsigned long long*)malloc((size_t)nwaves * 32);
		hipMemcpy(hs, sched, (size_t)nwaves * 32, hipMemcpyDeviceToHost);
		unsigned long long tmin = ~0ull, tmax = 0, smax = 0;
		static int per_simd[128 * 64];
		for (int i = 0; i < 128 * 64; i++) per_simd[i] = 0;
		for (int i = 0; i < nwaves; i++) {
			if (hs[4 * i] < tmin) tmin = hs[4 * i];
			if (hs[4 * i] > smax) smax = hs[4 * i];
			if (hs[4 * i + 1] > tmax) tmax = hs[4 * i + 1];
			const unsigned hw = (unsigned)hs[4 * i + 2], xcc = (unsigned)hs[4 * i + 3] & 15u;
			const unsigned simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
			per_simd[((xcc & 7u) * 16 + se * 2 + sh) * 64 + cu * 4 + simd]++;
		}
		int hist[64] = {0}, used = 0;
		for (int i = 0; i < 128 * 64; i++) if (per_simd[i]) { used++; hist[per_simd[i] < 63 ? per_simd[i] : 63]++; }
		printf("   schedule: %d waves on %d distinct SIMDs; waves per SIMD histogram:", nwaves, used);
		for (int i = 1; i < 64; i++) if (hist[i]) printf(" %dx%d", hist[i], i);
		printf("; first start -> last start %.1f us, first start -> last end %.1f us\n", (smax - tmin) * 0.01, (tmax - tmin) * 0.01);
		free(hs);
	}
	// the roof in time units, free of any clock estimate: wave64 VALU instructions the whole chip issued per second
	const double ginstr_per_s = instr_per_wave * nwaves / (ms * 1e-3) * 1e-9;
	printf("%-22s w=%d  kernel %.3f ms  clock %.2f GHz  cycles/instr: per wave %.2f, per SIMD %.2f (in-kernel stamps) %.2f (kernel wall time)  chip rate %.0f G instr/s\n",
	       KNAME[KIND], w, ms, clock_ghz, cyc_per_instr_wave, cyc_per_instr_simd, wall_cyc_per_instr_simd, ginstr_per_s);
}

int main()
{
	float* out; unsigned long long* stamps;
	hipMalloc(&out, 8192 * 64 * 4);
	hipMalloc(&stamps, 8192 * 16);
	unsigned long long* sched;
	hipMalloc(&sched, 8192 * 32);
	unsigned long long* h = (unsigned long long*)malloc(8192 * 16);
	const int ws[] = {1, 4, 5, 6, 8};
	for (int wi = 0; wi < 5; wi++) {
		const int w = ws[wi];
		run<K_FMA>(w, out, stamps, h, sched);
		run<K_PKFMA>(w, out, stamps, h, sched);
		run<K_EXP>(w, out, stamps, h, sched);
		run<K_RCP>(w, out, stamps, h, sched);
		run<K_MIX>(w, out, stamps, h, sched);
		run<K_CNDMASK>(w, out, stamps, h, sched);
		run<K_DPP>(w, out, stamps, h, sched);
		run<K_CNDMASK_S>(w, out, stamps, h, sched);
		run<K_CMP_CND>(w, out, stamps, h, sched);
		run<K_PKMUL>(w, out, stamps, h, sched);
		run<K_CMP>(w, out, stamps, h, sched);
		run<K_CMP_S>(w, out, stamps, h, sched);
		run<K_MOV>(w, out, stamps, h, sched);
		run<K_MAX>(w, out, stamps, h, sched);
		run<K_PERM32>(w, out, stamps, h, sched);
		run<K_MUL2>(w, out, stamps, h, sched);
		run<K_FMAC2>(w, out, stamps, h, sched);
		run<K_BLEND>(w, out, stamps, h, sched);
		run<K_ADD2>(w, out, stamps, h, sched);
		run<K_PKADD>(w, out, stamps, h, sched);
		run<K_FWDMIX>(w, out, stamps, h, sched);
		run<K_BWDMIX>(w, out, stamps, h, sched);
	}
	return 0;
}
