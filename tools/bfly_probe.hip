// Probe: what do v_permlane32_swap / v_permlane16_swap and the butterfly of render_common.h really do?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../gaussian-splatting_cc-comments_amd/csrc/render_common.h"
__global__ void probe(float* out) {
	const int lane = threadIdx.x;
	// raw swaps: a = lane, b = 100 + lane
	auto r = __builtin_amdgcn_permlane32_swap((unsigned)lane, (unsigned)(100 + lane), false, false);
	out[lane] = (float)r[0]; out[64 + lane] = (float)r[1];
	auto q = __builtin_amdgcn_permlane16_swap((unsigned)lane, (unsigned)(100 + lane), false, false);
	out[128 + lane] = (float)q[0]; out[192 + lane] = (float)q[1];
	float v[8];
	for (int i = 0; i < 8; i++) v[i] = (float)(1 << (3 * i)) * (lane == 5 * i + 3 ? 1.f : 0.f);  // value i is nonzero in ONE lane
	out[256 + lane] = gsr_bfly8(v, lane);
	const float w0 = gsr_fold32(v[0], v[1]), w1 = gsr_fold32(v[2], v[3]);
	const float w2 = gsr_fold32(v[4], v[5]), w3 = gsr_fold32(v[6], v[7]);
	const float u0 = gsr_fold16(w0, w1), u1 = gsr_fold16(w2, w3);
	out[320 + lane] = w0; out[384 + lane] = u0; out[448 + lane] = u1;
	out[512 + lane] = gsr_fold8(u0, u1, (lane & 8) != 0);
	out[576 + lane] = gsr_sum8((float)(lane == 3));
}
int main() {
	float* d; hipMalloc(&d, 640 * 4);
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
	float h[640]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	const char* names[] = {"p32.a'", "p32.b'", "p16.a'", "p16.b'", "bfly8", "w0", "u0", "u1", "fold8", "sum8"};
	for (int k = 4; k < 10; k++) { printf("%s:", names[k]); for (int l = 0; l < 64; l++) printf(" %g", h[64 * k + l]); printf("\n"); }
	return 0;
}
