// shim_demo.cpp -- a C++ caller of the reference's library surface, compiled against include/gsr_rasterizer.hpp.
//
//   shim_demo <inputs.bin> <outputs.bin>
//
// Reads one scene + camera + upstream gradient (written by tests/test_cpp_shim.py), runs
// CudaRasterizer::Rasterizer::{markVisible, forward, backward} the way rasterize_points.cu:87-129,156-213 does --
// three growable device buffers handed over as std::function<char*(size_t)> callbacks, zero-initialised gradient
// tensors -- and writes every output.  No Python, no torch: hipMalloc / hipMemcpy only.  The test compares the file
// with the Python binding's results bit for bit.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gsr_rasterizer.hpp"

#define HIP_OK(x)                                                                          \
	do {                                                                                   \
		hipError_t e_ = (x);                                                               \
		if (e_ != hipSuccess) {                                                            \
			fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                 \
			exit(2);                                                                       \
		}                                                                                  \
	} while (0)

// a device buffer that grows on demand: the role of resizeFunctional(torch::Tensor&) (rasterize_points.cu:28-36)
struct Growable {
	char* p = nullptr;
	size_t cap = 0;
	int calls = 0;
	std::function<char*(size_t)> functional()
	{
		return [this](size_t n) {
			calls++;
			if (n > cap) {
				if (p) HIP_OK(hipFree(p));
				HIP_OK(hipMalloc((void**)&p, n));
				cap = n;
			}
			return p;
		};
	}
};

static std::vector<float> read_f(FILE* f, size_t n)
{
	std::vector<float> v(n);
	if (n && fread(v.data(), 4, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
	return v;
}
static float* to_dev(const std::vector<float>& v)
{
	if (v.empty()) return nullptr;
	float* d = nullptr;
	HIP_OK(hipMalloc((void**)&d, v.size() * 4));
	HIP_OK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
	return d;
}
static float* dev_zeros(size_t n)
{
	float* d = nullptr;
	HIP_OK(hipMalloc((void**)&d, (n ? n : 1) * 4));
	HIP_OK(hipMemset(d, 0, (n ? n : 1) * 4));
	return d;
}
static void write_dev(FILE* f, const void* d, size_t bytes)
{
	std::vector<char> h(bytes);
	if (bytes) HIP_OK(hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost));
	fwrite(h.data(), 1, bytes, f);
}

int main(int argc, char** argv)
{
	if (argc != 3) { fprintf(stderr, "usage: %s inputs.bin outputs.bin\n", argv[0]); return 2; }
	FILE* f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	int hdr[5];
	float fl[3];
	if (fread(hdr, 4, 5, f) != 5 || fread(fl, 4, 3, f) != 3) { fprintf(stderr, "bad header\n"); return 2; }
	const int P = hdr[0], D = hdr[1], M = hdr[2], W = hdr[3], H = hdr[4];
	const float tanfovx = fl[0], tanfovy = fl[1], scale_modifier = fl[2];
	float* bg = to_dev(read_f(f, 3));
	float* means3D = to_dev(read_f(f, (size_t)P * 3));
	float* shs = to_dev(read_f(f, (size_t)P * M * 3));
	float* opac = to_dev(read_f(f, (size_t)P));
	float* scales = to_dev(read_f(f, (size_t)P * 3));
	float* rots = to_dev(read_f(f, (size_t)P * 4));
	float* view = to_dev(read_f(f, 16));
	float* proj = to_dev(read_f(f, 16));
	float* campos = to_dev(read_f(f, 3));
	float* dL_dpix = to_dev(read_f(f, (size_t)3 * H * W));
	fclose(f);

	using CudaRasterizer::Rasterizer;
	bool* present = nullptr;
	HIP_OK(hipMalloc((void**)&present, P ? P : 1));
	Rasterizer::markVisible(P, means3D, view, proj, present);

	Growable geom, binning, img;
	float* out_color = dev_zeros((size_t)3 * H * W);   // torch::full({3, H, W}, 0.0), rasterize_points.cu:79
	int* radii = reinterpret_cast<int*>(dev_zeros((size_t)P));
	int rendered = 0;
	try {
		rendered = Rasterizer::forward(geom.functional(), binning.functional(), img.functional(), P, D, M, bg, W, H, means3D, shs, nullptr,
		                               opac, scales, scale_modifier, rots, nullptr, view, proj, campos, tanfovx, tanfovy, false, out_color,
		                               radii, false);
	} catch (const std::exception& e) {
		fprintf(stderr, "forward: %s\n", e.what());
		return 3;
	}
	// rasterize_points.cu:168-178: zero-initialised gradient tensors
	float *dL_dmeans3D = dev_zeros((size_t)P * 3), *dL_dmeans2D = dev_zeros((size_t)P * 3), *dL_dcolors = dev_zeros((size_t)P * 3);
	float *dL_dconic = dev_zeros((size_t)P * 4), *dL_dopacity = dev_zeros((size_t)P), *dL_dcov3D = dev_zeros((size_t)P * 6);
	float *dL_dsh = dev_zeros((size_t)P * M * 3), *dL_dscales = dev_zeros((size_t)P * 3), *dL_drotations = dev_zeros((size_t)P * 4);
	try {
		Rasterizer::backward(P, D, M, rendered, bg, W, H, means3D, shs, nullptr, scales, scale_modifier, rots, nullptr, view, proj, campos,
		                     tanfovx, tanfovy, radii, geom.p, binning.p, img.p, dL_dpix, dL_dmeans2D, dL_dconic, dL_dopacity, dL_dcolors,
		                     dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, false);
		// a second backward with a caller-provided scratch (the fourth callback) must give the same bits: done into the
		// same tensors after zeroing one of them, and compared by the test through that tensor
		Growable scratch;
		Rasterizer::scratchBuffer() = scratch.functional();
		HIP_OK(hipMemset(dL_dopacity, 0, (size_t)(P ? P : 1) * 4));
		Rasterizer::backward(P, D, M, rendered, bg, W, H, means3D, shs, nullptr, scales, scale_modifier, rots, nullptr, view, proj, campos,
		                     tanfovx, tanfovy, radii, geom.p, binning.p, img.p, dL_dpix, dL_dmeans2D, dL_dconic, dL_dopacity, dL_dcolors,
		                     dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, true);
		Rasterizer::scratchBuffer() = nullptr;
		if (scratch.calls != 1) { fprintf(stderr, "scratch callback called %d times\n", scratch.calls); return 3; }
		HIP_OK(hipDeviceSynchronize());
		if (scratch.p) HIP_OK(hipFree(scratch.p));
	} catch (const std::exception& e) {
		fprintf(stderr, "backward: %s\n", e.what());
		return 3;
	}
	HIP_OK(hipDeviceSynchronize());
	if (geom.calls != 1 || binning.calls != 1 || img.calls != 1) { fprintf(stderr, "each buffer callback must be called exactly once\n"); return 3; }

	FILE* o = fopen(argv[2], "wb");
	if (!o) { perror(argv[2]); return 2; }
	fwrite(&rendered, 4, 1, o);
	write_dev(o, present, (size_t)P);
	write_dev(o, out_color, (size_t)3 * H * W * 4);
	write_dev(o, radii, (size_t)P * 4);
	write_dev(o, dL_dmeans2D, (size_t)P * 3 * 4);
	write_dev(o, dL_dcolors, (size_t)P * 3 * 4);
	write_dev(o, dL_dopacity, (size_t)P * 4);
	write_dev(o, dL_dmeans3D, (size_t)P * 3 * 4);
	write_dev(o, dL_dcov3D, (size_t)P * 6 * 4);
	write_dev(o, dL_dsh, (size_t)P * M * 3 * 4);
	write_dev(o, dL_dscales, (size_t)P * 3 * 4);
	write_dev(o, dL_drotations, (size_t)P * 4 * 4);
	write_dev(o, dL_dconic, (size_t)P * 4 * 4);
	fclose(o);
	if (gsr_thread_release() != GSR_OK) { fprintf(stderr, "gsr_thread_release: %s\n", gsr_last_error()); return 3; }
	printf("shim_demo ok: P=%d rendered=%d\n", P, rendered);
	return 0;
}
