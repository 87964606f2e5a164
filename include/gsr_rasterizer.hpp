/*
 * gsr_rasterizer.hpp -- the C++ surface of the reference's rasterizer library on top of the C ABI (gsr.h).
 *
 * Header-only.  Declares CudaRasterizer::Rasterizer with exactly the three static functions and argument lists of
 * submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:20-85 -- what rasterize_points.cu:87-129,
 * :156-213, :218-237 and the SIBR viewer call -- so that a C++ caller of the reference recompiles against
 * libgsr_hip.so unchanged: include this header instead of "cuda_rasterizer/rasterizer.h", link -lgsr_hip.
 *
 * Differences a caller can observe (all within the reference's own contract):
 *   * the three std::function<char*(size_t)> callbacks are asked for THIS library's blob sizes (gsr_*_bytes) plus
 *     256 bytes; like the reference (obtain(), rasterizer_impl.h:53-60) the library aligns inside what it is given,
 *     and backward() applies the same alignment to the raw char* it receives;
 *   * backward() needs gsr_backward_scratch_bytes(P, R) bytes of device scratch for the call.  They come from
 *     Rasterizer::scratchBuffer() when the caller installed one (a fourth callback of the same shape, e.g. another
 *     resizeFunctional tensor), otherwise from hipMalloc / hipFree around the call;
 *   * work is enqueued on Rasterizer::stream() (default: the null stream, where the reference launches);
 *   * errors throw std::runtime_error with gsr_last_error() (the reference throws from CHECK_CUDA only with debug);
 *   * backward() overwrites every output element (the reference accumulates into zero-initialised tensors,
 *     rasterize_points.cu:168-178: same values when the caller zeroed them, as the reference requires).
 */
#ifndef GSR_RASTERIZER_HPP_INCLUDED
#define GSR_RASTERIZER_HPP_INCLUDED

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>

#include "gsr.h"

namespace CudaRasterizer
{
	class Rasterizer
	{
	public:
		/* optional: where backward() takes its scratch from (called once per backward with the byte count) */
		static std::function<char*(size_t)>& scratchBuffer()
		{
			static thread_local std::function<char*(size_t)> f;
			return f;
		}
		/* stream all calls of this host thread are enqueued on (a hipStream_t; default nullptr = the null stream) */
		static void*& stream()
		{
			static thread_local void* s = nullptr;
			return s;
		}

		/* rasterizer.h:24-29 */
		static void markVisible(int P, float* means3D, float* viewmatrix, float* projmatrix, bool* present)
		{
			static_assert(sizeof(bool) == 1, "present is written as one byte per Gaussian");
			check(gsr_mark_visible(P, means3D, viewmatrix, projmatrix, reinterpret_cast<uint8_t*>(present), stream()));
		}

		/* rasterizer.h:31-53; returns num_rendered */
		static int forward(std::function<char*(size_t)> geometryBuffer, std::function<char*(size_t)> binningBuffer,
		                   std::function<char*(size_t)> imageBuffer, const int P, int D, int M, const float* background,
		                   const int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
		                   const float* opacities, const float* scales, const float scale_modifier, const float* rotations,
		                   const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
		                   const float tan_fovx, float tan_fovy, const bool prefiltered, float* out_color, int* radii = nullptr,
		                   bool debug = false)
		{
			if (P == 0) return 0;  // (rasterize_points.cu:94 never calls with P == 0; nothing to do)
			char* geom = aligned(geometryBuffer(gsr_geometry_bytes(P) + kPad));
			char* img = aligned(imageBuffer(gsr_image_bytes(width, height) + kPad));
			int64_t num_rendered = 0;
			check(gsr_forward_preprocess(P, D, M, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
			                             rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy,
			                             prefiltered ? 1 : 0, radii, geom, &num_rendered, stream(), debug ? GSR_DEBUG_SYNC : 0));
			if (num_rendered > 0x7fffffffLL) throw std::runtime_error("num_rendered does not fit the reference's int return value");
			char* binning = aligned(binningBuffer(gsr_binning_bytes(P, num_rendered, width, height) + kPad));
			check(gsr_forward_render(P, num_rendered, width, height, background, radii, geom, binning, img, out_color, stream(),
			                         debug ? GSR_DEBUG_SYNC : 0));
			return (int)num_rendered;
		}

		/* rasterizer.h:55-84 */
		static void backward(const int P, int D, int M, int R, const float* background, const int width, int height,
		                     const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
		                     const float scale_modifier, const float* rotations, const float* cov3D_precomp,
		                     const float* viewmatrix, const float* projmatrix, const float* campos, const float tan_fovx,
		                     float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
		                     const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
		                     float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, bool debug)
		{
			if (P == 0) return;
			const size_t need = gsr_backward_scratch_bytes(P, R);
			char* scratch = nullptr;
			bool own = false;
			if (need) {
				if (scratchBuffer()) {
					scratch = aligned(scratchBuffer()(need + kPad));
				} else {
					void* p = nullptr;
					if (hipMalloc(&p, need) != hipSuccess) throw std::runtime_error("CudaRasterizer::Rasterizer::backward: hipMalloc of the scratch failed");
					scratch = static_cast<char*>(p);
					own = true;
				}
			}
			const int rc = gsr_backward(P, D, M, R, width, height, background, means3D, shs, colors_precomp, scales, scale_modifier,
			                            rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii,
			                            aligned(geom_buffer), aligned(binning_buffer), aligned(image_buffer), scratch, dL_dpix,
			                            dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
			                            dL_drot, stream(), debug ? GSR_DEBUG_SYNC : 0);
			if (own) {  // the kernels that read the scratch must have finished before it goes back
				(void)hipStreamSynchronize(static_cast<hipStream_t>(stream()));
				(void)hipFree(scratch);
			}
			check(rc);
		}

	private:
		static constexpr size_t kPad = 256;
		static char* aligned(char* p)
		{
			return reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + (kPad - 1)) & ~static_cast<uintptr_t>(kPad - 1));
		}
		static void check(int rc)
		{
			if (rc != GSR_OK) throw std::runtime_error(std::string("gsr error ") + std::to_string(rc) + ": " + gsr_last_error());
		}
	};
}  // namespace CudaRasterizer

#endif /* GSR_RASTERIZER_HPP_INCLUDED */
