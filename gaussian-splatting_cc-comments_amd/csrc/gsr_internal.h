// gsr_internal.h -- host-side glue shared by the translation units of libgsr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/gsr.h"
#include "gsr_device.h"

#define GSR_PREPROCESS_BLOCK 256

static inline size_t gsr_align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
static inline int gsr_grid_x(int W) { return (W + GSR_TILE_X - 1) / GSR_TILE_X; }
static inline int gsr_grid_y(int H) { return (H + GSR_TILE_Y - 1) / GSR_TILE_Y; }

// Typed views of the opaque blobs (offsets published through gsr_*_layout_of()).
struct GsrGeometry {
	GsrSplat* splat;
	float* depths;
	uint32_t* tiles_touched;
	uint32_t* point_offsets;
	uint8_t* clamped;
	uint32_t* status;       // [0] prefiltered trap, [1] num_rendered
	uint32_t* block_sums;   // scan temp: per-preprocess-block tile counts, then exclusive offsets
};

struct GsrImage {
	float* final_T;
	uint32_t* n_contrib;
	uint2* ranges;
	uint32_t* tile_max_contrib;
};

struct GsrBinning {
	uint32_t* point_list;
	uint32_t* point_list_unsorted;
	uint64_t* keys;
	uint64_t* keys_unsorted;
	void* sort_temp;
	size_t sort_temp_bytes;
};

GsrGeometry gsr_geometry_view(void* blob, int P);
GsrImage gsr_image_view(void* blob, int W, int H);
GsrBinning gsr_binning_view(void* blob, int P, int64_t R, int W, int H);

// error plumbing (api.hip)
int gsr_fail(int code, const char* fmt, ...);
int gsr_check_hip(hipError_t e, const char* what);
// Called after each stage: in debug mode synchronises the stream and reports kernel errors.
int gsr_stage_done(hipStream_t s, int debug, const char* stage);
// Per-kernel event profiling (api.hip); no-ops unless gsr_profile_begin() was called.
void gsr_prof_mark_begin(hipStream_t s, const char* name);
void gsr_prof_mark_end(hipStream_t s);

struct GsrProfScope {
	hipStream_t s;
	GsrProfScope(hipStream_t st, const char* name) : s(st) { gsr_prof_mark_begin(s, name); }
	~GsrProfScope() { gsr_prof_mark_end(s); }
};

// ---- kernel launchers ------------------------------------------------------------------------
struct GsrPreprocessArgs {
	int P, D, M, W, H;
	const float* means3D;
	const float* shs;
	const float* colors_precomp;
	const float* opacities;
	const float* scales;
	float scale_modifier;
	const float* rotations;
	const float* cov3D_precomp;
	const float* viewmatrix;
	const float* projmatrix;
	const float* cam_pos;
	float tan_fovx, tan_fovy, focal_x, focal_y;
	int prefiltered;
	int* radii;
	GsrGeometry g;
};

// preprocess.hip
void gsr_launch_preprocess(const GsrPreprocessArgs& a, hipStream_t s);
void gsr_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);

// binning.hip
void gsr_launch_scan_block_sums(GsrGeometry g, int P, hipStream_t s);         // block_sums -> exclusive, total -> status[1]
void gsr_launch_finalize_offsets(GsrGeometry g, int P, hipStream_t s);        // point_offsets (inclusive) + splat.slot_base
void gsr_launch_duplicate_keys(GsrGeometry g, const int* radii, int P, int W, int H, GsrBinning b, hipStream_t s);
int gsr_sort_pairs(GsrBinning b, int64_t R, int end_bit, hipStream_t s);
size_t gsr_sort_temp_bytes(int64_t R);
void gsr_launch_tile_ranges(const uint64_t* keys, int64_t R, uint2* ranges, int ntiles, hipStream_t s);

// render_forward.hip
void gsr_launch_render_forward(int W, int H, GsrImage img, const uint32_t* point_list, const GsrSplat* splat,
                               const float* bg, float* out_color, hipStream_t s);

// render_backward.hip
void gsr_launch_render_backward(int W, int H, GsrImage img, const uint32_t* point_list, const GsrSplat* splat,
                                const float* bg, const float* dL_dpix, GsrGradSlot* slots, uint8_t* slot_valid,
                                hipStream_t s);

// gaussian_backward.hip
struct GsrGaussianBackwardArgs {
	int P, D, M, W, H;
	const float* means3D;
	const float* shs;
	const float* colors_precomp;
	const float* scales;
	float scale_modifier;
	const float* rotations;
	const float* cov3D_precomp;
	const float* viewmatrix;
	const float* projmatrix;
	const float* cam_pos;
	float tan_fovx, tan_fovy, focal_x, focal_y;
	const int* radii;
	GsrGeometry g;
	const GsrGradSlot* slots;
	const uint8_t* slot_valid;
	float* dL_dmean2D;
	float* dL_dconic;
	float* dL_dopacity;
	float* dL_dcolor;
	float* dL_dmean3D;
	float* dL_dcov3D;
	float* dL_dsh;
	float* dL_dscale;
	float* dL_drot;
};
void gsr_launch_gaussian_backward(const GsrGaussianBackwardArgs& a, hipStream_t s);
