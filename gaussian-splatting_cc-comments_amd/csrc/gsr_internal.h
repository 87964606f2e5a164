// gsr_internal.h -- host-side glue shared by the translation units of libgsr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/gsr.h"
#include "gsr_device.h"

#define GSR_PREPROCESS_BLOCK 256  // Gaussians per workgroup of the binning kernels (granularity of their scans)
// The instance count (num_rendered) is accumulated by the preprocess workgroups with one atomic add each,
// spread over this many words so that no address sees more than a few dozen (3 900 adds to ONE word cost
// 9 us of serialisation); the host adds the parts after the read-back.
#define GSR_COUNT_PARTS 64
// status words: [0] prefiltered trap, [2] 1 = the depth sort's result is in the ping-pong partners (perm_alt, depth_keys_alt),
// [3] 1 = slot_base is final and numbers the gradient slots in INDEX order (the bucket depth sort's first kernel did it); 0 = the
// binning numbers them in depth order itself (tilebin.hip pass 1 / the key emission),
// [4, 68) partial instance counts, [68, 132) partial maxima of ~depth_key, [132, 196) partial maxima of depth_key
// (visible Gaussians only; the depth sort orders key - min, so only the bits of max - min need passes)
#define GSR_STATUS_NEGMIN (4 + GSR_COUNT_PARTS)
#define GSR_STATUS_MAX (4 + 2 * GSR_COUNT_PARTS)
#define GSR_STATUS_WORDS (4 + 3 * GSR_COUNT_PARTS)

static inline size_t gsr_align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
static inline int gsr_grid_x(int W) { return (W + GSR_TILE_X - 1) / GSR_TILE_X; }
static inline int gsr_grid_y(int H) { return (H + GSR_TILE_Y - 1) / GSR_TILE_Y; }

// Typed views of the opaque blobs (offsets published through gsr_*_layout_of()).
struct GsrGeometry {
	GsrSplat* splat;
	uint32_t* depth_keys;      // depth bits per Gaussian (0xFFFFFFFF = culled); after the depth sort: sorted keys
	uint32_t* depth_keys_alt;  // ping-pong partner
	uint32_t* perm;            // identity; after the depth sort: Gaussian ids in (depth, id) order
	uint32_t* perm_alt;
	uint32_t* tiles_touched;
	uint2* rect;               // dense copy of the tile rectangle {x | y << 16, w | h << 16}: what the depth-ordered kernels gather
	uint2* rshape;             // {rectangle in one word, trim word} (gsr_rect_trim.h): what the depth sort carries along and the column-pair binning reads
	uint32_t* slot_base;       // first (Gaussian,tile) gradient slot: exclusive prefix of tiles_touched in index order (final once status word 3 is set)
	uint8_t* clamped;
	float* sh_ddir;            // [9][P] d(colour channel c)/d(unit view direction) of the visible Gaussians (plane 3c + {x,y,z})
	uint32_t* status;             // GSR_STATUS_* words

	uint32_t* sorted_block_sums;  // per-workgroup tile counts in depth order (the key emission takes their prefix sums itself)
	uint32_t* block_tiles;        // tile counts of the preprocess kernel's workgroups (256 Gaussians each, index order): slot_base's prefix
	void* sort_table;             // radix histogram table for the P-sized depth sort
	void* col_table;              // tilebin.hip, pass 1 (column pairs by tile column): chunk sums, block rows, digit totals
};

// Depth checkpoints of heavy tiles (lists of at least 2 * GSR_CKPT_STRIDE instances): every GSR_CKPT_STRIDE instances the forward
// blend stores each pixel's running (T, C) BEFORE the instance at that position; the backward blend can then start in the middle
// of a list (render_backward.hip).  One 4-KB record (256 pixels x float4) per checkpoint, indexed by the checkpoint's position in the
// global sorted instance list: record (range.x + p) / GSR_CKPT_STRIDE for local position p -- tiles own disjoint ranges and only
// tiles of at least two strides store any, so no two checkpoints share a record.
#ifndef GSR_CKPT_STRIDE
#define GSR_CKPT_STRIDE 512    // (1024 until the lists were trimmed: gsr_rect_trim.h -- a list position now holds a contributor 1.6 x as often)
#endif
#ifndef GSR_SPLIT_MIN_LIST
#define GSR_SPLIT_MIN_LIST 640 // forward: shortest list that is handed out as four band waves (1024 until the lists were trimmed)
#endif
#define GSR_CKPT_MAX_SEGMENTS 15   // segments a heavy tile's walk is cut into at most (4 bits of the dispatch entry, 0 = whole tile)
static inline size_t gsr_checkpoint_records(int64_t R) { return (size_t)(R / GSR_CKPT_STRIDE) + 2; }

struct GsrImage {
	float* final_C;            // [3][W*H] accumulated colour without the background term (heavy tiles only: the backward's segments need it)
	float* final_T;
	uint32_t* n_contrib;
	uint2* ranges;
	uint32_t* tile_max_contrib;
	uint32_t* tile_order;      // tiles in descending order of backward work (longest-processing-time-first dispatch)
};

struct GsrBinning {
	uint32_t* point_list;      // sorted Gaussian ids (final)
	uint32_t* point_list_alt;  // ping-pong partner
	uint32_t* tile_keys;       // sorted tile ids (final)
	uint32_t* tile_keys_alt;   // after the forward: [R] validity BYTES of the backward's gradient slots (zeroed by tile_ranges)
	void* sort_table;          // radix histogram table for the R-sized tile sort
	float4* checkpoints;       // [gsr_checkpoint_records(R)][256] depth checkpoints of heavy tiles (see GSR_CKPT_STRIDE)
};

GsrGeometry gsr_geometry_view(void* blob, int P);
GsrImage gsr_image_view(void* blob, int W, int H);
GsrBinning gsr_binning_view(void* blob, int P, int64_t R, int W, int H);

// error plumbing (api.hip)
int gsr_fail(int code, const char* fmt, ...);
int gsr_check_hip(hipError_t e, const char* what);
// Called after each stage: in debug mode synchronises the stream and reports kernel errors.
int gsr_stage_done(hipStream_t s, int debug, const char* stage);
// Per-kernel event profiling (api.hip); no-ops unless gsr_profile_begin() was called for this stream.
void gsr_prof_mark_begin(hipStream_t s, const char* name);
void gsr_prof_mark_end(hipStream_t s);
bool gsr_prof_kernel_events(hipStream_t s, const char* name, hipEvent_t* a, hipEvent_t* b);

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
	int trim;   // 0: GSR_DEBUG_NO_TRIM
	int* radii;
	GsrGeometry g;
	// leaf mode (gsr_forward_preprocess_leaf): shs = _features_dc, shs_rest = _features_rest,
	// opacities / scales / rotations are the raw leaves and are activated inside the kernel
	int leaf;
	const float* shs_rest;
};

// preprocess.hip
void gsr_launch_preprocess(const GsrPreprocessArgs& a, hipStream_t s, hipEvent_t done = nullptr);
void gsr_launch_zero_status(uint32_t* status, hipStream_t s, hipEvent_t done = nullptr);
bool gsr_preprocess_needs_color(const GsrPreprocessArgs& a);
void gsr_launch_preprocess_color(const GsrPreprocessArgs& a, hipStream_t s, int wgs_per_cu = 0);
void gsr_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);

// binning.hip
void gsr_launch_sorted_block_sums(GsrGeometry g, int P, int result_in_alt, hipStream_t s);
void gsr_launch_duplicate_keys(GsrGeometry g, int P, int W, void* keys, int key_bytes, uint32_t* vals, uint32_t* clear, size_t clear_words, hipStream_t s);
void gsr_launch_tile_ranges(const void* tile_keys, int key_bytes, int64_t R, uint2* ranges, int ntiles, uint32_t* valid, hipStream_t s);
void gsr_launch_tile_order(GsrImage img, int ntiles, bool backward, int64_t num_rendered, bool split, hipStream_t s, bool normalise_empty = false);
uint32_t gsr_tile_order_max_split(int ntiles);      // forward: tiles that may be split into four band entries
uint32_t gsr_tile_order_max_segments(int ntiles);   // backward: extra entries for the depth segments of heavy tiles

// tilebin.hip: the sorted instance list by column pairs (images of at most 256 x 256 tiles)
bool gsr_tilebin_applies(int W, int H);
size_t gsr_tilebin_col_clear_words(size_t P);   // leading words of col_table that the preprocess kernel zeroes
size_t gsr_tilebin_col_table_bytes(size_t P);
size_t gsr_tilebin_row_clear_words(size_t R);
size_t gsr_tilebin_row_table_bytes(size_t R);
void gsr_launch_tilebin_col_hist(GsrGeometry g, int P, int result_in_alt, hipStream_t s, bool seg_ready = false);
uint4* gsr_tilebin_seg(GsrGeometry g, int P);
uint4* gsr_tilebin_recs(GsrGeometry g, int P);
void gsr_launch_tilebin_col_scatter(GsrGeometry g, int P, GsrBinning b, int64_t R, hipStream_t s);
void gsr_launch_tilebin_row_hist(GsrGeometry g, int P, GsrBinning b, int64_t R, hipStream_t s);
void gsr_launch_tilebin_row_scatter(GsrGeometry g, int P, GsrBinning b, int64_t R, uint2* ranges, int W, int H, hipStream_t s);

// sort.hip
int gsr_radix_num_passes(int nbits_total);
size_t gsr_radix_table_bytes(size_t n);
size_t gsr_radix_clear_words(size_t n);  // leading words of the table that the producer of the keys must zero
void gsr_launch_slot_base_finish(uint32_t* slot_base, const uint32_t* block_tiles, uint32_t* status, size_t n, hipStream_t s);
void gsr_radix_sort_passes(void* k0, uint32_t* v0, void* k1, uint32_t* v1, size_t n, int nbits_total, int npass_total,
                           int pass_first, int pass_count, void* table_mem, const uint32_t* bias, int key_bytes, hipStream_t s);
void gsr_radix_sort_u32(void* k0, uint32_t* v0, void* k1, uint32_t* v1, size_t n, int nbits_total, void* table_mem,
                        int* result_in_first, int clear_table, int key_bytes, hipStream_t s);
void gsr_radix_top_pass(const uint32_t* k0, uint32_t* k1, const uint2* rec_in, uint4* rec_out, size_t n, void* table_mem, const uint32_t* bias, hipStream_t s,
                        uint32_t* slot_base, const uint32_t* block_tiles, uint32_t* status);
int gsr_radix_top_chunks(size_t n);
// depthsort.hip: depth order in three launches (top-digit buckets, then every bucket sorted inside LDS) for up to this many Gaussians
#ifndef GSR_BUCKET_SORT_MAX_P
#define GSR_BUCKET_SORT_MAX_P (2 << 20)
#endif
bool gsr_bucket_sort_applies(int P);
void gsr_launch_depth_bucket_sort(GsrGeometry g, int P, uint4* seg, hipStream_t s);
// bytes per tile key of an instance-sized sort: 2 when every tile id fits 16 bits AND the sort runs the instance-sized
// kernels (sort.hip), else 4
int gsr_tile_key_bytes(int ntiles, size_t num_rendered);

// render_forward.hip
void gsr_launch_render_forward(int W, int H, GsrImage img, const uint32_t* point_list, const GsrSplat* splat, float4* checkpoints,
                               const float* bg, float* out_color, bool ordered, bool cull, hipStream_t s);   // ordered: tile_order holds ntiles + 3 * gsr_tile_order_max_split(ntiles) entries

// render_backward.hip
void gsr_launch_render_backward(int W, int H, GsrImage img, const uint32_t* point_list, const GsrSplat* splat, const float4* checkpoints,
                                const uint32_t* slot_base, const float* bg, const float* dL_dpix, GsrGradSlot* slots,
                                uint8_t* slot_valid, bool cull, hipStream_t s, hipEvent_t t_start = nullptr, hipEvent_t t_stop = nullptr);   // t_*: taken by the kernel's own dispatch packet

// gaussian_backward.hip
struct GsrGaussianBackwardArgs {
	int P, D, M, W, H;
	int first, count;          // the Gaussians [first, first + count) are processed (first is a multiple of 64)
	int out_row0;              // gradient outputs are written at row (index - out_row0): 0, or `first` for per-part buffers
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
	const int* radii;          // may be NULL: visibility is then taken from tiles_touched (same predicate)
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
	// leaf mode (gsr_backward_leaf): inputs as in GsrPreprocessArgs; dL_dsh = grad of _features_dc,
	// dL_dsh_rest = grad of _features_rest, dL_dopacity/dL_dscale/dL_drot are gradients w.r.t. the raw
	// leaves; dL_dconic, dL_dcolor, dL_dcov3D may be NULL (not written)
	int leaf;
	const float* shs_rest;
	float* dL_dsh_rest;
	// densification statistics (train.py:157-159, scene/gaussian_model.py:599-602), each may be NULL
	float* stat_xyz_gradient_accum;  // [P] += ||dL_dmean2D.xy|| for visible Gaussians
	float* stat_denom;               // [P] += 1 for visible Gaussians
	float* stat_max_radii2D;         // [P] = max(itself, radius) for visible Gaussians
};
void gsr_launch_gaussian_backward(const GsrGaussianBackwardArgs& a, hipStream_t s);
void gsr_launch_sh_grad_from_views(int P, int D, int M, int V, const float* means3D, const float* cam_pos, const float* dL_dRGB,
                                   int64_t view_stride, float* dL_dsh, hipStream_t s);

// loss.hip
size_t gsr_loss_scratch_layout(int C, int H, int W, size_t* maps_off, size_t* partial_off, int* ntiles);
void gsr_launch_l1_ssim(int C, int H, int W, const float* img, const float* gt, float lambda, float* loss_out, float* dL_dimg,
                        void* scratch, hipStream_t s);

// optimizer.hip
int gsr_launch_adam(int ngroups, const gsr_adam_group* groups, double beta1, double beta2, double eps, const int* radii, hipStream_t s);

// knn.hip
size_t gsr_knn_scratch_size(int P);
void gsr_launch_knn(int P, const float* points, float* mean_dist2, void* scratch, hipStream_t s);
