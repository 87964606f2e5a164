/*
 * gsr.h -- C ABI of the MI355X (gfx950) differentiable Gaussian rasterizer, libgsr_hip.so.
 *
 * This is the drop-in boundary for the reference's torch extension
 * `diff_gaussian_rasterization._C` (submodules/diff-gaussian-rasterization/ext.cpp:18-22).
 * Every entry point takes plain device/host pointers and sizes; the library allocates nothing
 * on the device, keeps no state between calls and borrows pointers only for the duration of a
 * call.  All device work is enqueued on the caller's `stream` (a hipStream_t passed as void*).
 * Pointers are DEVICE pointers unless the parameter name ends in `_host`.
 *
 * "Not provided" convention (rasterize_points.cu:108-125): pass NULL for shs / colors_precomp /
 * scales / rotations / cov3D_precomp exactly where the reference passes an empty tensor.
 *
 * Return value: 0 on success, otherwise a negative gsr_status; gsr_last_error() returns a
 * thread-local message for the last failure.  `debug` is a bit mask (GSR_DEBUG_*): the reference's
 * bool `debug` is GSR_DEBUG_SYNC = 1 -- every stage is followed by a stream synchronise + error
 * check, like CHECK_CUDA (cuda_rasterizer/auxiliary.h:177-184); the other bits are diagnostics that
 * travel with the call (the library reads nothing from the environment and keeps no switches).
 *
 * The three state buffers (geometry / binning / image) are opaque byte blobs that only travel
 * forward -> backward, like the reference's geomBuffer / binningBuffer / imgBuffer
 * (rasterize_points.cu:82-91).  Their layout is this library's own; gsr_*_layout() publishes the
 * byte offsets so tests can inspect intermediates without extra kernels.
 */
#ifndef GSR_H_INCLUDED
#define GSR_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
	GSR_OK = 0,
	GSR_ERR_INVALID_ARGUMENT = -1,
	GSR_ERR_HIP = -2,          /* a HIP runtime call or kernel failed */
	GSR_ERR_PREFILTERED = -3,  /* prefiltered=1 but a point was culled (auxiliary.h:167-171 __trap) */
	GSR_ERR_BUFFER_TOO_SMALL = -4
} gsr_status;

/* bits of the `debug` argument of every entry point that has one (and of gsr_backward_args.debug) */
#define GSR_DEBUG_SYNC    1  /* the reference's debug = true: synchronise + check after every stage */
#define GSR_DEBUG_NO_CULL 2  /* blend kernels evaluate every (instance, pixel band) pair of a tile's list, like the reference,
                                instead of skipping the pairs proven to contribute nothing: bisects a suspected culling error */
#define GSR_DEBUG_SERIAL  4  /* gsr_forward_preprocess*: the SH colour kernel runs in line on `stream`, not on the helper stream */
#define GSR_DEBUG_TILE_SORT 16 /* gsr_forward_preprocess* AND gsr_forward_render (pass it to both or to neither): instances are emitted
                                 with their tile ids and radix-sorted (the path of images beyond 256 x 256 tiles) instead of being
                                 binned by column pairs; same point_list, same ranges */
#define GSR_DEBUG_RADIX_DEPTH 32 /* gsr_forward_preprocess*: the Gaussians are put in depth order by three or four global radix passes (the path
                                  of more than 2 Mi Gaussians) instead of top-digit buckets sorted inside LDS; same order */
#define GSR_DEBUG_NO_TRIM 64 /* gsr_forward_preprocess*: every tile of a Gaussian's rectangle is binned, like the reference does; default: tiles that
                                the splat's ellipse at alpha = 1 / 255 provably misses are left out of point_list (csrc/gsr_rect_trim.h: same
                                image, radii, gradients and num_rendered, bit for bit; shorter tile lists) */
#define GSR_DEBUG_NO_SPLIT 8 /* gsr_forward_render: heavy tiles (instance lists >= 640 and >= 2x the mean) are blended by one wave like
                                every other tile, not by four waves of one 16x4-pixel band each (same results either way) */

/* Message of the last failing call on this thread ("" if none). */
const char* gsr_last_error(void);

/* Library / build identification, e.g. "gsr-hip gfx950 r1". */
const char* gsr_version(void);

/* Frees what the calling host thread's earlier calls created and kept for reuse: per device one helper stream with
 * its two events, one stream + event for the read-back of the instance count, one pinned 784-byte landing buffer for it and
 * its event (all created on first use by gsr_forward_preprocess*).  Optional -- a later call simply creates them again; meant for worker threads that
 * end, and for leak checkers.  Call it when none of this thread's library calls is still executing. */
int gsr_thread_release(void);

/* ---- buffer sizing (replaces required<GeometryState/ImageState/BinningState>(),
 *      cuda_rasterizer/rasterizer_impl.h:65-73 and the resize callbacks rasterize_points.cu:28-36) */
size_t gsr_geometry_bytes(int P);
size_t gsr_image_bytes(int width, int height);
size_t gsr_binning_bytes(int P, int64_t num_rendered, int width, int height);
/* Scratch for one gsr_backward() call: [R] 48-byte per-(Gaussian,tile) gradient records written
 * by the backward blend.  Not kept after the call. */
size_t gsr_backward_scratch_bytes(int P, int64_t num_rendered);

/* Byte offsets of the typed arrays inside each blob (introspection for tests / debuggers). */
typedef struct {
	size_t splat;          /* [P] 48-byte records: xy(2f) conic a b c + opacity(4f) rect_min(u16x2) rect_wh(u16x2) rgb(3f) unused(u32) */
	size_t depth_keys;     /* [P] u32: after the call, depth bits sorted ascending (culled = 0xFFFFFFFF last) -- in this array or in */
	size_t depth_keys_alt; /* [P] u32 its ping-pong partner: status word 2 says which (0 always after the bucket sort of up to 2 Mi Gaussians;
	                          with the global radix passes 1 = the _alt pair: three passes sufficed) */
	size_t perm;           /* [P] u32 Gaussian ids in (depth, id) order (or in perm_alt, see above) */
	size_t perm_alt;       /* [P] u32 sort ping-pong */
	size_t tiles_touched;  /* [P] u32 */
	size_t rect;           /* [P] uint2: tile rectangle {min x | min y << 16, width | height << 16} (dense copy of the record's) */
	size_t slot_base;      /* [P] u32: first gradient slot of the Gaussian: exclusive prefix of tiles_touched in index order (status word 3 = 1
	                          once forward stage 1 has run); a Gaussian's slots are contiguous */
	size_t clamped;        /* [P] u8 bit c set = colour channel c was clamped at 0 */
	size_t sh_ddir;        /* [9][P] f32 d(colour channel c)/d(unit view direction x,y,z) of the visible Gaussians, left by the forward so
	                          that the backward does not read the SH rows again (backward.cu:98-132) */
	size_t status;         /* u32 device status words (0: prefiltered trap; 2: depth order in the _alt pair; 3: slots numbered in index order; 4..67: partial instance
	                          counts; 68..131 / 132..195: partial maxima of ~depth key / depth key of the visible Gaussians) */
	size_t scan_temp;      /* per-workgroup tile counts in depth order (tile-sort path: the key emission's offsets), then the preprocess
	                          workgroups' tile counts in index order (the gradient slots' numbering) */
	size_t sort_table;     /* radix histogram table of the depth sort */
	size_t col_table;      /* column-pair binning, pass 1 (by tile column): chunk sums, per-workgroup digit rows, 256 digit totals, the
	                          Gaussians' records {rectangle in one word, trim word, id, -} (see rshape) in depth order and (bucket depth sort) in
	                          bucket order, 16 B each */
	size_t rshape;         /* [P] uint2 {rectangle in one word: x | y << 8 | (w - 1) << 16 | (h - 1) << 24, 0xFFFFFFFF = no tiles; trim word:
	                          a nibble per tile column of the rectangle (per group of 2 .. 32 columns when it is wider than 8), rows left out at its top
	                          (2 bits) and bottom (2 bits; in units of 2 .. 16 rows when it is taller than 16)}: what the
	                          column-pair binning reads (csrc/gsr_rect_trim.h) */
	size_t total;
} gsr_geometry_layout;

typedef struct {
	size_t final_C;        /* [3][W*H] f32 accumulated colour without the background term; written for heavy tiles only */
	size_t final_T;        /* [W*H] f32 */
	size_t n_contrib;      /* [W*H] u32 */
	size_t ranges;         /* [tiles] uint2 */
	size_t tile_max_contrib; /* [tiles] u32 = max n_contrib over the tile's pixels */
	size_t tile_order;     /* u32 dispatch list of the blend kernels: tile ids in descending order of work; the forward's list has up to
	                          3 * min(2048, tiles / 4) further entries (heavy tiles as four band entries: tile | (band + 1) << 28) */
	size_t total;
} gsr_image_layout;

typedef struct {
	size_t point_list;     /* [R] u32 Gaussian ids sorted by (tile, depth, id) -- the reference's point_list; with the column-pair binning its
	                          first L <= R entries: the instances in tiles their splat provably cannot reach at alpha = 1 / 255 are left out
	                          (gsr_geometry_layout.rshape's trim words; L = sum of the tile ranges' lengths; L = R with GSR_DEBUG_NO_TRIM) */
	size_t point_list_alt; /* [R] u32 sort ping-pong */
	size_t tile_keys;      /* [R] tile id of each sorted instance (the high word of the reference's key), tile_key_bytes each */
	size_t tile_keys_alt;  /* [R] sort ping-pong; after the forward its first R bytes are the backward's slot validity flags */
	size_t sort_table;     /* radix histogram table of the tile sort */
	size_t checkpoints;    /* [R / 512 + 2][256] float4: per-pixel (T, C) of heavy tiles every 512 list positions of their walk */
	size_t total;
	size_t tile_key_bytes; /* 2 (uint16_t: every tile id of the image is below 65 536) or 4 (uint32_t); both arrays are sized for 4 */
	size_t column_pairs;   /* 1: images of at most 256 x 256 tiles are binned by column pairs (no per-instance tile keys exist: the tile
	                          of sorted instance i follows from `ranges`; [point_list_alt, tile_keys_alt) then holds up to R interleaved
	                          8-byte records {u32 y0 | (h - 1) << 8, u32 Gaussian id}, sorted by tile column, then depth) unless
	                          GSR_DEBUG_TILE_SORT is passed; 0: always the tile sort */
} gsr_binning_layout;

int gsr_geometry_layout_of(int P, gsr_geometry_layout* out);
int gsr_image_layout_of(int width, int height, gsr_image_layout* out);
int gsr_binning_layout_of(int P, int64_t num_rendered, int width, int height, gsr_binning_layout* out);

/*
 * Forward, stage 1 of 2: per-Gaussian preprocess, instance count and its read-back, and the
 * per-Gaussian half of the sort (depth order of the Gaussians + their slot offsets), which keeps
 * running on the stream while the host already has the count.  Replaces the first half of
 * CudaRasterizer::Rasterizer::forward (cuda_rasterizer/rasterizer_impl.cu:227-331:
 * FORWARD::preprocess, cub InclusiveSum, the blocking cudaMemcpy of num_rendered).  Blocks only
 * until `*num_rendered_host` is valid.
 *   radii [P] int32 out (0 for culled Gaussians), or NULL when the caller does not want them
 *   (`int* radii = nullptr`, cuda_rasterizer/rasterizer.h:52); geometry: gsr_geometry_bytes(P) bytes.
 * Streams: all work is ordered after what `stream` holds on entry and is complete, in `stream`'s order, for whatever
 * the caller enqueues after the call.  Inside the call the SH colour kernel runs on a helper stream of the library (one
 * per host thread and device, non-blocking, forked from and joined into `stream` with events) beside the geometry
 * kernel and the depth sort; GSR_DEBUG_SERIAL or GSR_DEBUG_SYNC in `debug` and an all-stages gsr_profile_begin() keep
 * it on `stream`.  Every return of the call, error returns included, leaves `stream` ordered after the helper stream's
 * work, so `geometry` may be released or reused on `stream` as soon as the call has returned.  The read-back of the count
 * travels on a second library stream behind the geometry kernel (beside the depth sort, not in front of it); the host has
 * waited for it when the call returns.
 */
int gsr_forward_preprocess(
	int P, int D, int M,
	int width, int height,
	const float* means3D,        /* [P][3] */
	const float* shs,            /* [P][M][3] or NULL */
	const float* colors_precomp, /* [P][3] or NULL */
	const float* opacities,      /* [P] */
	const float* scales,         /* [P][3] or NULL */
	float scale_modifier,
	const float* rotations,      /* [P][4] or NULL */
	const float* cov3D_precomp,  /* [P][6] or NULL */
	const float* viewmatrix,     /* [16] */
	const float* projmatrix,     /* [16] */
	const float* cam_pos,        /* [3] */
	float tan_fovx, float tan_fovy,
	int prefiltered,
	int* radii,
	void* geometry,
	int64_t* num_rendered_host,
	void* stream, int debug);

/*
 * Forward, stage 2 of 2: (tile, Gaussian) instance emission in depth order, stable sort by tile,
 * tile ranges, per-tile blend.  Replaces rasterizer_impl.cu:333-410 (duplicateWithKeys, cub
 * SortPairs on bits [0,32+bit), identifyTileRanges, FORWARD::render); the sorted instance list
 * and the ranges are identical to the reference's.  binning: gsr_binning_bytes(P, num_rendered, w, h) bytes;
 * image: gsr_image_bytes(w, h) bytes; out_color [3][H][W] fully written (background where empty).
 */
int gsr_forward_render(
	int P, int64_t num_rendered,
	int width, int height,
	const float* background,     /* [3] */
	const int* radii,            /* unused (kept for symmetry with Rasterizer::forward); may be NULL */
	void* geometry, void* binning, void* image,
	float* out_color,
	void* stream, int debug);

/*
 * Backward.  Replaces CudaRasterizer::Rasterizer::backward (rasterizer_impl.cu:416-518:
 * BACKWARD::render, computeCov2DCUDA, BACKWARD::preprocess) and the zero-initialised gradient
 * tensors of rasterize_points.cu:168-178: every element of the eight outputs is written by this
 * call (zeros for culled Gaussians), so the caller may pass uninitialised memory.
 *   dL_dpix [3][H][W]; outputs: dL_dmean2D [P][3], dL_dconic [P][4] (the reference's (P,2,2)),
 *   dL_dopacity [P], dL_dcolor [P][3], dL_dmean3D [P][3], dL_dcov3D [P][6], dL_dsh [P][M][3]
 *   (may be NULL when M == 0), dL_dscale [P][3], dL_drot [P][4].
 * dL_dconic (an intermediate of the reference), dL_dcolor when the colours came from SH and dL_dcov3D
 * when the covariances came from scales/rotations have no consumer: pass NULL to skip those writes.
 * View-parallel mode: with shs given and dL_dsh == NULL the SH gradient is not produced and
 * dL_dcolor receives dL/dRGB with the channels the forward clamped at 0 zeroed -- the 3 floats per
 * Gaussian that gsr_sh_grad_from_views() needs; every other output is unchanged.
 * radii may be NULL (`const int* radii = nullptr`, cuda_rasterizer/rasterizer.h:69): visibility is then
 * taken from the geometry state (radii > 0 <=> tiles_touched > 0, forward.cu:300-301).
 */
int gsr_backward(
	int P, int D, int M, int64_t num_rendered,
	int width, int height,
	const float* background,
	const float* means3D,
	const float* shs,
	const float* colors_precomp,
	const float* scales,
	float scale_modifier,
	const float* rotations,
	const float* cov3D_precomp,
	const float* viewmatrix,
	const float* projmatrix,
	const float* cam_pos,
	float tan_fovx, float tan_fovy,
	const int* radii,
	void* geometry, void* binning, void* image,
	void* scratch,               /* gsr_backward_scratch_bytes(P, num_rendered) bytes */
	const float* dL_dpix,
	float* dL_dmean2D,
	float* dL_dconic,
	float* dL_dopacity,
	float* dL_dcolor,
	float* dL_dmean3D,
	float* dL_dcov3D,
	float* dL_dsh,
	float* dL_dscale,
	float* dL_drot,
	void* stream, int debug);

/*
 * The backward in two stages, for callers that pipeline it.  gsr_backward() / gsr_backward_leaf() are
 * gsr_backward_blend() followed by gsr_backward_gaussians() over all P Gaussians.
 *   gsr_backward_blend      BACKWARD::render (rasterizer_impl.cu:470-495): per-tile blend gradients into `scratch`.
 *   gsr_backward_gaussians  the per-Gaussian rest (computeCov2DCUDA + BACKWARD::preprocess, :500-517) for the
 *                           Gaussians [first, first + count), first a multiple of 64 (count may end anywhere).
 *                           A view-parallel caller runs it part by part and starts the gradient exchange of a
 *                           finished part while the next one computes.  Gradient outputs are written at row
 *                           (index - out_row0): out_row0 = 0 for whole-scene tensors, = first when each part
 *                           has a buffer of its own; inputs and statistics are always indexed by Gaussian.
 * leaf = 0: the fields mean what gsr_backward()'s arguments of the same name mean (shs_rest, dL_dsh_rest unused);
 * leaf = 1: gsr_backward_leaf() semantics: means3D = xyz, shs = features_dc, shs_rest = features_rest,
 *           scales = log_scales, rotations = raw_rotations, dL_dsh / dL_dsh_rest = the two feature gradients,
 *           dL_dopacity / dL_dscale / dL_drot = gradients w.r.t. the raw leaves, dL_dcolor = optional dL_dRGB.
 * Densification statistics (train.py:157-159, scene/gaussian_model.py:599-602), each [P] or NULL,
 * updated in place for the Gaussians visible in this view (radii > 0):
 *   stat_max_radii2D = max(stat_max_radii2D, radii);  stat_xyz_gradient_accum += ||dL_dmean2D.xy||;  stat_denom += 1
 * (stat_max_radii2D needs radii != NULL).
 */
typedef struct {
	int P, D, M;
	int64_t num_rendered;
	int width, height;
	int leaf;
	const float* background;
	const float* means3D;
	const float* shs;
	const float* shs_rest;
	const float* colors_precomp;
	const float* scales;
	float scale_modifier;
	const float* rotations;
	const float* cov3D_precomp;
	const float* viewmatrix;
	const float* projmatrix;
	const float* cam_pos;
	float tan_fovx, tan_fovy;
	const int* radii;
	void* geometry; void* binning; void* image; void* scratch;
	const float* dL_dpix;
	float* dL_dmean2D; float* dL_dconic; float* dL_dopacity; float* dL_dcolor; float* dL_dmean3D;
	float* dL_dcov3D; float* dL_dsh; float* dL_dsh_rest; float* dL_dscale; float* dL_drot;
	float* stat_xyz_gradient_accum; float* stat_denom; float* stat_max_radii2D;
	void* stream;
	int debug;
} gsr_backward_args;
int gsr_backward_blend(const gsr_backward_args* args);
int gsr_backward_gaussians(const gsr_backward_args* args, int first, int count, int out_row0);

/*
 * View-parallel SH gradient (no reference counterpart; the reference is single-view, single-GPU).
 * dL/dsh of one view is the outer product basis(dir) x dL/dRGB per Gaussian (backward.cu:45-96), so
 * the sum over V views is rebuilt from V x 3 floats per Gaussian instead of exchanging 3*M:
 *   dL_dsh[g][k][c] = sum_v basis_k(normalize(means3D[g] - cam_pos[v])) * dL_dRGB[v][g][c]   (k < (D+1)^2)
 * dL_dRGB [V][P][3] are the clamp-masked colour gradients of the V views (gsr_backward, view-parallel
 * mode), view v starting at dL_dRGB + v * view_stride floats (0 = densely packed, 3 * P; an
 * all-gather of per-rank blocks that carry a trailer can be consumed in place), cam_pos [V][3];
 * dL_dsh [P][M][3] is fully written (rows >= (D+1)^2 zero).  1 <= M <= 16.
 */
int gsr_sh_grad_from_views(int P, int D, int M, int V, const float* means3D, const float* cam_pos,
                           const float* dL_dRGB, int64_t view_stride, float* dL_dsh, void* stream);

/*
 * Training loss next to the path: replaces train.py:126-128 with
 * utils/loss_utils.py:16-63 -- loss = (1 - lambda) * mean|img - gt| + lambda * (1 - SSIM(img, gt)),
 * 11x11 Gaussian window, sigma 1.5, zero padding -- AND its backward: dL_dimg [C][H][W] is dloss/dimg
 * (pass NULL for loss only), i.e. the dL_dpix that gsr_backward() consumes.  loss_out: 3 device
 * floats {loss, l1, ssim}.  scratch: gsr_loss_scratch_bytes(C, H, W) bytes, free after the call.
 */
size_t gsr_loss_scratch_bytes(int C, int H, int W);
int gsr_l1_ssim_loss(int C, int H, int W, const float* img, const float* gt, float lambda_dssim, float* loss_out,
                     float* dL_dimg, void* scratch, void* stream);

/*
 * Leaf-parameter mode: the same forward stage 1 and backward, but fed with the
 * optimiser's raw leaves of scene/gaussian_model.py:243-250 instead of their activated copies, and
 * returning gradients w.r.t. those leaves.  Replaces, around the rasterizer call, the property getters
 * of gaussian_model.py:114-135 and their autograd backward:
 *     scales    = exp(log_scales)                       rotations = F.normalize(raw_rotations)
 *     opacities = sigmoid(opacity_logits)                shs       = cat(features_dc, features_rest, dim=1)
 * features_dc [P][1][3], features_rest [P][M-1][3] (NULL when M == 1).  Stage 2 is unchanged
 * (gsr_forward_render).  Backward outputs, each fully written: dL_dmean2D [P][3] (screen-space, for
 * the densification statistic), dL_dxyz [P][3], dL_dfeatures_dc [P][1][3], dL_dfeatures_rest
 * [P][M-1][3], dL_dopacity_logits [P], dL_dlog_scales [P][3], dL_draw_rotations [P][4].
 * dL_dRGB [P][3] is optional; view-parallel mode as in gsr_backward: pass both feature gradients as
 * NULL and dL_dRGB non-NULL.
 */
int gsr_forward_preprocess_leaf(
	int P, int D, int M, int width, int height,
	const float* xyz, const float* features_dc, const float* features_rest,
	const float* opacity_logits, const float* log_scales, float scale_modifier, const float* raw_rotations,
	const float* viewmatrix, const float* projmatrix, const float* cam_pos,
	float tan_fovx, float tan_fovy, int prefiltered,
	int* radii, void* geometry, int64_t* num_rendered_host, void* stream, int debug);
int gsr_backward_leaf(
	int P, int D, int M, int64_t num_rendered, int width, int height, const float* background,
	const float* xyz, const float* features_dc, const float* features_rest,
	const float* log_scales, float scale_modifier, const float* raw_rotations,
	const float* viewmatrix, const float* projmatrix, const float* cam_pos,
	float tan_fovx, float tan_fovy, const int* radii,
	void* geometry, void* binning, void* image, void* scratch, const float* dL_dpix,
	float* dL_dmean2D, float* dL_dxyz, float* dL_dfeatures_dc, float* dL_dfeatures_rest,
	float* dL_dopacity_logits, float* dL_dlog_scales, float* dL_draw_rotations, float* dL_dRGB,
	void* stream, int debug);

/*
 * One-launch Adam over up to GSR_ADAM_MAX_GROUPS parameter tensors: replaces
 * torch.optim.Adam.step() as the reference configures it (gaussian_model.py:243-252: betas
 * (0.9, 0.999), eps 1e-15, no weight decay, no amsgrad), same arithmetic per element:
 *     exp_avg += (1-b1)*(grad-exp_avg); exp_avg_sq = exp_avg_sq*b2 + (1-b2)*grad^2;
 *     param  -= lr/(1-b1^step) * exp_avg / (sqrt(exp_avg_sq)/sqrt(1-b2^step) + eps)
 * `step` is the group's 1-based step count of THIS update.  radii: NULL (reference semantics: every
 * element is updated) or [P] int -- then rows (numel/row Gaussians of `row` floats) whose radii <= 0
 * are left untouched ("visible-only" Adam; changes the optimisation, opt-in).
 */
#define GSR_ADAM_MAX_GROUPS 8
typedef struct {
	float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
	int64_t numel; int64_t step; double lr; int32_t row;
} gsr_adam_group;
int gsr_adam_step(int ngroups, const gsr_adam_group* groups, double beta1, double beta2, double eps,
                  const int* radii, void* stream);

/*
 * Scale initialisation from the input cloud: replaces simple_knn._C.distCUDA2
 * (submodules/simple-knn/spatial.cu:14-25, simple_knn.cu:175-220), used by create_from_pcd
 * (scene/gaussian_model.py:215).  mean_dist2[i] = (d0 + d1 + d2) / 3 with d0 <= d1 <= d2 the three
 * smallest squared fp32 distances from points[i] to the OTHER points (coincident points count with 0;
 * fewer than 4 points leave FLT_MAX terms in the sum -- ~1.1e38 or +inf -- as in the reference).  Exact search.
 * points [P][3], mean_dist2 [P]; scratch: gsr_knn_scratch_bytes(P) bytes, free after the call.
 */
size_t gsr_knn_scratch_bytes(int P);
int gsr_knn_mean_dist2(int P, const float* points, float* mean_dist2, void* scratch, void* stream);

/* Replaces CudaRasterizer::Rasterizer::markVisible (rasterizer_impl.cu:162-174):
 * present[i] = 1 iff the view-space z of means3D[i] is > 0.2. */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/* rasterizer_impl.cu:37-52 (host helper; exported for tests). */
uint32_t gsr_get_higher_msb(uint32_t n);

/*
 * Per-kernel timing hook for benchmarks, per stream: between gsr_profile_begin(stream) and
 * gsr_profile_end(stream, ...) every library call that is given THIS stream records HIP events around each
 * of its stages (from whatever host thread it is made: PyTorch runs backward on its own thread);
 * gsr_profile_end() synchronises and returns (name, milliseconds) pairs.  Calls on other streams are not
 * recorded and pay nothing.  Not part of the reference surface.
 */
typedef struct { const char* name; float ms; } gsr_kernel_time;
int gsr_profile_begin(void* stream);
/* Same, recording only the stage called `stage` ("render_backward", "sort", ...): two event records per
 * step instead of two per stage, for timed regions (each record costs ~5 us of queue drain on the GPU). */
int gsr_profile_begin_only(void* stream, const char* stage);
int gsr_profile_end(void* stream, gsr_kernel_time* out, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H_INCLUDED */
