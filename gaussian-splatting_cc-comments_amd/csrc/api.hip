// api.hip -- the C ABI of libgsr_hip.so (include/gsr.h): buffer layouts, stage orchestration,
// error reporting.  Orchestration replaces CudaRasterizer::Rasterizer::{forward,backward,markVisible}
// (cuda_rasterizer/rasterizer_impl.cu:162-174, 227-411, 416-518).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "gsr_internal.h"

#define GSR_MAX_DEVICES 64
// Beside the depth sort the SH colour kernel is held to two workgroups per CU (unused dynamic LDS on top of its staging area): it has
// until the end of the depth sort to finish, and at full occupancy its memory traffic doubled the latency-bound first launches of
// that sort (histogram 6 -> 15 us, scatter 13 -> 25).  Measured at C3: step 1.167 -> 1.160 ms with two workgroups per CU (three:
// 1.161, one: 1.161).  Only while the colour kernel is the shorter of the two: its time grows with P, the depth sort's barely (C5,
// 6M Gaussians: colour 0.42 ms against 0.25 ms of sort -- there it keeps the whole chip).  (Round 4: the limit used to be a fixed
// 40 KB of dynamic LDS, which on top of the 52 KB the LEAF kernel then staged left ONE workgroup per CU -- 0.224 ms instead of 0.06, and
// the training iteration's binning waited 110 us for it: profiles/r4_train_iteration_timeline_before.txt; the leaf kernel now stages
// in halves like the packed one.)
#ifndef GSR_COLOR_BESIDE_WGS
#define GSR_COLOR_BESIDE_WGS 2
#endif
// (Measured and not kept, round 4: the colour kernel in two pieces -- 40 / 55 / 70 % of its workgroups beside the geometry kernel, the
// rest held back by an event until the bucket depth sort's last kernel had finished, i.e. beside the binning kernels instead of the
// sort's round trips -- step 1.092 -> 1.107-1.110 ms at C3: the held-back piece ends after the binning does and the join waits for it.)
#define GSR_COLOR_BESIDE_MAX_P 1500000

// ---- errors ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int gsr_fail(int code, const char* fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}

int gsr_check_hip(hipError_t e, const char* what)
{
	if (e == hipSuccess) return GSR_OK;
	return gsr_fail(GSR_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

int gsr_stage_done(hipStream_t s, int debug, const char* stage)
{
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return gsr_fail(GSR_ERR_HIP, "launch of %s failed: %s", stage, hipGetErrorString(e));
	if (debug & GSR_DEBUG_SYNC) {  // CHECK_CUDA semantics, auxiliary.h:177-184
		e = hipStreamSynchronize(s);
		if (e != hipSuccess) return gsr_fail(GSR_ERR_HIP, "[HIP ERROR] in stage %s: %s", stage, hipGetErrorString(e));
	}
	return GSR_OK;
}

extern "C" const char* gsr_last_error(void) { return g_err; }
extern "C" const char* gsr_version(void) { return "gsr-hip gfx950 r4"; }

// ---- per-kernel event profiling ----------------------------------------------------------------
// One recorder per stream that asked for it (gsr_profile_begin(stream)).  A stage looks its stream up; with no
// recorder active anywhere that is one relaxed atomic load.  Events come from a pool that only grows, so
// recording inside a timed region costs two hipEventRecord per stage and no allocation after the first step.
// Nothing here is keyed by thread: PyTorch runs the backward on its own autograd thread, one per device, and
// each of them is told apart by the stream it passes.
struct ProfEntry { const char* name; hipEvent_t a, b; };
struct ProfRecorder {
	hipStream_t stream;
	bool on = false, open = false;
	std::vector<ProfEntry> ev;
	size_t used = 0;
	char only[64] = "";  // when non-empty: record this stage only (every hipEventRecord drains the queue ~5 us)
};
static std::mutex g_prof_mu;
static std::vector<ProfRecorder*> g_prof;  // guarded by g_prof_mu; recorders are kept for reuse of their events
static std::atomic<int> g_prof_active{0};

static ProfRecorder* prof_find(hipStream_t s, bool create)
{
	for (ProfRecorder* r : g_prof)
		if (r->stream == s) return r;
	if (!create) return nullptr;
	ProfRecorder* r = new ProfRecorder();
	r->stream = s;
	g_prof.push_back(r);
	return r;
}

// Single-stage recording of a stage that is ONE kernel: the events its launcher hands to hipExtLaunchKernelGGL as start / stop
// events -- the kernel's own dispatch packet takes the two timestamps, no hipEventRecord (a barrier packet that costs the
// stream's next launch 6-8 us) stands in front of or behind it.  false: not recording this stage that way.
bool gsr_prof_kernel_events(hipStream_t s, const char* name, hipEvent_t* a, hipEvent_t* b)
{
	if (g_prof_active.load(std::memory_order_relaxed) == 0) return false;
	std::lock_guard<std::mutex> lk(g_prof_mu);
	ProfRecorder* r = prof_find(s, false);
	if (!r || !r->on || !r->only[0] || strcmp(r->only, name) != 0) return false;
	if (r->used == r->ev.size()) {
		ProfEntry e;
		e.name = name;
		(void)hipEventCreate(&e.a);
		(void)hipEventCreate(&e.b);
		r->ev.push_back(e);
	}
	ProfEntry& e = r->ev[r->used++];
	e.name = name;
	r->open = false;
	*a = e.a;
	*b = e.b;
	return true;
}

void gsr_prof_mark_begin(hipStream_t s, const char* name)
{
	if (!name || g_prof_active.load(std::memory_order_relaxed) == 0) return;
	std::lock_guard<std::mutex> lk(g_prof_mu);
	ProfRecorder* r = prof_find(s, false);
	if (!r || !r->on) return;
	r->open = false;
	if (r->only[0] && strcmp(r->only, name) != 0) return;
	r->open = true;
	if (r->used == r->ev.size()) {
		ProfEntry e;
		e.name = name;
		(void)hipEventCreate(&e.a);
		(void)hipEventCreate(&e.b);
		r->ev.push_back(e);
	}
	ProfEntry& e = r->ev[r->used++];
	e.name = name;
	(void)hipEventRecord(e.a, s);
}

void gsr_prof_mark_end(hipStream_t s)
{
	if (g_prof_active.load(std::memory_order_relaxed) == 0) return;
	std::lock_guard<std::mutex> lk(g_prof_mu);
	ProfRecorder* r = prof_find(s, false);
	if (!r || !r->on || !r->open || r->used == 0) return;
	r->open = false;
	(void)hipEventRecord(r->ev[r->used - 1].b, s);
}

// true while a recorder on this stream records EVERY stage (bench.py's per-kernel table): stages that normally
// overlap on a helper stream then run in line, so that each gets a time of its own
static bool gsr_prof_records_all(hipStream_t s)
{
	if (g_prof_active.load(std::memory_order_relaxed) == 0) return false;
	std::lock_guard<std::mutex> lk(g_prof_mu);
	ProfRecorder* r = prof_find(s, false);
	return r && r->on && !r->only[0];
}

extern "C" int gsr_profile_begin_only(void* stream, const char* stage)
{
	std::lock_guard<std::mutex> lk(g_prof_mu);
	ProfRecorder* r = prof_find((hipStream_t)stream, true);
	if (!r->on) g_prof_active.fetch_add(1);
	r->on = true;
	r->open = false;
	r->used = 0;
	r->only[0] = 0;
	if (stage) { strncpy(r->only, stage, sizeof r->only - 1); r->only[sizeof r->only - 1] = 0; }
	return GSR_OK;
}

extern "C" int gsr_profile_begin(void* stream) { return gsr_profile_begin_only(stream, nullptr); }

extern "C" int gsr_profile_end(void* stream, gsr_kernel_time* out, int capacity)
{
	std::lock_guard<std::mutex> lk(g_prof_mu);
	ProfRecorder* r = prof_find((hipStream_t)stream, false);
	if (!r || !r->on) return 0;
	r->on = false;
	g_prof_active.fetch_sub(1);
	int n = 0;
	for (size_t k = 0; k < r->used; k++) {
		ProfEntry& e = r->ev[k];
		(void)hipEventSynchronize(e.b);
		float ms = 0.f;
		(void)hipEventElapsedTime(&ms, e.a, e.b);
		if (out && n < capacity) { out[n].name = e.name; out[n].ms = ms; n++; }
	}
	r->used = 0;
	return n;
}

// ---- layouts -----------------------------------------------------------------------------------
extern "C" int gsr_geometry_layout_of(int P, gsr_geometry_layout* o)
{
	if (P < 0 || !o) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_geometry_layout_of: bad arguments");
	const size_t n = (size_t)P;
	const size_t nb = (n + GSR_PREPROCESS_BLOCK - 1) / GSR_PREPROCESS_BLOCK;
	size_t off = 0;
	o->splat = off;          off = gsr_align_up(off + n * sizeof(GsrSplat));
	o->depth_keys = off;     off = gsr_align_up(off + n * 4);
	o->depth_keys_alt = off; off = gsr_align_up(off + n * 4);
	o->perm = off;           off = gsr_align_up(off + n * 4);
	o->perm_alt = off;       off = gsr_align_up(off + n * 4);
	o->tiles_touched = off;  off = gsr_align_up(off + n * 4);
	o->rect = off;           off = gsr_align_up(off + n * 8);
	o->slot_base = off;      off = gsr_align_up(off + n * 4);
	o->clamped = off;        off = gsr_align_up(off + n);
	o->sh_ddir = off;        off = gsr_align_up(off + n * 36);
	o->status = off;         off = gsr_align_up(off + GSR_STATUS_WORDS * 4);
	o->scan_temp = off;      off = gsr_align_up(off + 2 * gsr_align_up(nb * 4));   // depth-ordered block sums, then the preprocess workgroups' tile counts
	o->sort_table = off;     off = gsr_align_up(off + gsr_radix_table_bytes(n));
	o->col_table = off;      off = gsr_align_up(off + gsr_tilebin_col_table_bytes(n));
	o->rshape = off;         off = gsr_align_up(off + n * 8);
	o->total = off;
	return GSR_OK;
}

extern "C" int gsr_image_layout_of(int W, int H, gsr_image_layout* o)
{
	if (W < 0 || H < 0 || !o) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_image_layout_of: bad arguments");
	const size_t N = (size_t)W * H, T = (size_t)gsr_grid_x(W) * gsr_grid_y(H);
	size_t off = 0;
	o->final_C = off;          off = gsr_align_up(off + 3 * N * 4);
	o->final_T = off;          off = gsr_align_up(off + N * 4);
	o->n_contrib = off;        off = gsr_align_up(off + N * 4);
	o->ranges = off;           off = gsr_align_up(off + T * 8);
	o->tile_max_contrib = off; off = gsr_align_up(off + T * 4);
	{
		const size_t extra_f = 3 * (size_t)gsr_tile_order_max_split((int)T), extra_b = gsr_tile_order_max_segments((int)T);
		o->tile_order = off;   off = gsr_align_up(off + (T + (extra_f > extra_b ? extra_f : extra_b) + 1) * 4);   // + the segment coarseness word
	}
	o->total = off;
	return GSR_OK;
}

extern "C" int gsr_binning_layout_of(int P, int64_t R, int W, int H, gsr_binning_layout* o)
{
	(void)P;
	if (R < 0 || W < 0 || H < 0 || !o) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_binning_layout_of: bad arguments");
	const size_t n = (size_t)R;
	size_t off = 0;
	o->point_list = off;     off = gsr_align_up(off + n * 4);
	o->point_list_alt = off; off = gsr_align_up(off + n * 4);
	o->tile_keys = off;      off = gsr_align_up(off + n * 4);
	o->tile_keys_alt = off;  off = gsr_align_up(off + n * 4);
	{
		const size_t a = gsr_radix_table_bytes(n), b = gsr_tilebin_row_table_bytes(n);   // tile sort (sort.hip) / pass 2 of tilebin.hip
		o->sort_table = off; off = gsr_align_up(off + (a > b ? a : b));
	}
	o->checkpoints = off;    off = gsr_align_up(off + gsr_checkpoint_records(R) * 256 * sizeof(float4));
	o->total = off;
	o->tile_key_bytes = (size_t)gsr_tile_key_bytes(gsr_grid_x(W) * gsr_grid_y(H), n);
	o->column_pairs = gsr_tilebin_applies(W, H) ? 1 : 0;
	return GSR_OK;
}

extern "C" size_t gsr_geometry_bytes(int P)
{
	gsr_geometry_layout l;
	return gsr_geometry_layout_of(P, &l) == GSR_OK ? l.total : 0;
}
extern "C" size_t gsr_image_bytes(int W, int H)
{
	gsr_image_layout l;
	return gsr_image_layout_of(W, H, &l) == GSR_OK ? l.total : 0;
}
extern "C" size_t gsr_binning_bytes(int P, int64_t R, int W, int H)
{
	gsr_binning_layout l;
	return gsr_binning_layout_of(P, R, W, H, &l) == GSR_OK ? l.total : 0;
}
extern "C" size_t gsr_backward_scratch_bytes(int P, int64_t R)
{
	(void)P;
	if (R < 0) return 0;
	return gsr_align_up((size_t)R * sizeof(GsrGradSlot));
}

GsrGeometry gsr_geometry_view(void* blob, int P)
{
	gsr_geometry_layout l;
	gsr_geometry_layout_of(P, &l);
	char* b = (char*)blob;
	GsrGeometry g;
	g.splat = (GsrSplat*)(b + l.splat);
	g.depth_keys = (uint32_t*)(b + l.depth_keys);
	g.depth_keys_alt = (uint32_t*)(b + l.depth_keys_alt);
	g.perm = (uint32_t*)(b + l.perm);
	g.perm_alt = (uint32_t*)(b + l.perm_alt);
	g.tiles_touched = (uint32_t*)(b + l.tiles_touched);
	g.rect = (uint2*)(b + l.rect);
	g.slot_base = (uint32_t*)(b + l.slot_base);
	g.clamped = (uint8_t*)(b + l.clamped);
	g.sh_ddir = (float*)(b + l.sh_ddir);
	g.status = (uint32_t*)(b + l.status);
	g.sorted_block_sums = (uint32_t*)(b + l.scan_temp);
	g.block_tiles = (uint32_t*)(b + l.scan_temp + gsr_align_up(((size_t)P + GSR_PREPROCESS_BLOCK - 1) / GSR_PREPROCESS_BLOCK * 4));
	g.sort_table = (void*)(b + l.sort_table);
	g.col_table = (void*)(b + l.col_table);
	g.rshape = (uint2*)(b + l.rshape);
	return g;
}

GsrImage gsr_image_view(void* blob, int W, int H)
{
	gsr_image_layout l;
	gsr_image_layout_of(W, H, &l);
	char* b = (char*)blob;
	GsrImage im;
	im.final_C = (float*)(b + l.final_C);
	im.final_T = (float*)(b + l.final_T);
	im.n_contrib = (uint32_t*)(b + l.n_contrib);
	im.ranges = (uint2*)(b + l.ranges);
	im.tile_max_contrib = (uint32_t*)(b + l.tile_max_contrib);
	im.tile_order = (uint32_t*)(b + l.tile_order);
	return im;
}

GsrBinning gsr_binning_view(void* blob, int P, int64_t R, int W, int H)
{
	gsr_binning_layout l;
	gsr_binning_layout_of(P, R, W, H, &l);
	char* b = (char*)blob;
	GsrBinning bn;
	bn.point_list = (uint32_t*)(b + l.point_list);
	bn.point_list_alt = (uint32_t*)(b + l.point_list_alt);
	bn.tile_keys = (uint32_t*)(b + l.tile_keys);
	bn.tile_keys_alt = (uint32_t*)(b + l.tile_keys_alt);
	bn.sort_table = (void*)(b + l.sort_table);
	bn.checkpoints = (float4*)(b + l.checkpoints);
	return bn;
}

// rasterizer_impl.cu:37-52
extern "C" uint32_t gsr_get_higher_msb(uint32_t n)
{
	uint32_t msb = sizeof(n) * 4;
	uint32_t step = msb;
	while (step > 1) {
		step /= 2;
		if (n >> msb) msb += step; else msb -= step;
	}
	if (n >> msb) msb++;
	return msb;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

// ---- per-thread helper resources -----------------------------------------------------------------
// One helper stream (+ fork / join events) and one pinned landing buffer (+ its event) per host thread and device, created
// on first use by gsr_forward_preprocess*.  They are the only things the library keeps between calls; a thread that is
// done with the library returns them with gsr_thread_release().
struct GsrThreadDevice {
	hipStream_t aux_stream = nullptr;
	hipEvent_t aux_fork = nullptr, aux_join = nullptr;
	uint32_t* status_host = nullptr;
	hipEvent_t status_event = nullptr;
	hipStream_t copy_stream = nullptr;   // the count's read-back travels beside the depth sort, not in front of it
	hipEvent_t copy_fork = nullptr;
};
struct GsrThreadState { GsrThreadDevice dev[GSR_MAX_DEVICES]; };
static thread_local GsrThreadState g_thread;

extern "C" int gsr_thread_release(void)
{
	g_err[0] = 0;
	int rc = GSR_OK;
	for (int d = 0; d < GSR_MAX_DEVICES; d++) {
		GsrThreadDevice& t = g_thread.dev[d];
		// the helper stream's work was joined into the caller's stream by every call; draining it here covers a caller
		// that releases while its own stream still runs
		if (t.aux_stream) { (void)hipStreamSynchronize(t.aux_stream); if (hipStreamDestroy(t.aux_stream) != hipSuccess) rc = GSR_ERR_HIP; }
		if (t.aux_fork && hipEventDestroy(t.aux_fork) != hipSuccess) rc = GSR_ERR_HIP;
		if (t.aux_join && hipEventDestroy(t.aux_join) != hipSuccess) rc = GSR_ERR_HIP;
		if (t.status_event && hipEventDestroy(t.status_event) != hipSuccess) rc = GSR_ERR_HIP;
		if (t.copy_stream) { (void)hipStreamSynchronize(t.copy_stream); if (hipStreamDestroy(t.copy_stream) != hipSuccess) rc = GSR_ERR_HIP; }
		if (t.copy_fork && hipEventDestroy(t.copy_fork) != hipSuccess) rc = GSR_ERR_HIP;
		if (t.status_host && hipHostFree(t.status_host) != hipSuccess) rc = GSR_ERR_HIP;
		t = GsrThreadDevice();
	}
	if (rc) { (void)hipGetLastError(); return gsr_fail(rc, "gsr_thread_release: a HIP resource could not be freed"); }
	return GSR_OK;
}

// Makes `stream` wait for the helper stream's join event when the scope ends, unless now() already did.
struct GsrJoinOnExit {
	hipStream_t s;
	hipEvent_t ev = nullptr;
	explicit GsrJoinOnExit(hipStream_t st) : s(st) {}
	void arm(hipEvent_t e) { ev = e; }
	int now()
	{
		if (!ev) return GSR_OK;
		hipEvent_t e = ev;
		ev = nullptr;
		if (hipStreamWaitEvent(s, e, 0) == hipSuccess) return GSR_OK;
		(void)hipEventSynchronize(e);  // the stream could not be made to wait: the host waits instead
		return gsr_fail(GSR_ERR_HIP, "hipStreamWaitEvent(join) failed");
	}
	~GsrJoinOnExit()
	{
		if (ev && hipStreamWaitEvent(s, ev, 0) != hipSuccess) (void)hipEventSynchronize(ev);
	}
};

// Drains the count's read-back stream when the scope ends, unless the host has already waited for the copy: the blit reads
// geometry->status, and include/gsr.h promises that nothing of it outlives the call -- on the error returns too.
struct GsrDrainOnExit {
	hipStream_t st = nullptr;
	void arm(hipStream_t s) { st = s; }
	void disarm() { st = nullptr; }
	~GsrDrainOnExit() { if (st) (void)hipStreamSynchronize(st); }
};

// What stage 1 last chose on this host thread (binning by column pairs or by the tile sort), and for which geometry buffer: stage 2
// re-derives the choice from its own `debug` argument, and a caller that passes GSR_DEBUG_TILE_SORT to one call only would otherwise
// get a point_list built from tables that were never filled.
struct GsrLastStage1 { const void* geometry = nullptr; bool col_pairs = false; };
static thread_local GsrLastStage1 g_last_stage1;

// ---- forward, stage 1 --------------------------------------------------------------------------
static int gsr_forward_preprocess_impl(int P, int D, int M, int width, int height, const float* means3D,
                                       const float* shs, const float* shs_rest, int leaf, const float* colors_precomp,
                                       const float* opacities, const float* scales, float scale_modifier,
                                       const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                       const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                                       int prefiltered, int* radii, void* geometry, int64_t* num_rendered_host,
                                       void* stream, int debug)
{
	g_err[0] = 0;
	hipStream_t s = (hipStream_t)stream;
	if (!num_rendered_host) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "num_rendered_host is NULL");
	*num_rendered_host = 0;
	if (P < 0 || width <= 0 || height <= 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "bad P / image size");
	if (P == 0) return GSR_OK;  // rasterize_points.cu:94: nothing is launched for an empty scene
	if (!means3D || !opacities || !viewmatrix || !projmatrix || !geometry)  // radii is optional (rasterizer.h:52)
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_forward_preprocess: required pointer is NULL");
	if (!colors_precomp && !shs)  // rasterizer_impl.cu:281-284
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "For non-RGB, provide precomputed Gaussian colors!");
	if (!colors_precomp && (M <= 0 || (D + 1) * (D + 1) > M || D < 0 || D > 3 || !cam_pos))
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "SH degree %d needs %d coefficients, M = %d", D, (D + 1) * (D + 1), M);
	if (!cov3D_precomp && (!scales || !rotations))
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "provide scales+rotations or cov3D_precomp");
	if (width > 65535 * GSR_TILE_X || height > 65535 * GSR_TILE_Y)
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "image too large for 16-bit tile coordinates");
	if (!aligned16(geometry)) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "geometry buffer must be 16-byte aligned");

	if (leaf && (M > 1 && !shs_rest)) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "leaf mode: features_rest is NULL");
	if (leaf && M > 16) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "leaf mode: at most 16 SH coefficients (degree 3), M = %d", M);
	GsrPreprocessArgs a = {};
	a.leaf = leaf; a.shs_rest = shs_rest;
	a.P = P; a.D = D; a.M = M; a.W = width; a.H = height;
	a.means3D = means3D; a.shs = shs; a.colors_precomp = colors_precomp; a.opacities = opacities;
	a.scales = scales; a.scale_modifier = scale_modifier; a.rotations = rotations; a.cov3D_precomp = cov3D_precomp;
	a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; a.cam_pos = cam_pos;
	a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
	a.focal_y = height / (2.0f * tan_fovy);  // rasterizer_impl.cu:251-252
	a.focal_x = width / (2.0f * tan_fovx);
	a.prefiltered = prefiltered;
	a.radii = radii;
	a.g = gsr_geometry_view(geometry, P);

	int rc;
	int device = 0;
	if ((rc = gsr_check_hip(hipGetDevice(&device), "hipGetDevice"))) return rc;
	if (device < 0 || device >= GSR_MAX_DEVICES) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "device index %d not supported", device);
	// The SH colours are needed by nothing before the blend: their kernel runs on a helper stream (one per host thread and
	// device, created on first use, freed by gsr_thread_release()) beside the geometry kernel and the depth sort, whose
	// launches are short and leave most of the chip idle, and is joined before this call's last kernel.  In line instead
	// when every stage is being timed, with GSR_DEBUG_SYNC (a device sync follows every stage) and with GSR_DEBUG_SERIAL.
	GsrThreadDevice& td = g_thread.dev[device];
	const bool col_pairs = gsr_tilebin_applies(width, height) && !(debug & GSR_DEBUG_TILE_SORT);   // (stage 2 decides the same way)
	a.trim = (col_pairs && !(debug & GSR_DEBUG_NO_TRIM)) ? 1 : 0;   // tiles a splat provably misses are left out of the column-pair binning (gsr_rect_trim.h); the tile sort bins them all
	const bool color = gsr_preprocess_needs_color(a);
	// Depth order: up to GSR_BUCKET_SORT_MAX_P Gaussians in three launches whatever the depth range (depthsort.hip: top-digit buckets,
	// then every bucket sorted inside LDS; the result lands in (depth_keys, perm) and the rectangles in depth order come with it).
	// Beyond, and with GSR_DEBUG_RADIX_DEPTH: the global LSD passes.
#ifdef GSR_AB_FORCE_RADIX_DEPTH   // (A/B builds: csrc/Makefile `variant`)
	const bool bucket = false;
#else
	const bool bucket = gsr_bucket_sort_applies(P) && !(debug & GSR_DEBUG_RADIX_DEPTH);
#endif
	bool beside = color && !(debug & (GSR_DEBUG_SYNC | GSR_DEBUG_SERIAL)) && !gsr_prof_records_all(s);
	if (beside && !td.aux_stream) {
		hipStream_t st = nullptr;
		hipEvent_t f = nullptr, j = nullptr;
		// (Measured and not kept, round 3: a lowest-priority helper stream changes nothing -- the depth sort's first histogram
		// and scatter still take 15 + 25 us beside the colour kernel instead of 6 + 13 alone; a helper stream confined to every
		// other CU with hipExtStreamCreateWithCUMask made the step 0.16 ms slower.)
		if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&f, hipEventDisableTiming) == hipSuccess &&
		    hipEventCreateWithFlags(&j, hipEventDisableTiming) == hipSuccess) {
			td.aux_stream = st; td.aux_fork = f; td.aux_join = j;
		} else {
			(void)hipGetLastError();
			if (st) (void)hipStreamDestroy(st);
			if (f) (void)hipEventDestroy(f);
			if (j) (void)hipEventDestroy(j);
			beside = false;  // no helper stream: the colour kernel runs in line
		}
	}
	// From the fork on, EVERY return -- the error returns too -- first makes the caller's stream wait for the colour kernel:
	// the caller is free to release or reuse `geometry` on `stream` the moment this call returns (include/gsr.h, stream
	// contract), and the helper stream may still be writing rgb / clamp bits / sh_ddir into it.
	GsrJoinOnExit join(s);
	// The status words start at zero (a one-workgroup kernel); its dispatch packet signals the helper stream's fork event, so the
	// helper stream starts where the caller's stream stands now (its inputs are ready there) without a barrier packet of its own
	gsr_launch_zero_status(a.g.status, s, beside ? td.aux_fork : nullptr);
	if (beside) {
		if ((rc = gsr_check_hip(hipStreamWaitEvent(td.aux_stream, td.aux_fork, 0), "hipStreamWaitEvent(fork)"))) return rc;
		gsr_launch_preprocess_color(a, td.aux_stream, P <= GSR_COLOR_BESIDE_MAX_P ? GSR_COLOR_BESIDE_WGS : 0);
		if (hipEventRecord(td.aux_join, td.aux_stream) != hipSuccess) {
			(void)hipStreamSynchronize(td.aux_stream);  // no event to wait for: wait on the host instead, then report
			return gsr_fail(GSR_ERR_HIP, "hipEventRecord(join) failed");
		}
		join.arm(td.aux_join);
	}
	// The count's read-back (a 6 us blit) goes to a stream of its own behind the geometry kernel, so that the depth sort's first
	// launch follows that kernel directly; the stream waits for an event that the geometry kernel's own dispatch packet signals
	// (hipExtLaunchKernelGGL) -- a hipEventRecord behind the kernel is a barrier packet and cost the sort's first launch ~8 us.
	// The host waits for the copy before this call returns, so nothing of it outlives the call.  In line with GSR_DEBUG_SYNC /
	// GSR_DEBUG_SERIAL, when every stage is being timed, or when the stream cannot be had.
	bool copy_beside = !(debug & (GSR_DEBUG_SYNC | GSR_DEBUG_SERIAL)) && !gsr_prof_records_all(s);
	if (copy_beside && !td.copy_stream) {
		hipStream_t st = nullptr;
		hipEvent_t f = nullptr;
		if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&f, hipEventDisableTiming) == hipSuccess) {
			td.copy_stream = st; td.copy_fork = f;
		} else {
			(void)hipGetLastError();
			if (st) (void)hipStreamDestroy(st);
			if (f) (void)hipEventDestroy(f);
			copy_beside = false;
		}
	}
	{
		GsrProfScope p(s, "preprocess");
		gsr_launch_preprocess(a, s, copy_beside ? td.copy_fork : nullptr);
	}
	if ((rc = gsr_stage_done(s, debug, "preprocess"))) return rc;
	if (color && !beside) {
		GsrProfScope p(s, "preprocess_color");
		gsr_launch_preprocess_color(a, s);
	}
	if ((rc = gsr_stage_done(s, debug, "preprocess_color"))) return rc;

	// read num_rendered back; the per-Gaussian half of the sort is enqueued behind the copy and keeps
	// the GPU busy while the host waits on the event, allocates the binning buffer and launches stage 2
	// pinned landing buffer + event, created once per (thread, device) and reused
	if (!td.status_host) {
		if ((rc = gsr_check_hip(hipHostMalloc((void**)&td.status_host, GSR_STATUS_WORDS * 4, hipHostMallocDefault), "hipHostMalloc"))) return rc;
	}
	if (!td.status_event) {
		if ((rc = gsr_check_hip(hipEventCreateWithFlags(&td.status_event, hipEventDisableTiming), "hipEventCreate"))) return rc;
	}
	uint32_t* status_host = td.status_host;
	hipEvent_t ev = td.status_event;
	if (copy_beside && (rc = gsr_check_hip(hipStreamWaitEvent(td.copy_stream, td.copy_fork, 0), "hipStreamWaitEvent(copy fork)"))) return rc;
	hipStream_t cs = copy_beside ? td.copy_stream : s;
	GsrDrainOnExit drain;   // from here to the host's wait below every return drains the copy's stream: the copy must not outlive the call
	if (copy_beside) drain.arm(td.copy_stream);
	if ((rc = gsr_check_hip(hipMemcpyAsync(status_host, a.g.status, GSR_STATUS_WORDS * 4, hipMemcpyDeviceToHost, cs), "hipMemcpyAsync(num_rendered)"))) return rc;
	if ((rc = gsr_check_hip(hipEventRecord(ev, cs), "hipEventRecord"))) return rc;
	// The depth sort orders key - min (keys = float bits of the view-space depth; min / max: partial maxima in the status
	// words, reduced by every sort workgroup).  Its first three 8-bit passes are always needed and are enqueued at once;
	// whether bits 24..31 of max - min are populated is known once the status block has landed on the host.
	const uint32_t* bias = a.g.status + GSR_STATUS_NEGMIN;
	{
		GsrProfScope p(s, "depth_sort");
		if (bucket) {
			gsr_launch_depth_bucket_sort(a.g, P, col_pairs ? gsr_tilebin_seg(a.g, P) : nullptr, s);
			if (col_pairs) gsr_launch_tilebin_col_hist(a.g, P, 0, s, true);
			else gsr_launch_sorted_block_sums(a.g, P, 0, s);
		} else {
			gsr_radix_sort_passes(a.g.depth_keys, a.g.perm, a.g.depth_keys_alt, a.g.perm_alt, (size_t)P, 32, 4, 0, 3, a.g.sort_table, bias, 4, s);
			// ... and so are the block sums over the three-pass result (the common case) -- with them, for images the column-pair
			// binning handles, the histogram of its first pass (tilebin.hip): the stream then holds work until the host, back from
			// the wait below, has launched stage 2.  A fourth pass redoes them.
			if (col_pairs) gsr_launch_tilebin_col_hist(a.g, P, 1, s);
			else gsr_launch_sorted_block_sums(a.g, P, 1, s);
			// The gradient slots are numbered in index order here too (the bucket sort's first kernel does it on the way): a kernel of its
			// own, and LAST in stage 1 -- nothing reads slot_base before the backward, and in front of the passes it ran beside the colour
			// kernel, which streams the SH rows: 48 MB took 0.10-0.125 ms there at 6 M Gaussians, 0.025 alone.
			gsr_launch_slot_base_finish(a.g.slot_base, a.g.block_tiles, a.g.status, (size_t)P, s);
		}
	}
	if ((rc = gsr_stage_done(s, debug, "depth_sort"))) return rc;
	if ((rc = gsr_check_hip(hipEventSynchronize(ev), "hipEventSynchronize(num_rendered)"))) return rc;
	drain.disarm();   // (the copy has landed)
	g_last_stage1.geometry = geometry;
	g_last_stage1.col_pairs = col_pairs;
	if (status_host[0] & 1u)  // (the join guard makes `stream` wait for the colour kernel first)
		return gsr_fail(GSR_ERR_PREFILTERED, "Point is filtered although prefiltered is set. This shouldn't happen!");
	int64_t total = 0;
	uint32_t negmin = 0, kmax = 0;
	for (int k = 0; k < GSR_COUNT_PARTS; k++) {
		total += (int64_t)status_host[4 + k];
		negmin = status_host[GSR_STATUS_NEGMIN + k] > negmin ? status_host[GSR_STATUS_NEGMIN + k] : negmin;
		kmax = status_host[GSR_STATUS_MAX + k] > kmax ? status_host[GSR_STATUS_MAX + k] : kmax;
	}
	*num_rendered_host = total;
	const uint32_t kmin = ~negmin;
	const uint32_t culled_value = kmax >= kmin ? (kmax - kmin) + 1u : 0u;   // largest biased key (what culled Gaussians sort as)
	const int fourth = !bucket && ((culled_value >> 24) != 0u || (kmax >= kmin && kmax - kmin == 0xFFFFFFFFu));
	// join: everything the caller enqueues after this call comes after the colour kernel too
	if ((rc = join.now())) return rc;
	if (fourth) {
		GsrProfScope p(s, "depth_sort");
		gsr_radix_sort_passes(a.g.depth_keys, a.g.perm, a.g.depth_keys_alt, a.g.perm_alt, (size_t)P, 32, 4, 3, 1, a.g.sort_table, bias, 4, s);
		// three passes leave the order in (depth_keys_alt, perm_alt), four in (depth_keys, perm): recorded in status[2]
		if (col_pairs) {   // the histogram's chunk sums were added to by the run over the three-pass order
			if ((rc = gsr_check_hip(hipMemsetAsync(a.g.col_table, 0, gsr_tilebin_col_clear_words((size_t)P) * 4, s), "hipMemsetAsync(col_table)"))) return rc;
			gsr_launch_tilebin_col_hist(a.g, P, 0, s);
		} else {
			gsr_launch_sorted_block_sums(a.g, P, 0, s);   // (their prefix sums are taken by the key emission itself)
		}
	}
	return gsr_stage_done(s, debug, "depth_sort");
}

extern "C" int gsr_forward_preprocess(int P, int D, int M, int width, int height, const float* means3D,
                                      const float* shs, const float* colors_precomp, const float* opacities,
                                      const float* scales, float scale_modifier, const float* rotations,
                                      const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                      const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                                      int* radii, void* geometry, int64_t* num_rendered_host, void* stream, int debug)
{
	return gsr_forward_preprocess_impl(P, D, M, width, height, means3D, shs, nullptr, 0, colors_precomp, opacities, scales,
	                                   scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx,
	                                   tan_fovy, prefiltered, radii, geometry, num_rendered_host, stream, debug);
}

extern "C" int gsr_forward_preprocess_leaf(int P, int D, int M, int width, int height, const float* xyz,
                                           const float* features_dc, const float* features_rest,
                                           const float* opacity_logits, const float* log_scales, float scale_modifier,
                                           const float* raw_rotations, const float* viewmatrix, const float* projmatrix,
                                           const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                                           int* radii, void* geometry, int64_t* num_rendered_host, void* stream,
                                           int debug)
{
	return gsr_forward_preprocess_impl(P, D, M, width, height, xyz, features_dc, features_rest, 1, nullptr, opacity_logits,
	                                   log_scales, scale_modifier, raw_rotations, nullptr, viewmatrix, projmatrix, cam_pos,
	                                   tan_fovx, tan_fovy, prefiltered, radii, geometry, num_rendered_host, stream, debug);
}

// ---- forward, stage 2 --------------------------------------------------------------------------
extern "C" int gsr_forward_render(int P, int64_t R, int width, int height, const float* background,
                                  const int* radii, void* geometry, void* binning, void* image, float* out_color,
                                  void* stream, int debug)
{
	g_err[0] = 0;
	hipStream_t s = (hipStream_t)stream;
	if (P < 0 || R < 0 || width <= 0 || height <= 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "bad sizes");
	if (P == 0) return GSR_OK;  // image stays as allocated by the caller (zero-filled in the reference)
	if (!background || !geometry || !image || !out_color || (R > 0 && !binning))
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_forward_render: required pointer is NULL");
	if (!aligned16(geometry) || !aligned16(image) || !aligned16(binning))
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "state buffers must be 16-byte aligned");
	if (R > 0xffffffffLL) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "num_rendered exceeds 32-bit offsets");
	if ((int64_t)gsr_grid_x(width) * gsr_grid_y(height) >= (1 << 28))  // dispatch-list entries keep the tile id in 28 bits
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "image too large: more than 2^28 tiles");

	GsrGeometry g = gsr_geometry_view(geometry, P);
	GsrImage im = gsr_image_view(image, width, height);
	const int ntiles = gsr_grid_x(width) * gsr_grid_y(height);
	int rc;
	int key_bytes = 4;
	GsrBinning b;
	memset(&b, 0, sizeof b);
	const bool col_pairs = gsr_tilebin_applies(width, height) && !(debug & GSR_DEBUG_TILE_SORT);   // (as stage 1 decided)
	if (g_last_stage1.geometry == geometry && g_last_stage1.col_pairs != col_pairs)
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_forward_render: GSR_DEBUG_TILE_SORT must be passed to both forward calls or to neither "
		                "(gsr_forward_preprocess on this thread prepared this geometry buffer for the other binning)");
	if (R > 0) b = gsr_binning_view(binning, P, R, width, height);
	if (R > 0 && col_pairs) {
		// column pairs by tile column, their instances by tile row (tilebin.hip): point_list, ranges and the cleared validity
		// bytes come out of the second pass; no per-instance key is ever stored
		{
			GsrProfScope p(s, "col_scatter");
			gsr_launch_tilebin_col_scatter(g, P, b, R, s);
		}
		if ((rc = gsr_stage_done(s, debug, "col_scatter"))) return rc;
		{
			GsrProfScope p(s, "row_hist");
			gsr_launch_tilebin_row_hist(g, P, b, R, s);
		}
		if ((rc = gsr_stage_done(s, debug, "row_hist"))) return rc;
		{
			GsrProfScope p(s, "row_scatter");
			gsr_launch_tilebin_row_scatter(g, P, b, R, im.ranges, width, height, s);
		}
		if ((rc = gsr_stage_done(s, debug, "row_scatter"))) return rc;
	} else {
		if (R > 0) {
			const int bit = (int)gsr_get_higher_msb((uint32_t)ntiles);  // key bits above the depth word, rasterizer_impl.cu:355
			// emit into whichever buffer pair makes the ping-pong sort finish in (tile_keys, point_list)
			const bool even = gsr_radix_num_passes(bit) % 2 == 0;
			uint32_t *k0 = even ? b.tile_keys : b.tile_keys_alt, *v0 = even ? b.point_list : b.point_list_alt;
			uint32_t *k1 = even ? b.tile_keys_alt : b.tile_keys, *v1 = even ? b.point_list_alt : b.point_list;
			key_bytes = gsr_tile_key_bytes(ntiles, (size_t)R);  // 16-bit tile ids whenever they fit: a quarter less traffic in emission, sort, ranges
			{
				GsrProfScope p(s, "duplicate_keys");
				gsr_launch_duplicate_keys(g, P, width, k0, key_bytes, v0, (uint32_t*)b.sort_table, gsr_radix_clear_words((size_t)R), s);
			}
			if ((rc = gsr_stage_done(s, debug, "duplicate_keys"))) return rc;
			{
				GsrProfScope p(s, "sort");
				int in_first = 1;
				gsr_radix_sort_u32(k0, v0, k1, v1, (size_t)R, bit, b.sort_table, &in_first, 0, key_bytes, s);  // chunk sums zeroed by duplicate_keys
			}
			if ((rc = gsr_stage_done(s, debug, "sort"))) return rc;
		}
		{
			GsrProfScope p(s, "tile_ranges");
			gsr_launch_tile_ranges(b.tile_keys, key_bytes, R, im.ranges, ntiles, b.tile_keys_alt, s);
		}
		if ((rc = gsr_stage_done(s, debug, "tile_ranges"))) return rc;
	}
	if (R > 0) {  // (measured at C3: the forward's own order is worth 33 us of blend time for ~12 us of this kernel and its launch)
		GsrProfScope p(s, "tile_order");
		gsr_launch_tile_order(im, ntiles, false, R, !(debug & GSR_DEBUG_NO_SPLIT), s, col_pairs);
	}
	if ((rc = gsr_stage_done(s, debug, "tile_order"))) return rc;
	{
		GsrProfScope p(s, "render_forward");
				gsr_launch_render_forward(width, height, im, b.point_list, g.splat, b.checkpoints, background, out_color, R > 0, !(debug & GSR_DEBUG_NO_CULL), s);
	}
	return gsr_stage_done(s, debug, "render_forward");
}

// ---- backward ----------------------------------------------------------------------------------
static int gsr_backward_check(const gsr_backward_args& a, const char* who)
{
	if (a.P < 0 || a.num_rendered < 0 || a.width <= 0 || a.height <= 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "%s: bad sizes", who);
	if (a.P == 0) return GSR_OK;
	if (!a.background || !a.means3D || !a.viewmatrix || !a.projmatrix || !a.geometry || !a.image || !a.dL_dpix ||
	    !a.dL_dmean2D || !a.dL_dopacity || !a.dL_dmean3D || !a.dL_dscale || !a.dL_drot || (a.num_rendered > 0 && (!a.binning || !a.scratch)))
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: required pointer is NULL");
	// dL_dconic is an intermediate; dL_dcolor / dL_dcov3D are only results when the colours / covariances were
	// inputs (or, dL_dcolor, in view-parallel mode): NULL = not written
	if (!a.leaf && ((a.colors_precomp && !a.dL_dcolor) || (a.cov3D_precomp && !a.dL_dcov3D) || (a.shs && !a.dL_dsh && !a.dL_dcolor)))
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: the gradient of a provided input is NULL");
	if (a.leaf) {
		if (!a.shs || !a.scales || !a.rotations || (a.M > 1 && !a.shs_rest))
			return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward_leaf: a leaf tensor is NULL");
		if (a.M > 16) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward_leaf: at most 16 SH coefficients (degree 3), M = %d", a.M);
		if ((a.dL_dsh == nullptr) != (a.dL_dsh_rest == nullptr) && a.M > 1)
			return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward_leaf: pass both feature gradients or neither");
		if (!a.dL_dsh && !a.dL_dcolor)
			return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward_leaf: without feature gradients dL_dRGB is required");
	}
	if (a.stat_max_radii2D && !a.radii)
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: stat_max_radii2D needs radii");
	if (!aligned16(a.geometry) || !aligned16(a.image) || !aligned16(a.binning) || !aligned16(a.scratch))
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "state buffers must be 16-byte aligned");
	return GSR_OK;
}

// Validity bytes of the gradient slots: the tile sort's dead ping-pong buffer, cleared at the end of the forward.
// Which slots get written depends on the forward alone (not on dL_dpix), so a second backward over the
// same forward state (retain_graph) finds exactly the bytes it would set itself.
static uint8_t* gsr_slot_valid_of(const gsr_backward_args& a)
{
	if (a.num_rendered <= 0) return nullptr;
	return (uint8_t*)gsr_binning_view(a.binning, a.P, a.num_rendered, a.width, a.height).tile_keys_alt;
}

extern "C" int gsr_backward_blend(const gsr_backward_args* args)
{
	g_err[0] = 0;
	if (!args) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward_blend: args is NULL");
	const gsr_backward_args& a = *args;
	int rc;
	if ((rc = gsr_backward_check(a, "gsr_backward_blend"))) return rc;
	if (a.P == 0 || a.num_rendered == 0) return GSR_OK;
	hipStream_t s = (hipStream_t)a.stream;
	GsrGeometry g = gsr_geometry_view(a.geometry, a.P);
	GsrImage im = gsr_image_view(a.image, a.width, a.height);
	GsrBinning b = gsr_binning_view(a.binning, a.P, a.num_rendered, a.width, a.height);
	{
		GsrProfScope p(s, "tile_order");
		gsr_launch_tile_order(im, gsr_grid_x(a.width) * gsr_grid_y(a.height), true, a.num_rendered, !(a.debug & GSR_DEBUG_NO_SPLIT), s);
	}
	if ((rc = gsr_stage_done(s, a.debug, "tile_order"))) return rc;
	{
		// (recorded alone -- bench.py's timed region -- the kernel's dispatch packet takes the timestamps itself)
		hipEvent_t t0 = nullptr, t1 = nullptr;
		const bool own = gsr_prof_kernel_events(s, "render_backward", &t0, &t1);
		GsrProfScope p(s, own ? nullptr : "render_backward");
		gsr_launch_render_backward(a.width, a.height, im, b.point_list, g.splat, b.checkpoints, g.slot_base, a.background, a.dL_dpix,
		                           (GsrGradSlot*)a.scratch, (uint8_t*)b.tile_keys_alt, !(a.debug & GSR_DEBUG_NO_CULL), s, t0, t1);
	}
	return gsr_stage_done(s, a.debug, "render_backward");
}

extern "C" int gsr_backward_gaussians(const gsr_backward_args* args, int first, int count, int out_row0)
{
	g_err[0] = 0;
	if (!args) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward_gaussians: args is NULL");
	const gsr_backward_args& b = *args;
	int rc;
	if ((rc = gsr_backward_check(b, "gsr_backward_gaussians"))) return rc;
	if (first < 0 || count < 0 || (int64_t)first + count > b.P || (first & 63) || (out_row0 != 0 && out_row0 != first))
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward_gaussians: bad range [%d, %d + %d) of %d Gaussians (first must be a multiple "
		                "of 64, out_row0 must be 0 or first)", first, first, count, b.P);
	if (count == 0) return GSR_OK;
	hipStream_t s = (hipStream_t)b.stream;
	GsrGaussianBackwardArgs a = {};
	a.leaf = b.leaf; a.shs_rest = b.shs_rest; a.dL_dsh_rest = b.dL_dsh_rest;
	a.P = b.P; a.D = b.D; a.M = b.M; a.W = b.width; a.H = b.height;
	a.first = first; a.count = count; a.out_row0 = out_row0;
	a.means3D = b.means3D; a.shs = b.shs; a.colors_precomp = b.colors_precomp; a.scales = b.scales;
	a.scale_modifier = b.scale_modifier; a.rotations = b.rotations; a.cov3D_precomp = b.cov3D_precomp;
	a.viewmatrix = b.viewmatrix; a.projmatrix = b.projmatrix; a.cam_pos = b.cam_pos;
	a.tan_fovx = b.tan_fovx; a.tan_fovy = b.tan_fovy;
	a.focal_y = b.height / (2.0f * b.tan_fovy);
	a.focal_x = b.width / (2.0f * b.tan_fovx);
	a.radii = b.radii; a.g = gsr_geometry_view(b.geometry, b.P);
	a.slots = (const GsrGradSlot*)b.scratch; a.slot_valid = gsr_slot_valid_of(b);
	a.dL_dmean2D = b.dL_dmean2D; a.dL_dconic = b.dL_dconic; a.dL_dopacity = b.dL_dopacity; a.dL_dcolor = b.dL_dcolor;
	a.dL_dmean3D = b.dL_dmean3D; a.dL_dcov3D = b.dL_dcov3D; a.dL_dsh = b.dL_dsh; a.dL_dscale = b.dL_dscale; a.dL_drot = b.dL_drot;
	a.stat_xyz_gradient_accum = b.stat_xyz_gradient_accum; a.stat_denom = b.stat_denom; a.stat_max_radii2D = b.stat_max_radii2D;
	{
		GsrProfScope p(s, "gaussian_backward");
		gsr_launch_gaussian_backward(a, s);
	}
	return gsr_stage_done(s, b.debug, "gaussian_backward");
}

static int gsr_backward_whole(const gsr_backward_args& a)
{
	int rc;
	if ((rc = gsr_backward_blend(&a))) return rc;
	return gsr_backward_gaussians(&a, 0, a.P, 0);
}

extern "C" int gsr_backward(int P, int D, int M, int64_t R, int width, int height, const float* background,
                            const float* means3D, const float* shs, const float* colors_precomp,
                            const float* scales, float scale_modifier, const float* rotations,
                            const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                            const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii, void* geometry,
                            void* binning, void* image, void* scratch, const float* dL_dpix, float* dL_dmean2D,
                            float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                            float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, void* stream,
                            int debug)
{
	gsr_backward_args a = {};
	a.P = P; a.D = D; a.M = M; a.num_rendered = R; a.width = width; a.height = height; a.leaf = 0;
	a.background = background; a.means3D = means3D; a.shs = shs; a.colors_precomp = colors_precomp; a.scales = scales;
	a.scale_modifier = scale_modifier; a.rotations = rotations; a.cov3D_precomp = cov3D_precomp; a.viewmatrix = viewmatrix;
	a.projmatrix = projmatrix; a.cam_pos = cam_pos; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.radii = radii;
	a.geometry = geometry; a.binning = binning; a.image = image; a.scratch = scratch; a.dL_dpix = dL_dpix;
	a.dL_dmean2D = dL_dmean2D; a.dL_dconic = dL_dconic; a.dL_dopacity = dL_dopacity; a.dL_dcolor = dL_dcolor;
	a.dL_dmean3D = dL_dmean3D; a.dL_dcov3D = dL_dcov3D; a.dL_dsh = dL_dsh; a.dL_dscale = dL_dscale; a.dL_drot = dL_drot;
	a.stream = stream; a.debug = debug;
	return gsr_backward_whole(a);
}

extern "C" int gsr_backward_leaf(int P, int D, int M, int64_t R, int width, int height, const float* background,
                                 const float* xyz, const float* features_dc, const float* features_rest,
                                 const float* log_scales, float scale_modifier, const float* raw_rotations,
                                 const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                                 float tan_fovy, const int* radii, void* geometry, void* binning, void* image,
                                 void* scratch, const float* dL_dpix, float* dL_dmean2D, float* dL_dxyz,
                                 float* dL_dfeatures_dc, float* dL_dfeatures_rest, float* dL_dopacity_logits,
                                 float* dL_dlog_scales, float* dL_draw_rotations, float* dL_dRGB, void* stream, int debug)
{
	gsr_backward_args a = {};
	a.P = P; a.D = D; a.M = M; a.num_rendered = R; a.width = width; a.height = height; a.leaf = 1;
	a.background = background; a.means3D = xyz; a.shs = features_dc; a.shs_rest = features_rest; a.scales = log_scales;
	a.scale_modifier = scale_modifier; a.rotations = raw_rotations; a.viewmatrix = viewmatrix; a.projmatrix = projmatrix;
	a.cam_pos = cam_pos; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.radii = radii;
	a.geometry = geometry; a.binning = binning; a.image = image; a.scratch = scratch; a.dL_dpix = dL_dpix;
	a.dL_dmean2D = dL_dmean2D; a.dL_dopacity = dL_dopacity_logits; a.dL_dcolor = dL_dRGB; a.dL_dmean3D = dL_dxyz;
	a.dL_dsh = dL_dfeatures_dc; a.dL_dsh_rest = dL_dfeatures_rest; a.dL_dscale = dL_dlog_scales; a.dL_drot = dL_draw_rotations;
	a.stream = stream; a.debug = debug;
	return gsr_backward_whole(a);
}

extern "C" size_t gsr_loss_scratch_bytes(int C, int H, int W)
{
	if (C <= 0 || H <= 0 || W <= 0) return 0;
	size_t a, b;
	int n;
	return gsr_loss_scratch_layout(C, H, W, &a, &b, &n);
}

extern "C" int gsr_l1_ssim_loss(int C, int H, int W, const float* img, const float* gt, float lambda_dssim, float* loss_out,
                                float* dL_dimg, void* scratch, void* stream)
{
	g_err[0] = 0;
	if (C <= 0 || H <= 0 || W <= 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_l1_ssim_loss: bad image size");
	if (!img || !gt || !loss_out || !scratch) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_l1_ssim_loss: NULL pointer");
	gsr_launch_l1_ssim(C, H, W, img, gt, lambda_dssim, loss_out, dL_dimg, scratch, (hipStream_t)stream);
	return gsr_stage_done((hipStream_t)stream, 0, "l1_ssim_loss");
}

extern "C" size_t gsr_knn_scratch_bytes(int P) { return gsr_knn_scratch_size(P); }

extern "C" int gsr_knn_mean_dist2(int P, const float* points, float* mean_dist2, void* scratch, void* stream)
{
	g_err[0] = 0;
	if (P < 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_knn_mean_dist2: negative point count");
	if (P == 0) return GSR_OK;
	if (!points || !mean_dist2 || !scratch) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_knn_mean_dist2: required pointer is NULL");
	if (!aligned16(scratch)) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_knn_mean_dist2: scratch must be 16-byte aligned");
	gsr_launch_knn(P, points, mean_dist2, scratch, (hipStream_t)stream);
	return gsr_check_hip(hipGetLastError(), "knn kernels");
}

extern "C" int gsr_adam_step(int ngroups, const gsr_adam_group* groups, double beta1, double beta2, double eps,
                             const int* radii, void* stream)
{
	g_err[0] = 0;
	if (ngroups < 0 || ngroups > GSR_ADAM_MAX_GROUPS) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_adam_step: 0..%d groups", GSR_ADAM_MAX_GROUPS);
	if (ngroups == 0) return GSR_OK;
	if (!groups) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_adam_step: groups is NULL");
	for (int k = 0; k < ngroups; k++) {
		const gsr_adam_group& g = groups[k];
		if (g.numel < 0 || g.step < 1 || (g.numel > 0 && (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq)))
			return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_adam_step: group %d: NULL tensor, negative size or step < 1", k);
		if (g.numel > ((int64_t)1 << 41)) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_adam_step: group %d too large", k);
		if (radii && (g.row <= 0 || g.numel % g.row != 0))
			return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_adam_step: group %d: numel is not a multiple of row", k);
	}
	int64_t total = 0;
	for (int k = 0; k < ngroups; k++) total += groups[k].numel;
	if (total == 0) return GSR_OK;  // nothing to launch
	hipStream_t s = (hipStream_t)stream;
	{
		GsrProfScope p(s, "adam");
		gsr_launch_adam(ngroups, groups, beta1, beta2, eps, radii, s);
	}
	return gsr_check_hip(hipGetLastError(), "gsr_adam_kernel launch");
}

extern "C" int gsr_sh_grad_from_views(int P, int D, int M, int V, const float* means3D, const float* cam_pos,
                                      const float* dL_dRGB, int64_t view_stride, float* dL_dsh, void* stream)
{
	g_err[0] = 0;
	if (P < 0 || V < 0 || D < 0 || D > 3 || M <= 0 || M > 16 || (D + 1) * (D + 1) > M)
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_sh_grad_from_views: bad sizes (P %d, V %d, D %d, M %d)", P, V, D, M);
	if (P == 0) return GSR_OK;
	if (!means3D || !dL_dsh || (V > 0 && (!cam_pos || !dL_dRGB)))
		return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_sh_grad_from_views: NULL pointer");
	if (view_stride == 0) view_stride = (int64_t)P * 3;
	if (view_stride < (int64_t)P * 3) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_sh_grad_from_views: view_stride < 3 * P");
	gsr_launch_sh_grad_from_views(P, D, M, V, means3D, cam_pos, dL_dRGB, view_stride, dL_dsh, (hipStream_t)stream);
	return gsr_stage_done((hipStream_t)stream, 0, "sh_grad_from_views");
}

extern "C" int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                                uint8_t* present, void* stream)
{
	g_err[0] = 0;
	(void)projmatrix;
	if (P < 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "bad P");
	if (P == 0) return GSR_OK;
	if (!means3D || !viewmatrix || !present) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "gsr_mark_visible: NULL pointer");
	gsr_launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
	return gsr_stage_done((hipStream_t)stream, 0, "mark_visible");
}
