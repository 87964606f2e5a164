"""Per-tile wave clocks of the two blend kernels: where does a launch's time go once the last tiles have started?

Uses the diagnostic twin of the library (`make -C gaussian-splatting_cc-comments_amd/csrc tile_clock` ->
libgsr_hip_tileclock.so, render_common.h GSR_TILE_CLOCK): every wave (= tile) records its start and end on the
100 MHz constant clock and the hardware ids of the SIMD it ran on.  From one recorded step the script derives, per
kernel:

  * launch span (first start -> last end), the sum of wave durations, the mean number of resident waves;
  * per SIMD: busy span, finishing time; the distribution of finishing times relative to the end of the launch
    (the "tail": wave slots that sit empty while the last tiles run);
  * an LPT bound: sum(wave time) / (SIMDs x waves per SIMD), i.e. the span a perfectly packed launch of the same
    waves (at the same per-wave speed) would need.

  usage (GPU box, repo root):  python tools/tile_clock.py [--config C3] [--out profiles/r2_tile_clock.txt]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian-splatting_cc-comments_amd")
TWIN = os.path.join(PKG, "libgsr_hip_tileclock.so")
sys.path.insert(0, PKG)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def analyse(name, rec, waves_per_simd, out, forward=False):
    rec = rec.astype(np.int64)
    rec = rec[rec[:, 1] > 0]
    t0, t1 = rec[:, 0], rec[:, 1]
    hw, xcc = rec[:, 2], rec[:, 3] & 15
    simd = ((xcc & 7) << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3)
    start, end = t0.min(), t1.max()
    span = (end - start) * 0.01  # us
    dur = (t1 - t0) * 0.01
    ids, inv = np.unique(simd, return_inverse=True)
    nsimd = len(ids)
    finish = np.zeros(nsimd)
    first = np.full(nsimd, 1e18)
    busy = np.zeros(nsimd)
    count = np.zeros(nsimd, np.int64)
    np.maximum.at(finish, inv, (t1 - start) * 0.01)
    np.minimum.at(first, inv, (t0 - start) * 0.01)
    np.add.at(busy, inv, dur)
    np.add.at(count, inv, 1)
    # resident waves over time (10 ns steps): how long the chip ran with fewer waves than it has slots
    ticks = int(end - start) + 1
    delta = np.zeros(ticks + 1, np.int64)
    np.add.at(delta, (t0 - start), 1)
    np.add.at(delta, (t1 - start), -1)
    resident = np.cumsum(delta)[:ticks]
    slots = nsimd * waves_per_simd
    print(f"== {name}: {len(rec)} waves on {nsimd} SIMDs, launch span {span:.1f} us", file=out)
    if rec.shape[1] >= 6:   # shader clock over the waves' own lifetimes (s_memtime ticks per 100 MHz tick)
        ghz = float((rec[:, 5] - rec[:, 4]).sum()) / float((t1 - t0).sum()) * 0.1
        print(f"   shader clock while the waves ran: {ghz:.3f} GHz (s_memtime / s_memrealtime over all waves)", file=out)
    if forward and rec.shape[1] >= 8 and rec[:, 7].any():   # the forward's record: dispatch entry, then its work counters
        sb = rec[:, 7]
        staged, bands, hits = int((sb & 0xFFFFF).sum()), int(((sb >> 20) & 0x3FFFFF).sum()), int(((sb >> 42) & 0x3FFFFF).sum())
        whole = (rec[:, 6] >> 28) == 0
        print(f"   work: {staged} instances staged after the band cull ({staged / max(len(rec), 1):.0f} per wave), {bands} bands of 16x4 pixels evaluated "
              f"({bands / max(staged, 1):.2f} per staged instance), {hits} of them with at least one pixel blending the instance ({hits / max(bands, 1):.2f}); "
              f"{int(whole.sum())} whole-tile waves, {int((~whole).sum())} one-band waves", file=out)
        # what a wave's duration is made of: least squares of the duration on (1, staged, bands)
        A = np.stack([np.ones(len(rec)), (sb & 0xFFFFF).astype(float), ((sb >> 20) & 0x3FFFFF).astype(float)], axis=1)
        d_us = (t1 - t0) * 0.01
        coef, *_ = np.linalg.lstsq(A, d_us, rcond=None)
        pred = A @ coef
        print(f"   duration ~ {coef[0]:.1f} us + {coef[1] * 1e3:.1f} ns per staged instance + {coef[2] * 1e3:.1f} ns per evaluated band  "
              f"(R^2 {1 - ((d_us - pred) ** 2).sum() / ((d_us - d_us.mean()) ** 2).sum():.2f}: the rest is who shared the SIMD)", file=out)
    elif rec.shape[1] >= 8 and rec[:, 6].any():
        sa, sb = rec[:, 6], rec[:, 7]
        staged, pairs, pairs_hit = int((sa & 0xFFFFF).sum()), int(((sa >> 20) & 0x3FFFFF).sum()), int(((sa >> 42) & 0x3FFFFF).sum())
        red, lanes = int((sb & 0xFFFFFF).sum()), int((sb >> 24).sum())
        print(f"   work: {staged} instances staged after the band cull, {pairs} pixel pairs evaluated ({pairs / max(staged, 1):.2f} per staged instance), "
              f"{pairs_hit} of them with a hit ({pairs_hit / max(pairs, 1):.2f}), {lanes / max(pairs_hit, 1):.1f} of 64 lanes hit per such pair, "
              f"{red} wave reductions ({red / max(staged, 1):.2f} per staged instance)", file=out)
    print(f"   wave duration: mean {dur.mean():.1f}  median {np.median(dur):.1f}  p90 {np.percentile(dur, 90):.1f}  max {dur.max():.1f} us;"
          f"  waves per SIMD: min {count.min()} mean {count.mean():.2f} max {count.max()}", file=out)
    print(f"   sum of wave durations {dur.sum() / 1e3:.2f} ms = {dur.sum() / span:.0f} resident waves on average "
          f"({dur.sum() / span / nsimd:.2f} per SIMD of {waves_per_simd} slots)", file=out)
    print(f"   packed bound: sum / ({nsimd} SIMDs x {waves_per_simd}) = {dur.sum() / slots:.1f} us  ->  the launch is {span / (dur.sum() / slots):.2f}x that", file=out)
    for frac in (1.0, 0.9, 0.75, 0.5):
        below = np.nonzero(resident < frac * slots)[0]
        # time from the moment the resident count last drops below frac*slots for good, to the end
        last_full = np.nonzero(resident >= frac * slots)[0]
        tail = (ticks - 1 - last_full.max()) * 0.01 if len(last_full) else span
        print(f"   resident waves < {frac:.2f} x slots during {len(below) * 0.01:.1f} us in total; for good over the last {tail:.1f} us", file=out)
    q = np.percentile(span - finish, [0, 10, 50, 90, 100])
    print(f"   a SIMD's last wave ends before the launch does by: min {q[0]:.1f}  p10 {q[1]:.1f}  median {q[2]:.1f}  p90 {q[3]:.1f}  max {q[4]:.1f} us"
          f"  (mean {np.mean(span - finish):.1f} us = {np.mean(span - finish) / span * 100:.1f} % of the span)", file=out)
    print(f"   first wave of a SIMD starts after: median {np.median(first):.1f}  max {first.max():.1f} us", file=out)
    # speed of a wave as a function of how crowded its SIMD was is not separable here; report the per-SIMD busy share
    print(f"   per-SIMD sum of wave durations / span: min {(busy / span).min():.2f} mean {(busy / span).mean():.2f} max {(busy / span).max():.2f}", file=out)
    # dispatch order check: start time against duration rank (LPT: long tiles first)
    order = np.argsort(t0, kind="stable")
    k = len(order) // 4
    print(f"   mean duration of the first / last quarter of waves to start: {dur[order[:k]].mean():.1f} / {dur[order[-k:]].mean():.1f} us", file=out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--out", default=None)
    ap.add_argument("--dump", default=None, help="npz with the raw clocks and the dispatch keys")
    ap.add_argument("--scene", default="uniform", choices=["uniform", "blob", "lowop"],
                    help="uniform: the benchmark scene of --config; blob / lowop: the non-uniform 1M-Gaussian scenes of tools/skew_bench.py")
    ap.add_argument("--forward-key", action="store_true", help="also dispatch the forward by the work / durations just measured")
    ap.add_argument("--backward-key-length", action="store_true", help="also dispatch the backward by the forward's key and by the tile id")
    ap.add_argument("--measured-key", action="store_true", help="also dispatch the backward by its own measured durations")
    a = ap.parse_args()
    if not os.path.exists(TWIN):
        raise SystemExit(f"{TWIN} missing: make -C gaussian-splatting_cc-comments_amd/csrc tile_clock")
    import gsr_scene
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
    _C.use_library(TWIN)   # the diagnostic twin instead of the product library
    L = _C.lib()
    dev = torch.device("cuda:0")
    P, W, H, D, mu = gsr_scene.CONFIGS[a.config]
    scene = gsr_scene.make_scene(P, mu, D, seed=0)
    if a.scene != "uniform":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import skew_bench
        scene = skew_bench.make_skewed_scene(a.scene, P, mu, D)
    cam = gsr_scene.make_camera(W, H)
    to = lambda t: t.to(dev)
    params = dict(means3D=to(scene.means3D).requires_grad_(True), shs=to(scene.shs).requires_grad_(True),
                  opacities=to(scene.opacities).requires_grad_(True), scales=to(scene.scales).requires_grad_(True),
                  rotations=to(scene.rotations).requires_grad_(True))
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=to(scene.bg), scale_modifier=1.0,
        viewmatrix=to(cam.world_view_transform), projmatrix=to(cam.full_proj_transform), sh_degree=D,
        campos=to(cam.camera_center), prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(settings)
    dpix = to(torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)))
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    # one record per DISPATCH ENTRY (a heavy tile is several: four band waves in the forward, one wave per depth segment in the
    # backward): room for the whole lists; unused entries keep t1 = 0 and are dropped by analyse()
    nrec = ntiles + 3 * min(2048, ntiles // 4) + min(4096, ntiles // 2) + 1
    bufs = {k: torch.zeros(nrec, 8, dtype=torch.int64, device=dev) for k in ("forward", "backward")}

    captured = {}
    orig = _C.rasterize_gaussians

    def spy(*args):
        r = orig(*args)
        captured["img"] = r[5]
        return r
    _C.rasterize_gaussians = spy

    def step():
        for p in params.values():
            p.grad = None
        means2D = torch.zeros_like(params["means3D"], requires_grad=True)
        color, _ = rasterizer(means3D=params["means3D"], means2D=means2D, **{k: v for k, v in params.items() if k != "means3D"})
        color.backward(dpix)

    for _ in range(40):   # the first steps of a fresh process run at a lower shader clock (bench.py, settling steps)
        step()
    torch.cuda.synchronize()
    for k, b in bufs.items():
        fn = getattr(L, f"gsr_debug_tile_clock_{k}")
        fn.argtypes = [ctypes.c_void_p]
        assert fn(b.data_ptr()) == 0
    step()
    torch.cuda.synchronize()
    for k in bufs:
        getattr(L, f"gsr_debug_tile_clock_{k}")(None)
    out = open(a.out, "w") if a.out else sys.stdout
    print(f"# tools/tile_clock.py --config {a.config} --scene {a.scene}: P={P} {W}x{H}, {ntiles} tiles, one wave64 per tile; clocks on the 100 MHz constant clock", file=out)
    analyse("render_forward (6 waves per SIMD, 80 VGPRs)", bufs["forward"].cpu().numpy(), 6, out, forward=True)
    analyse("render_backward (4 waves per SIMD, 128 VGPRs)", bufs["backward"].cpu().numpy(), 4, out)
    # how well does the dispatch key predict a tile's duration?
    il = _C.image_layout(W, H)
    img = captured["img"].cpu().numpy()
    ranges = img[il.ranges:il.ranges + 8 * ntiles].view(np.uint32).reshape(ntiles, 2).astype(np.int64)
    tmc = img[il.tile_max_contrib:il.tile_max_contrib + 4 * ntiles].view(np.uint32).astype(np.int64)
    length = ranges[:, 1] - ranges[:, 0]
    staged = np.minimum(length, tmc)
    if a.scene != "uniform":   # (per-tile keys against per-ENTRY durations only line up when nothing is split)
        if a.out:
            out.close()
            print(open(a.out).read())
        return
    f, b = bufs["forward"].cpu().numpy()[:ntiles], bufs["backward"].cpu().numpy()[:ntiles]
    # records are in dispatch order: bring them into tile order through the dispatch lists (whole-tile entries only)
    order_f = f[:, 6] & 0x0FFFFFFF
    tmp = np.zeros_like(f); tmp[order_f] = f; f = tmp
    tile_order_b = img[il.tile_order:il.tile_order + 4 * ntiles].view(np.uint32).astype(np.int64) & 0x0FFFFFFF
    tmp = np.zeros_like(b); tmp[tile_order_b] = b; b = tmp
    fd, bd = (f[:, 1] - f[:, 0]) * 0.01, (b[:, 1] - b[:, 0]) * 0.01
    rank = lambda x: np.argsort(np.argsort(x))
    rc = lambda x, y: float(np.corrcoef(rank(x), rank(y))[0, 1])
    fwork = ((f[:, 7] >> 20) & 0x3FFFFF).astype(np.int64)   # bands evaluated by the tile's wave
    print(f"== forward: rank correlation of a wave's duration with its evaluated bands {rc(fwork, fd):.2f}, with its staged instances "
          f"{rc((f[:, 7] & 0xFFFFF).astype(np.int64), fd):.2f}; of the evaluated bands with the range length {rc(fwork, length):.2f}, with the largest n_contrib {rc(fwork, tmc):.2f}", file=out)
    print(f"== dispatch keys: rank correlation with the wave's duration: forward, range length {rc(length, fd):.2f}; "
          f"backward, staged instances {rc(staged, bd):.2f}; forward duration vs backward duration {rc(fd, bd):.2f}; "
          f"largest n_contrib vs forward duration {rc(tmc, fd):.2f}", file=out)
    if a.forward_key:
        # Would a better dispatch key shorten the forward?  Dispatch it by what the forward just counted per tile -- the bands it
        # evaluated, i.e. its real work, which no key known BEFORE the forward predicts (range length: see above) but which the
        # previous forward of the same view would -- and by its measured durations, and compare the launch spans.
        L.gsr_debug_forward_key.argtypes = [ctypes.c_void_p]
        span = lambda rec: (rec[:, 1].max() - rec[:, 0].min()) * 0.01
        print(f"== forward dispatched by keys only a previous forward of the same view can supply: span with the product key (range length) {span(f):.1f} us", file=out)
        for label, keyv in (("bands evaluated", fwork), ("largest n_contrib", tmc), ("measured duration", ((f[:, 1] - f[:, 0]) // 4))):
            key = torch.from_numpy(np.asarray(keyv).astype(np.int32)).to(dev)
            assert L.gsr_debug_forward_key(key.data_ptr()) == 0
            spans = []
            for rep in range(3):
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                fnf = L.gsr_debug_tile_clock_forward
                assert fnf(bufs["forward"].data_ptr()) == 0
                step()
                torch.cuda.synchronize()
                fnf(None)
                rec = bufs["forward"].cpu().numpy()[:ntiles]
                spans.append(span(rec))
            print(f"   by {label}: span {' / '.join(f'{x:.1f}' for x in spans)} us", file=out)
        L.gsr_debug_forward_key(None)
    if a.backward_key_length:
        # Does the backward need an order of its own?  Dispatch it by the FORWARD's key (range length, known before the forward) and
        # by the tile id, and compare with its own key (staged instances = min(range, largest n_contrib), known after the forward:
        # an order kernel of 16 us between the two blend kernels).
        L.gsr_debug_backward_key.argtypes = [ctypes.c_void_p]
        span = lambda rec: (rec[:, 1].max() - rec[:, 0].min()) * 0.01
        print(f"== backward dispatched by other keys: span with the product key (staged instances) {span(b):.1f} us", file=out)
        for label, keyv in (("range length (the forward's key)", length), ("tile id (no order)", np.arange(ntiles)[::-1].copy()), ("staged instances again", staged)):
            key = torch.from_numpy(np.asarray(keyv).astype(np.int32)).to(dev)
            assert L.gsr_debug_backward_key(key.data_ptr()) == 0
            spans = []
            for rep in range(3):
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                fnb = L.gsr_debug_tile_clock_backward
                assert fnb(bufs["backward"].data_ptr()) == 0
                step()
                torch.cuda.synchronize()
                fnb(None)
                spans.append(span(bufs["backward"].cpu().numpy()[:ntiles]))
            print(f"   by {label}: span {' / '.join(f'{x:.1f}' for x in spans)} us", file=out)
        L.gsr_debug_backward_key(None)
    if a.measured_key:
        # Is a better dispatch key to be had?  Dispatch the backward by the durations just measured (three rounds: the durations
        # change with the order) and compare the launch spans.  (Measured at C3: 543 us with the product key, 629 / 594 / 592
        # by durations -- a wave's duration says more about its neighbours than about its tile; and by the kernel's own count
        # of its issue slots per tile, which correlates 0.93 with the product key: no better either.)
        L.gsr_debug_backward_key.argtypes = [ctypes.c_void_p]
        span = lambda rec: (rec[:, 1].max() - rec[:, 0].min()) * 0.01
        print(f"== backward dispatched by its own measured durations (bound on what a better key can give): span with the product key {span(b):.1f} us", file=out)
        cur = b
        for rnd in range(3):
            key = torch.from_numpy(((cur[:, 1] - cur[:, 0]) // 4).astype(np.int32)).to(dev)
            assert L.gsr_debug_backward_key(key.data_ptr()) == 0
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            fnb = L.gsr_debug_tile_clock_backward
            assert fnb(bufs["backward"].data_ptr()) == 0
            step()
            torch.cuda.synchronize()
            fnb(None)
            rec = bufs["backward"].cpu().numpy()[:ntiles]
            lst = captured["img"].cpu().numpy()[il.tile_order:il.tile_order + 4 * ntiles].view(np.uint32).astype(np.int64) & 0x0FFFFFFF
            cur = np.zeros_like(rec)
            cur[lst] = rec   # dispatch order -> tile order
            print(f"   round {rnd + 1}, by the durations of the previous round: span {span(cur):.1f} us", file=out)
        L.gsr_debug_backward_key(None)
    if a.dump:
        np.savez_compressed(a.dump, forward=f, backward=b, ranges=ranges, tile_max_contrib=tmc)
    if a.out:
        out.close()
        print(open(a.out).read())


if __name__ == "__main__":
    main()
