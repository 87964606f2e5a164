"""GPU: properties of the C-ABI boundary itself -- caller's stream, re-entrancy from several host
threads, optional radii, the exactness of the band culling, and the N > 1 launcher of bench.py on one GPU."""
import ctypes
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

import gsr_scene
import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _direct(scene, cam, D, dpix, dev, debug=0, **kw):
    """forward + backward through the binding (no autograd): every tensor the reference's extension returns.
    debug: the C ABI's mask (include/gsr.h GSR_DEBUG_*), e.g. _C.DEBUG_NO_CULL."""
    from diff_gaussian_rasterization import _C
    st = util.hip_settings(scene, cam, D, dev)
    e = torch.empty(0, device=dev)
    t = {k: getattr(scene, k).to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(
        st.bg, t["means3D"], e, t["opacities"], t["scales"], t["rotations"], 1.0, e, st.viewmatrix, st.projmatrix, st.tanfovx,
        st.tanfovy, st.image_height, st.image_width, t["shs"], D, st.campos, False, debug)
    grads = _C.rasterize_gaussians_backward(
        st.bg, t["means3D"], radii, e, t["scales"], t["rotations"], 1.0, e, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy,
        dpix, t["shs"], D, st.campos, geom, R, binning, img, debug, **kw)
    return (color, radii) + tuple(grads)


def _same(a, b):
    for x, y in zip(a, b):
        if x is None or y is None:
            assert x is None and y is None
        else:
            assert torch.equal(x, y)


def test_non_default_stream_gives_identical_results():
    """All device work goes to the caller's stream (torch.cuda.current_stream): a side stream must give the same
    bits as the default stream, with no implicit use of stream 0."""
    _need_gpu()
    dev = torch.device("cuda:0")
    scene = gsr_scene.make_scene(9000, -3.0, sh_degree=3, seed=41)
    cam = gsr_scene.ring_camera(320, 200, 1, 8)
    dpix = torch.randn(3, 200, 320, generator=torch.Generator().manual_seed(8)).to(dev)
    ref = _direct(scene, cam, 3, dpix, dev)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        got = _direct(scene, cam, 3, dpix, dev)
        # autograd surface on the side stream too
        from diff_gaussian_rasterization import GaussianRasterizer
        p = {k: getattr(scene, k).to(dev).clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        m2 = torch.zeros_like(p["means3D"], requires_grad=True)
        color, _ = GaussianRasterizer(util.hip_settings(scene, cam, 3, dev))(means2D=m2, **p)
        color.backward(dpix)
    side.synchronize()
    _same(ref, got)
    assert torch.equal(color, ref[0]) and torch.equal(p["means3D"].grad, ref[5]) and torch.equal(p["shs"].grad, ref[7])


def test_two_host_threads_call_the_library_concurrently():
    """Re-entrancy: two Python threads, each on its own stream, run forward + backward of different scenes
    through the C ABI at the same time (ctypes releases the GIL during the calls); every iteration must reproduce the
    serial result bit for bit.  One of the threads also records per-kernel events: profiling state is per stream."""
    _need_gpu()
    from diff_gaussian_rasterization import _C
    dev = torch.device("cuda:0")
    jobs = []
    for seed, P, W, H in ((1, 12000, 400, 240), (2, 5000, 256, 144)):
        scene = gsr_scene.make_scene(P, -3.0, sh_degree=3, seed=seed)
        cam = gsr_scene.ring_camera(W, H, seed, 8)
        dpix = torch.randn(3, H, W, generator=torch.Generator().manual_seed(seed)).to(dev)
        jobs.append((scene, cam, dpix, _direct(scene, cam, 3, dpix, dev)))
    torch.cuda.synchronize()
    errors, recorded = [], {}

    def work(k):
        try:
            scene, cam, dpix, ref = jobs[k]
            s = torch.cuda.Stream(dev)
            with torch.cuda.stream(s):
                if k == 0:
                    _C.profile_begin(device=dev)
                for _ in range(6):
                    got = _direct(scene, cam, 3, dpix, dev)
                    s.synchronize()
                    _same(ref, got)
                if k == 0:
                    recorded["names"] = [n for n, _ in _C.profile_end(1024, device=dev)]
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    # thread 0's recorder saw exactly its own 6 steps (its stream), none of thread 1's
    assert recorded["names"].count("render_backward") == 6 and recorded["names"].count("preprocess") == 6


def test_radii_may_be_null_like_the_reference_default():
    """`int* radii = nullptr` (cuda_rasterizer/rasterizer.h:52,69): forward and backward accept a NULL radii."""
    _need_gpu()
    from diff_gaussian_rasterization import _C
    L = _C.lib()
    dev = torch.device("cuda:0")
    scene = gsr_scene.make_scene(4000, -3.0, sh_degree=3, seed=12)
    cam = gsr_scene.ring_camera(200, 120, 3, 8, radius=2.5)   # part of the scene is behind the camera
    st = util.hip_settings(scene, cam, 3, dev)
    dpix = torch.randn(3, 120, 200, generator=torch.Generator().manual_seed(3)).to(dev)
    ref = _direct(scene, cam, 3, dpix, dev)
    P, W, H = 4000, 200, 120
    t = {k: getattr(scene, k).to(dev).contiguous() for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    byte = dict(dtype=torch.uint8, device=dev)
    geom = torch.empty(L.gsr_geometry_bytes(P), **byte)
    img = torch.empty(L.gsr_image_bytes(W, H), **byte)
    color = torch.empty(3, H, W, device=dev)
    R = ctypes.c_int64(0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ptr = lambda x: x.data_ptr()
    _C._check(L.gsr_forward_preprocess(P, 3, 16, W, H, ptr(t["means3D"]), ptr(t["shs"]), None, ptr(t["opacities"]), ptr(t["scales"]), 1.0,
                                       ptr(t["rotations"]), None, ptr(st.viewmatrix), ptr(st.projmatrix), ptr(st.campos), st.tanfovx,
                                       st.tanfovy, 0, None, ptr(geom), ctypes.byref(R), stream, 0))
    R = int(R.value)
    binning = torch.empty(L.gsr_binning_bytes(P, R, W, H), **byte)
    _C._check(L.gsr_forward_render(P, R, W, H, ptr(st.bg), None, ptr(geom), ptr(binning), ptr(img), ptr(color), stream, 0))
    assert torch.equal(color, ref[0])
    f32 = dict(dtype=torch.float32, device=dev)
    g = dict(m2=torch.empty(P, 3, **f32), op=torch.empty(P, 1, **f32), m3=torch.empty(P, 3, **f32), sh=torch.empty(P, 16, 3, **f32),
             sc=torch.empty(P, 3, **f32), rot=torch.empty(P, 4, **f32))
    scratch = torch.empty(L.gsr_backward_scratch_bytes(P, R), **byte)
    _C._check(L.gsr_backward(P, 3, 16, R, W, H, ptr(st.bg), ptr(t["means3D"]), ptr(t["shs"]), None, ptr(t["scales"]), 1.0, ptr(t["rotations"]),
                             None, ptr(st.viewmatrix), ptr(st.projmatrix), ptr(st.campos), st.tanfovx, st.tanfovy, None, ptr(geom),
                             ptr(binning), ptr(img), ptr(scratch), ptr(dpix), ptr(g["m2"]), None, ptr(g["op"]), None, ptr(g["m3"]), None,
                             ptr(g["sh"]), ptr(g["sc"]), ptr(g["rot"]), stream, 0))
    torch.cuda.synchronize()
    # tuple of _direct: color, radii, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations
    for got, want in ((g["m2"], ref[2]), (g["op"], ref[4]), (g["m3"], ref[5]), (g["sh"], ref[7]), (g["sc"], ref[8]), (g["rot"], ref[9])):
        assert torch.equal(got, want)


@pytest.mark.parametrize("case", ["fuzz", "C2", "C3"])
def test_band_culling_changes_nothing(case):
    """The staging lanes drop (instance, 16x4-pixel band) pairs that cannot reach alpha = 1/255 (render_common.h
    gsr_tile_band_mask).  GSR_DEBUG_NO_CULL in the call's debug mask evaluates every pair like the reference: images,
    n_contrib and every gradient must be bit-identical with and without the culling -- the direct check of its safety
    margin, also at the headline workload (C3: 9.2M instances, 4.6x C2's)."""
    _need_gpu()
    from diff_gaussian_rasterization import _C
    dev = torch.device("cuda:0")
    if case in ("C2", "C3"):
        scene, cam, D = gsr_scene.make_config(case)
        runs = [(scene, cam, D)]
    else:
        runs = []
        for seed in range(6):   # the randomised configurations of test_fuzz_gpu: cameras inside the cloud, odd sizes, big splats
            rng = np.random.default_rng(100 + seed)
            P = int(rng.integers(500, 6000))
            W, H = int(rng.integers(40, 400)), int(rng.integers(30, 260))
            D = int(rng.integers(0, 4))
            mu = float(rng.uniform(-3.5, -1.0))
            scene = gsr_scene.make_scene(P, mu, sh_degree=D, seed=seed)
            cam = gsr_scene.ring_camera(W, H, int(rng.integers(0, 8)), 8, radius=float(rng.uniform(0.3, 4.5)))
            runs.append((scene, cam, D))
    for scene, cam, D in runs:
        dpix = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(1)).to(dev)
        a = _direct(scene, cam, D, dpix, dev)
        b = _direct(scene, cam, D, dpix, dev, debug=_C.DEBUG_NO_CULL)
        _same(a, b)


def _heavy_scene():
    """A dense, low-opacity blob in the middle of the view: a few tiles carry instance lists many times the mean and are
    walked thousands of instances deep."""
    P, W, H, D = 60_000, 320, 200, 2
    scene = gsr_scene.make_scene(P, -3.6, sh_degree=D, seed=9)
    g = torch.Generator().manual_seed(10)
    means = scene.means3D.clone()
    means[: P * 3 // 4] = torch.randn(P * 3 // 4, 3, generator=g) * torch.tensor([0.12, 0.08, 0.3])
    opac = scene.opacities.clone()
    opac[: P * 3 // 4] = torch.sigmoid(torch.randn(P * 3 // 4, 1, generator=g) - 3.5)
    return scene._replace(means3D=means.contiguous(), opacities=opac.contiguous()), gsr_scene.make_camera(W, H), D


def test_heavy_tiles_split_in_bands_and_in_depth():
    """Heavy tiles are split twice.  FORWARD: four entries, one wave per 16x4-pixel band (binning.hip gsr_tile_order_kernel);
    a tile's pixels are independent, so image, radii and all state must be bit-identical to the unsplit run
    (GSR_DEBUG_NO_SPLIT).  BACKWARD: one wave per depth segment of 512 list positions, each starting from the per-pixel (T, C)
    checkpoint the forward left (render_backward.hip); accum_rec then comes from a difference of forward sums instead of the
    reference's recurrence, so the gradients agree with the unsplit run to rounding (<= 2e-6 of the largest element on the blend sums), not bit
    for bit -- and both must sit inside the oracle's bars (check_grads).  Both splits must actually have happened."""
    _need_gpu()
    from diff_gaussian_rasterization import _C
    from test_parity_gpu import check_forward, check_grads
    dev = torch.device("cuda:0")
    scene, cam, D = _heavy_scene()
    W, H = cam.image_width, cam.image_height
    o = util.oracle_forward(scene, cam, D)
    dpix_cpu = util.fragile_free_dpix(o, cam)
    dpix = dpix_cpu.to(dev)
    a = _direct(scene, cam, D, dpix, dev)
    b = _direct(scene, cam, D, dpix, dev, debug=_C.DEBUG_NO_SPLIT)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])      # image, radii: bit-identical
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
    for n, x, y in zip(names, a[2:], b[2:]):
        if x is None or x.numel() == 0:
            continue
        e = float((x - y).abs().max()) / max(float(y.abs().max()), 1e-30)
        print(f"split vs unsplit {n}: {e:.2e}")
        # measured 5e-7 ... 7e-7 on the blend sums and what follows them linearly; the conic -> covariance -> scale / quaternion
        # chain amplifies last-bit differences of the sums: 1e-6 ... 4e-6 there
        assert e <= (5e-5 if n in ("dL_dcov3D", "dL_dscales", "dL_drotations") else 2e-6), (n, e)
    # the dispatch lists: the forward's (band entries) is read after a forward, the backward's (segment entries) after a backward
    st = util.hip_settings(scene, cam, D, dev)
    e = torch.empty(0, device=dev)
    t = {k: getattr(scene, k).to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(st.bg, t["means3D"], e, t["opacities"], t["scales"], t["rotations"], 1.0, e,
                                                                 st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, H, W, t["shs"], D,
                                                                 st.campos, False, False)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    il = _C.image_layout(W, H)

    def entries(count):
        v = img[il.tile_order:il.tile_order + 4 * count].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        return v[v != 0xFFFFFFFF]
    valid = entries(T + 3 * min(2048, T // 4))
    band = valid >> 28
    nsplit = int((band > 0).sum()) // 4
    rng = img[il.ranges:il.ranges + 8 * T].view(torch.int32).view(T, 2)
    lens = (rng[:, 1] - rng[:, 0]).to(torch.int64)
    assert nsplit >= 1 and int((band > 0).sum()) == 4 * nsplit, "the blob's tiles must have been split into bands"
    assert valid.numel() == T + 3 * nsplit
    tiles = valid & 0x0FFFFFFF
    # heavy: at least 640 listed instances and at least twice the mean LISTED range (binning.hip; the order kernel bins lengths by 16)
    assert int(lens[tiles[band > 0]].min()) >= max(640, 2 * (int(lens.sum()) // T)) - 16, "only heavy tiles are split"
    tmc = img[il.tile_max_contrib:il.tile_max_contrib + 4 * T].view(torch.int32).to(torch.int64)
    walked = torch.minimum(lens, tmc)
    _C.rasterize_gaussians_backward(st.bg, t["means3D"], radii, e, t["scales"], t["rotations"], 1.0, e, st.viewmatrix, st.projmatrix, st.tanfovx,
                                    st.tanfovy, dpix, t["shs"], D, st.campos, geom, R, binning, img, False)
    torch.cuda.synchronize()
    valid = entries(T + min(4096, T // 2))
    seg = valid >> 28
    tiles = valid & 0x0FFFFFFF
    cut = torch.unique(tiles[seg > 0])
    coarse = int(img[il.tile_order + 4 * (T + min(4096, T // 2)):il.tile_order + 4 * (T + min(4096, T // 2)) + 4].view(torch.int32)[0])
    assert coarse in (1, 2, 4, 8) and cut.numel() >= 1, "tiles walked deep must have been cut in depth"
    assert torch.equal(torch.sort(cut).values, torch.nonzero(walked >= (coarse + 1) * 512).flatten())   # (GSR_CKPT_STRIDE = 512 list positions)
    assert valid.numel() == T + int((seg > 0).sum()) - cut.numel()
    print(f"coarseness {coarse}; {nsplit} of {T} tiles split in bands (longest list {int(lens.max())}, mean {R // T}); {cut.numel()} cut in depth into "
          f"{int((seg > 0).sum())} segments (deepest walk {int(walked.max())})")
    # against the oracle, through the autograd surface: same bars as every other case
    h = util.hip_forward_backward(scene, cam, D, dpix_cpu)
    check_forward(h, o, cam)
    check_grads(h, o, dpix_cpu, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"], label="heavy_tiles")


def test_colour_kernel_beside_or_in_line_is_the_same():
    """The SH colour kernel of the forward runs on a helper stream beside the geometry kernel and the depth sort, forked
    from and joined into the caller's stream inside gsr_forward_preprocess (api.hip).  GSR_DEBUG_SERIAL runs it in line.
    Both must give bit-identical images, state and gradients -- also when calls follow each other without a host sync in
    between and reuse the same buffers (a missed join would show as a stale colour in the next blend)."""
    _need_gpu()
    from diff_gaussian_rasterization import _C
    dev = torch.device("cuda:0")
    scene, cam, D = gsr_scene.make_config("C2")
    dpix = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(1)).to(dev)
    ref = _direct(scene, cam, D, dpix, dev, debug=_C.DEBUG_SERIAL)
    for _ in range(5):   # back to back: the caching allocator hands the same buffers to consecutive calls
        _same(ref, _direct(scene, cam, D, dpix, dev))
    # a second scene in between changes every colour: nothing of it may survive into the next call
    scene2 = gsr_scene.make_scene(scene.means3D.shape[0], -2.2, sh_degree=D, seed=7)
    _direct(scene2, cam, D, dpix, dev)
    _same(ref, _direct(scene, cam, D, dpix, dev))
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        got = _direct(scene, cam, D, dpix, dev)
    side.synchronize()
    _same(ref, got)


def test_error_return_after_the_fork_still_joins_and_thread_release_frees():
    """gsr_forward_preprocess forks the SH colour kernel onto a helper stream; a call that fails AFTER the fork
    (prefiltered = 1 with a culled point -> GSR_ERR_PREFILTERED) must still leave the caller's stream ordered after that
    kernel, because the caller may free `geometry` right away.  Here the failed call's geometry blob is handed back to
    the allocator and immediately reused by a correct call on the same stream: its results must be the reference bits.
    gsr_thread_release() then frees the thread's helper stream / events / pinned buffer, and the next call recreates them."""
    _need_gpu()
    from diff_gaussian_rasterization import _C
    dev = torch.device("cuda:0")
    scene = gsr_scene.make_scene(60_000, -3.5, sh_degree=3, seed=5)
    cam = gsr_scene.ring_camera(320, 200, 1, 8, radius=1.0)   # camera inside the cloud: some points are culled
    dpix = torch.randn(3, 200, 320, generator=torch.Generator().manual_seed(2)).to(dev)
    ref = _direct(scene, cam, 3, dpix, dev, debug=_C.DEBUG_SERIAL)
    st = util.hip_settings(scene, cam, 3, dev)
    e = torch.empty(0, device=dev)
    t = {k: getattr(scene, k).to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    for _ in range(3):
        with pytest.raises(RuntimeError, match="filtered although prefiltered"):
            _C.rasterize_gaussians(st.bg, t["means3D"], e, t["opacities"], t["scales"], t["rotations"], 1.0, e, st.viewmatrix,
                                   st.projmatrix, st.tanfovx, st.tanfovy, st.image_height, st.image_width, t["shs"], 3, st.campos,
                                   True, False)
        _same(ref, _direct(scene, cam, 3, dpix, dev))   # reuses the blob the failed call just dropped
    torch.cuda.synchronize()
    _C.thread_release()
    _C.thread_release()   # idempotent
    _same(ref, _direct(scene, cam, 3, dpix, dev))
    torch.cuda.synchronize()
    _C.thread_release()


def test_bench_line_carries_the_contract_fields():
    """`python bench.py` prints ONE JSON line with the fields the task statement names, including `roofline` for the
    dominant kernel and `cpu_baseline` (here at C1, where the CPU legs take a second)."""
    _need_gpu()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C1", "--steps", "4", "--warmup", "1",
                        "--settle-steps", "3", "--no-train-step"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "it/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1
    assert d["settle"]["steps"] == 3 and d["step_ms"]["n"] == 4


def test_bench_gpus_2_on_one_gpu_over_gloo():
    """`python bench.py --gpus 2` with no launcher starts two ranks itself; here both share the one GPU of the box and
    exchange over gloo (GSR_BENCH_BACKEND=gloo) -- the rehearsal of the view-parallel step (rasterize_view_parallel,
    both exchange modes) that is possible without a second GPU."""
    _need_gpu()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GSR_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C1", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["views_per_step"] == 2
    assert d["alt_exchange"]["mode"] == "allreduce" and d["alt_exchange"]["value"] > 0
    assert d["data"] == "synthetic" and d["kernels"]["render_backward"]["ms"] > 0


def _vp_worker(rank, world, port, sh_mode, out_dir, backend="gloo"):
    import datetime
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        import view_parallel
        from diff_gaussian_rasterization import GaussianRasterizer
        dev = torch.device("cuda:0")
        P = 5003
        scene = gsr_scene.make_scene(P, -3.0, sh_degree=3, seed=91)
        names = ("means3D", "shs", "opacities", "scales", "rotations")

        def view(r):
            cam = gsr_scene.ring_camera(208, 120, r, 8)
            dpix = torch.randn(3, 120, 208, generator=torch.Generator().manual_seed(40 + r)).to(dev)
            return util.hip_settings(scene, cam, 3, dev), dpix

        ex = view_parallel.GradientExchange(P, 16, dev, sh_mode=sh_mode, parts=2)
        st, dpix = view(rank)
        q = {k: getattr(scene, k).to(dev).clone().requires_grad_(True) for k in names}
        m2 = torch.zeros_like(q["means3D"], requires_grad=True)
        color, radii = view_parallel.rasterize_view_parallel(q["means3D"], m2, q["shs"], q["opacities"], q["scales"], q["rotations"], st, ex)
        color.backward(dpix)
        torch.cuda.synchronize()
        # what the sum over the two views must be: both views rendered here with the plain rasterizer
        want = {k: 0 for k in names}
        for r in range(world):
            st_r, dpix_r = view(r)
            p = {k: getattr(scene, k).to(dev).clone().requires_grad_(True) for k in names}
            n2 = torch.zeros_like(p["means3D"], requires_grad=True)
            c, _ = GaussianRasterizer(st_r)(means2D=n2, **p)
            c.backward(dpix_r)
            for k in names:
                want[k] = want[k] + p[k].grad
            if r == rank:
                assert torch.equal(c, color) and torch.equal(n2.grad, m2.grad)   # this rank's own image and screen-space gradient
        for k in names:
            err = float((q[k].grad - want[k]).abs().max()) / max(float(want[k].abs().max()), 1e-30)
            assert err <= 1e-6, (k, err)   # fp32 sum of two views in either order
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sh_mode", ["compact", "allreduce"])
def test_view_parallel_two_ranks_sum_the_gradients_of_two_views(sh_mode, tmp_path):
    """Two ranks (sharing the box's one GPU, collectives over gloo) each render their own view through
    view_parallel.rasterize_view_parallel; every parameter gradient must equal the sum of the two views' gradients from
    the plain rasterizer, and the image / screen-space gradient of a rank its own view's."""
    _need_gpu()
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_vp_worker, args=(2, port, sh_mode, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


@pytest.mark.parametrize("sh_mode", ["compact", "allreduce"])
def test_view_parallel_step_over_rccl_one_rank(sh_mode, tmp_path):
    """The view-parallel step with its collectives executed by RCCL (backend "nccl"): a ONE-rank communicator, which is all
    a one-GPU box allows (RCCL refuses two ranks on one device) -- init with a finite timeout, the per-part all-gather /
    all-reduce calls of GradientExchange with exactly the tensors, dtypes and shapes of the N-rank run, and the teardown.
    With one rank the summed gradients are this view's own."""
    _need_gpu()
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_vp_worker, args=(1, port, sh_mode, str(tmp_path), "nccl"), nprocs=1, join=True)
    assert (tmp_path / "ok0").exists()


def test_bench_view_parallel_path_over_rccl_one_rank():
    """`bench.py` on its N > 1 code path -- process group over RCCL, rasterize_view_parallel with both exchange modes, the
    max-over-ranks all-reduce of the timing -- forced on with one rank (GSR_BENCH_FORCE_GROUP=1)."""
    _need_gpu()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GSR_BENCH_FORCE_GROUP"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "C1", "--steps", "3", "--warmup", "1",
                        "--settle-steps", "2", "--no-extras", "--no-cpu-baseline", "--collective-timeout-s", "120", "--deadline-s", "400"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "view-parallel" in d["config"]["parallelism"]
    assert d["alt_exchange"]["mode"] == "allreduce" and d["alt_exchange"]["value"] > 0


def test_stage2_refuses_a_binning_choice_stage1_did_not_make():
    """GSR_DEBUG_TILE_SORT selects the instance emission + tile sort in BOTH forward calls; stage 2 re-derives the choice from its own
    debug argument.  Passed to one call only, stage 2 would bin from tables stage 1 never filled: it must refuse instead
    (GSR_ERR_INVALID_ARGUMENT), both ways round, and the matching pairs must still run."""
    _need_gpu()
    from diff_gaussian_rasterization import _C
    L = _C.lib()
    dev = torch.device("cuda:0")
    scene = gsr_scene.make_scene(3000, -3.0, sh_degree=1, seed=5)
    cam = gsr_scene.make_camera(160, 96)
    st = util.hip_settings(scene, cam, 1, dev)
    t = {k: getattr(scene, k).to(dev).contiguous() for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    P, W, H = 3000, 160, 96
    stream = torch.cuda.current_stream(dev).cuda_stream
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    for d1, d2, ok in ((0, 0, True), (_C.DEBUG_TILE_SORT, _C.DEBUG_TILE_SORT, True), (_C.DEBUG_TILE_SORT, 0, False), (0, _C.DEBUG_TILE_SORT, False)):
        geom = torch.empty(L.gsr_geometry_bytes(P), dtype=torch.uint8, device=dev)
        img = torch.empty(L.gsr_image_bytes(W, H), dtype=torch.uint8, device=dev)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        out = torch.empty(3, H, W, device=dev)
        R = ctypes.c_int64(0)
        rc = L.gsr_forward_preprocess(P, 1, 4, W, H, p(t["means3D"]), p(t["shs"]), None, p(t["opacities"]), p(t["scales"]), 1.0, p(t["rotations"]),
                                      None, p(st.viewmatrix), p(st.projmatrix), p(st.campos), float(st.tanfovx), float(st.tanfovy), 0, p(radii),
                                      p(geom), ctypes.byref(R), stream, d1)
        assert rc == 0, L.gsr_last_error()
        binning = torch.empty(L.gsr_binning_bytes(P, R.value, W, H), dtype=torch.uint8, device=dev)
        rc = L.gsr_forward_render(P, R.value, W, H, p(st.bg), p(radii), p(geom), p(binning), p(img), p(out), stream, d2)
        torch.cuda.synchronize()
        if ok:
            assert rc == 0, L.gsr_last_error()
        else:
            assert rc == -1 and b"GSR_DEBUG_TILE_SORT" in L.gsr_last_error()


def test_two_views_in_flight_accumulate_the_sequential_gradients_bit_for_bit():
    """view_parallel.ViewsInFlight: forward + backward of two views at a time on two streams.  The library keeps no state between
    calls and works on the caller's stream; autograd adds the views' gradients in the order of the backward calls: images and
    accumulated gradients must be exactly those of rendering the views one after the other (three views: a full pair and a rest)."""
    _need_gpu()
    import view_parallel
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    scene = gsr_scene.make_scene(60_000, -4.0, sh_degree=3, seed=12)
    W, H = 640, 360
    cams = [gsr_scene.ring_camera(W, H, k, 8) for k in (0, 3, 5)]
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    params = {k: getattr(scene, k).to(dev).requires_grad_(True) for k in names}
    rasts = [GaussianRasterizer(util.hip_settings(scene, c, 3, dev)) for c in cams]
    dpixs = [torch.randn(3, H, W, generator=torch.Generator().manual_seed(20 + k)).to(dev) for k in range(3)]

    def fn(r):
        def f():
            m2 = torch.zeros_like(params["means3D"], requires_grad=True)
            return r(means3D=params["means3D"], means2D=m2, **{k: v for k, v in params.items() if k != "means3D"})[0]
        return f
    fns = [fn(r) for r in rasts]
    images_seq = []
    for f, dp in zip(fns, dpixs):
        img = f()
        img.backward(dp)
        images_seq.append(img.detach().clone())
    torch.cuda.synchronize()
    ref = {k: p.grad.clone() for k, p in params.items()}
    for trial in range(3):
        for p in params.values():
            p.grad = None
        images = view_parallel.ViewsInFlight(dev, 2).forward_backward(fns, dpixs)
        torch.cuda.synchronize()
        for a, b in zip(images, images_seq):
            assert torch.equal(a, b)
        for k, p in params.items():
            assert torch.equal(p.grad, ref[k]), k
