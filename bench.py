"""Benchmark of the rasterizer hot path: train iters/sec (fwd+bwd rasterize) @1980x1080, 1M Gaussians.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3]      (N > 1: starts N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one forward + one backward of the drop-in GaussianRasterizer over one synthetic view
(inputs resident in HBM).  With N > 1 every rank renders its own view of the replicated scene and
the per-step parameter gradients (59 floats per Gaussian) are summed over the ranks inside the backward
(view_parallel.rasterize_view_parallel: RCCL collectives per part of the per-Gaussian backward, overlapped
with the next part); value = views per second over all ranks (weak scaling).  Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "gaussian-splatting_cc-comments_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import gsr_scene  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, corrected as
# MI355X_MICROARCH.md "HBM" prescribes) of this same command, summarised by tools/pmc_summary.py
PMC_SUMMARY = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r4_pmc_summary.json", "r3b_pmc_summary.json", "r3_pmc_summary.json", "r2_pmc_summary.json", "r1_pmc_summary.json"))
                    if os.path.exists(f)), os.path.join(ROOT, "profiles", "r3_pmc_summary.json"))
KERNEL_SYMBOL = {"render_backward": "gsr_render_backward_wave_kernel", "render_forward": "gsr_render_forward_wave_kernel",
                 "gaussian_backward": "gsr_gaussian_backward_kernel", "preprocess": "gsr_preprocess_kernel",
                 "preprocess_color": "gsr_preprocess_color_kernel",
                 "duplicate_keys": "gsr_duplicate_keys_kernel", "tile_ranges": "gsr_tile_ranges_kernel",
                 "col_scatter": "gsr_tb_col_scatter_kernel", "row_hist": "gsr_tb_row_hist_kernel", "row_scatter": "gsr_tb_row_scatter_kernel"}


DOMINANT_STAGE = "render_backward"   # the kernel with the largest launch time in every configuration measured


def _pmc_entry(stage, workload):
    d = json.load(open(PMC_SUMMARY))
    if d.get("_workload") != workload:
        raise KeyError(workload)
    sym = KERNEL_SYMBOL[stage]
    for k, v in d.items():   # templated kernels appear as name<args>
        if isinstance(v, dict) and (k == sym or k.startswith(sym + "<")):
            return v
    raise KeyError(sym)


def pmc_traffic(stage, workload):
    try:
        return _pmc_entry(stage, workload).get("hbm_traffic_bytes")
    except (OSError, KeyError, ValueError):
        return None


def algorithmic_bytes(P, V, R, Rp, N, T, M):
    """SURVEY.md section 8(d) per-stage algorithmic bytes (deg-3 SH: 4*3*M = 192 B)."""
    sh = 12 * M
    fwd = {
        "preprocess": 44 * P + (8 * P + 54 * V),       # geometry kernel; with the colour kernel: SURVEY's (44 P + sh V) + (8 P + 67 V)
        "preprocess_color": sh * V + 13 * V,            # SH rows in, colour + clamp flags out
        "scan": 8 * P,
        "duplicate_keys": 20 * P + 12 * R,
        "sort": 24 * R,
        "tile_ranges": 8 * R + 16 * T,
        # What the binning chain of this library REPLACES, priced at the SURVEY's bytes for the reference's stages: InclusiveSum +
        # duplicateWithKeys + SortPairs (all 64 key bits: the depth sort does the depth half) + identifyTileRanges
        # (rasterizer_impl.cu:323, 78-159, 357-385) = 28 P + 44 R + 16 T.  A comparison figure ("vs replaced stages"), not the
        # bytes these kernels move: see binning_moved_bytes()
        "binning_chain": 8 * P + (20 * P + 12 * R) + 24 * R + (8 * R + 16 * T),
        "render_forward": 8 * T + 40 * Rp + 20 * N,
    }
    bwd = {
        "render_backward": 8 * T + 40 * R + 20 * N + 44 * V,
        "gaussian_backward": 92 * V + (sh + 339) * V + 300 * P,   # includes re-reading the SH row, which this kernel does not do
    }
    return fwd, bwd


def binning_moved_bytes(P, V, R, L, T, pairs):
    """Bytes the column-pair binning's three stage-2 kernels have to move by their own design (csrc/tilebin.hip): pass 1's scatter
    reads the 16-byte depth-ordered records of all P Gaussians and writes one 8-byte column pair per (Gaussian, tile column that keeps
    a row) and the visible Gaussians' slot bases; pass 2 reads the pairs twice (histogram, scatter) and writes 4 bytes per LISTED
    instance (point_list: L of the R instances -- tiles a splat provably misses are left out, csrc/gsr_rect_trim.h), one validity
    byte per gradient slot (R) and 8 bytes per tile (ranges).  No per-instance key exists."""
    return dict(col_scatter=16 * P + 8 * pairs + 4 * V, row_hist=8 * pairs, row_scatter=8 * pairs + 4 * L + R + 8 * T)


def listed_counts(geom, img, P, W, H):
    """(column pairs, instances) the column-pair binning actually lists for this view, from the geometry buffer's trim words and the
    image buffer's tile ranges (torch restatement of gsr_trim_columns, csrc/gsr_rect_trim.h)."""
    from diff_gaussian_rasterization import _C
    gl, il = _C.geometry_layout(P), _C.image_layout(W, H)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    rs = geom[gl.rshape:gl.rshape + 8 * P].view(torch.int32).view(P, 2).to(torch.int64) & 0xFFFFFFFF
    packed, trim = rs[:, 0], rs[:, 1]
    none = packed == 0xFFFFFFFF
    w, h = ((packed >> 16) & 255) + 1, (packed >> 24) + 1
    cs, rsh = torch.zeros_like(w), torch.zeros_like(w)   # log2 of the columns per nibble / rows per unit
    for _ in range(6):
        cs = torch.where(((w + (1 << cs) - 1) >> cs) > 8, cs + 1, cs)
        rsh = torch.where(((h + (1 << rsh) - 1) >> rsh) > 16, rsh + 1, rsh)
    groups = (w + (1 << cs) - 1) >> cs
    first, last = torch.full_like(w, 99), torch.full_like(w, -1)
    for c in range(8):
        nib = (trim >> (4 * c)) & 15
        keeps = (c < groups) & ((((nib & 3) + (nib >> 2)) << rsh) < h)
        first = torch.where(keeps & (first == 99), torch.full_like(w, c), first)
        last = torch.where(keeps, torch.full_like(w, c), last)
    active = trim != 0
    kept_cols = torch.clamp(torch.minimum((last + 1) << cs, w) - (first << cs), min=0)
    wt = torch.where(active, torch.where(last >= 0, kept_cols, torch.zeros_like(w)), w)
    pairs = int(wt[~none].sum().item())
    rng = img[il.ranges:il.ranges + 8 * T].view(torch.int32).view(T, 2).to(torch.int64)
    return pairs, int((rng[:, 1] - rng[:, 0]).sum().item())


def host_cores():
    """CPUs this process may actually use: cgroup quota if any, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(scene, cam, D):
    """The CPU oracle (a port: the reference has no CPU path and its CUDA sources cannot be built
    here) timed on one full step of the same workload, all host cores via OpenMP."""
    from oracle import oracle
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # read by libgomp when the oracle library is first loaded
    t0 = time.time()
    o = oracle.forward(scene.means3D.numpy(), scene.opacities.numpy(), cam.world_view_transform.numpy(),
                       cam.full_proj_transform.numpy(), cam.camera_center.numpy(), scene.bg.numpy(), cam.image_width,
                       cam.image_height, cam.tanfovx, cam.tanfovy, D, shs=scene.shs.numpy(), scales=scene.scales.numpy(),
                       rotations=scene.rotations.numpy())
    g = torch.Generator().manual_seed(1)
    dpix = torch.randn(3, cam.image_height, cam.image_width, generator=g).numpy()
    oracle.backward(o, dpix)
    dt = time.time() - t0
    return dict(value=1.0 / dt, unit="it/s", cores=cores, kind="port",
                sample="1 full fwd+bwd step of the same workload (C oracle, OpenMP, all host cores)"), o


def cpu_baseline_torch():
    """The CPU baseline north_star names: a PyTorch autograd splat on the box's host cores, at BASELINE.json
    configs[0] (C1: 10k Gaussians, SH degree 0, 256x256, one view) -- oracle/torch_tile_splat.py, fp32, all cores."""
    from oracle import torch_tile_splat
    cores = host_cores()
    scene, cam, D = gsr_scene.make_config("C1")
    t, R, _ = torch_tile_splat.time_forward_backward(scene, cam, D, repeats=5, threads=cores)
    return dict(value=round(1.0 / t, 4), unit="it/s", cores=cores, kind="port",
                sample=f"C1 (10000 Gaussians, SH deg 0, 256x256, R={R}): fwd+bwd through torch.autograd, median of 5, "
                       f"torch.set_num_threads({cores}); NOT the headline workload (C3 needs ~160x the pair evaluations)")


def parity_figures(o, color, params, dpix, rasterizer, H, W):
    """What the parity bars leave out, made visible: the oracle flags pixels whose accept/reject decisions sit within
    2e-5 of a threshold ("fragile": a 1-ulp exp difference may flip alpha < 1/255 or T(1-alpha) < 1e-4 there); the
    tests compare images on the others and zero the upstream gradient on the fragile ones."""
    import numpy as np
    from oracle import oracle
    ok = (o["fragile"] == 0).reshape(H, W)
    diff = np.abs(color.detach().cpu().numpy() - o["color"])
    fig = dict(image_L1=float(diff.mean()), image_maxabs_nonfragile=float(diff[:, ok].max()),
               image_maxabs_all_pixels=float(diff.max()), fragile_pixel_fraction=float(1.0 - ok.mean()))
    og = oracle.backward(o, dpix.cpu().numpy())   # UNMASKED upstream gradient: fragile pixels included
    names = dict(means3D="dL_dmeans3D", shs="dL_dsh", opacities="dL_dopacity", scales="dL_dscales", rotations="dL_drotations")
    for p in params.values():
        p.grad = None
    means2D = torch.zeros_like(params["means3D"], requires_grad=True)
    c, _ = rasterizer(means3D=params["means3D"], means2D=means2D, **{k: v for k, v in params.items() if k != "means3D"})
    c.backward(dpix)
    torch.cuda.synchronize()
    fig["grad_relerr_unmasked_dpix"] = {k: float(np.abs(params[k].grad.cpu().numpy().reshape(-1) - og[n].reshape(-1)).max() /
                                                 max(np.abs(og[n]).max(), 1e-30)) for k, n in names.items()}
    return fig


def bench_loss(image, dev, iters=20):
    """Not part of `value`: the fused L1+SSIM loss fwd+bwd (include/gsr.h gsr_l1_ssim_loss)
    next to the stock-PyTorch sequence of train.py:126-128 on the same (3,H,W) image."""
    import fused_loss
    from diff_gaussian_rasterization import _C
    g = torch.Generator().manual_seed(5)
    gt = (image + 0.05 * torch.randn(image.shape, generator=g).to(dev)).clamp(0, 1)
    img = image.clone().requires_grad_(True)

    def fused():
        img.grad = None
        fused_loss.l1_ssim_loss(img, gt, 0.2).backward()

    taps = torch.tensor([math.exp(-(k - 5) ** 2 / (2 * 1.5 ** 2)) for k in range(11)])
    taps = taps / taps.sum()
    w = (taps[:, None] @ taps[None, :]).to(dev).expand(3, 1, 11, 11).contiguous()
    F = torch.nn.functional

    def stock():  # utils/loss_utils.py:16-63 + train.py:126-128
        img.grad = None
        x = img[None]
        y = gt[None]
        mu1, mu2 = F.conv2d(x, w, padding=5, groups=3), F.conv2d(y, w, padding=5, groups=3)
        s11 = F.conv2d(x * x, w, padding=5, groups=3) - mu1.pow(2)
        s22 = F.conv2d(y * y, w, padding=5, groups=3) - mu2.pow(2)
        s12 = F.conv2d(x * y, w, padding=5, groups=3) - mu1 * mu2
        ssim = (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1.pow(2) + mu2.pow(2) + 1e-4) * (s11 + s22 + 9e-4))).mean()
        (0.8 * torch.abs(img - gt).mean() + 0.2 * (1.0 - ssim)).backward()

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    t_fused, t_stock = timeit(fused), timeit(stock)
    _C.profile_begin()
    fused()
    kt = dict(_C.profile_end())
    n = image.numel()
    # bytes per kernel: forward reads x, y and writes the three maps the backward needs; backward reads the maps, x, y and writes the gradient
    kb = dict(ssim_forward=n * 4 * (2 + 3), ssim_backward=n * 4 * (3 + 2 + 1))
    kern = {k: dict(ms=round(v, 4), algorithmic_bytes=int(kb[k]), GBps=round(kb[k] / (v * 1e-3) / 1e9, 1), hbm_frac=round(kb[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 3))
            for k, v in kt.items() if k in kb and v > 0}
    return dict(fused_ms=round(t_fused, 4), stock_pytorch_ms=round(t_stock, 4),
                kernels_ms={k: round(v, 4) for k, v in kt.items()}, kernels=kern,
                algorithmic_bytes=int(n * 4 * (2 + 3 + 3 + 2 + 1)),  # A: read x,y write 3 maps; B: read 3 maps, x, y, write grad
                bound="LDS / vector ALU, not HBM: per output pixel and map the separable 11-tap window costs 22 multiply-adds out of LDS-staged "
                      "tiles with an 11-pixel halo (csrc/loss.hip); profiles/r4_pmc_summary_extras.json holds the kernels' counters",
                note="fwd+bwd of loss = 0.8*L1 + 0.2*(1-SSIM) on the rendered (3,H,W) image; not included in `value`")


def bench_train_step(scene, settings, D, dev, iters=10, modes=("stock_around", "fused_loss", "all_fused")):
    """Whole training iteration (render -> L1+SSIM loss -> backward -> Adam step, train.py:93-131,179-181) three
    ways on the benchmark scene; an extra next to the headline metric, not part of `value`:
      stock_around : PyTorch activations, stock-PyTorch loss and torch.optim.Adam around the HIP rasterizer
                     (= the reference's loop with only diff_gaussian_rasterization swapped)
      fused_loss   : as above with the fused L1+SSIM loss (8f-2)
      all_fused    : leaf-parameter rasterizer + fused loss + one-launch Adam (8f-3)"""
    import torch.nn.functional as F
    import fused_loss
    import gsr_model
    from diff_gaussian_rasterization import GaussianRasterizer
    from fused_params import FusedAdam, rasterize_leaf_gaussians
    lrs = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=2.5e-3 / 20, opacity=0.05, scaling=5e-3, rotation=1e-3)
    gt = torch.rand(3, settings.image_height, settings.image_width, device=dev)
    rast = GaussianRasterizer(settings)
    g1 = torch.exp(-(torch.arange(11, dtype=torch.float32) - 5) ** 2 / (2 * 1.5 ** 2))
    g1 = g1 / g1.sum()
    window = (g1[:, None] @ g1[None, :]).expand(3, 1, 11, 11).contiguous().to(dev)

    def stock_loss(image):  # utils/loss_utils.py:16-63 + train.py:126-127
        x, y = image[None], gt[None]
        mu1, mu2 = F.conv2d(x, window, padding=5, groups=3), F.conv2d(y, window, padding=5, groups=3)
        mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
        s1 = F.conv2d(x * x, window, padding=5, groups=3) - mu1_sq
        s2 = F.conv2d(y * y, window, padding=5, groups=3) - mu2_sq
        s12 = F.conv2d(x * y, window, padding=5, groups=3) - mu1_mu2
        ssim = (((2 * mu1_mu2 + 0.01 ** 2) * (2 * s12 + 0.03 ** 2)) / ((mu1_sq + mu2_sq + 0.01 ** 2) * (s1 + s2 + 0.03 ** 2))).mean()
        return 0.8 * torch.abs(image - gt).mean() + 0.2 * (1.0 - ssim)

    def make(opt_cls):
        pc = gsr_model.GaussianParams.from_activated(scene.means3D, scene.shs, scene.scales, scene.rotations, scene.opacities,
                                                     device=dev, active_sh_degree=D)
        names = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity")
        groups = [{"params": [torch.nn.Parameter(p.detach())], "lr": lrs[n], "name": n} for n, p in zip(names, pc.parameters())]
        pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity = (g["params"][0] for g in groups)
        return pc, opt_cls(groups, lr=0.0, eps=1e-15)

    def run(mode):
        pc, opt = make(FusedAdam if mode == "all_fused" else torch.optim.Adam)

        def it():
            m2 = torch.zeros_like(pc._xyz, requires_grad=True)
            if mode == "all_fused":
                image, radii = rasterize_leaf_gaussians(pc._xyz, m2, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling,
                                                        pc._rotation, settings)
            else:
                image, radii = rast(means3D=pc.get_xyz, means2D=m2, shs=pc.get_features, opacities=pc.get_opacity,
                                    scales=pc.get_scaling, rotations=pc.get_rotation)
            loss = stock_loss(image) if mode == "stock_around" else fused_loss.l1_ssim_loss(image, gt, 0.2)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
        for _ in range(3):
            it()
        # three blocks of `iters` iterations, the fastest counts; all three are reported, with the hipMallocs PyTorch's caching allocator
        # made during each (none in any run looked at).  Round 4 saw this extra at 2.5-2.9 ms instead of 1.5-1.6 in three single-block
        # runs on two boxes whose other timings stalled as well (C2's timed region 0.84 ms around per-step times of 0.53): a host that
        # stalls for 10 ms doubles a 16-ms block, so one block is not a measurement
        blocks, allocs = [], []
        for _ in range(3):
            torch.cuda.synchronize()
            a0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) if dev.type == "cuda" else 0
            t0 = time.perf_counter()
            for _ in range(iters):
                it()
            torch.cuda.synchronize()
            blocks.append(round((time.perf_counter() - t0) / iters * 1e3, 4))
            allocs.append((torch.cuda.memory_stats(dev).get("num_device_alloc", 0) if dev.type == "cuda" else 0) - a0)
        ms = min(blocks)
        block_ms[mode] = dict(ms=blocks, device_allocations=allocs)
        if mode == "all_fused":
            # what the iteration is made of: every library stage of one more iteration, timed in line (HIP events around each stage --
            # the colour kernel then runs in line too, so the stages add up to MORE than the iteration, in which it overlaps)
            from diff_gaussian_rasterization import _C
            _C.profile_begin()
            it()
            torch.cuda.synchronize()
            kt = {}
            for name, t in _C.profile_end():
                kt[name] = kt.get(name, 0.0) + t
            nfloat = sum(p.numel() for g in opt.param_groups for p in g["params"])
            adam_bytes = 28 * nfloat   # per float: parameter, gradient, two moments read (16 B), parameter and moments written (12 B)
            extra["all_fused_stages_ms"] = {k: round(v, 4) for k, v in kt.items()}
            extra["all_fused_stages_sum_ms"] = round(sum(kt.values()), 4)
            if kt.get("adam", 0) > 0:
                extra["adam"] = dict(ms=round(kt["adam"], 4), floats=nfloat, algorithmic_bytes=adam_bytes, GBps=round(adam_bytes / (kt["adam"] * 1e-3) / 1e9, 1),
                                     hbm_frac=round(adam_bytes / (kt["adam"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3), bound="hbm",
                                     note="one launch over the six parameter groups (59 floats per Gaussian), 28 B per float: the streaming minimum")
        return ms

    extra = {}
    block_ms = {}
    res = {m: run(m) for m in modes}
    return dict(ms_per_iteration=res, blocks_ms=block_ms, **extra,
                note="render + 0.8*L1+0.2*(1-SSIM) + backward + Adam(6 groups, eps 1e-15) + zero_grad; parameters move, so V and R drift "
                     "slightly from the headline workload.  ms_per_iteration = the fastest of three blocks of 10 iterations (blocks_ms: all three, with the device "
                     "allocations PyTorch's caching allocator made during each).  all_fused_stages_ms: the library's stages of one all_fused iteration timed in line "
                     "(their sum exceeds the iteration by what overlaps in it: the colour kernel); what the iteration holds beyond them is the "
                     "caller's PyTorch work (zeros_like, zero_grad, the allocator) and launch gaps -- profiles/r4_train_iteration_timeline.txt")


def bench_views_in_flight(scene, D, W, H, dev, iters=20, nviews=2):
    """Not part of `value`: two views of the ring (BASELINE.json configs[3]'s cameras) per step on ONE GPU, rendered one after the
    other and two in flight on two streams (view_parallel.ViewsInFlight) -- the per-rank mode of view-parallel training with
    gradient accumulation.  Same work, same gradients (checked here bit for bit); what changes is what the chip does during a
    view's latency-bound binning chain and the blend kernels' tails."""
    import view_parallel
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    to = lambda t: t.to(dev)
    params = dict(means3D=to(scene.means3D).requires_grad_(True), shs=to(scene.shs).requires_grad_(True),
                  opacities=to(scene.opacities).requires_grad_(True), scales=to(scene.scales).requires_grad_(True),
                  rotations=to(scene.rotations).requires_grad_(True))
    cams = [gsr_scene.ring_camera(W, H, k=v, n=8) for v in range(nviews)]
    rasts = [GaussianRasterizer(GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=to(scene.bg), scale_modifier=1.0, viewmatrix=to(c.world_view_transform),
        projmatrix=to(c.full_proj_transform), sh_degree=D, campos=to(c.camera_center), prefiltered=False, debug=False)) for c in cams]
    dpixs = [to(torch.randn(3, H, W, generator=torch.Generator().manual_seed(11 + v))) for v in range(nviews)]

    def fn(r):
        def f():
            m2 = torch.zeros_like(params["means3D"], requires_grad=True)
            return r(means3D=params["means3D"], means2D=m2, **{k: v for k, v in params.items() if k != "means3D"})[0]
        return f
    fns = [fn(r) for r in rasts]
    vif = view_parallel.ViewsInFlight(dev, 2)
    vif_st = view_parallel.ViewsInFlight(dev, 2, staggered=True)

    def staggered():
        for p in params.values():
            p.grad = None
        vif_st.forward_backward(fns, dpixs)

    def sequential():
        for p in params.values():
            p.grad = None
        for f, dp in zip(fns, dpixs):
            f().backward(dp)

    def interleaved():
        for p in params.values():
            p.grad = None
        vif.forward_backward(fns, dpixs)

    def timeit(step):
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3
    sequential()
    torch.cuda.synchronize()
    ref = {k: p.grad.clone() for k, p in params.items()}
    interleaved()
    torch.cuda.synchronize()
    same = all(torch.equal(ref[k], p.grad) for k, p in params.items())
    t_seq, t_int = timeit(sequential), timeit(interleaved)
    t_seq2, t_int2 = timeit(sequential), timeit(interleaved)
    t_seq, t_int = min(t_seq, t_seq2), min(t_int, t_int2)
    staggered()
    torch.cuda.synchronize()
    same = same and all(torch.equal(ref[k], p.grad) for k, p in params.items())
    t_stag = min(timeit(staggered), timeit(staggered))
    return dict(views_per_step=nviews, sequential_ms_per_step=round(t_seq, 4), in_flight_ms_per_step=round(t_int, 4), staggered_ms_per_step=round(t_stag, 4),
                sequential_views_per_s=round(nviews / t_seq * 1e3, 1), in_flight_views_per_s=round(nviews / t_int * 1e3, 1),
                speedup=round(t_seq / t_int, 4), gradients_bit_identical=bool(same),
                note="two ring views per step on one GPU, gradients accumulated: one view after the other vs two in flight on two streams "
                     "(view_parallel.ViewsInFlight; bench.py --views-per-rank 2 --views-in-flight 2 is the same as a timed region); an extra, "
                     "not the headline")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=list(gsr_scene.CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the loss and training-iteration extras (profiling runs)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the whole-training-iteration extra (8f rows)")
    ap.add_argument("--sh-exchange", default="compact", choices=["compact", "allreduce"],
                    help="N > 1: 'compact' all-gathers 3 floats/Gaussian/view and rebuilds the summed SH gradient "
                         "locally (view_parallel.GradientExchange); 'allreduce' sums all 59 floats/Gaussian.  The timed "
                         "region uses this mode; the other one is timed afterwards and reported under 'alt_exchange'")
    ap.add_argument("--parts", type=int, default=0, help="N > 1: parts of the per-Gaussian backward whose exchange is "
                                                         "started while the next part computes; 0 (default) = chosen from the "
                                                         "measured cost of a collective in this process group: 2 when a "
                                                         "small all-reduce costs the stream less than 35 us (the second part's "
                                                         "~70 us of kernel then hides more than its two extra collectives cost), "
                                                         "else 1")
    ap.add_argument("--views-per-rank", type=int, default=1,
                    help="views every rank renders per step (gradient accumulation over k local views, then ONE exchange of the "
                         "59 floats/Gaussian through view_parallel.GradientBucket).  Default 1 = BASELINE.json configs[3] (one view "
                         "per rank and step, exchange pipelined with the per-Gaussian backward); k > 1 shows the amortised case: "
                         "value counts views, so it stays comparable")
    ap.add_argument("--views-in-flight", type=int, default=1,
                    help="with --views-per-rank k >= 2: forward + backward of this many consecutive views at a time on as many streams "
                         "(view_parallel.ViewsInFlight): a view's depth sort / binning runs under the other view's blend kernels.  "
                         "Gradients are bit for bit those of the sequential loop.  Default 1 = one view after the other")
    ap.add_argument("--staggered", action="store_true",
                    help="with --views-in-flight: every view's forward and backward are issued before the next view's forward, views "
                         "alternating over the streams (view_parallel.ViewsInFlight(staggered=True))")
    ap.add_argument("--collective-timeout-s", type=float, default=180.0,
                    help="N > 1: timeout of the process group (init and every collective); a rank that waits longer exits non-zero "
                         "with a message and the launcher stops its siblings")
    ap.add_argument("--deadline-s", type=float, default=1500.0,
                    help="N > 1: hard wall-clock limit of a rank; past it the rank prints what it was doing and exits with code 4 "
                         "(a wrong collective must fail, not hang the node)")
    ap.add_argument("--settle-steps", type=int, default=60,
                    help="steps run before the W warm-up steps (untimed; 0 = off): lets the device reach its steady state for "
                         "this workload, reported as `settle` in the JSON line")
    ap.add_argument("--debug-mask", type=int, default=0,
                    help="diagnostics only: bits of the C ABI's `debug` mask (include/gsr.h GSR_DEBUG_*) passed with every call, e.g. 32 = "
                         "global radix passes for the depth order, 16 = instance emission + tile sort: A/B of the alternate paths on "
                         "one box.  Reported in config; the headline line is the run with 0")
    ap.add_argument("--library", default=None,
                    help="diagnostics only: path of another build of the C ABI (csrc/Makefile `variant`) to bind instead of the product "
                         "library, for A/B runs on one box.  Reported in config; the headline line is the run without it")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearsal of the N > 1 plumbing WITHOUT the rasterizer (no GPU needed): ranks are spawned, the "
                         "process group is formed and every step runs only the gradient exchange on synthetic buffers")
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N child ranks of this same script (one per GPU) and
    relay rank 0's JSON line.  This parent never makes a HIP call and never re-execs: the children are fresh
    processes that initialise the GPU themselves."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        for p in procs:
            code = p.wait()
            if code != 0 and rc == 0:
                rc = code
                for q in procs:   # one rank failed: the others would wait in a collective forever
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


def step_stats(ms):
    ms = sorted(ms)
    n = len(ms)
    if n == 0:
        return None
    med = ms[n // 2] if n % 2 else 0.5 * (ms[n // 2 - 1] + ms[n // 2])
    return dict(min=round(ms[0], 4), median=round(med, 4), max=round(ms[-1], 4), n=n)


def valu_issue_fraction(stage, workload, kernel_ms=None):
    """Fraction of the chip's vector-ALU issue roof the kernel reaches, from the committed PMC summary (tools/profile_run.sh ->
    tools/pmc_summary.py): roof time = the kernel's wave64 VALU instructions (SQ_INSTS_VALU, exact: checked against kernels of
    known instruction counts, profiles/r3_pmc_calibration.txt) / the rate the chip SUSTAINS on a synthetic, dependency-free
    stream with the same instruction-class mix at the same occupancy (tools/valu_probe.hip "backward blend class mix": 588 G
    instr/s at 4 waves per SIMD; the class mix from SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32; profiles/r3_valu_probe.txt), divided by
    the kernel's duration.  Time-domain on both sides, so no clock estimate enters.  With kernel_ms given (this run's own duration
    of the kernel) the fraction is taken against it, else against the duration recorded with the counters."""
    try:
        e = _pmc_entry(stage, workload)
        ms = kernel_ms if kernel_ms else e["kernel_ms"]
        return dict(frac=round(e["valu_roof_ms"] / ms, 3), roof_ms=round(e["valu_roof_ms"], 4), kernel_ms=round(ms, 4),
                    valu_insts=int(e["SQ_INSTS_VALU"]), class_counts={k: int(v) for k, v in e.get("valu_class_counts", {}).items()},
                    avg_waves_per_simd=round(e.get("avg_waves_per_simd", 0.0), 2), model=e.get("valu_roof_model", "additive per-class costs"),
                    source=os.path.basename(PMC_SUMMARY))
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; pass --gpus {world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GSR_BENCH_BACKEND=gloo (debug only): lets several ranks share ONE GPU to rehearse the N > 1 code path
    backend = os.environ.get("GSR_BENCH_BACKEND", "gloo" if args.dry_run else "nccl")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL; must be set before the first HIP call
    if args.dry_run:
        dev = torch.device("cpu")
    else:
        if backend != "nccl":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)  # before any collective: RCCL binds the communicator to the current device
    phase = {"now": "start"}
    # GSR_BENCH_FORCE_GROUP=1 (tests): form the process group and take the view-parallel step even with ONE rank -- on a
    # one-GPU box that is the only way to run this code over RCCL (two ranks cannot share a device under RCCL)
    grouped = world > 1 or os.environ.get("GSR_BENCH_FORCE_GROUP") == "1"
    if grouped:
        import datetime
        import threading
        if env_world is None:   # forced group without a launcher: the rendezvous of a one-rank group
            os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))

        def _deadline():
            print(f"bench.py rank {rank}: --deadline-s {args.deadline_s:.0f} exceeded during '{phase['now']}'; exiting", file=sys.stderr, flush=True)
            os._exit(4)
        watchdog = threading.Timer(args.deadline_s, _deadline)
        watchdog.daemon = True
        watchdog.start()
        # a finite timeout for init and every collective: RCCL's watchdog (TORCH_NCCL_ASYNC_ERROR_HANDLING, on by default)
        # aborts the process when a collective exceeds it, gloo raises -- either way a non-zero exit, never a hang
        timeout = datetime.timedelta(seconds=args.collective_timeout_s)
        phase["now"] = "init_process_group"
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=timeout)
            else:
                dist.init_process_group(backend, timeout=timeout)
        except Exception as ex:  # noqa: BLE001
            print(f"bench.py rank {rank}: process group ({backend}, world {world}) could not be formed within "
                  f"{args.collective_timeout_s:.0f} s: {ex!r}", file=sys.stderr, flush=True)
            sys.exit(3)
    try:
        run_rank(args, rank, world, dev, phase, grouped)
    except Exception as ex:  # noqa: BLE001
        if grouped:   # a failed or timed-out collective: say so and leave with a code; the launcher stops the other ranks
            import traceback
            traceback.print_exc()
            print(f"bench.py rank {rank}: failed during '{phase['now']}': {ex!r}", file=sys.stderr, flush=True)
            os._exit(3)   # not sys.exit: destroy_process_group() on a broken communicator can itself block
        raise
    if grouped:
        phase["now"] = "destroy_process_group"
        dist.destroy_process_group()


def run_rank(args, rank, world, dev, phase=None, grouped=None):
    import view_parallel
    phase = phase if phase is not None else {}
    grouped = (world > 1) if grouped is None else grouped
    kviews = max(1, int(args.views_per_rank))
    P, W, H, D, mu = gsr_scene.CONFIGS[args.config]
    M = (D + 1) ** 2
    sync = (lambda: None) if args.dry_run else torch.cuda.synchronize

    def barrier():
        sync()
        if grouped:
            dist.barrier()

    # --parts 0: from what a collective costs THIS process group (every rank measures, the maximum over the ranks decides, so
    # all ranks cut their Gaussians alike)
    parts_note = None
    if args.parts <= 0:
        args.parts = 2
        if grouped and kviews == 1:
            phase["now"] = "measuring the cost of a small all-reduce (choice of --parts)"
            t = torch.zeros(256, device=dev if not args.dry_run else "cpu")
            for _ in range(5):
                dist.all_reduce(t)
            sync()
            t0 = time.perf_counter()
            for _ in range(20):
                dist.all_reduce(t)
            sync()
            cost = torch.tensor([(time.perf_counter() - t0) / 20 * 1e6], dtype=torch.float64, device=t.device)
            dist.all_reduce(cost, op=dist.ReduceOp.MAX)
            us = float(cost.item())
            args.parts = 2 if us < 35.0 else 1
            parts_note = f"--parts chosen = {args.parts}: a 1-KB all-reduce costs {us:.1f} us in this process group (threshold 35 us)"

    if args.dry_run:
        # plumbing rehearsal: the exchange of one step on synthetic per-part buffers, nothing else
        ex = {m: view_parallel.GradientExchange(P, M, dev, sh_mode=m, parts=args.parts) for m in ("compact", "allreduce")}
        campos = torch.zeros(3)

        def make_step(mode):
            e = ex[mode]

            def step():
                e.begin_step()
                for k in range(len(e.ranges)):
                    e.bucket[k].fill_(float(rank + 1))
                    e.submit(k, campos)
                e.finish(None, D, rebuild_sh=False)
            return step
        steps = {m: make_step(m) for m in ex}
        settings = None
    else:
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
        if args.library:
            _C.use_library(args.library)
        _C.lib()  # fail loudly if the HIP library is missing
        scene = gsr_scene.make_scene(P, mu, D, seed=0)          # replicated parameters
        # views of this rank: BASELINE.json configs[3] puts 8 cameras on a ring; rank r renders views r, r + world, ...
        nviews = world * kviews
        my_views = view_parallel.shard_views(nviews, rank, world)
        cams = [gsr_scene.ring_camera(W, H, k=v, n=max(nviews, 8)) for v in my_views] if nviews > 1 else [gsr_scene.make_camera(W, H)]
        cam = cams[0]
        M = scene.shs.shape[1]
        to = lambda t: t.to(dev)
        params = dict(means3D=to(scene.means3D).requires_grad_(True), shs=to(scene.shs).requires_grad_(True),
                      opacities=to(scene.opacities).requires_grad_(True), scales=to(scene.scales).requires_grad_(True),
                      rotations=to(scene.rotations).requires_grad_(True))

        def settings_of(c):
            return GaussianRasterizationSettings(
                image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=to(scene.bg), scale_modifier=1.0,
                viewmatrix=to(c.world_view_transform), projmatrix=to(c.full_proj_transform), sh_degree=D,
                campos=to(c.camera_center), prefiltered=False, debug=(args.debug_mask or False))
        all_settings = [settings_of(c) for c in cams]
        settings = all_settings[0]
        rasterizers = [GaussianRasterizer(st) for st in all_settings]
        rasterizer = rasterizers[0]
        dpixs = [to(torch.randn(3, H, W, generator=torch.Generator().manual_seed(1 + v))) for v in my_views]
        dpix = dpixs[0]
        state = {}
        ex = {m: view_parallel.GradientExchange(P, M, dev, sh_mode=m, parts=args.parts) for m in ("compact", "allreduce")} \
            if (grouped and kviews == 1) else {}
        bucket = view_parallel.GradientBucket(list(params.values())) if (grouped and kviews > 1) else None

        vif = view_parallel.ViewsInFlight(dev, args.views_in_flight, staggered=args.staggered) if (args.views_in_flight > 1 and kviews > 1 and not (grouped and kviews == 1)) else None

        def render_fn(rast):
            def f():
                means2D = torch.zeros_like(params["means3D"], requires_grad=True)
                color, radii = rast(means3D=params["means3D"], means2D=means2D, **{k: v for k, v in params.items() if k != "means3D"})
                state["radii"] = radii
                return color
            return f
        render_fns = [render_fn(r) for r in rasterizers]

        def make_step(mode):
            def step():
                for p in params.values():
                    p.grad = None
                if vif is not None:   # k local views, several in flight (their gradients accumulate into .grad in view order)
                    state["color"] = vif.forward_backward(render_fns, dpixs)[-1]
                    if bucket is not None:
                        bucket.all_reduce()
                    return
                for rast, st, dp in zip(rasterizers, all_settings, dpixs):
                    means2D = torch.zeros_like(params["means3D"], requires_grad=True)  # gaussian_renderer/__init__.py:37
                    if grouped and kviews == 1:
                        # view-parallel: this rank's view; the backward exchanges the 59 floats/Gaussian of parameter
                        # gradients part by part (view_parallel.GradientExchange) and returns their sum over the ranks
                        color, radii = view_parallel.rasterize_view_parallel(params["means3D"], means2D, params["shs"], params["opacities"],
                                                                             params["scales"], params["rotations"], st, ex[mode])
                    else:
                        color, radii = rast(means3D=params["means3D"], means2D=means2D, **{k: v for k, v in params.items() if k != "means3D"})
                    color.backward(dp)   # k > 1: autograd accumulates the k local views into .grad
                if bucket is not None:
                    bucket.all_reduce()  # ONE flat all-reduce of the 59 floats/Gaussian per step, after the k local views
                state["color"], state["radii"] = color, radii
            return step
        steps = {m: make_step(m) for m in ("compact", "allreduce")}

    def timed(step, nsteps, only=None, per_step=False):
        """K steps between barriers.  per_step: one event per step boundary as well (K + 1 records) for the per-step times -- an
        event record is a barrier packet in the stream (6-8 us in front of the next launch), so the timed region proper runs
        without them and the per-step distribution comes from a second pass of the same K steps."""
        barrier()
        if not args.dry_run and only is not None:
            _C.profile_begin(only=only)
        marks = []

        def mark():
            if args.dry_run:
                marks.append(time.perf_counter())
            elif per_step:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append(e)
        t0 = time.perf_counter()
        mark()
        for _ in range(nsteps):
            step()
            mark()
        barrier()
        elapsed = time.perf_counter() - t0
        if args.dry_run:
            per = [(b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:])]
        else:
            per = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
        if grouped:
            tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed, per

    mode = args.sh_exchange
    step = steps[mode]
    # Settling phase, untimed and before the W warm-up steps: the first ~20 steps of a fresh process run up to 7 % slower
    # than the steady state (1.31 -> 1.22 ms at C3, decaying over ~25 ms of GPU time; a matmul loop beforehand does not
    # remove it, steps of this workload do), so a short K would measure the ramp instead of the training loop's rate.
    # No cyclic garbage collection from here to the end of the timed regions: a collection is a host pause of milliseconds
    # (seen: one 4.8 ms step among 30 of 1.23), and one that runs between the settling steps and the timed ones lets the GPU
    # idle, which restarts the ramp (measured: 1.61 -> 1.21 ms over the 20 timed steps; it is the two instruction-bound
    # blend kernels that run slower after an idle period, i.e. the shader clock).
    gc.collect()
    gc.disable()
    settle_steps = 0 if args.dry_run else max(0, args.settle_steps)
    phase["now"] = "settling steps (first collectives of the run)"
    for _ in range(settle_steps):   # a count, the same on every rank (the steps of N > 1 contain collectives)
        step()
    sync()
    phase["now"] = "warm-up steps"
    for _ in range(args.warmup):
        step()
    phase["now"] = "timed steps"
    # Timed region: HIP events only around the dominant kernel (every event record drains the queue for
    # ~5 us; bracketing all stages costs ~80 us per step, 4 % of it) plus one event per step boundary.  The full
    # per-kernel table comes from a second, untimed pass below.
    elapsed, per_step = timed(step, args.steps, only=DOMINANT_STAGE)
    dom_records = []
    if not args.dry_run:
        dom_records = _C.profile_end(capacity=4 * max(args.steps, 1))   # the dominant kernel's launches inside the timed region
        _, per_step = timed(step, args.steps, per_step=True)            # the per-step distribution: the same K steps once more (see timed())
    if os.environ.get("GSR_BENCH_DUMP_STEPS") == "1" and rank == 0:
        print("per-step ms:", " ".join(f"{x:.3f}" for x in per_step), file=sys.stderr, flush=True)
    dom_times, ktimes, table_steps = [], [], 0
    if not args.dry_run:
        dom_times = [ms for name, ms in dom_records if name == DOMINANT_STAGE]
        table_steps = max(3, min(args.steps, 10))
        _C.profile_begin()
        for _ in range(table_steps):
            step()
        torch.cuda.synchronize()
        ktimes = _C.profile_end(capacity=64 * table_steps)
    alt = None
    phase["now"] = "per-kernel table / alternative exchange mode"
    if grouped and (args.dry_run or kviews == 1):
        other = "allreduce" if mode == "compact" else "compact"
        for _ in range(max(1, args.warmup)):
            steps[other]()
        e2, per2 = timed(steps[other], args.steps)
        alt = dict(mode=other, value=round(world * args.steps / e2, 3), ms_per_step=round(e2 / args.steps * 1e3, 4), step_ms=step_stats(per2))

    gc.enable()
    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    out = dict(metric="train iters/sec (fwd+bwd rasterize) @1980x1080, 1M Gaussians", value=round(world * kviews * args.steps / elapsed, 3),
               unit="it/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 4),
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
               step_ms=dict(step_stats(per_step) or {}, note="a second pass of the same K steps with one event record per step boundary (each record is a "
                                                              "barrier packet, 6-8 us in front of the next launch: the timed region runs without them)"),
               settle=dict(steps=settle_steps,
                           note="untimed steps of the same workload before the W warm-up steps (device reaches its steady state: the "
                                "first ~20 steps of a fresh process run up to 7 % slower); --settle-steps 0 disables"))
    if alt is not None:
        out["alt_exchange"] = alt
    if parts_note:
        out["parts_choice"] = parts_note
    if grouped and kviews == 1:
        parallelism = f"view-parallel x{world}, one view per rank, SH gradient exchange: {mode}, backward in {args.parts} parts"
    elif grouped:
        parallelism = (f"view-parallel x{world}, {kviews} views per rank and step (local gradient accumulation), one flat all-reduce of "
                       "59 floats/Gaussian per step")
    else:
        parallelism = "single view" if kviews == 1 else f"{kviews} views per step, gradients accumulated"
    if kviews > 1:
        out["views_in_flight"] = args.views_in_flight if (args.views_in_flight > 1) else 1
        if args.views_in_flight > 1 and args.staggered:
            out["views_in_flight_schedule"] = "staggered"
        out["views_per_rank"] = kviews
        out["value_note"] = "value = views (fwd+bwd rasterize iterations) per second over all ranks; one step = views_per_rank views per rank"
    if args.dry_run:
        out.update(data="dry-run: process-group plumbing and gradient exchange only, NO rasterizer (not a measurement)",
                   config=dict(workload=f"{args.config}: exchange buffers of {P} Gaussians", P=P, views_per_step=world * kviews, parallelism=parallelism))
        print(json.dumps(out), flush=True)
        return

    # workload statistics for the algorithmic byte counts
    radii = state["radii"]
    V = int((radii > 0).sum().item())
    N, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    # R and R' (instances actually staged by the forward blend) from the state buffers
    cap = {}
    orig = _C.rasterize_gaussians

    def spy(*a):
        r = orig(*a)
        cap["R"], cap["img"], cap["geom"] = r[0], r[5], r[3]
        return r
    _C.rasterize_gaussians = spy
    with torch.no_grad():
        rasterizer(means3D=params["means3D"], means2D=torch.zeros_like(params["means3D"]),
                   **{k: v for k, v in params.items() if k != "means3D"})
    _C.rasterize_gaussians = orig
    R = int(cap["R"])
    il = _C.image_layout(W, H)
    tmc = cap["img"][il.tile_max_contrib:il.tile_max_contrib + 4 * T].view(torch.int32)
    rng = cap["img"][il.ranges:il.ranges + 8 * T].view(torch.int32).view(T, 2)
    # forward stages whole 256-batches until every pixel of the tile is done
    staged = torch.minimum(((tmc + 256) // 256) * 256, rng[:, 1] - rng[:, 0])
    Rp = int(staged.clamp(min=0).sum().item())
    fwd_b, bwd_b = algorithmic_bytes(P, V, R, Rp, N, T, M)
    per_kernel = {}
    for name, ms in ktimes:
        per_kernel.setdefault(name, []).append(ms)
    kern = {}
    allb = dict(fwd_b, **bwd_b)
    for name, v in per_kernel.items():
        avg = sum(v) / table_steps   # per step (a stage may be recorded more than once per step: tile_order, depth_sort)
        kern[name] = dict(ms=round(avg, 4), launches=len(v), algorithmic_bytes=allb.get(name),
                          GBps=round(allb[name] / (avg * 1e-3) / 1e9, 1) if allb.get(name) and avg > 0 else None,
                          hbm_frac=round(allb[name] / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 3) if allb.get(name) and avg > 0 else None)
    # every kernel also against the bytes the PMC counters saw it move (profiles/, same workload): a kernel that moves less than the
    # reference's stage would (or more) is then read against what it does, not against what it replaces
    for name, e in kern.items():
        tr = pmc_traffic(name, args.config) if name in KERNEL_SYMBOL else None
        if tr and e["ms"] > 0:
            e.update(pmc_traffic_bytes=int(tr), traffic_GBps=round(tr / (e["ms"] * 1e-3) / 1e9, 1), traffic_frac=round(tr / (e["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3))
    if "gaussian_backward" in kern:
        kern["gaussian_backward"]["note"] = ("algorithmic_bytes / hbm_frac are SURVEY 8(d)'s, which re-read the 192-byte SH row per visible Gaussian; this "
                                             "kernel reads the 36 bytes of colour derivatives the forward left instead, so it is to be read against what it "
                                             "moves: pmc_traffic_bytes / traffic_frac")
    tb = [n for n in ("col_scatter", "row_hist", "row_scatter") if n in kern]
    if len(tb) == 3:
        # column pairs and instances of this view as the binning lists them (the geometry buffer's trim words, the tile ranges)
        pairs, listed = listed_counts(cap["geom"], cap["img"], P, W, H)
        moved = binning_moved_bytes(P, V, R, listed, T, pairs)
        for n in tb:
            kern[n].update(algorithmic_bytes=moved[n], GBps=round(moved[n] / (kern[n]["ms"] * 1e-3) / 1e9, 1),
                           hbm_frac=round(moved[n] / (kern[n]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3))
        ms = sum(kern[n]["ms"] for n in tb)
        mb = sum(moved.values())
        kern["binning"] = dict(ms=round(ms, 4), launches=None, column_pairs=pairs, instances_listed=listed, num_rendered=R, algorithmic_bytes=mb, GBps=round(mb / (ms * 1e-3) / 1e9, 1),
                               hbm_frac=round(mb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                               note="col_scatter + row_hist + row_scatter of the column-pair binning against the bytes THEY have to move "
                                    "(16 P + 4 V + 24 B per column pair + 4 B per listed instance + R + 8 T: no per-instance key exists; instances_listed of the "
                                    "num_rendered instances are listed -- the others lie in tiles their splat provably misses, csrc/gsr_rect_trim.h); these kernels are bound by "
                                    "their LDS instruction stream and dependent round trips, not by bytes")
        if "depth_sort" in kern:
            cms = ms + kern["depth_sort"]["ms"]
            cb = allb["binning_chain"]
            kern["binning_chain"] = dict(ms=round(cms, 4), launches=None, vs_replaced_stages_bytes=cb,
                                         vs_replaced_stages_GBps=round(cb / (cms * 1e-3) / 1e9, 1),
                                         vs_replaced_stages_frac=round(cb / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                                         note="depth_sort (depth order of the Gaussians + pass 1's histogram) + col_scatter + row_hist + row_scatter = "
                                              "everything that replaces InclusiveSum + duplicateWithKeys + SortPairs + identifyTileRanges of the reference, "
                                              "against SURVEY 8(d)'s bytes for THOSE stages (28 P + 44 R + 16 T): a comparison with what the reference's "
                                              "stages would need at HBM speed, not this chain's own traffic; kernel times of the serial per-kernel table "
                                              "(in the timed region the depth sort shares the chip with the colour kernel)")
    step_bytes = sum(v for k, v in fwd_b.items() if k != "binning_chain") + sum(bwd_b.values())
    dom = max((k for k in kern if k not in ("binning", "binning_chain")), key=lambda k: kern[k]["ms"]) if kern else None
    roofline = None
    if dom:
        if dom == DOMINANT_STAGE and dom_times:   # its launches inside the timed region
            avg = sum(dom_times) / len(dom_times)
            kern[dom].update(ms=round(avg, 4), launches=len(dom_times), measured_in="timed region",
                             GBps=round(allb[dom] / (avg * 1e-3) / 1e9, 1) if allb.get(dom) and avg > 0 else None)
        a = kern[dom]["GBps"] or 0.0
        roofline = dict(kernel=dom, bound="hbm", binding_roof="valu_issue", achieved=a, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(a / HBM_PEAK_GBS, 4), traffic=pmc_traffic(dom, args.config),
                        valu_issue=valu_issue_fraction(dom, args.config, kern[dom]["ms"]),
                        note="bound = the roof that achieved / peak / frac of this line are measured against (the line's vocabulary is hbm | mfma: "
                             "algorithmic bytes per launch over the launch's duration against 8 TB/s); binding_roof = the roof that actually limits "
                             "this kernel: vector-ALU issue (valu_issue.frac: its instructions, counted by the SQ counters, over the rate this chip "
                             "sustains on a stream of the same class mix, against its duration), so its HBM fraction is small by construction; "
                             "'kernels' lists the streaming stages with their own HBM fractions",
                        avg_launch_ms=kern[dom]["ms"], algorithmic_bytes_per_launch=kern[dom]["algorithmic_bytes"],
                        step_algorithmic_bytes=step_bytes,
                        step_GBps=round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                        step_frac=round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    out.update(config=dict(workload=f"{args.config}: {P} Gaussians, SH deg {D}, {W}x{H}, mu={mu}, seed 0",
                           P=P, V=V, R=R, R_staged_fwd=Rp, views_per_step=world * kviews, parallelism=parallelism),
               roofline=roofline, kernels=kern,
               kernels_note=f"per-kernel ms: HIP events on the launch stream; '{DOMINANT_STAGE}' over the timed region, the "
                            f"others over {table_steps} extra untimed steps (bracketing every stage inside the timed region "
                            "would add ~80 us of event drains per step); 'preprocess_color' is timed in line here -- in the "
                            "timed region it runs on the library's helper stream beside 'preprocess' and 'depth_sort', so the "
                            "kernels add up to more than the step")
    if args.debug_mask:
        out["config"]["debug_mask"] = args.debug_mask   # a diagnostic run of an alternate path, not the headline
    if args.library:
        out["config"]["library"] = os.path.basename(args.library)   # a diagnostic run of an A/B build, not the headline
    if world == 1 and not args.no_extras:
        out["two_views_in_flight"] = bench_views_in_flight(scene, D, W, H, dev)
        out["loss_l1_ssim"] = bench_loss(state["color"].detach(), dev)
        if not args.no_train_step:
            for p in params.values():
                p.grad = None
            out["train_iteration"] = bench_train_step(scene, settings, D, dev)
    if world == 1 and kviews == 1 and not args.no_cpu_baseline:
        import numpy as np
        cb, o = cpu_baseline(scene, cam, D)
        out["cpu_baseline"] = cb
        out["cpu_baseline_torch"] = cpu_baseline_torch()
        out["parity_vs_oracle"] = parity_figures(o, state["color"], params, dpix, rasterizer, H, W)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
