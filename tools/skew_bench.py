"""Kernel times on deliberately NON-uniform scenes: real captures have a heavy-tailed tile-length distribution, the
uniform benchmark scene does not, and one wave per tile is the design most exposed to it.

  blob    half of the 1M Gaussians in a dense blob at the centre of the view: a few hundred tiles carry instance lists
          20-30x the mean; early termination bounds how far the blend kernels walk them (tile_max_contrib << range)
  lowop   the same blob with low opacities (sigmoid(N(-3,1))): pixels saturate late, so the heavy tiles are walked (almost)
          to the end of their lists in both directions -- the case early termination does not rescue

    python tools/skew_bench.py [--variant blob|lowop|uniform] [--out gpurun_out/r3_skew.jsonl]
Prints (and appends to --out) one JSON line: step time, per-kernel times, tile-length statistics."""
import argparse
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "gaussian-splatting_cc-comments_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import gsr_scene  # noqa: E402


def make_skewed_scene(variant, P=1_000_000, mu=-5.0, D=3):
    sc = gsr_scene.make_scene(P, mu, D, seed=0)
    if variant == "uniform":
        return sc
    g = torch.Generator().manual_seed(1)
    means = sc.means3D.clone()
    means[: P // 2] = torch.randn(P // 2, 3, generator=g) * torch.tensor([0.25, 0.15, 0.4])   # dense blob in the centre
    opac = sc.opacities.clone()
    if variant == "lowop":
        opac[: P // 2] = torch.sigmoid(torch.randn(P // 2, 1, generator=g) - 3.0)
    elif variant != "blob":
        raise ValueError(variant)
    return sc._replace(means3D=means.contiguous(), opacities=opac.contiguous())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="blob", choices=["blob", "lowop", "uniform"])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
    dev = torch.device("cuda:0")
    P, W, H, D = 1_000_000, 1980, 1080, 3
    sc = make_skewed_scene(a.variant, P, -5.0, D)
    cam = gsr_scene.make_camera(W, H)
    to = lambda t: t.to(dev)
    params = dict(means3D=to(sc.means3D).requires_grad_(True), shs=to(sc.shs).requires_grad_(True), opacities=to(sc.opacities).requires_grad_(True),
                  scales=to(sc.scales).requires_grad_(True), rotations=to(sc.rotations).requires_grad_(True))
    st = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, to(sc.bg), 1.0, to(cam.world_view_transform), to(cam.full_proj_transform), D,
                                       to(cam.camera_center), False, False)
    rast = GaussianRasterizer(st)
    dpix = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)

    def step():
        for p in params.values():
            p.grad = None
        c, r = rast(means2D=torch.zeros_like(params["means3D"], requires_grad=True), **params)
        c.backward(dpix)
        return r
    for _ in range(40):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20 * 1e3
    _C.profile_begin()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    k = {}
    for n, ms in _C.profile_end(4096):
        k.setdefault(n, []).append(ms)
    cap = {}
    orig = _C.rasterize_gaussians

    def spy(*args):
        o = orig(*args)
        cap["R"], cap["img"] = o[0], o[5]
        return o
    _C.rasterize_gaussians = spy
    with torch.no_grad():
        rast(means2D=torch.zeros_like(params["means3D"]), **params)
    _C.rasterize_gaussians = orig
    T = ((W + 15) // 16) * ((H + 15) // 16)
    il = _C.image_layout(W, H)
    rng = cap["img"][il.ranges:il.ranges + 8 * T].view(torch.int32).view(T, 2)
    lens = (rng[:, 1] - rng[:, 0]).float()
    tmc = cap["img"][il.tile_max_contrib:il.tile_max_contrib + 4 * T].view(torch.int32).float()
    walked = torch.minimum(lens, tmc)
    q = lambda t, p: float(torch.quantile(t, p))
    line = dict(variant=a.variant, ms_per_step=round(dt, 4), R=int(cap["R"]), tiles=T,
                tile_len=dict(mean=round(float(lens.mean()), 1), p99=round(q(lens, 0.99), 1), max=float(lens.max())),
                walked_backward=dict(mean=round(float(walked.mean()), 1), p99=round(q(walked, 0.99), 1), max=float(walked.max()),
                                     max_over_mean=round(float(walked.max() / walked.mean()), 2)),
                kernels_ms={n: round(sum(v) / 5, 4) for n, v in k.items()})
    s = json.dumps(line)
    print(s)
    if a.out:
        with open(a.out, "a") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
