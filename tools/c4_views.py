"""Per-view figures of BASELINE.json configs[3] (8 ring views of the 1M-Gaussian scene) on ONE GPU: V, R and the
steady-state fwd+bwd time of every view.  On the 8-GPU run each rank renders one of these views and every step ends in a
collective, so max / mean of the per-view step time is the load-imbalance bound of that run (what the slowest rank
costs the others), before any exchange time.

    python tools/c4_views.py gpurun_out/r3_c4_views.json [uniform|blob|lowop]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting_cc-comments_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import gsr_scene  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402


def main():
    out = sys.argv[1]
    variant = sys.argv[2] if len(sys.argv) > 2 else "uniform"   # uniform | blob | lowop: the non-symmetric scenes of tools/skew_bench.py
    dev = torch.device("cuda:0")
    P, W, H, D, mu = gsr_scene.CONFIGS["C3"]
    scene = gsr_scene.make_scene(P, mu, D, seed=0)
    if variant != "uniform":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import skew_bench
        scene = skew_bench.make_skewed_scene(variant, P, mu, D)
    to = lambda t: t.to(dev)
    params = {k: to(getattr(scene, k)).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    views = []
    for v in range(8):
        cam = gsr_scene.ring_camera(W, H, k=v, n=8)
        st = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, to(scene.bg), 1.0, to(cam.world_view_transform),
                                           to(cam.full_proj_transform), D, to(cam.camera_center), False, False)
        rast = GaussianRasterizer(st)
        dpix = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1 + v)).to(dev)
        state = {}

        def step():
            for p in params.values():
                p.grad = None
            c, r = rast(means2D=torch.zeros_like(params["means3D"], requires_grad=True), **params)
            c.backward(dpix)
            state["radii"] = r
        for _ in range(40):   # settle (bench.py: the first steps of a workload run slower)
            step()
        torch.cuda.synchronize()
        marks = []
        for _ in range(21):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e)
            step()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:]))
        from diff_gaussian_rasterization import _C
        cap = {}
        orig = _C.rasterize_gaussians

        def spy(*a):
            o = orig(*a)
            cap["R"] = o[0]
            return o
        _C.rasterize_gaussians = spy
        with torch.no_grad():
            rast(means2D=torch.zeros_like(params["means3D"]), **params)
        _C.rasterize_gaussians = orig
        views.append(dict(view=v, V=int((state["radii"] > 0).sum()), R=int(cap["R"]), step_ms_median=round(ms[len(ms) // 2], 4),
                          step_ms_min=round(ms[0], 4), step_ms_max=round(ms[-1], 4)))
        print(views[-1], flush=True)
    med = [v["step_ms_median"] for v in views]
    res = dict(scene=variant, workload="C4: 8 ring cameras (gsr_scene.ring_camera(k, 8), radius 4) of the C3 scene (1M Gaussians, SH deg 3, 1980x1080), "
                        "rendered one after the other on one MI355X; fwd+bwd of the drop-in rasterizer, 21 timed steps after 40 settling steps",
               views=views, step_ms_mean=round(sum(med) / 8, 4), step_ms_max=max(med),
               imbalance_max_over_mean=round(max(med) / (sum(med) / 8), 4),
               note="on the 8-GPU run every step ends in a collective, so the slowest view sets the step: 8 views per max(step_ms) "
                    "is the compute-side ceiling of the view-parallel rate, before any exchange time")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "views"}))


if __name__ == "__main__":
    main()
