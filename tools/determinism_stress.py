"""Is every output the same bits every time?  Many scenes, each rendered forward + backward several times through the product path
(GaussianRasterizer, lean mode, column-pair binning, bucket depth sort, helper streams) -- image, radii and every gradient must be
bit-identical from run to run, and identical to one run through the global radix depth passes (GSR_DEBUG_RADIX_DEPTH) and one
through the instance emission + tile sort (GSR_DEBUG_TILE_SORT).  The library has no atomics on floats and fixed summation orders,
so anything else is a race.  (Round 4: a barrier without its LDS wait in the depth sort showed only as "one run in ten differs" on
one scene kind -- tools/depth_sort_stress.py found it; this tool looks for its like across the whole path.)

Scene kinds: uniform cube; a dense clump (tile lists of tens of thousands); the clump at low opacity (deep walks: the forward's band
split and the backward's depth segments); Gaussians on a handful of distinct depths (the depth sort's copy levels); slabs of
thousands of Gaussians on a few float steps; most Gaussians culled.

The default binning leaves out tiles a splat provably misses (csrc/gsr_rect_trim.h); the tile-sort path bins every tile of the
rectangle like the reference.  The image must not notice, and neither do the gradients -- except where a heavy tile's backward walk
is cut into depth segments (checkpoints every 1024th LIST POSITION: the cuts fall on other instances when the list is shorter, and a
segment's starting state is a difference of forward sums, 5e-7 away from the recurrence).  With base_mask = 8 (GSR_DEBUG_NO_SPLIT in
every run) nothing is cut and every gradient must have the same bits on all three paths; without it the tile-sort comparison is
bitwise on image and radii and to 1e-3 of the gradient's largest magnitude otherwise (5e-7 on the blend sums, times the covariance chain).

    python tools/determinism_stress.py [n_scenes=60] [runs=6] [seed0=0] [base_mask=0]        (GPU box, repo root)"""
import os
import sys

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R_, os.path.join(R_, "gaussian-splatting_cc-comments_amd"), os.path.join(R_, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import gsr_scene
import util
from diff_gaussian_rasterization import GaussianRasterizer, _C

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
base_mask = int(sys.argv[4]) if len(sys.argv) > 4 else 0
NAMES = ("means3D", "shs", "opacities", "scales", "rotations")


def render(scene_dev, st, dpix):
    leaves = {k: v.clone().requires_grad_(True) for k, v in scene_dev.items()}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    color, radii = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=means2D, **{k: v for k, v in leaves.items() if k != "means3D"})
    color.backward(dpix)
    out = {"color": color.detach(), "radii": radii, "dL_dmeans2D": means2D.grad}
    out.update({"dL_d" + k: v.grad for k, v in leaves.items()})
    return out


def same(a, b, close=False):
    d = [k for k in a if not torch.equal(a[k].contiguous().view(torch.uint8), b[k].contiguous().view(torch.uint8))]
    if close:   # gradients: to 1e-3 of the tensor's largest magnitude (the image and the radii stay bitwise)
        d = [k for k in d if not k.startswith("dL_") or float((a[k] - b[k]).abs().max()) > 1e-3 * max(float(a[k].abs().max()), 1e-30)]
    return d


bad = 0
kinds = {}
for k in range(n):
    r = np.random.default_rng(700000 + seed0 + k)
    kind = int(r.integers(0, 6))
    P = int(np.exp(r.uniform(np.log(20_000), np.log(1_200_000))))
    D = int(r.integers(0, 4))
    W, H = (int(r.integers(200, 2000)), int(r.integers(100, 1100)))
    scene = gsr_scene.make_scene(P, float(r.uniform(-5.0, -2.0)), sh_degree=D, seed=int(r.integers(1 << 30)))
    if kind in (1, 2):   # half of the Gaussians in a clump; kind 2: nearly transparent, so the lists are walked deep
        g = torch.Generator().manual_seed(k)
        nb = P // 2
        means = scene.means3D.clone()
        means[:nb] = torch.tensor(r.uniform(-0.4, 0.4, 3), dtype=torch.float32) + torch.randn(nb, 3, generator=g) * float(r.uniform(0.05, 0.3))
        scene = scene._replace(means3D=means.contiguous())
        if kind == 2:
            scene = scene._replace(opacities=torch.sigmoid(torch.randn(P, 1, generator=g) - 3.0).contiguous())
    elif kind in (3, 4, 5):
        import test_depth_sort_gpu as T
        if kind == 3:
            z = r.choice(r.uniform(0.5, 50.0, int(r.integers(1, 9))), P)
        elif kind == 4:
            z = r.uniform(1.0, 9.0, P)
            for _ in range(int(r.integers(1, 4))):
                m = r.random(P) < r.uniform(0.05, 0.5)
                z0 = np.float32(r.uniform(1.0, 9.0))
                z[m] = z0 + r.integers(0, int(r.choice([1, 3, 40, 3000])), int(m.sum())) * np.spacing(z0)
        else:
            z = r.uniform(2.5, 5.5, P)
            z[r.random(P) < r.uniform(0.5, 0.99)] = -1.0
        sc = T._scene_with_depths(P, np.asarray(z, dtype=np.float64).copy(), seed=k)
        scene = scene._replace(means3D=sc.means3D)   # (the depths; SH degree, scales, opacities stay the random scene's)
        W, H = 203 * int(r.integers(1, 5)), 117 * int(r.integers(1, 5))
    kinds[kind] = kinds.get(kind, 0) + 1
    cam = gsr_scene.make_camera(W, H)
    scene_dev = {name: getattr(scene, name).to(dev).contiguous() for name in NAMES}
    dpix = torch.randn(3, H, W, generator=torch.Generator().manual_seed(k)).to(dev)
    msg = []
    try:
        first = render(scene_dev, util.hip_settings(scene, cam, D, dev, debug=base_mask), dpix)
        for it in range(1, runs):
            d = same(first, render(scene_dev, util.hip_settings(scene, cam, D, dev, debug=base_mask), dpix))
            if d:
                msg.append(f"run {it} differs from run 0 in {d}")
                break
        for name, mask in (("radix depth passes", _C.DEBUG_RADIX_DEPTH), ("tile sort", _C.DEBUG_TILE_SORT)):
            d = same(first, render(scene_dev, util.hip_settings(scene, cam, D, dev, debug=mask | base_mask), dpix),
                     close=(mask == _C.DEBUG_TILE_SORT and not (base_mask & _C.DEBUG_NO_SPLIT)))
            if d:
                msg.append(f"{name}: differs in {d}")
        torch.cuda.synchronize()
    except Exception as ex:  # noqa: BLE001
        msg.append(repr(ex)[:300])
    if msg:
        bad += 1
        print(f"scene {seed0 + k}: kind {kind}, P {P}, D {D}, {W}x{H}: {msg}", flush=True)
    if (k + 1) % 10 == 0:
        print(f"{k + 1} scenes x {runs} runs (+ 2 alternate paths), {bad} bad", flush=True)
print(f"{n} scenes ({dict(sorted(kinds.items()))}) x {runs} runs: " + ("FAILED" if bad else "every output bit-identical in every run and on both alternate paths"))
sys.exit(1 if bad else 0)
