"""Randomised A/B of the two binning paths (column pairs, csrc/tilebin.hip, against instance emission + tile sort): many scenes,
sizes from one tile to 4096 pixels a side, cameras outside and inside the cloud (rectangles of one tile up to the whole image), dense
clumps (tile lists of tens of thousands).  Every scene: image, radii, the sorted instance list, the ranges, the gradient slots'
numbering and n_contrib must be bit-identical (both paths binning every tile of every rectangle, GSR_DEBUG_NO_TRIM on the column pairs), and the
default run -- tiles a splat provably misses left out -- must give the same image.  No oracle involved, so a scene costs milliseconds.

    python tools/binning_stress.py [n_scenes=300] [seed0=0]        (GPU box, repo root)"""
import os, sys, math
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R_, os.path.join(R_, "gaussian-splatting_cc-comments_amd"), os.path.join(R_, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import gsr_scene
import util
from diff_gaussian_rasterization import GaussianRasterizer, _C

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
tot_R = 0
for k in range(n):
    r = np.random.default_rng(100000 + seed0 + k)
    shape = r.integers(0, 5)
    if shape == 0:
        W, H = int(r.integers(1, 64)), int(r.integers(1, 64))
    elif shape == 1:
        W, H = int(r.integers(1000, 4097)), int(r.integers(1, 80))
    elif shape == 2:
        W, H = int(r.integers(1, 80)), int(r.integers(1000, 4097))
    else:
        W, H = int(r.integers(64, 2000)), int(r.integers(64, 1200))
    P = int(r.choice([1, 7, 64, 65, 1000, 1023, 1024, 1025, 5000, 30000, 120000]))
    D = int(r.integers(0, 4))
    mu = float(r.uniform(-5.0, -1.5))
    scene = gsr_scene.make_scene(P, mu, sh_degree=D, seed=int(r.integers(1 << 30)))
    if r.random() < 0.3 and P >= 1000:   # a dense clump
        g = torch.Generator().manual_seed(k)
        nb = P // 2
        means = scene.means3D.clone()
        means[:nb] = torch.tensor(r.uniform(-0.5, 0.5, 3), dtype=torch.float32) + torch.randn(nb, 3, generator=g) * 0.1
        scene = scene._replace(means3D=means.contiguous())
    inside = r.random() < 0.3
    eye = r.normal(size=3)
    eye = eye / np.linalg.norm(eye) * (float(r.uniform(0.1, 1.2)) if inside else float(r.uniform(2.0, 6.0)))
    fwd = r.uniform(-0.4, 0.4, 3) - eye
    fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, 1.0, 0.0]) if abs(fwd[1]) < 0.95 else np.array([1.0, 0.0, 0.0])
    right = np.cross(up, fwd); right /= np.linalg.norm(right)
    Rm = np.stack([right, np.cross(fwd, right), fwd], axis=1)
    cam = gsr_scene.make_camera(W, H, fovx=float(r.uniform(0.05, 2.4)), R=Rm, T=-Rm.T @ eye)
    dpix = torch.randn(3, H, W, generator=torch.Generator().manual_seed(k))
    try:
        # a: column pairs binning every tile of every rectangle (GSR_DEBUG_NO_TRIM), b: instance emission + tile sort -- the reference's
        # lists both; d: the default, column pairs without the tiles a splat provably misses (csrc/gsr_rect_trim.h): same image
        a = util.hip_forward_backward(scene, cam, D, dpix, debug=_C.DEBUG_NO_TRIM)
        b = util.hip_forward_backward(scene, cam, D, dpix, debug=_C.DEBUG_TILE_SORT)
        d = util.hip_forward_backward(scene, cam, D, None)
        keys = ["color", "radii", "final_T", "n_contrib", "ranges"] + (["point_list"] if a["num_rendered"] > 0 else [])
        diff = [x for x in keys if not np.array_equal(a[x], b[x])] + [x for x in a["grads"] if not np.array_equal(a["grads"][x], b["grads"][x])]
        diff += ["default run: " + x for x in ("color", "radii", "final_T") if not np.array_equal(d[x], a[x])]
        if d["num_rendered"] != a["num_rendered"] or (a["num_rendered"] > 0 and len(d["point_list"]) > len(a["point_list"])):
            diff.append("default run: num_rendered / list length")
        if int(a["tiles_touched"].astype(np.int64).sum()) != int(a["num_rendered"]) or int(b["num_rendered"]) != int(a["num_rendered"]):
            diff.append("num_rendered != sum(tiles_touched): the count was read back before it was complete")
        vis = a["tiles_touched"] > 0   # (slot_base is written for Gaussians with tiles only)
        if not np.array_equal(a["slot_base"][vis], b["slot_base"][vis]):
            diff.append("slot_base")
        tot_R += a["num_rendered"]
    except Exception as ex:  # noqa: BLE001
        diff = [repr(ex)[:200]]
    if diff:
        bad += 1
        print(f"scene {seed0 + k}: {W}x{H} P={P} D={D} inside={inside}: DIFFERS in {diff}", flush=True)
    if (k + 1) % 50 == 0:
        print(f"{k + 1} scenes, {tot_R} instances in total, {bad} bad", flush=True)
print("FAILED" if bad else "all scenes identical on both paths")
sys.exit(1 if bad else 0)
