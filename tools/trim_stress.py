"""Randomised check of the trim words (csrc/gsr_rect_trim.h) beyond tests/test_trim_gpu.py: many small scenes with wide
distributions -- splats from a fraction of a pixel to the whole image, needles (one scale 30 x the others), opacities from 0 over
exactly 1 / 255 to 1, cameras far outside and inside the cloud, images from one tile to 700 pixels a side -- and for every scene
  * a float64 evaluation of alpha at EVERY pixel centre of EVERY tile of EVERY rectangle: no tile that holds a pixel with
    alpha >= (1 - 1e-4) / 255 and power <= 0 may be left out of the list;
  * the list length = the number of instances the trim words keep;
  * image, radii and final T of the default run = those of a run that bins every tile (GSR_DEBUG_NO_TRIM), bit for bit.

    python tools/trim_stress.py [n_scenes=200] [seed0=0]        (GPU box, repo root)"""
import os
import sys

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R_, os.path.join(R_, "gaussian-splatting_cc-comments_amd"), os.path.join(R_, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import gsr_scene
import util
from diff_gaussian_rasterization import _C

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
tot = kept_n = need_n = 0
for k in range(n):
    r = np.random.default_rng(800000 + seed0 + k)
    P = int(r.integers(50, 4000))
    W, H = int(r.integers(1, 700)), int(r.integers(1, 500))
    D = int(r.integers(0, 4))
    scene = gsr_scene.make_scene(P, float(r.uniform(-6.0, -0.5)), sh_degree=D, seed=int(r.integers(1 << 30)))
    g = torch.Generator().manual_seed(k)
    sc = scene.scales.clone()
    if r.random() < 0.5:   # needles and discs
        ax = torch.randint(0, 3, (P,), generator=g)
        sc[torch.arange(P), ax] *= float(r.choice([30.0, 1.0 / 30.0]))
    kind = int(r.integers(0, 4))
    if kind == 0:
        op = torch.rand(P, 1, generator=g)
    elif kind == 1:
        op = torch.sigmoid(torch.randn(P, 1, generator=g) * 2.0 - 4.0)
    elif kind == 2:
        op = torch.full((P, 1), 1.0)
    else:
        op = torch.rand(P, 1, generator=g) * (3.0 / 255.0)   # around the threshold
    op[::13] = 0.0
    op[3::13] = 1.0 / 255.0
    op[7::13] = 1.0
    scene = scene._replace(scales=sc.contiguous(), opacities=op.contiguous())
    cam = gsr_scene.ring_camera(W, H, int(r.integers(0, 8)), 8, radius=float(np.exp(r.uniform(np.log(0.05), np.log(8.0)))))
    msg = []
    try:
        gx = (W + 15) // 16
        h = util.hip_forward_backward(scene, cam, D, None)
        a = util.hip_forward_backward(scene, cam, D, None, debug=_C.DEBUG_NO_TRIM)
        for key in ("color", "radii", "final_T"):
            if not np.array_equal(h[key], a[key]):
                msg.append(f"{key} differs from the run that bins every tile")
        rs, m2, co = h["rshape"], h["means2D"].astype(np.float64), h["conic_opacity"].astype(np.float64)
        vis = np.nonzero(h["tiles_touched"] > 0)[0]
        if len(vis):
            packed = rs[vis, 0].astype(np.int64)
            x0, y0, w, hh = packed & 255, (packed >> 8) & 255, ((packed >> 16) & 255) + 1, (packed >> 24) + 1
            gi = np.repeat(vis, w * hh)
            first = np.cumsum(w * hh) - w * hh
            j = np.arange(len(gi)) - np.repeat(first, w * hh)
            wv, x0v, y0v = np.repeat(w, w * hh), np.repeat(x0, w * hh), np.repeat(y0, w * hh)
            ti = (y0v + j // wv) * gx + x0v + j % wv
            kept = util.trim_kept(rs, gi, ti, gx)
            can = np.zeros(len(gi), bool)
            for s0 in range(0, len(gi), 20000):   # largest alpha over the tile's 256 pixel centres, in slices
                sl = slice(s0, s0 + 20000)
                px = (ti[sl] % gx)[:, None] * 16 + np.arange(16)[None, :]
                py = (ti[sl] // gx)[:, None] * 16 + np.arange(16)[None, :]
                dx = m2[gi[sl], 0][:, None, None] - px[:, None, :]
                dy = m2[gi[sl], 1][:, None, None] - py[:, :, None]
                ca, cb, cc, o = (co[gi[sl], q][:, None, None] for q in range(4))
                power = -0.5 * (ca * dx * dx + cc * dy * dy) - cb * dx * dy
                alpha = np.minimum(0.99, o * np.exp(np.minimum(power, 50.0)))
                inside = (px[:, None, :] < W) & (py[:, :, None] < H)
                can[sl] = ((alpha >= (1.0 - 1e-4) / 255.0) & (power <= 0.0) & inside).any(axis=(1, 2))
            lost = can & ~kept
            if lost.any():
                msg.append(f"{int(lost.sum())} instances that can contribute are left out, e.g. Gaussian {gi[lost][0]} tile {ti[lost][0]}")
            listed = len(h["point_list"]) if h["num_rendered"] > 0 else 0
            if listed != int(kept.sum()):
                msg.append(f"the list holds {listed} instances, the trim words keep {int(kept.sum())}")
            tot += len(gi); kept_n += int(kept.sum()); need_n += int(can.sum())
    except Exception as ex:  # noqa: BLE001
        msg.append(repr(ex)[:300])
    if msg:
        bad += 1
        print(f"scene {seed0 + k}: P {P}, {W}x{H}, D {D}, opacity kind {kind}: {msg}", flush=True)
    if (k + 1) % 50 == 0:
        print(f"{k + 1} scenes, {tot} (Gaussian, tile) pairs, {bad} bad", flush=True)
print(f"{n} scenes, {tot} (Gaussian, tile) pairs: {need_n} can contribute ({need_n / max(tot, 1):.3f}), {kept_n} kept ({kept_n / max(tot, 1):.3f}); "
      + ("FAILED" if bad else "no pair that can contribute is left out, every image identical to the run that bins every tile"))
sys.exit(1 if bad else 0)
