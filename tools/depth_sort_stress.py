"""Randomised check of the bucket depth sort (csrc/depthsort.hip) against numpy's stable argsort: many sizes and depth
distributions -- uniform, log-uniform over many binades, mixtures with slabs of thousands of Gaussians on a few float steps, a
handful of distinct values, already sorted / reverse sorted input, most Gaussians culled -- so that one-item buckets, buckets cut
into parts, parts that overflow into the histogram levels and copied buckets all occur.  No oracle involved: the expected order
is np.argsort(depth bits, kind="stable") of the keys the library itself produced.

    python tools/depth_sort_stress.py [n_cases=200] [seed0=0]        (GPU box, repo root)"""
import os
import sys

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R_, os.path.join(R_, "gaussian-splatting_cc-comments_amd"), os.path.join(R_, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import gsr_scene
import test_depth_sort_gpu as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cam = gsr_scene.make_camera(203, 117)
bad = 0
tot = 0
kinds = {}
for k in range(n):
    r = np.random.default_rng(500000 + seed0 + k)
    P = int(np.exp(r.uniform(np.log(1), np.log(600_000))))
    kind = int(r.integers(0, 7))
    if kind == 0:
        z = r.uniform(2.5, 5.5, P)
    elif kind == 1:
        z = np.exp(r.uniform(np.log(0.25), np.log(float(r.choice([8.0, 300.0, 20000.0]))), P))
    elif kind == 2:   # slabs: a share of the Gaussians on a few float steps at random depths
        z = r.uniform(1.0, 9.0, P)
        for _ in range(int(r.integers(1, 4))):
            m = r.random(P) < r.uniform(0.05, 0.5)
            z0 = np.float32(r.uniform(1.0, 9.0))
            z[m] = z0 + r.integers(0, int(r.choice([1, 3, 40, 3000])), int(m.sum())) * np.spacing(z0)
    elif kind == 3:
        z = r.choice(r.uniform(0.5, 50.0, int(r.integers(1, 9))), P)
    elif kind == 4:
        z = np.sort(r.uniform(2.5, 5.5, P))[:: int(r.choice([1, -1]))]
    elif kind == 5:
        z = r.uniform(2.5, 5.5, P)
        z[r.random(P) < r.uniform(0.5, 0.999)] = -1.0
    else:   # one dense bucket and a thin spread over a wide range
        z = np.where(r.random(P) < 0.9, 4.0 + r.uniform(0, 1e-3, P), np.exp(r.uniform(np.log(0.3), np.log(500.0), P)))
    kinds[kind] = kinds.get(kind, 0) + 1
    scene = T._scene_with_depths(P, np.array(z, dtype=np.float64, copy=True, order="C").reshape(-1).copy(), seed=k)
    try:
        a = T._run(scene, cam, 0, 0)
    except AssertionError as e:
        bad += 1
        zz = np.asarray(z, dtype=np.float32)
        print(f"case {k}: kind {kind}, P {P}, visible-depth count {(zz > 0.2).sum()}, min/max {zz.min()} {zz.max()}: {e}", flush=True)
        continue
    bits = a["depth_bits"]
    want = np.argsort(bits, kind="stable").astype(np.uint32)
    ok = np.array_equal(a["perm"], want) and np.array_equal(a["sorted_depth_keys"], bits[want]) and a["depth_sort_result_in_alt"] == 0
    tot += P
    if not ok:
        bad += 1
        print(f"case {k}: kind {kind}, P {P}: MISMATCH", flush=True)
    if (k + 1) % 50 == 0:
        print(f"{k + 1} cases, {tot} Gaussians in total, {bad} bad", flush=True)
print(f"{n} cases ({kinds}), {tot} Gaussians in total: " + ("all orders equal numpy's stable argsort" if bad == 0 else f"{bad} MISMATCHES"))
sys.exit(1 if bad else 0)
