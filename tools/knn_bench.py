"""distCUDA2 timing on the GPU (uniform and clustered clouds) next to scipy's k-d tree on the host cores."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "gaussian-splatting_cc-comments_amd")): sys.path.insert(0, p)
import numpy as np, torch
from simple_knn._C import distCUDA2
from diff_gaussian_rasterization import _C
for P, kind in ((100_000, "uniform"), (1_000_000, "uniform"), (1_000_000, "clustered"), (6_000_000, "clustered")):
    r = np.random.default_rng(1)
    if kind == "uniform":
        pts = r.uniform(-3, 3, size=(P, 3)).astype(np.float32)
    else:
        c = r.normal(size=(max(P // 200, 1), 3)) * 4
        pts = (c[r.integers(0, len(c), P)] + r.normal(size=(P, 3)) * r.choice([0.01, 0.1, 1.0], size=(P, 1))).astype(np.float32)
    t = torch.from_numpy(pts).cuda()
    for _ in range(2): distCUDA2(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): d = distCUDA2(t)
    torch.cuda.synchronize(); gpu_ms = (time.perf_counter() - t0) / 5 * 1e3
    cpu_s = None
    if P <= 1_000_000:
        from scipy.spatial import cKDTree
        t0 = time.perf_counter(); cKDTree(pts).query(pts, k=4, workers=-1); cpu_s = time.perf_counter() - t0
    print(f"P={P} {kind}: GPU {gpu_ms:.2f} ms" + (f", scipy cKDTree (all host cores) {cpu_s:.2f} s" if cpu_s else ""))
