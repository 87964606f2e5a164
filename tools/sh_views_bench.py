"""Time of gsr_sh_grad_from_views (rebuild of the summed SH gradient from V views) at 1M Gaussians."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "gaussian-splatting_cc-comments_amd")): sys.path.insert(0, p)
import torch
from diff_gaussian_rasterization import _C
dev = torch.device("cuda:0"); P = 1_000_000
means = torch.randn(P, 3, device=dev)
for V in (1, 2, 8):
    rgb = torch.randn(V, P, 3, device=dev); cams = torch.randn(V, 3, device=dev) * 4
    for _ in range(3): _C.sh_grad_from_views(means, cams, rgb, 3, 16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): _C.sh_grad_from_views(means, cams, rgb, 3, 16)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"V={V}: {ms:.3f} ms  ({(V * 12 + 12 + 192) * P / ms / 1e6:.0f} GB/s algorithmic)")
