"""Drop-in for `simple_knn._C` (submodules/simple-knn/ext.cpp:15-17, spatial.cu:14-25):

    distCUDA2(points (P,3) float32 HIP tensor) -> (P,) float32: mean squared distance to the 3 nearest
    other points, the scale initialiser of create_from_pcd (scene/gaussian_model.py:215).

ctypes binding of include/gsr.h gsr_knn_mean_dist2; exact search on the GPU, no CPU fallback.
"""
import ctypes

import torch

from diff_gaussian_rasterization import _C as _gsr


def distCUDA2(points):
    if not points.is_cuda:
        raise RuntimeError("points must be a HIP (cuda) tensor; distCUDA2 has no CPU path")
    if points.dim() != 2 or points.size(1) != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    L = _gsr.lib()
    if not getattr(L, "_gsr_knn_bound", False):
        L.gsr_knn_scratch_bytes.restype = ctypes.c_size_t
        L.gsr_knn_scratch_bytes.argtypes = [ctypes.c_int]
        L.gsr_knn_mean_dist2.restype = ctypes.c_int
        L.gsr_knn_mean_dist2.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4
        L._gsr_knn_bound = True
    dev = points.device
    P = int(points.size(0))
    pts = points.detach().to(torch.float32).contiguous()
    with torch.cuda.device(dev):
        means = torch.empty((P,), dtype=torch.float32, device=dev)  # spatial.cu:19-20 (torch.full 0.0; fully overwritten)
        if P:
            scratch = torch.empty((L.gsr_knn_scratch_bytes(P),), dtype=torch.uint8, device=dev)
            _gsr._check(L.gsr_knn_mean_dist2(P, pts.data_ptr(), means.data_ptr(), scratch.data_ptr(),
                                             torch.cuda.current_stream(dev).cuda_stream))
            scratch.record_stream(torch.cuda.current_stream(dev))
    return means
