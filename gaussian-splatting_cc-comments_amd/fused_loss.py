"""Fused training loss of the reference's loop (train.py:126-128, utils/loss_utils.py:16-63):

    Ll1 = l1_loss(image, gt_image)
    loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - ssim(image, gt_image))
    loss.backward()

as ONE forward + ONE backward HIP kernel over the image (include/gsr.h gsr_l1_ssim_loss) instead of five
grouped conv2d, their autograd backward and the elementwise ops in between.  The backward's result is
the dL/dpix tensor the rasterizer's backward consumes.  HIP tensors only (no CPU fallback).
"""
import ctypes

import torch

from diff_gaussian_rasterization import _C


def _lib():
    L = _C.lib()
    if not getattr(L, "_gsr_loss_bound", False):
        L.gsr_loss_scratch_bytes.restype = ctypes.c_size_t
        L.gsr_loss_scratch_bytes.argtypes = [ctypes.c_int] * 3
        L.gsr_l1_ssim_loss.restype = ctypes.c_int
        L.gsr_l1_ssim_loss.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + [ctypes.c_float] + [ctypes.c_void_p] * 4
        L._gsr_loss_bound = True
    return L


def l1_ssim_loss_and_grad(image, gt, lambda_dssim=0.2, want_grad=True):
    """-> (vals (3,) device tensor {loss, l1, ssim}, dloss/dimage (C,H,W) or None)"""
    if not image.is_cuda or not gt.is_cuda:
        raise RuntimeError("image and gt must be HIP (cuda) tensors; the fused loss has no CPU path")
    if image.shape != gt.shape or image.dim() != 3:
        raise RuntimeError(f"image and gt must both be (C,H,W); got {tuple(image.shape)} and {tuple(gt.shape)}")
    L = _lib()
    dev = image.device
    C, H, W = (int(v) for v in image.shape)
    img = image.detach().to(torch.float32).contiguous()
    g = gt.detach().to(device=dev, dtype=torch.float32).contiguous()
    with torch.cuda.device(dev):
        vals = torch.empty(3, dtype=torch.float32, device=dev)
        grad = torch.empty((C, H, W), dtype=torch.float32, device=dev) if want_grad else None
        scratch = torch.empty(L.gsr_loss_scratch_bytes(C, H, W), dtype=torch.uint8, device=dev)
        rc = L.gsr_l1_ssim_loss(C, H, W, img.data_ptr(), g.data_ptr(), float(lambda_dssim), vals.data_ptr(),
                                grad.data_ptr() if want_grad else None, scratch.data_ptr(),
                                torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"gsr error {rc}: {L.gsr_last_error().decode()}")
        scratch.record_stream(torch.cuda.current_stream(dev))
    return vals, grad


class _L1SSIMLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        vals, grad = l1_ssim_loss_and_grad(image, gt, lambda_dssim, want_grad=image.requires_grad)
        ctx.save_for_backward(grad if grad is not None else torch.empty(0, device=image.device))
        loss, l1, s = vals[0], vals[1], vals[2]
        ctx.mark_non_differentiable(l1, s)   # the very objects that are returned: Ll1 and ssim are for logging only
        return loss, l1, s

    @staticmethod
    def backward(ctx, g_loss, g_l1, g_ssim):
        (grad,) = ctx.saved_tensors
        return (grad * g_loss if grad.numel() else None), None, None


def l1_ssim_loss(image, gt, lambda_dssim=0.2):
    """loss (scalar tensor, differentiable w.r.t. image) of train.py:127."""
    return _L1SSIMLoss.apply(image, gt, lambda_dssim)[0]


def l1_ssim_loss_terms(image, gt, lambda_dssim=0.2):
    """-> (loss, Ll1, ssim) as in train.py:126-127 (Ll1 and ssim are for logging, not differentiable)."""
    return _L1SSIMLoss.apply(image, gt, lambda_dssim)
