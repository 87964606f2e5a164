"""Drop-in replacement for the reference's `diff_gaussian_rasterization` package
(submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py): same names,
signatures, validation errors, saved-tensor set, gradient order and debug snapshot behaviour.
The only difference is underneath: `_C` is a ctypes binding of the gfx950 C-ABI library
(include/gsr.h) instead of a CUDA torch extension.

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


def cpu_deep_copy_tuple(input_tuple):
    """reference __init__.py:17-19"""
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, densify_stats=None):
    """reference __init__.py:22-45 (+ the optional densification-statistics tensors, see GaussianRasterizer)"""
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, densify_stats)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, densify_stats=None):
        # argument order of _C.rasterize_gaussians: reference __init__.py:64-84
        args = (
            raster_settings.bg,
            means3D,
            colors_precomp,
            opacities,
            scales,
            rotations,
            raster_settings.scale_modifier,
            cov3Ds_precomp,
            raster_settings.viewmatrix,
            raster_settings.projmatrix,
            raster_settings.tanfovx,
            raster_settings.tanfovy,
            raster_settings.image_height,
            raster_settings.image_width,
            sh,
            raster_settings.sh_degree,
            raster_settings.campos,
            raster_settings.prefiltered,
            raster_settings.debug,
        )
        if raster_settings.debug:  # reference __init__.py:87-94
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians(*args)

        ctx.raster_settings = raster_settings
        ctx.densify_stats = densify_stats
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        # no zero tensor for the (integer) radii output on the way back: autograd would fill P words per step for nothing
        ctx.set_materialize_grads(False)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        if grad_out_color is None:  # the image took no part in the loss: the zero gradient autograd would have materialised
            grad_out_color = torch.zeros((3, int(raster_settings.image_height), int(raster_settings.image_width)),
                                         dtype=torch.float32, device=ctx.saved_tensors[1].device)
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer) = ctx.saved_tensors

        # argument order of _C.rasterize_gaussians_backward: reference __init__.py:118-138
        args = (
            raster_settings.bg,
            means3D,
            radii,
            colors_precomp,
            scales,
            rotations,
            raster_settings.scale_modifier,
            cov3Ds_precomp,
            raster_settings.viewmatrix,
            raster_settings.projmatrix,
            raster_settings.tanfovx,
            raster_settings.tanfovy,
            grad_out_color,
            sh,
            raster_settings.sh_degree,
            raster_settings.campos,
            geomBuffer,
            num_rendered,
            binningBuffer,
            imgBuffer,
            raster_settings.debug,
        )
        if raster_settings.debug:  # reference __init__.py:141-148
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
                 grad_scales, grad_rotations) = _C.rasterize_gaussians_backward(*args, stats=ctx.densify_stats)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            # gradients of inputs that were not provided have no consumer below: the binding skips them
            (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
             grad_scales, grad_rotations) = _C.rasterize_gaussians_backward(*args, lean=True, stats=ctx.densify_stats)

        # gradient order: reference __init__.py:154-164
        grads = (
            grad_means3D,
            grad_means2D,
            grad_sh if (sh.numel() != 0 and grad_sh is not None) else None,
            grad_colors_precomp if colors_precomp.numel() != 0 else None,
            grad_opacities,
            grad_scales if scales.numel() != 0 else None,
            grad_rotations if rotations.numel() != 0 else None,
            grad_cov3Ds_precomp if cov3Ds_precomp.numel() != 0 else None,
            None,
            None,
        )
        return grads


class GaussianRasterizationSettings(NamedTuple):
    """reference __init__.py:168-180"""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    """reference __init__.py:182-258.

    densify_stats (extension, optional): (xyz_gradient_accum, denom, max_radii2D) float32 [P] tensors that the
    backward's per-Gaussian kernel updates in place for the Gaussians visible in this view -- the bookkeeping of
    train.py:157-159 / scene/gaussian_model.py:599-602 without separate passes (view_parallel.DensificationStats)."""

    def __init__(self, raster_settings, densify_stats=None):
        super().__init__()
        self.raster_settings = raster_settings
        self.densify_stats = densify_stats

    def markVisible(self, positions):
        # Mark visible points (based on frustum culling for camera) with a boolean
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])

        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, raster_settings, self.densify_stats)
