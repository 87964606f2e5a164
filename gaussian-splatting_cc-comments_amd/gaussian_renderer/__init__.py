"""Caller of the rasterizer hot path with the reference's contract
(gaussian_renderer/__init__.py:18-124): `render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier,
override_color)` -> {"render", "viewspace_points", "visibility_filter", "radii"}.

What the contract fixes, and this module keeps:
  * the rasterizer receives ACTIVATED parameters (`pc.get_*`), the camera's `world_view_transform`,
    `full_proj_transform`, `camera_center`, `tan(FoV / 2)` and `pc.active_sh_degree`;
  * a zero (P,3) tensor travels as `means2D`; its `.grad` after backward is dL/d(screen-space mean),
    which densification reads (train.py:157-159) -- it is returned as "viewspace_points";
  * `pipe.convert_SHs_python` evaluates the SH colours in PyTorch and hands them over as
    `colors_precomp`; `pipe.compute_cov3D_python` does the same for the covariance
    (`pc.get_covariance`); `override_color` replaces the colours altogether;
  * `visibility_filter` = `radii > 0`.
`pc` is anything with the read interface of scene/gaussian_model.py:114-138 -- gsr_model.GaussianParams
or the reference's own GaussianModel.

Extension (not in the reference): `pipe.fused_activations = True` renders straight from the optimiser
leaves (`pc._xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation`) with the activations and
their backward done inside the per-Gaussian kernels (fused_params.py) -- same result dict.
"""
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from gsr_model import eval_sh


def _settings_for(camera, pc, pipe, bg_color, scaling_modifier):
    return GaussianRasterizationSettings(
        image_height=int(camera.image_height), image_width=int(camera.image_width),
        tanfovx=math.tan(0.5 * camera.FoVx), tanfovy=math.tan(0.5 * camera.FoVy),
        bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=camera.world_view_transform, projmatrix=camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=camera.camera_center,
        prefiltered=False, debug=pipe.debug)


def _python_sh_colors(camera, pc):
    """SH -> RGB outside the kernel (the reference's convert_SHs_python branch)."""
    feats = pc.get_features                                            # (P, (Dmax+1)^2, 3)
    per_channel = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
    view_dir = pc.get_xyz - camera.camera_center.repeat(feats.shape[0], 1)
    view_dir = view_dir / view_dir.norm(dim=1, keepdim=True)
    return torch.clamp_min(eval_sh(pc.active_sh_degree, per_channel, view_dir) + 0.5, 0.0)


def _result(image, screenspace_points, radii):
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, densify_stats=None):
    """Render the scene seen from `viewpoint_camera`.  `bg_color` must live on the GPU.
    densify_stats (extension): see GaussianRasterizer -- the statistics of train.py:157-159 updated by the backward."""
    xyz = pc.get_xyz
    # carrier of the screen-space gradient: zeros, a non-leaf that keeps its grad
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    settings = _settings_for(viewpoint_camera, pc, pipe, bg_color, scaling_modifier)

    python_cov = bool(pipe.compute_cov3D_python)
    python_sh = bool(pipe.convert_SHs_python)
    if getattr(pipe, "fused_activations", False) and override_color is None and not python_cov and not python_sh:
        from fused_params import rasterize_leaf_gaussians
        image, radii = rasterize_leaf_gaussians(pc._xyz, screenspace_points, pc._features_dc, pc._features_rest, pc._opacity,
                                                pc._scaling, pc._rotation, settings, densify_stats)
        return _result(image, screenspace_points, radii)

    inputs = dict(means3D=xyz, means2D=screenspace_points, opacities=pc.get_opacity,
                  shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None)
    if python_cov:
        inputs["cov3D_precomp"] = pc.get_covariance(scaling_modifier)
    else:
        inputs["scales"], inputs["rotations"] = pc.get_scaling, pc.get_rotation
    if override_color is not None:
        inputs["colors_precomp"] = override_color
    elif python_sh:
        inputs["colors_precomp"] = _python_sh_colors(viewpoint_camera, pc)
    else:
        inputs["shs"] = pc.get_features

    image, radii = GaussianRasterizer(raster_settings=settings, densify_stats=densify_stats)(**inputs)
    return _result(image, screenspace_points, radii)
