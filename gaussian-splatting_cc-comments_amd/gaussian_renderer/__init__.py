"""Caller of the rasterizer hot path with the reference's contract
(gaussian_renderer/__init__.py:18-124): builds GaussianRasterizationSettings from a camera, passes
ACTIVATED parameters, creates the zero `screenspace_points` tensor whose .grad carries dL/dmean2D
back to densification, chooses SH-in-kernel vs SH-in-Python (`pipe.convert_SHs_python`) and
scale/rotation-in-kernel vs covariance-in-Python (`pipe.compute_cov3D_python`), and returns the
same dict.  `pc` is anything with the read interface of scene/gaussian_model.py:114-138
(get_xyz, get_opacity, get_scaling, get_rotation, get_features, get_covariance, active_sh_degree,
max_sh_degree) -- e.g. gsr_model.GaussianParams or the reference's own GaussianModel.

Extension (not in the reference): `pipe.fused_activations = True` renders straight from the optimiser
leaves (`pc._xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation`) with the activations and
their backward done inside the per-Gaussian kernels (fused_params.py) -- same result dict.
"""
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from gsr_model import eval_sh


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    """Render the scene.  Background tensor (bg_color) must be on the GPU."""
    # zero tensor used to make pytorch return gradients of the 2D (screen-space) means
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True,
                                          device=pc.get_xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)

    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=pipe.debug,
    )
    if getattr(pipe, "fused_activations", False) and override_color is None and not pipe.compute_cov3D_python \
            and not pipe.convert_SHs_python:
        from fused_params import rasterize_leaf_gaussians
        rendered_image, radii = rasterize_leaf_gaussians(pc._xyz, screenspace_points, pc._features_dc, pc._features_rest,
                                                         pc._opacity, pc._scaling, pc._rotation, raster_settings)
        return {"render": rendered_image,
                "viewspace_points": screenspace_points,
                "visibility_filter": radii > 0,
                "radii": radii}

    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    means3D = pc.get_xyz
    means2D = screenspace_points
    opacity = pc.get_opacity

    # 3D covariance: precomputed in Python if asked, otherwise from scaling / rotation by the rasterizer
    scales = None
    rotations = None
    cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling
        rotations = pc.get_rotation

    # colours: precomputed from SHs in Python if asked, otherwise SH -> RGB by the rasterizer
    shs = None
    colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = (pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1))
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized)
            colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    rendered_image, radii = rasterizer(
        means3D=means3D,
        means2D=means2D,
        shs=shs,
        colors_precomp=colors_precomp,
        opacities=opacity,
        scales=scales,
        rotations=rotations,
        cov3D_precomp=cov3D_precomp)

    # Gaussians that were frustum culled or had a radius of 0 were not visible
    return {"render": rendered_image,
            "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0,
            "radii": radii}
