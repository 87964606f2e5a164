"""The read interface of the reference's GaussianModel that the render caller uses
(scene/gaussian_model.py:40-50, 114-138), plus the two Python-side alternates of the hot path:
`eval_sh` (utils/sh_utils.py:57-112) and the covariance builder (utils/general_utils.py:78-150,
scene/gaussian_model.py:32-37) -- written device-agnostic (the reference hard-codes "cuda").
Training policy (optimiser, densification, PLY I/O) is out of scope.
"""
from types import SimpleNamespace

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def eval_sh(deg, sh, dirs):
    """sh [..., C, (deg+1)^2], dirs [..., 3] unit vectors -> [..., C]   (degrees 0..3)"""
    assert 0 <= deg <= 3 and sh.shape[-1] >= (deg + 1) ** 2
    result = C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            result = (result + C2[0] * xy * sh[..., 4] + C2[1] * yz * sh[..., 5] +
                      C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] + C2[3] * xz * sh[..., 7] + C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + C3[0] * y * (3 * xx - yy) * sh[..., 9] + C3[1] * xy * z * sh[..., 10] +
                          C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12] +
                          C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + C3[5] * z * (xx - yy) * sh[..., 14] +
                          C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return result


def build_rotation(r):
    q = r / torch.sqrt((r * r).sum(dim=1))[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1).view(-1, 3, 3)


def build_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    """(P,6) upper triangle of R S S^T R^T (strip_symmetric order xx xy xz yy yz zz)."""
    L = build_rotation(rotation) @ torch.diag_embed(scaling_modifier * scaling)
    cov = L @ L.transpose(1, 2)
    return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1)


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class GaussianParams:
    """Raw (pre-activation) parameters with the reference's activations."""

    def __init__(self, xyz, features_dc, features_rest, scaling_raw, rotation_raw, opacity_raw, max_sh_degree=3,
                 active_sh_degree=None):
        self._xyz, self._features_dc, self._features_rest = xyz, features_dc, features_rest
        self._scaling, self._rotation, self._opacity = scaling_raw, rotation_raw, opacity_raw
        self.max_sh_degree = max_sh_degree
        self.active_sh_degree = max_sh_degree if active_sh_degree is None else active_sh_degree

    @classmethod
    def from_activated(cls, means3D, shs, scales, rotations, opacities, max_sh_degree=3, active_sh_degree=None,
                       device=None, requires_grad=True):
        """Builds raw leaves whose activations reproduce the given activated tensors."""
        mk = lambda t: t.detach().to(device).clone().requires_grad_(requires_grad)
        return cls(mk(means3D), mk(shs[:, :1, :]), mk(shs[:, 1:, :]), mk(torch.log(scales)), mk(rotations),
                   mk(inverse_sigmoid(opacities)), max_sh_degree, active_sh_degree)

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity]

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        return build_covariance_from_scaling_rotation(self.get_scaling, scaling_modifier, self._rotation)

    def save_ply(self, path):
        """scene/gaussian_model.py:277-295"""
        import gsr_ply
        gsr_ply.save_gaussians_ply(path, self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation)

    @classmethod
    def load_ply(cls, path, max_sh_degree=3, device=None, requires_grad=True):
        """scene/gaussian_model.py:323-364 (active_sh_degree = max_sh_degree afterwards, :364)"""
        import gsr_ply
        lv = gsr_ply.load_gaussians_ply(path, max_sh_degree)
        mk = lambda a: torch.from_numpy(a).to(device).requires_grad_(requires_grad)
        return cls(mk(lv["xyz"]), mk(lv["features_dc"]), mk(lv["features_rest"]), mk(lv["scaling"]), mk(lv["rotation"]), mk(lv["opacity"]),
                   max_sh_degree, max_sh_degree)


def pipeline_params(convert_SHs_python=False, compute_cov3D_python=False, debug=False, fused_activations=False):
    """arguments/__init__.py:82-88 PipelineParams defaults (+ this build's `fused_activations` switch)."""
    return SimpleNamespace(convert_SHs_python=convert_SHs_python, compute_cov3D_python=compute_cov3D_python, debug=debug,
                           fused_activations=fused_activations)
