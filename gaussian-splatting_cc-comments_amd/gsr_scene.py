"""Seeded synthetic scenes and cameras for the rasterizer hot path.

Camera matrices follow the reference conventions exactly (scene/cameras.py:57-61,
utils/graphics_utils.py:38-75): `viewmatrix` = W2C transposed, `projmatrix` =
viewmatrix @ P^T, `campos` = inverse(viewmatrix)[3, :3].
"""
import math
from typing import NamedTuple, Optional

import numpy as np
import torch

# name: (P, W, H, sh_degree, mu)  -- BASELINE.md section 2
CONFIGS = {
    "C1": (10_000, 256, 256, 0, -3.5),
    "C2": (100_000, 1980, 1080, 3, -4.5),
    "C3": (1_000_000, 1980, 1080, 3, -5.0),
    "C5": (6_000_000, 3840, 2160, 3, -5.5),
}


class Camera(NamedTuple):
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor  # (4,4) W2C^T
    full_proj_transform: torch.Tensor   # (4,4)
    camera_center: torch.Tensor         # (3,)

    @property
    def tanfovx(self):
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.FoVy * 0.5)


def world_to_view(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """utils/graphics_utils.py:38-57 getWorld2View2 with translate=0, scale=1 (R is C2W rotation)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return np.float32(np.linalg.inv(np.linalg.inv(Rt)))


def projection_matrix(znear, zfar, fovX, fovY) -> torch.Tensor:
    """utils/graphics_utils.py:59-80 getProjectionMatrix."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(W, H, fovx=1.0, R: Optional[np.ndarray] = None, T: Optional[np.ndarray] = None,
                znear=0.01, zfar=100.0) -> Camera:
    """scene/cameras.py:57-61.  Default: camera at (0,0,-4) looking down +z."""
    if R is None:
        R = np.eye(3)
    if T is None:
        T = np.array([0.0, 0.0, 4.0])
    tanfovx = math.tan(fovx * 0.5)
    tanfovy = tanfovx * H / W
    fovy = 2.0 * math.atan(tanfovy)
    # exactly the reference's sequence of operations on exactly its tensor layouts: the inverse of the TRANSPOSED VIEW
    # (not of a contiguous copy: LAPACK then sees the other storage order and the centre differs in the last bit --
    # tests/golden/camera_class_golden.npz)
    wvt = torch.tensor(world_to_view(R, T)).transpose(0, 1)
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wvt.inverse()[3, :3].contiguous()
    return Camera(W, H, fovx, fovy, wvt.contiguous(), full, center)


def ring_camera(W, H, k, n=8, radius=4.0, fovx=1.0) -> Camera:
    """k-th of n cameras on a ring of `radius` around the origin, looking at it."""
    ang = 2.0 * math.pi * k / n
    c = np.array([radius * math.sin(ang), 0.0, -radius * math.cos(ang)])  # k=0 -> (0,0,-4)
    fwd = -c / np.linalg.norm(c)
    up = np.array([0.0, 1.0, 0.0])
    right = np.cross(up, fwd)
    right /= np.linalg.norm(right)
    up2 = np.cross(fwd, right)
    R = np.stack([right, up2, fwd], axis=1)  # C2W rotation (columns = camera axes in world)
    T = -R.T @ c
    return make_camera(W, H, fovx, R, T)


class Scene(NamedTuple):
    means3D: torch.Tensor    # (P,3)
    scales: torch.Tensor     # (P,3) activated (exp)
    rotations: torch.Tensor  # (P,4) normalised
    opacities: torch.Tensor  # (P,1) activated (sigmoid)
    shs: torch.Tensor        # (P,16,3) or (P,1,3) at degree 0
    bg: torch.Tensor         # (3,)


def make_scene(P, mu, sh_degree=3, seed=0, n_coeffs=None) -> Scene:
    """BASELINE.md section 2 / SURVEY 8d synthetic inputs (CPU tensors, fp32, torch generator)."""
    g = torch.Generator().manual_seed(seed)
    means = (torch.rand(P, 3, generator=g) * 3.0 - 1.5)
    scales = torch.exp(torch.randn(P, 3, generator=g) * 0.7 + mu)
    rot = torch.nn.functional.normalize(torch.randn(P, 4, generator=g), dim=1)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g))
    M = n_coeffs if n_coeffs is not None else (sh_degree + 1) ** 2
    shs = torch.randn(P, M, 3, generator=g)
    shs[:, 1:, :] *= 0.2
    bg = torch.tensor([0.1, 0.2, 0.3])
    return Scene(means.contiguous(), scales.contiguous(), rot.contiguous(), opac.contiguous(), shs.contiguous(), bg)


def make_config(name, seed=0):
    P, W, H, D, mu = CONFIGS[name]
    return make_scene(P, mu, D, seed), make_camera(W, H), D
