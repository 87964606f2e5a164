"""Generates tests/golden/*.npz by IMPORTING the reference's own Python helpers in this container
(the only part of the rasterizer path the reference offers as importable CPU code):

    utils/sh_utils.py:eval_sh                     -> SH colour and, through torch.autograd, its gradient
    utils/graphics_utils.py:getWorld2View2,
                            getProjectionMatrix   -> camera matrices
    utils/general_utils.py:build_rotation, build_scaling_rotation, strip_symmetric (:78-150)
                                                  -> the 3D covariance (pins computeCov3D, forward.cu:146-180, for unit
                                                     quaternions, and the compute_cov3D_python alternate)
    scene/cameras.py:16-61 Camera                 -> full_proj_transform / camera_center composition

These last files hard-code device="cuda" / .cuda(); the script redirects that to the CPU for the duration of the calls,
without touching the reference.  scene/gaussian_model.py itself is NOT imported (it needs `plyfile` and `simple_knn`,
which are not installed here, and nothing stands in for them): its covariance builder (:32-37) is the three calls
`L = build_scaling_rotation(scaling_modifier * scaling, rotation); strip_symmetric(L @ L.transpose(1, 2))` of the
general_utils functions above, made here in that order.  scene/cameras.py is loaded as a file, because its package
__init__ imports the dataset readers (plyfile again).

Run:  python tests/golden/make_golden.py     (needs /root/reference; the fixtures are committed,
the tests only read the .npz files).  Fixtures are data: inputs and the reference's outputs.
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
from utils.graphics_utils import getProjectionMatrix, getWorld2View2  # noqa: E402
from utils.loss_utils import l1_loss, ssim  # noqa: E402
from utils.sh_utils import eval_sh  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def sh_fixture():
    g = torch.Generator().manual_seed(1234)
    n = 257
    out = {}
    campos = torch.tensor([0.3, -0.2, -4.0])
    pos = torch.rand(n, 3, generator=g) * 3 - 1.5
    shs = torch.randn(n, 16, 3, generator=g)
    shs[:, 1:] *= 0.4
    shs[::7, 0, :] -= 3.0   # force some negative colours so the clamp (and its flags) is exercised
    dL_dcolor = torch.randn(n, 3, generator=g)
    out.update(campos=campos.numpy(), pos=pos.numpy(), shs=shs.numpy(), dL_dcolor=dL_dcolor.numpy())
    for deg in range(4):
        p = pos.clone().requires_grad_(True)
        s = shs.clone().requires_grad_(True)
        # gaussian_renderer/__init__.py:93-98 (convert_SHs_python branch)
        shs_view = s.transpose(1, 2).reshape(-1, 3, 16)
        dir_pp = p - campos.repeat(n, 1)
        dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        sh2rgb = eval_sh(deg, shs_view, dir_pp_normalized)
        colors = torch.clamp_min(sh2rgb + 0.5, 0.0)
        (colors * dL_dcolor).sum().backward()
        out[f"rgb_deg{deg}"] = colors.detach().numpy()
        out[f"raw_deg{deg}"] = (sh2rgb + 0.5).detach().numpy()
        out[f"dL_dsh_deg{deg}"] = s.grad.numpy()
        out[f"dL_dpos_deg{deg}"] = p.grad.numpy() if p.grad is not None else np.zeros((n, 3), np.float32)
    np.savez_compressed(os.path.join(HERE, "sh_golden.npz"), **out)


def camera_fixture():
    out = {}
    cases = []
    rng = np.random.default_rng(7)
    for k in range(6):
        # random rotation via QR, random translation
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        t = rng.normal(size=3) * 2
        fovx = float(rng.uniform(0.4, 1.4))
        fovy = float(rng.uniform(0.3, 1.2))
        cases.append((q, t, fovx, fovy))
    out["R"] = np.stack([c[0] for c in cases])
    out["T"] = np.stack([c[1] for c in cases])
    out["fovx"] = np.array([c[2] for c in cases])
    out["fovy"] = np.array([c[3] for c in cases])
    out["w2v"] = np.stack([getWorld2View2(c[0], c[1]) for c in cases])
    out["proj"] = np.stack([getProjectionMatrix(znear=0.01, zfar=100.0, fovX=c[2], fovY=c[3]).numpy() for c in cases])
    np.savez_compressed(os.path.join(HERE, "camera_golden.npz"), **out)


def loss_fixture():
    """train.py:126-127: loss = (1 - lambda) * l1_loss + lambda * (1 - ssim); gradient by torch.autograd."""
    g = torch.Generator().manual_seed(99)
    out = {}
    cases = {}
    a = torch.rand(3, 45, 70, generator=g)
    cases["noise"] = (a, torch.rand(3, 45, 70, generator=g))
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 64), torch.linspace(0, 1, 97), indexing="ij")
    base = torch.stack([0.5 + 0.5 * torch.sin(9 * xx + 3 * yy), xx * yy, (xx - yy).abs()])
    cases["smooth"] = ((base + 0.05 * torch.randn(3, 64, 97, generator=g)).clamp(0, 1), base)
    cases["equal"] = (base.clone(), base.clone())
    for name, (img, gt) in cases.items():
        x = img.clone().requires_grad_(True)
        Ll1 = l1_loss(x, gt)
        s = ssim(x, gt)
        loss = (1.0 - 0.2) * Ll1 + 0.2 * (1.0 - s)
        loss.backward()
        out[f"{name}_img"], out[f"{name}_gt"] = img.numpy(), gt.numpy()
        out[f"{name}_vals"] = np.array([loss.item(), Ll1.item(), s.item()], np.float64)
        out[f"{name}_grad"] = x.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "loss_golden.npz"), **out)


class _on_cpu:
    """Runs the reference's CUDA-only Python on the CPU: `device="cuda"` arguments of the torch factories and
    Tensor.cuda() are redirected for the duration of the block."""
    NAMES = ("zeros", "ones", "empty", "tensor", "full")

    def __enter__(self):
        self.saved = {n: getattr(torch, n) for n in self.NAMES}
        self.cuda = torch.Tensor.cuda

        def wrap(f):
            def g(*a, **k):
                if "device" in k and str(k["device"]).startswith("cuda"):
                    k["device"] = "cpu"
                return f(*a, **k)
            return g
        for n, f in self.saved.items():
            setattr(torch, n, wrap(f))
        torch.Tensor.cuda = lambda t, *a, **k: t
        return self

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(torch, n, f)
        torch.Tensor.cuda = self.cuda
        return False


def cov3d_fixture():
    from utils.general_utils import build_rotation, build_scaling_rotation, strip_symmetric
    g = torch.Generator().manual_seed(4321)
    n = 300
    scaling = torch.exp(torch.randn(n, 3, generator=g) * 0.8 - 2.0)
    scaling[::11] *= torch.tensor([30.0, 1.0, 0.05])       # needles and pancakes
    rot_raw = torch.randn(n, 4, generator=g)                  # unnormalised quaternions
    rot_unit = torch.nn.functional.normalize(rot_raw)
    out = dict(scaling=scaling.numpy(), rotation_raw=rot_raw.numpy(), rotation_unit=rot_unit.numpy())
    with _on_cpu():
        out["R_of_raw"] = build_rotation(rot_raw).numpy()                      # general_utils.py:105 (normalises inside)
        out["L_unit"] = build_scaling_rotation(scaling, rot_unit).numpy()      # general_utils.py:130
        for mod in (1.0, 1.7):
            # scene/gaussian_model.py:32-37 build_covariance_from_scaling_rotation, call for call
            L = build_scaling_rotation(mod * scaling, rot_unit)
            actual_covariance = L @ L.transpose(1, 2)
            out[f"cov_mod{mod}"] = strip_symmetric(actual_covariance).numpy()
    np.savez_compressed(os.path.join(HERE, "cov3d_golden.npz"), **out)


def camera_class_fixture():
    """scene/cameras.py loaded as a file (its package __init__ pulls in the dataset readers and plyfile)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_scene_cameras", os.path.join(REF, "scene", "cameras.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(11)
    out = {k: [] for k in ("R", "T", "fovx", "fovy", "world_view_transform", "projection_matrix", "full_proj_transform", "camera_center")}
    img = torch.zeros(3, 8, 12)
    with _on_cpu():
        for k in range(6):
            q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            if np.linalg.det(q) < 0:
                q[:, 0] = -q[:, 0]
            t = rng.normal(size=3) * 2
            fovx, fovy = float(rng.uniform(0.4, 1.4)), float(rng.uniform(0.3, 1.2))
            cam = mod.Camera(colmap_id=k, R=q, T=t, FoVx=fovx, FoVy=fovy, image=img, gt_alpha_mask=None, image_name=str(k), uid=k,
                             data_device="cpu")
            for name, v in (("R", q), ("T", t), ("fovx", fovx), ("fovy", fovy), ("world_view_transform", cam.world_view_transform.numpy()),
                            ("projection_matrix", cam.projection_matrix.numpy()), ("full_proj_transform", cam.full_proj_transform.numpy()),
                            ("camera_center", cam.camera_center.numpy())):
                out[name].append(v)
    np.savez_compressed(os.path.join(HERE, "camera_class_golden.npz"), **{k: np.stack([np.asarray(x) for x in v]) for k, v in out.items()})


if __name__ == "__main__":
    sh_fixture()
    camera_fixture()
    loss_fixture()
    cov3d_fixture()
    camera_class_fixture()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
