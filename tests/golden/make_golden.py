"""Generates tests/golden/*.npz by IMPORTING the reference's own Python helpers in this container
(the only part of the rasterizer path the reference offers as importable CPU code):

    utils/sh_utils.py:eval_sh                     -> SH colour and, through torch.autograd, its gradient
    utils/graphics_utils.py:getWorld2View2,
                            getProjectionMatrix   -> camera matrices

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


if __name__ == "__main__":
    sh_fixture()
    camera_fixture()
    loss_fixture()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
