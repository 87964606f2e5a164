"""CPU: the oracle's analytic backward (the formulas of backward.cu that the HIP kernels also
implement) against torch.autograd of an independent float64 restatement of the forward
(tests/torch_splat.py).  This is SURVEY.md 8c pin 2: it pins the hand-written gradient formulas to
the true derivative, with the reference's deliberate deviations encoded (Appendix A item 14)."""
import numpy as np
import pytest
import torch

import gsr_scene
import torch_splat
import util
from oracle import oracle


@pytest.mark.parametrize("P,W,H,D,mu,seed", [(150, 64, 48, 3, -2.2, 2), (300, 80, 48, 1, -2.6, 8), (60, 48, 32, 0, -1.6, 5)])
def test_oracle_backward_equals_autograd(P, W, H, D, mu, seed):
    scene = gsr_scene.make_scene(P, mu, sh_degree=D, seed=seed)
    # keep everything well inside the frustum so the +-1.3 tan(fov) clamp is inactive
    scene = scene._replace(means3D=(scene.means3D * 0.6).contiguous())
    cam = gsr_scene.make_camera(W, H)
    o = util.oracle_forward(scene, cam, D, margin=1e-3)
    dpix = util.fragile_free_dpix(o, cam, seed=4)
    og = oracle.backward(o, dpix.numpy())

    dt = torch.float64
    leaf = lambda t: t.to(dt).clone().requires_grad_(True)
    means, scales, rots, opac, shs = map(leaf, (scene.means3D, scene.scales, scene.rotations, scene.opacities, scene.shs))
    img, T_final, _ = torch_splat.render(o, means, scales, rots, opac, shs)
    ok = (o["fragile"] == 0).reshape(H, W)
    # forward agreement first (fp32 oracle vs fp64 restatement)
    err = np.abs(img.detach().numpy() - o["color"])[:, ok].max()
    assert err < 5e-5, err
    (img * dpix.to(dt)).sum().backward()

    def nerr(a, b):
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-20))

    pairs = dict(dL_dmeans3D=means.grad, dL_dscales=scales.grad, dL_drotations=rots.grad, dL_dopacity=opac.grad,
                 dL_dsh=shs.grad)
    for k, g in pairs.items():
        e = nerr(og[k].reshape(P, -1).astype(np.float64), g.numpy().reshape(P, -1))
        assert e < 2e-4, f"{k}: {e}"
