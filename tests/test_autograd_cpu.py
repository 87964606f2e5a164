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

# max|oracle - autograd| / max|autograd| per tensor.  The oracle computes in fp32 (the reference's arithmetic), the
# restatement in fp64: measured 2e-7 ... 3e-6 on positions / opacity / SH and up to 2.5e-5 on scales / rotations (the
# conic -> covariance -> quaternion chain amplifies fp32 rounding ~10x); the bars are ~4x those floors.
BARS = dict(dL_dmeans3D=1.5e-5, dL_dopacity=5e-6, dL_dsh=5e-6, dL_dscales=6e-5, dL_drotations=1e-4)


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
        assert e < BARS[k], f"{k}: {e}"


@pytest.mark.parametrize("seed,radius,mod", [(3, 0.8, 1.0), (4, 1.2, 1.3), (6, 0.5, 0.7)])
def test_oracle_backward_equals_autograd_inside_the_cloud(seed, radius, mod):
    """Cameras INSIDE the cloud: many Gaussians sit outside +-1.3 tan(fov), where the reference clamps t.xy in the EWA
    Jacobian (forward.cu:102-107), zeroes the x / y mean gradient through J (x_grad_mul, backward.cu:177-178,265-266)
    and keeps dJ02/dt.z with the clamped value held fixed (backward.cu:174-176); and scale_modifier != 1, where
    dL/dscale lacks the modifier (backward.cu:281-345).  The float64 autograd restatement encodes exactly those
    deviations (torch_splat.render) and nothing else of the hand-written backward -- an independent derivation of the
    branches that the HIP-vs-oracle tests alone could not tell from a shared transcription error."""
    P, W, H, D = 220, 72, 48, 2
    scene = gsr_scene.make_scene(P, -2.4, sh_degree=D, seed=seed)
    cam = gsr_scene.ring_camera(W, H, seed % 8, 8, radius=radius)
    o = util.oracle_forward(scene, cam, D, margin=1e-3, scale_modifier=mod)
    dpix = util.fragile_free_dpix(o, cam, seed=4)
    og = oracle.backward(o, dpix.numpy())
    dt = torch.float64
    leaf = lambda t: t.to(dt).clone().requires_grad_(True)
    means, scales, rots, opac, shs = map(leaf, (scene.means3D, scene.scales, scene.rotations, scene.opacities, scene.shs))
    img, _, _ = torch_splat.render(o, means, scales, rots, opac, shs, scale_modifier=mod)
    assert torch_splat.render.clamp_active >= 5, "the case must exercise the frustum clamp"
    ok = (o["fragile"] == 0).reshape(H, W)
    assert np.abs(img.detach().numpy() - o["color"])[:, ok].max() < 5e-5
    (img * dpix.to(dt)).sum().backward()
    pairs = dict(dL_dmeans3D=means.grad, dL_dscales=scales.grad, dL_drotations=rots.grad, dL_dopacity=opac.grad, dL_dsh=shs.grad)
    for k, g in pairs.items():
        a, b = og[k].reshape(P, -1).astype(np.float64), g.numpy().reshape(P, -1)
        e = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-20))
        assert e < BARS[k], f"{k}: {e}"


def test_oracle_equals_the_independent_tiled_pytorch_splat():
    """oracle/torch_tile_splat.py (the PyTorch-CPU autograd splat of bench.py's cpu_baseline_torch leg) shares nothing with
    the C oracle: its own projection, radius / rectangle arithmetic, (tile, depth) sort and blend, gradients by autograd.
    Same instance count, same radii, same image, gradients within fp32 noise -- with a camera close enough for the frustum
    clamp to be active."""
    from oracle import torch_tile_splat
    scene = gsr_scene.make_scene(3000, -3.0, sh_degree=3, seed=23)
    cam = gsr_scene.ring_camera(200, 120, 1, 8, radius=3.0)
    o = util.oracle_forward(scene, cam, 3, margin=1e-3)
    dpix = util.fragile_free_dpix(o, cam, seed=9)
    og = oracle.backward(o, dpix.numpy())
    leaves = [t.clone().requires_grad_(True) for t in (scene.means3D, scene.scales, scene.rotations, scene.opacities, scene.shs)]
    img, radii, R = torch_tile_splat.render(*leaves, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, scene.bg,
                                            cam.image_width, cam.image_height, cam.tanfovx, cam.tanfovy, 3)
    assert R == o["num_rendered"] and np.array_equal(radii.numpy(), o["radii"])
    ok = (o["fragile"] == 0).reshape(cam.image_height, cam.image_width)
    assert np.abs(img.detach().numpy() - o["color"])[:, ok].max() < 1e-5
    (img * dpix).sum().backward()
    for k, leaf in zip(("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dsh"), leaves):
        a, b = og[k].reshape(3000, -1), leaf.grad.numpy().reshape(3000, -1)
        e = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-20))
        assert e < 1e-4, (k, e)   # measured 9e-7 ... 1.3e-5
