"""GPU, end to end: a short optimisation with every piece built around the path -- leaf-parameter
rasterizer, fused L1+SSIM loss, one-launch Adam (train.py:93-131,179-181 with the 8f rows swapped in) --
next to the same loop written with what the reference uses (PyTorch activations, stock loss_utils code,
torch.optim.Adam) around the drop-in rasterizer.  The loss must fall, and the two loops must stay together."""
import pytest
import torch
import torch.nn.functional as F

import gsr_scene

pytestmark = pytest.mark.gpu

LRS = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=2.5e-3 / 20, opacity=0.05, scaling=5e-3, rotation=1e-3)  # arguments/__init__.py:71-81


def _stock_loss(image, gt, lam=0.2):
    """utils/loss_utils.py:16-63 + train.py:126-127"""
    g1 = torch.exp(-(torch.arange(11, dtype=torch.float32, device=image.device) - 5) ** 2 / (2 * 1.5 ** 2))
    g1 = g1 / g1.sum()
    window = (g1[:, None] @ g1[None, :]).expand(3, 1, 11, 11).contiguous()
    x, y = image[None], gt[None]
    mu1, mu2 = F.conv2d(x, window, padding=5, groups=3), F.conv2d(y, window, padding=5, groups=3)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(x * x, window, padding=5, groups=3) - mu1_sq
    s2 = F.conv2d(y * y, window, padding=5, groups=3) - mu2_sq
    s12 = F.conv2d(x * y, window, padding=5, groups=3) - mu12
    ssim = (((2 * mu12 + 0.01 ** 2) * (2 * s12 + 0.03 ** 2)) / ((mu1_sq + mu2_sq + 0.01 ** 2) * (s1 + s2 + 0.03 ** 2))).mean()
    return (1.0 - lam) * torch.abs(image - gt).mean() + lam * (1.0 - ssim)


def test_fused_training_loop_converges_and_tracks_the_pytorch_loop():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import fused_loss
    import gsr_model
    import util
    from diff_gaussian_rasterization import GaussianRasterizer
    from fused_params import FusedAdam, rasterize_leaf_gaussians
    dev = torch.device("cuda:0")
    W, H, D, P = 240, 136, 3, 6000
    target_scene = gsr_scene.make_scene(P, -2.8, sh_degree=D, seed=1)
    cams = [gsr_scene.ring_camera(W, H, k, 8) for k in (0, 2, 5)]
    settings = [util.hip_settings(target_scene, c, D, dev) for c in cams]
    to = lambda t: t.to(dev)
    with torch.no_grad():
        gts = [GaussianRasterizer(st)(means3D=to(target_scene.means3D), means2D=torch.zeros(P, 3, device=dev), shs=to(target_scene.shs),
                                      opacities=to(target_scene.opacities), scales=to(target_scene.scales),
                                      rotations=to(target_scene.rotations))[0] for st in settings]
    # start: the target with perturbed colours, opacities, sizes and positions
    g = torch.Generator().manual_seed(5)
    start = target_scene._replace(
        means3D=target_scene.means3D + 0.01 * torch.randn(P, 3, generator=g),
        shs=target_scene.shs + 0.3 * torch.randn(P, 16, 3, generator=g),
        opacities=(target_scene.opacities * (0.5 + 0.5 * torch.rand(P, 1, generator=g))).clamp(0.02, 0.98),
        scales=target_scene.scales * torch.exp(0.3 * torch.randn(P, 3, generator=g)))
    names = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity")

    def make(opt_cls):
        pc = gsr_model.GaussianParams.from_activated(start.means3D, start.shs, start.scales, start.rotations, start.opacities, device=dev)
        groups = [{"params": [torch.nn.Parameter(p.detach())], "lr": LRS[n] * 10, "name": n} for n, p in zip(names, pc.parameters())]
        pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity = (gr["params"][0] for gr in groups)
        return pc, opt_cls(groups, lr=0.0, eps=1e-15)

    def run(fused, iters=40):
        pc, opt = make(FusedAdam if fused else torch.optim.Adam)
        losses, first_grads = [], None
        for it in range(iters):
            v = it % len(cams)
            m2 = torch.zeros_like(pc._xyz, requires_grad=True)
            if fused:
                image, radii = rasterize_leaf_gaussians(pc._xyz, m2, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling,
                                                        pc._rotation, settings[v])
                loss = fused_loss.l1_ssim_loss(image, gts[v], 0.2)
            else:
                image, radii = GaussianRasterizer(settings[v])(means3D=pc.get_xyz, means2D=m2, shs=pc.get_features, opacities=pc.get_opacity,
                                                               scales=pc.get_scaling, rotations=pc.get_rotation)
                loss = _stock_loss(image, gts[v])
            loss.backward()
            assert m2.grad is not None and float(m2.grad[:, 2].abs().max()) == 0.0   # the densification statistic's carrier
            if it == 0:   # identical parameters on both sides: the gradients of this one step are compared directly
                first_grads = [gr["params"][0].grad.detach().clone() for gr in opt.param_groups]
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(float(loss.detach()))
        return losses, [p.detach().clone() for p in pc.parameters()], first_grads

    lf, pf, gf = run(True)
    lp, pp, gp = run(False)
    # (1) one identical step: fused leaf path (activations + their backward inside the kernels, fused loss) against PyTorch
    # activations + stock loss around the drop-in rasterizer -- same parameters, so any difference here is arithmetic, not
    # optimiser dynamics.  What is left for (2) below is Adam's amplification alone.
    # Measured: xyz 2.9e-5, the others below (the two sides differ in the loss kernel too: dL/dpix agrees to ~1e-6 and the
    # position gradients are cancellation-heavy sums of it over every covered pixel) -- bar 5e-5 of the largest element.
    step0 = {n: float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30) for n, a, b in zip(names, gf, gp)}
    print("step-0 gradient, max|diff| / max|g|:", ", ".join(f"{n} {e:.2e}" for n, e in step0.items()))
    assert max(step0.values()) <= 5e-5, step0
    first, last = sum(lf[:3]) / 3, sum(lf[-3:]) / 3
    assert last < 0.6 * first, (first, last)
    # same mathematics, different fp32 evaluation order; Adam divides by sqrt(v) ~ |g| in its first steps,
    # which turns last-bit differences of tiny gradients into full-size steps of those elements, so the
    # trajectories separate slowly: tight at the start, loose at the end
    rel = [abs(a - b) / abs(b) for a, b in zip(lf, lp)]
    print("loss fused  ", [round(v, 5) for v in lf[::4]])
    print("loss pytorch", [round(v, 5) for v in lp[::4]])
    print("rel diff    ", [f"{v:.1e}" for v in rel[::4]])
    # (2) the trajectories: Adam's normalisation turns last-bit differences of tiny gradients into full-size steps of those
    # elements, so the two loops drift apart slowly -- tight at the start, within 3 % over the first half, and bounded at
    # the end of the 40 iterations
    assert max(rel[:8]) <= 1e-3, rel[:8]
    assert max(rel[:20]) <= 3e-2, rel[:20]
    assert max(rel) <= 1e-1, rel
    for n, a, b in zip(names, pf, pp):
        d = float((a - b).abs().mean()) / max(float(b.abs().mean()), 1e-12)
        print(n, f"mean |diff| / mean |param| = {d:.2e}")
        assert d < 2e-2, (n, d)
