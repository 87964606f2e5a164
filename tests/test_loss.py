"""The fused L1 + SSIM loss.  CPU: the float64 oracle against golden vectors produced
by the reference's own utils/loss_utils.py (l1_loss, ssim + torch.autograd).  GPU: the HIP kernels
against the oracle and the golden vectors.  Bars: loss values 1e-5 absolute; gradient
max|d| <= 1e-5 * max|g|, or three times the distance between the reference's own fp32 result and
exact arithmetic where that is larger (its E[x^2] - mu^2 cancels in fp32 on smooth images)."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_golden.npz")
CASES = ("noise", "smooth", "equal")


def _nerr(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(np.abs(b).max(), 1e-30))


def test_oracle_matches_reference_loss_utils():
    g = np.load(GOLD)
    for name in CASES:
        loss, l1, ss, grad = oracle.l1_ssim(g[f"{name}_img"], g[f"{name}_gt"], 0.2)
        ref = g[f"{name}_vals"]
        assert abs(loss - ref[0]) < 1e-5 and abs(l1 - ref[1]) < 1e-6 and abs(ss - ref[2]) < 1e-5, (name, loss, l1, ss, ref)
        rg = g[f"{name}_grad"]
        if name == "equal":  # SSIM is at its maximum and L1 at its kink: the exact gradient is 0
            assert np.abs(grad).max() < 1e-12 and np.abs(rg).max() < 1e-7
        else:
            assert _nerr(grad, rg) < 5e-4, (name, _nerr(grad, rg))


@pytest.mark.gpu
def test_hip_loss_matches_oracle_and_reference():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import fused_loss
    dev = torch.device("cuda:0")
    g = np.load(GOLD)
    for name in CASES:
        img = torch.from_numpy(g[f"{name}_img"]).to(dev).requires_grad_(True)
        gt = torch.from_numpy(g[f"{name}_gt"]).to(dev)
        loss, l1, ss = fused_loss.l1_ssim_loss_terms(img, gt, 0.2)
        (loss * 3.0).backward()  # a non-unit upstream factor
        o_loss, o_l1, o_ss, o_grad = oracle.l1_ssim(g[f"{name}_img"], g[f"{name}_gt"], 0.2)
        assert abs(loss.item() - o_loss) < 1e-5 and abs(l1.item() - o_l1) < 1e-6 and abs(ss.item() - o_ss) < 1e-5
        ref = g[f"{name}_vals"]
        assert abs(loss.item() - ref[0]) < 1e-5
        got = img.grad.cpu().numpy() / 3.0
        ref_noise = _nerr(g[f"{name}_grad"], o_grad) if name != "equal" else 0.0
        if name == "equal":
            assert np.abs(got).max() < 1e-7
        else:
            assert _nerr(got, o_grad) <= max(1e-5, 3.0 * ref_noise), (name, _nerr(got, o_grad), ref_noise)
    # a full-size image: finite, matches a stock-PyTorch evaluation of the same formula
    gen = torch.Generator().manual_seed(3)
    gt = torch.rand(3, 1080, 1980, generator=gen).to(dev)
    img = (gt + 0.1 * torch.randn(3, 1080, 1980, generator=gen).to(dev)).clamp(0, 1).requires_grad_(True)
    loss = fused_loss.l1_ssim_loss(img, gt, 0.2)
    loss.backward()
    assert torch.isfinite(img.grad).all() and 0.0 < loss.item() < 1.0
    taps = torch.tensor([np.exp(-(k - 5) ** 2 / (2 * 1.5 ** 2)) for k in range(11)], dtype=torch.float32)
    taps = taps / taps.sum()
    w = (taps[:, None] @ taps[None, :]).to(dev).expand(3, 1, 11, 11).contiguous()
    x = img.detach().clone().requires_grad_(True)
    F = torch.nn.functional
    mu1, mu2 = F.conv2d(x[None], w, padding=5, groups=3), F.conv2d(gt[None], w, padding=5, groups=3)
    s11 = F.conv2d((x * x)[None], w, padding=5, groups=3) - mu1 ** 2
    s22 = F.conv2d((gt * gt)[None], w, padding=5, groups=3) - mu2 ** 2
    s12 = F.conv2d((x * gt)[None], w, padding=5, groups=3) - mu1 * mu2
    ssim = (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 ** 2 + mu2 ** 2 + 1e-4) * (s11 + s22 + 9e-4))).mean()
    ref_loss = 0.8 * (x - gt).abs().mean() + 0.2 * (1 - ssim)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    assert float((img.grad - x.grad).abs().max()) <= 2e-3 * float(x.grad.abs().max())
