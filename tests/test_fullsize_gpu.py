"""GPU, BASELINE.json's full sizes.  C2 (100k Gaussians, SH deg 3, 1980x1080): complete parity with the
oracle.  C3 (1M, the headline workload) and C5 (6M Gaussians at 4K): size-independent properties --
sorted instance list, ranges that partition it, bitwise run-to-run determinism (the reference's
atomics cannot offer it), exact linearity of the backward in the upstream gradient."""
import numpy as np
import pytest
import torch

import gsr_scene
import util
from test_parity_gpu import check_forward, check_grads

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


NAMES = ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"]


def exclusion_figures(name, scene, cam, D, o, h):
    """What the parity bars leave out, printed next to them: the fragile-pixel fraction, the image error over ALL pixels
    (fragile ones included), and the gradient error when the upstream gradient is NOT zeroed on the fragile pixels.
    A fragile pixel is one where a 1-ulp difference of exp() legitimately flips alpha < 1/255 or T(1-alpha) < 1e-4; the
    pixel then moves by up to alpha*T (~4e-3), which is why these are reported, and bounded loosely, not held to 1e-5."""
    ok = o["fragile"] == 0
    diff = np.abs(h["color"].reshape(3, -1) - o["color"].reshape(3, -1))
    frag = float(1.0 - ok.mean())
    full_max, ok_max = float(diff.max()), float(diff[:, ok].max())
    g = torch.Generator().manual_seed(1)
    dpix = torch.randn(3, cam.image_height, cam.image_width, generator=g)   # unmasked
    hu = util.hip_forward_backward(scene, cam, D, dpix)
    og = util.oracle.backward(o, dpix.numpy())
    rel = {k: float(np.abs(hu["grads"][k].astype(np.float64) - og[k]).max() / max(np.abs(og[k]).max(), 1e-30)) for k in NAMES}
    print(f"[{name}] fragile pixels {frag:.3e} of the image; image max-abs error: all pixels {full_max:.3e}, non-fragile {ok_max:.3e}; "
          "gradient error with UNMASKED dL/dpix (max-abs / max|g|): " + ", ".join(f"{k} {v:.2e}" for k, v in rel.items()))
    assert frag < 5e-3 and full_max < 2e-2 and max(rel.values()) < 2e-3
    return frag, full_max, rel


def test_c2_full_parity_with_oracle():
    _need_gpu()
    scene, cam, D = gsr_scene.make_config("C2")
    o = util.oracle_forward(scene, cam, D)
    dpix = util.fragile_free_dpix(o, cam)
    h = util.hip_forward_backward(scene, cam, D, dpix)
    check_forward(h, o, cam)
    check_grads(h, o, dpix, NAMES)
    exclusion_figures("C2", scene, cam, D, o, h)


def test_c3_full_parity_with_oracle():
    """The headline workload (1M Gaussians, 1980x1080, SH degree 3) against the oracle in full: every integer output
    exact, image and gradients to the same bars as the small cases."""
    _need_gpu()
    scene, cam, D = gsr_scene.make_config("C3")
    o = util.oracle_forward(scene, cam, D)
    dpix = util.fragile_free_dpix(o, cam)
    h = util.hip_forward_backward(scene, cam, D, dpix)
    check_forward(h, o, cam)
    check_grads(h, o, dpix, NAMES)
    exclusion_figures("C3", scene, cam, D, o, h)


def test_c5_integer_state_parity_with_oracle():
    """6M Gaussians at 3840x2160: radii, tiles, sorted instance list, 64-bit keys, tile ranges exact; image 1e-5 and
    n_contrib exact on the non-fragile pixels (check_forward).  Gradients at this size: properties only (below)."""
    _need_gpu()
    scene, cam, D = gsr_scene.make_config("C5")
    o = util.oracle_forward(scene, cam, D)
    h = util.hip_forward_backward(scene, cam, D, None)
    check_forward(h, o, cam)


def _properties(name, check_linearity=True):
    from diff_gaussian_rasterization import GaussianRasterizer, _C
    dev = torch.device("cuda:0")
    scene, cam, D = gsr_scene.make_config(name)
    P, W, H = scene.means3D.shape[0], cam.image_width, cam.image_height
    T = ((W + 15) // 16) * ((H + 15) // 16)
    settings = util.hip_settings(scene, cam, D, dev)
    leaves = dict(means3D=scene.means3D, shs=scene.shs, opacities=scene.opacities, scales=scene.scales,
                  rotations=scene.rotations)
    g = torch.Generator().manual_seed(2)
    dpix = torch.randn(3, H, W, generator=g).to(dev)

    def run(scale):
        p = {k: v.to(dev).clone().requires_grad_(True) for k, v in leaves.items()}
        means2D = torch.zeros_like(p["means3D"], requires_grad=True)
        cap = {}
        orig = _C.rasterize_gaussians

        def spy(*a):
            r = orig(*a)
            cap["R"], cap["geom"], cap["binning"], cap["img"] = r[0], r[3], r[4], r[5]
            return r
        _C.rasterize_gaussians = spy
        try:
            color, radii = GaussianRasterizer(settings)(means2D=means2D, **p)
        finally:
            _C.rasterize_gaussians = orig
        color.backward(dpix * scale)
        torch.cuda.synchronize()
        return color.detach(), radii, {k: v.grad for k, v in p.items()}, means2D.grad, cap

    color, radii, grads, g2d, cap = run(1.0)
    R = cap["R"]
    gl, il, bl = _C.geometry_layout(P), _C.image_layout(W, H), _C.binning_layout(P, R, W, H)
    u32 = lambda buf, off, n: buf[off:off + 4 * n].view(torch.int32)
    tiles_touched = u32(cap["geom"], gl.tiles_touched, P).to(torch.int64)
    assert int(tiles_touched.sum()) == R
    assert torch.equal(tiles_touched > 0, radii > 0)
    tile_keys = u32(cap["binning"], bl.tile_keys, R).to(torch.int64)
    plist = u32(cap["binning"], bl.point_list, R).to(torch.int64)
    assert bool((tile_keys[1:] >= tile_keys[:-1]).all()), "instances sorted by tile"
    assert int(tile_keys.max()) < T
    # depth bits per Gaussian from the depth sort's outputs
    in_alt = int(u32(cap["geom"], gl.status, 4)[2])   # the depth sort's result: in the ping-pong partners after three passes
    perm = u32(cap["geom"], gl.perm_alt if in_alt else gl.perm, P).to(torch.int64)
    skeys = u32(cap["geom"], gl.depth_keys_alt if in_alt else gl.depth_keys, P).to(torch.int64) & 0xFFFFFFFF
    assert bool((skeys[1:] >= skeys[:-1]).all()), "Gaussians sorted by depth"
    assert torch.equal(torch.sort(perm).values, torch.arange(P, device=dev)), "perm is a permutation"
    depth_of = torch.empty(P, dtype=torch.int64, device=dev)
    depth_of[perm] = skeys
    d = depth_of[plist]
    same_tile = tile_keys[1:] == tile_keys[:-1]
    assert bool((d[1:][same_tile] >= d[:-1][same_tile]).all()), "depth-sorted inside every tile"
    tie = same_tile & (d[1:] == d[:-1])
    assert bool((plist[1:][tie] > plist[:-1][tie]).all()), "stable: equal keys keep ascending Gaussian index"
    ranges = u32(cap["img"], il.ranges, 2 * T).view(T, 2).to(torch.int64)
    lens = ranges[:, 1] - ranges[:, 0]
    assert int(lens.sum()) == R and bool((lens >= 0).all())
    counts = torch.bincount(tile_keys, minlength=T)
    assert torch.equal(counts, lens)
    ne = lens > 0
    assert torch.equal(tile_keys[ranges[ne, 0]], torch.nonzero(ne).flatten())
    final_T = cap["img"][il.final_T:il.final_T + 4 * W * H].view(torch.float32)
    assert bool(torch.isfinite(color).all()) and float(final_T.min()) >= 1e-4 * (1 - 1e-6) and float(final_T.max()) <= 1.0
    n_contrib = u32(cap["img"], il.n_contrib, W * H).to(torch.int64).view(H, W)
    ty, tx = torch.arange(H, device=dev) // 16, torch.arange(W, device=dev) // 16
    assert bool((n_contrib <= lens[(ty[:, None] * ((W + 15) // 16) + tx[None, :])]).all())
    for k, v in grads.items():
        assert bool(torch.isfinite(v).all()), k
    assert bool((grads["means3D"][radii == 0] == 0).all()) and bool((grads["shs"][radii == 0] == 0).all())
    assert bool((g2d[:, 2] == 0).all())
    # bitwise determinism (no atomics anywhere on the path)
    color2, radii2, grads2, g2d2, _ = run(1.0)
    assert torch.equal(color, color2) and torch.equal(radii, radii2) and torch.equal(g2d, g2d2)
    for k in grads:
        assert torch.equal(grads[k], grads2[k]), f"{k} differs between two identical runs"
    if check_linearity:  # scaling dL/dpix by a power of two scales every gradient exactly
        _, _, grads4, g2d4, _ = run(4.0)
        assert torch.equal(g2d4, g2d * 4.0)
        for k in grads:
            assert torch.equal(grads4[k], grads[k] * 4.0), k
    return R


def test_c3_headline_workload_properties():
    _need_gpu()
    R = _properties("C3")
    assert abs(R - 9.16e6) / 9.16e6 < 0.01  # R measured with the reference's own preprocess


def test_c5_6m_gaussians_4k_properties():
    _need_gpu()
    R = _properties("C5", check_linearity=False)
    assert abs(R - 7.14e7) / 7.14e7 < 0.02
