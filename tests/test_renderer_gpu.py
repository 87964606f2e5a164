"""GPU: the render() caller (gaussian_renderer/__init__.py:18-124 contract) and its two Python-side
alternates, plus edge cases of the drop-in surface: clones with identical depth (sort stability),
opacities 0 and 1, debug mode and the snapshot dump."""
import os

import numpy as np
import pytest
import torch

import gsr_scene
import util

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _cam_to(cam, dev):
    return cam._replace(world_view_transform=cam.world_view_transform.to(dev), full_proj_transform=cam.full_proj_transform.to(dev),
                        camera_center=cam.camera_center.to(dev))


def test_render_contract_and_python_alternates_agree():
    _need_gpu()
    import gsr_model
    from gaussian_renderer import render
    dev = torch.device("cuda:0")
    scene = gsr_scene.make_scene(5000, -3.2, sh_degree=3, seed=12)
    cam = _cam_to(gsr_scene.ring_camera(240, 136, 2, 8), dev)
    g = torch.Generator().manual_seed(3)
    dpix = torch.randn(3, 136, 240, generator=g).to(dev)
    outs = {}
    for name, kw in (("kernel", {}), ("sh_python", dict(convert_SHs_python=True)), ("cov_python", dict(compute_cov3D_python=True))):
        pc = gsr_model.GaussianParams.from_activated(scene.means3D, scene.shs, scene.scales, scene.rotations, scene.opacities,
                                                     device=dev, active_sh_degree=2)
        r = render(cam, pc, gsr_model.pipeline_params(**kw), scene.bg.to(dev))
        assert set(r) == {"render", "viewspace_points", "visibility_filter", "radii"}
        assert r["render"].shape == (3, 136, 240) and r["radii"].dtype == torch.int32
        assert torch.equal(r["visibility_filter"], r["radii"] > 0)
        r["render"].backward(dpix)
        assert r["viewspace_points"].grad is not None and r["viewspace_points"].grad.shape == (5000, 3)
        assert float(r["viewspace_points"].grad[:, 2].abs().max()) == 0.0
        outs[name] = (r["render"].detach(), [p.grad.clone() for p in pc.parameters()], r["radii"], r["viewspace_points"].grad.clone())
    img0, grads0, radii0, vs0 = outs["kernel"]
    for name in ("sh_python", "cov_python"):
        img, grads, radii, vs = outs[name]
        # different roundings of colour / covariance may move a radius by one pixel or flip a threshold
        assert float((radii != radii0).float().mean()) < 1e-3
        d = (img - img0).abs()
        assert float(d.mean()) < 1e-6 and float((d > 1e-4).float().mean()) < 1e-3, (name, float(d.max()))
        for a, b in zip(grads, grads0):
            assert float((a - b).abs().max()) <= 2e-3 * max(1e-12, float(b.abs().max())), name


def test_clones_with_identical_depth_keep_index_order():
    """Densification clones share xyz (scene/gaussian_model.py:555): equal (tile, depth) keys must
    resolve to ascending Gaussian index exactly like cub's stable sort."""
    _need_gpu()
    from test_parity_gpu import check_forward, check_grads
    base = gsr_scene.make_scene(800, -2.8, sh_degree=1, seed=31)
    rep = lambda t: torch.cat([t, t[:300], t[100:250]]).contiguous()
    g = torch.Generator().manual_seed(1)
    scene = gsr_scene.Scene(rep(base.means3D), rep(base.scales) * (0.5 + torch.rand(1250, 3, generator=g)),
                            rep(base.rotations), rep(base.opacities), rep(base.shs) + 0.1 * torch.randn(1250, 4, 3, generator=g), base.bg)
    cam = gsr_scene.make_camera(160, 96)
    o = util.oracle_forward(scene, cam, 1)
    keys = o["keys"]
    assert int((keys[1:] == keys[:-1]).sum()) > 100, "fixture must contain equal keys"
    dpix = util.fragile_free_dpix(o, cam)
    h = util.hip_forward_backward(scene, cam, 1, dpix)
    check_forward(h, o, cam)
    check_grads(h, o, dpix, ["dL_dmeans3D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"])


def test_opacity_zero_and_one():
    _need_gpu()
    from test_parity_gpu import check_forward, check_grads
    scene = gsr_scene.make_scene(1500, -2.8, sh_degree=0, seed=41)
    op = scene.opacities.clone()
    op[::3] = 0.0      # never reaches 1/255: still binned (parity of point_list), contributes nothing
    op[1::3] = 1.0     # alpha clamps at 0.99 near the centre
    scene = scene._replace(opacities=op)
    cam = gsr_scene.make_camera(128, 80)
    o = util.oracle_forward(scene, cam, 0)
    dpix = util.fragile_free_dpix(o, cam)
    h = util.hip_forward_backward(scene, cam, 0, dpix)
    check_forward(h, o, cam)
    check_grads(h, o, dpix, ["dL_dmeans3D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"])
    assert np.all(h["grads"]["dL_dopacity"][::3] == 0), "a splat that never reaches alpha 1/255 gets no gradient at all"


def test_debug_mode_and_snapshot_dump(tmp_path, monkeypatch):
    _need_gpu()
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    scene = gsr_scene.make_scene(500, -3.0, sh_degree=0, seed=5)
    cam = gsr_scene.make_camera(64, 48)
    o = util.oracle_forward(scene, cam, 0)
    dpix = util.fragile_free_dpix(o, cam)
    h = util.hip_forward_backward(scene, cam, 0, dpix, debug=True)  # sync + check after every stage
    np.testing.assert_array_equal(h["radii"], o["radii"])
    # a failing forward in debug mode writes snapshot_fw.dump into the cwd (reference __init__.py:87-94)
    monkeypatch.chdir(tmp_path)
    cam2 = gsr_scene.ring_camera(32, 32, 1, 8, radius=1.0)
    r = GaussianRasterizer(util.hip_settings(scene, cam2, 0, dev, prefiltered=True, debug=True))
    m = scene.means3D.to(dev)
    with pytest.raises(RuntimeError, match="filtered although prefiltered"):
        r(means3D=m, means2D=m, opacities=scene.opacities.to(dev), shs=scene.shs.to(dev), scales=scene.scales.to(dev),
          rotations=scene.rotations.to(dev))
    assert os.path.exists(tmp_path / "snapshot_fw.dump")
    args = torch.load(tmp_path / "snapshot_fw.dump", weights_only=False)  # our own file, written a moment ago
    assert len(args) == 19 and torch.equal(args[1], scene.means3D)


def test_giant_elongated_splats_are_never_culled_wrongly():
    """Needle-shaped splats hundreds of pixels long: the quadratic form evaluated by the culling test
    cancels by many orders of magnitude far from the centre; the result must still match the oracle."""
    _need_gpu()
    from test_parity_gpu import check_forward, check_grads
    g = torch.Generator().manual_seed(77)
    P = 300
    base = gsr_scene.make_scene(P, -3.0, sh_degree=0, seed=77)
    scales = base.scales.clone()
    scales[:, 0] = torch.exp(torch.randn(P, generator=g) * 0.5 + 0.3)      # ~1.3 world units along one axis
    scales[:, 1:] = torch.exp(torch.randn(P, 2, generator=g) * 0.3 - 5.0)  # ~0.007 across
    scene = base._replace(scales=scales.contiguous(), opacities=torch.full((P, 1), 0.95))
    cam = gsr_scene.make_camera(640, 360)
    o = util.oracle_forward(scene, cam, 0)
    assert int(o["radii"].max()) > 300
    dpix = util.fragile_free_dpix(o, cam)
    h = util.hip_forward_backward(scene, cam, 0, dpix)
    check_forward(h, o, cam)
    check_grads(h, o, dpix, ["dL_dmeans3D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"])


def test_view_parallel_compact_sh_gradient_equals_sum_of_per_view_gradients():
    """Rebuilding the summed SH gradient from 3 floats per Gaussian per view (skip_sh mode of the backward +
    gsr_sh_grad_from_views) gives exactly the fixed-order sum of the per-view dL_dsh the rasterizer itself produces,
    and leaves every other gradient untouched."""
    _need_gpu()
    import view_parallel
    from diff_gaussian_rasterization import GaussianRasterizer, _C
    dev = torch.device("cuda:0")
    for D, M in ((3, 16), (1, 16), (2, 9)):
        scene = gsr_scene.make_scene(6000, -3.0, sh_degree=3, seed=50 + D, n_coeffs=M)
        cams = [gsr_scene.ring_camera(200, 120, k, 8) for k in (0, 3, 5)]
        g = torch.Generator().manual_seed(9)
        dpix = torch.randn(3, 120, 200, generator=g).to(dev)

        def run(cam, skip):
            """plain autograd run, or a direct call of the binding's backward with skip_sh=True on the same state"""
            p = {k: getattr(scene, k).to(dev).clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
            m2 = torch.zeros_like(p["means3D"], requires_grad=True)
            st = util.hip_settings(scene, cam, D, dev)
            if not skip:
                color, _ = GaussianRasterizer(st)(means2D=m2, **p)
                color.backward(dpix)
                return {k: v.grad for k, v in p.items()}, None
            e = torch.empty(0, device=dev)
            R, color, radii, geom, binning, img = _C.rasterize_gaussians(
                st.bg, p["means3D"].detach(), e, p["opacities"].detach(), p["scales"].detach(), p["rotations"].detach(), 1.0, e,
                st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, st.image_height, st.image_width, p["shs"].detach(), D,
                st.campos, False, False)
            d2, drgb, dop, d3, dcov, dsh, dsc, drot = _C.rasterize_gaussians_backward(
                st.bg, p["means3D"].detach(), radii, e, p["scales"].detach(), p["rotations"].detach(), 1.0, e, st.viewmatrix,
                st.projmatrix, st.tanfovx, st.tanfovy, dpix, p["shs"].detach(), D, st.campos, geom, R, binning, img, False,
                lean=True, skip_sh=True)
            assert dsh is None
            return dict(means3D=d3, opacities=dop, scales=dsc, rotations=drot), drgb

        ref = [run(c, False)[0] for c in cams]
        skp = [run(c, True) for c in cams]
        for (g_ref, (g_skip, rgb)) in zip(ref, skp):
            assert rgb is not None and rgb.shape == (6000, 3)
            for k in ("means3D", "opacities", "scales", "rotations"):
                assert torch.equal(g_ref[k], g_skip[k]), k
        want = (ref[0]["shs"] + ref[1]["shs"]) + ref[2]["shs"]
        rgb_all = torch.stack([s[1] for s in skp])
        cam_all = torch.stack([c.camera_center for c in cams]).to(dev)
        got = _C.sh_grad_from_views(scene.means3D.to(dev), cam_all, rgb_all, D, M)
        assert got.shape == (6000, M, 3)
        assert torch.equal(got, want), float((got - want).abs().max())
        # views handed over as strided blocks with a trailer row (the layout the exchange all-gathers): consumed in place
        blocks = torch.full((3, 6001, 3), float("nan"), device=dev)
        blocks[:, :6000] = rgb_all
        assert torch.equal(_C.sh_grad_from_views(scene.means3D.to(dev), cam_all, blocks[:, :6000, :], D, M), want)
        # single-process form of the exchange (world size 1)
        one = view_parallel.exchange_sh_gradient(scene.means3D.to(dev), cam_all[0], rgb_all[0], D, M)
        assert torch.equal(one, ref[0]["shs"])


@pytest.mark.parametrize("sh_mode", ["compact", "allreduce"])
def test_view_parallel_rasterizer_world1_equals_plain_rasterizer(sh_mode):
    """view_parallel.rasterize_view_parallel with a single rank: the part-by-part backward with the exchange buffers
    (two parts, outputs written straight into the buckets) returns bit for bit what the drop-in rasterizer returns."""
    _need_gpu()
    import view_parallel
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    P = 7013
    scene = gsr_scene.make_scene(P, -3.0, sh_degree=3, seed=77)
    cam = gsr_scene.ring_camera(240, 136, 2, 8)
    st = util.hip_settings(scene, cam, 3, dev)
    dpix = torch.randn(3, 136, 240, generator=torch.Generator().manual_seed(3)).to(dev)
    names = ("means3D", "shs", "opacities", "scales", "rotations")

    def leaves():
        return {k: getattr(scene, k).to(dev).clone().requires_grad_(True) for k in names}
    p = leaves()
    m2 = torch.zeros_like(p["means3D"], requires_grad=True)
    color, radii = GaussianRasterizer(st)(means2D=m2, **p)
    color.backward(dpix)
    ex = view_parallel.GradientExchange(P, 16, dev, sh_mode=sh_mode, parts=2)
    assert len(ex.ranges) == 2 and ex.ranges[0][1] % 256 == 0 and sum(c for _, c in ex.ranges) == P
    for _ in range(2):  # the exchange object is reused step after step
        q = leaves()
        n2 = torch.zeros_like(q["means3D"], requires_grad=True)
        color2, radii2 = view_parallel.rasterize_view_parallel(q["means3D"], n2, q["shs"], q["opacities"], q["scales"], q["rotations"], st, ex)
        color2.backward(dpix)
        assert torch.equal(color, color2) and torch.equal(radii, radii2)
        assert torch.equal(m2.grad, n2.grad)
        for k in names:
            assert torch.equal(p[k].grad, q[k].grad), k


def test_densification_statistics_from_the_backward_epilogue():
    """the backward's per-Gaussian kernel accumulates, per view, what train.py:157-159 and
    scene/gaussian_model.py:599-602 compute from (viewspace_points.grad, radii) -- checked against that sequential
    bookkeeping over three views, for the drop-in rasterizer, the leaf-mode one and the view-parallel one."""
    _need_gpu()
    import gsr_model
    import view_parallel
    from diff_gaussian_rasterization import GaussianRasterizer
    from fused_params import rasterize_leaf_gaussians
    dev = torch.device("cuda:0")
    P = 5000
    scene = gsr_scene.make_scene(P, -3.0, sh_degree=3, seed=31)
    cams = [gsr_scene.ring_camera(200, 120, k, 8, radius=2.5) for k in (0, 2, 5)]   # some Gaussians behind the camera
    dpix = torch.randn(3, 120, 200, generator=torch.Generator().manual_seed(6)).to(dev)
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    # the reference's bookkeeping
    max_radii2D, accum, denom = torch.zeros(P, device=dev), torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev)
    for cam in cams:
        p = {k: getattr(scene, k).to(dev).clone().requires_grad_(True) for k in names}
        m2 = torch.zeros_like(p["means3D"], requires_grad=True)
        color, radii = GaussianRasterizer(util.hip_settings(scene, cam, 3, dev))(means2D=m2, **p)
        color.backward(dpix)
        vis = radii > 0
        assert 0 < int(vis.sum()) < P
        max_radii2D[vis] = torch.max(max_radii2D[vis], radii[vis].float())                  # train.py:157
        accum[vis] += torch.norm(m2.grad[vis, :2], dim=-1, keepdim=True)                     # gaussian_model.py:600
        denom[vis] += 1                                                                      # gaussian_model.py:601

    def check(stats):
        stats.sync()
        assert torch.equal(stats.max_radii2D, max_radii2D)
        assert torch.equal(stats.denom, denom)
        assert torch.allclose(stats.xyz_gradient_accum, accum, rtol=2e-6, atol=0)

    stats = view_parallel.DensificationStats(P, dev)
    for cam in cams:
        p = {k: getattr(scene, k).to(dev).clone().requires_grad_(True) for k in names}
        m2 = torch.zeros_like(p["means3D"], requires_grad=True)
        color, _ = GaussianRasterizer(util.hip_settings(scene, cam, 3, dev), densify_stats=stats.kernel_tensors())(means2D=m2, **p)
        color.backward(dpix)
    check(stats)

    stats = view_parallel.DensificationStats(P, dev)
    ex = view_parallel.GradientExchange(P, 16, dev, parts=3)
    for cam in cams:
        p = {k: getattr(scene, k).to(dev).clone().requires_grad_(True) for k in names}
        m2 = torch.zeros_like(p["means3D"], requires_grad=True)
        color, _ = view_parallel.rasterize_view_parallel(p["means3D"], m2, p["shs"], p["opacities"], p["scales"], p["rotations"],
                                                          util.hip_settings(scene, cam, 3, dev), ex, stats.kernel_tensors())
        color.backward(dpix)
    check(stats)

    stats = view_parallel.DensificationStats(P, dev)
    for cam in cams:
        pc = gsr_model.GaussianParams.from_activated(scene.means3D, scene.shs, scene.scales, scene.rotations, scene.opacities, device=dev)
        m2 = torch.zeros_like(pc._xyz, requires_grad=True)
        color, _ = rasterize_leaf_gaussians(pc._xyz, m2, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation,
                                            util.hip_settings(scene, cam, 3, dev), stats.kernel_tensors())
        color.backward(dpix)
    stats.sync()
    assert torch.equal(stats.max_radii2D, max_radii2D) and torch.equal(stats.denom, denom)
    # leaf mode starts from log / logit / raw leaves whose activations reproduce the scene only to fp32 rounding
    assert float((stats.xyz_gradient_accum - accum).abs().max()) <= 1e-4 * float(accum.max())


def test_backward_twice_over_the_same_forward_state():
    """retain_graph: the backward reads the forward's state buffers (and the slot validity bytes that live in
    the sort's ping-pong buffer) a second time -- with a DIFFERENT upstream gradient -- and must give exactly
    what a fresh forward + backward gives for that gradient."""
    _need_gpu()
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    scene = gsr_scene.make_scene(7000, -3.0, sh_degree=3, seed=17)
    cam = gsr_scene.ring_camera(256, 144, 4, 8)
    st = util.hip_settings(scene, cam, 3, dev)
    g = torch.Generator().manual_seed(4)
    d1 = torch.randn(3, 144, 256, generator=g).to(dev)
    d2 = torch.randn(3, 144, 256, generator=g).to(dev) * (torch.rand(3, 144, 256, generator=g).to(dev) > 0.5)   # zeros on half of the pixels
    names = ("means3D", "shs", "opacities", "scales", "rotations")

    def fresh():
        p = {k: getattr(scene, k).to(dev).clone().requires_grad_(True) for k in names}
        color, _ = GaussianRasterizer(st)(means2D=torch.zeros_like(p["means3D"], requires_grad=True), **p)
        return p, color

    p, color = fresh()
    color.backward(d1, retain_graph=True)
    first = {k: p[k].grad.clone() for k in names}
    for k in names:
        p[k].grad = None
    color.backward(d2)
    second = {k: p[k].grad.clone() for k in names}
    q, color2 = fresh()
    color2.backward(d2)
    for k in names:
        assert torch.equal(second[k], q[k].grad), k
        assert not torch.equal(first[k], second[k]), k
