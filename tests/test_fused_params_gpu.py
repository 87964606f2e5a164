"""GPU: leaf-parameter mode (activations folded into the per-Gaussian kernels) and the one-launch Adam
(SURVEY.md 8f-3).  What they replace is PyTorch itself -- the property getters of
scene/gaussian_model.py:114-135 with their autograd backward, and torch.optim.Adam.step() as
gaussian_model.py:243-252 configures it -- so the checker is that PyTorch code run on the same device,
in front of the rasterizer path whose own parity against the oracle is tests/test_parity_gpu.py; the
forward is additionally checked against the CPU oracle fed with the activated values."""
import numpy as np
import pytest
import torch

import gsr_scene
import util

pytestmark = pytest.mark.gpu

GRAD_RTOL = 1e-5  # of the largest gradient magnitude of the tensor, as in test_parity_gpu.py


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _leaves(scene, dev, seed):
    """Raw leaves whose activations are (close to) the scene's tensors; rotations deliberately unnormalised."""
    import gsr_model
    g = torch.Generator().manual_seed(seed)
    rot = scene.rotations * (0.4 + 1.6 * torch.rand(scene.rotations.shape[0], 1, generator=g))
    mk = lambda t: t.to(dev).contiguous().clone().requires_grad_(True)
    return dict(xyz=mk(scene.means3D), features_dc=mk(scene.shs[:, :1, :]), features_rest=mk(scene.shs[:, 1:, :]),
                opacity=mk(gsr_model.inverse_sigmoid(scene.opacities.clamp(1e-4, 1 - 1e-4))), scaling=mk(torch.log(scene.scales)),
                rotation=mk(rot))


def _activated(lv):
    return dict(means3D=lv["xyz"], shs=torch.cat((lv["features_dc"], lv["features_rest"]), dim=1), opacities=torch.sigmoid(lv["opacity"]),
                scales=torch.exp(lv["scaling"]), rotations=torch.nn.functional.normalize(lv["rotation"]))


@pytest.mark.parametrize("P,D,M,size", [(5003, 3, 16, (200, 120)), (3000, 2, 9, (160, 96)), (2500, 0, 1, (128, 80)),
                                        (4100, 1, 16, (176, 100))])
def test_leaf_mode_equals_pytorch_activations_around_the_rasterizer(P, D, M, size):
    _need_gpu()
    from diff_gaussian_rasterization import GaussianRasterizer, _C
    from fused_params import rasterize_leaf_gaussians
    dev = torch.device("cuda:0")
    scene = gsr_scene.make_scene(P, -3.0, sh_degree=3, seed=100 + D, n_coeffs=M)
    cam = gsr_scene.ring_camera(size[0], size[1], 1, 8)
    st = util.hip_settings(scene, cam, D, dev)
    g = torch.Generator().manual_seed(5)
    dpix = torch.randn(3, size[1], size[0], generator=g).to(dev)

    # reference chain: PyTorch activations -> rasterizer -> autograd
    lv_ref = _leaves(scene, dev, 7)
    act = _activated(lv_ref)
    m2_ref = torch.zeros_like(lv_ref["xyz"], requires_grad=True)
    cap = {}
    orig = _C.rasterize_gaussians

    def spy(*a):
        r = orig(*a)
        cap["R"], cap["geom"], cap["binning"], cap["img"] = r[0], r[3], r[4], r[5]
        return r
    _C.rasterize_gaussians = spy
    try:
        color_ref, radii_ref = GaussianRasterizer(st)(means2D=m2_ref, **act)
    finally:
        _C.rasterize_gaussians = orig
    color_ref.backward(dpix)

    # leaf mode
    lv = _leaves(scene, dev, 7)
    m2 = torch.zeros_like(lv["xyz"], requires_grad=True)
    color, radii = rasterize_leaf_gaussians(lv["xyz"], m2, lv["features_dc"], lv["features_rest"], lv["opacity"], lv["scaling"],
                                            lv["rotation"], st)
    color.backward(dpix)
    torch.cuda.synchronize()

    # exp, sigmoid and the 4-element norm are evaluated exactly as PyTorch-ROCm's kernels evaluate them
    # (gsr_device.h gsr_act_*), so the kernels see bit-identical activated values: identical forward
    assert torch.equal(radii, radii_ref)
    assert torch.equal(color.detach(), color_ref.detach())
    for name, a, b in [("xyz", lv["xyz"].grad, lv_ref["xyz"].grad), ("means2D", m2.grad, m2_ref.grad),
                       ("features_dc", lv["features_dc"].grad, lv_ref["features_dc"].grad),
                       ("opacity", lv["opacity"].grad, lv_ref["opacity"].grad), ("scaling", lv["scaling"].grad, lv_ref["scaling"].grad),
                       ("rotation", lv["rotation"].grad, lv_ref["rotation"].grad)] + \
                      ([("features_rest", lv["features_rest"].grad, lv_ref["features_rest"].grad)] if M > 1 else []):
        assert a.shape == b.shape, name
        bar = GRAD_RTOL * max(float(b.abs().max()), 1e-12)
        # only the activation backward (a few fp32 operations per element) may round differently
        assert float((a - b).abs().max()) <= bar, (name, float((a - b).abs().max()), bar)
    if M == 1:
        assert lv["features_rest"].grad is None or lv["features_rest"].grad.numel() == 0

    # forward against the CPU oracle fed with the activated values (PyTorch's, copied to the host)
    scene_act = gsr_scene.Scene(act["means3D"].detach().cpu(), act["scales"].detach().cpu(), act["rotations"].detach().cpu(),
                                act["opacities"].detach().cpu(), act["shs"].detach().cpu(), scene.bg)
    o = util.oracle_forward(scene_act, cam, D)
    np.testing.assert_array_equal(radii.cpu().numpy(), o["radii"])
    ok = (o["fragile"] == 0).reshape(size[1], size[0])
    err = np.abs(color.detach().cpu().numpy() - o["color"].reshape(3, size[1], size[0]))[:, ok]
    assert float(np.mean(err > 1e-5)) < 1e-4, float(err.max())
    # the opacity the blend kernels see is torch.sigmoid's, bit for bit
    splat_ref = util.unpack_state(cap, P, size[0], size[1])["conic_opacity"][:, 3]
    vis = radii_ref.cpu().numpy() > 0
    assert np.array_equal(splat_ref[vis], act["opacities"].detach().cpu().numpy()[vis, 0])


def test_leaf_mode_through_render_and_view_parallel_switch():
    """gaussian_renderer.render(pipe.fused_activations=True) returns the reference's dict; the two-stage C ABI in leaf
    mode without feature gradients hands dL/dRGB over instead (view-parallel mode)."""
    _need_gpu()
    import gsr_model
    import view_parallel
    from gaussian_renderer import render
    dev = torch.device("cuda:0")
    scene = gsr_scene.make_scene(4000, -3.0, sh_degree=3, seed=21)
    cam = gsr_scene.ring_camera(192, 108, 2, 8)
    cam = cam._replace(world_view_transform=cam.world_view_transform.to(dev), full_proj_transform=cam.full_proj_transform.to(dev),
                       camera_center=cam.camera_center.to(dev))
    dpix = torch.randn(3, 108, 192, generator=torch.Generator().manual_seed(2)).to(dev)
    res = {}
    for fused in (False, True):
        pc = gsr_model.GaussianParams.from_activated(scene.means3D, scene.shs, scene.scales, scene.rotations, scene.opacities, device=dev)
        r = render(cam, pc, gsr_model.pipeline_params(fused_activations=fused), scene.bg.to(dev))
        assert set(r) == {"render", "viewspace_points", "visibility_filter", "radii"}
        r["render"].backward(dpix)
        res[fused] = (r["render"].detach(), r["radii"], [p.grad for p in pc.parameters()], r["viewspace_points"].grad)
    assert torch.equal(res[True][1], res[False][1])
    assert float((res[True][0] - res[False][0]).abs().max()) <= 1e-5
    for a, b in zip(res[True][2], res[False][2]):
        assert float((a - b).abs().max()) <= GRAD_RTOL * float(b.abs().max())
    assert float((res[True][3] - res[False][3]).abs().max()) <= GRAD_RTOL * float(res[False][3].abs().max())

    # view-parallel mode of the leaf backward: no feature gradients, dL/dRGB instead, everything else unchanged
    import fused_params
    from diff_gaussian_rasterization import _C
    pc = gsr_model.GaussianParams.from_activated(scene.means3D, scene.shs, scene.scales, scene.rotations, scene.opacities, device=dev)
    st = util.hip_settings(scene, cam, 3, dev)
    with torch.no_grad():
        R, color, radii, geom, binning, img, M, (xyz, fdc, frest, scaling, rotation) = fused_params.leaf_forward(
            pc._xyz, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation, st)
    P = xyz.shape[0]
    f32 = dict(dtype=torch.float32, device=dev)
    out = dict(dL_dmean2D=torch.empty(P, 3, **f32), dL_dmean3D=torch.empty(P, 3, **f32), dL_dopacity=torch.empty(P, 1, **f32),
               dL_dscale=torch.empty(P, 3, **f32), dL_drot=torch.empty(P, 4, **f32), dL_dcolor=torch.empty(P, 3, **f32))
    scratch = torch.empty(_C.lib().gsr_backward_scratch_bytes(P, R), dtype=torch.uint8, device=dev)
    a = fused_params.leaf_backward_args(st, R, M, xyz, fdc, frest, scaling, rotation, radii, geom, binning, img, scratch, dpix)
    _C.set_backward_outputs(a, **out)
    _C.backward_blend(a)
    _C.backward_gaussians(a, 0, P, 0)
    sh_grad = _C.sh_grad_from_views(xyz, cam.camera_center[None], out["dL_dcolor"][None], 3, 16)
    want = torch.cat((res[True][2][1], res[True][2][2]), dim=1)
    assert torch.equal(sh_grad, want)
    assert torch.equal(out["dL_dmean3D"], res[True][2][0])


def _adam_problem(dev, P, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = dict(xyz=(P, 3), f_dc=(P, 1, 3), f_rest=(P, 15, 3), opacity=(P, 1), scaling=(P, 3), rotation=(P, 4))
    lrs = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=2.5e-3 / 20, opacity=0.05, scaling=5e-3, rotation=1e-3)  # arguments/__init__.py:71-81
    params = {k: torch.randn(s, generator=g).to(dev) for k, s in shapes.items()}
    grads = [{k: (torch.randn(s, generator=g) * 10 ** float(torch.randint(-6, 1, (1,), generator=g))).to(dev) for k, s in shapes.items()}
             for _ in range(6)]
    return shapes, lrs, params, grads


def test_fused_adam_matches_torch_adam():
    _need_gpu()
    from fused_params import FusedAdam
    dev = torch.device("cuda:0")
    P = 3001  # numel of several groups is not a multiple of 4: exercises the scalar tail
    shapes, lrs, params, grads = _adam_problem(dev, P, 3)
    pa = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    pb = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    mk = lambda ps: [{"params": [ps[k]], "lr": lrs[k], "name": k} for k in shapes]
    ref = torch.optim.Adam(mk(pa), lr=0.0, eps=1e-15)       # gaussian_model.py:252
    fused = FusedAdam(mk(pb), lr=0.0, eps=1e-15)
    for it, gr in enumerate(grads):
        if it == 3:  # update_learning_rate(): the xyz group's lr changes between steps (gaussian_model.py:254-260)
            for opt in (ref, fused):
                for grp in opt.param_groups:
                    if grp["name"] == "xyz":
                        grp["lr"] = 0.9e-4
        for k in shapes:
            pa[k].grad = gr[k].clone()
            pb[k].grad = gr[k].clone()
        if it == 4:
            pa["opacity"].grad.zero_(); pb["opacity"].grad.zero_()   # zero gradient: momentum still moves the parameter
        ref.step()
        fused.step()
    torch.cuda.synchronize()
    for k in shapes:
        sa, sb = ref.state[pa[k]], fused.state[pb[k]]
        assert int(sb["step"]) == len(grads) == int(sa["step"])
        for name, a, b in (("param", pa[k].data, pb[k].data), ("exp_avg", sa["exp_avg"], sb["exp_avg"]),
                           ("exp_avg_sq", sa["exp_avg_sq"], sb["exp_avg_sq"])):
            # fp32 rounding of a different but equivalent evaluation order, accumulated over 6 steps
            tol = 2e-6 * float(a.abs().max()) if name != "param" else 1e-6 * max(1.0, float(a.abs().max()))
            assert float((a - b).abs().max()) <= tol, (k, name, float((a - b).abs().max()), tol)


def test_fused_adam_state_survives_densification_style_edits_and_visible_only_mode():
    _need_gpu()
    from fused_params import FusedAdam
    dev = torch.device("cuda:0")
    P = 1000
    shapes, lrs, params, grads = _adam_problem(dev, P, 11)
    ps = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    opt = FusedAdam([{"params": [ps[k]], "lr": lrs[k], "name": k} for k in shapes], lr=0.0, eps=1e-15)
    for k in shapes:
        ps[k].grad = grads[0][k].clone()
    opt.step()
    # gaussian_model.py:436-460 cat_tensors_to_optimizer: grow every tensor, zero moments for the new rows
    for grp in opt.param_groups:
        old = grp["params"][0]
        st = opt.state.get(old)
        ext = torch.zeros((50,) + tuple(old.shape[1:]), device=dev)
        st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
        st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
        del opt.state[old]
        grp["params"][0] = torch.nn.Parameter(torch.cat((old.data, ext), dim=0))
        opt.state[grp["params"][0]] = st
    P2 = P + 50
    radii = torch.zeros(P2, dtype=torch.int32, device=dev)
    radii[::3] = 5
    before = {grp["name"]: (grp["params"][0].data.clone(), opt.state[grp["params"][0]]["exp_avg"].clone()) for grp in opt.param_groups}
    dense = {}
    for grp in opt.param_groups:  # what a dense step would give (on copies)
        p = grp["params"][0]
        p.grad = torch.randn(p.shape, generator=torch.Generator().manual_seed(5)).to(dev)
        st = opt.state[p]
        q = torch.nn.Parameter(p.data.clone())
        q.grad = p.grad.clone()
        o2 = FusedAdam([{"params": [q], "lr": grp["lr"]}], lr=0.0, eps=1e-15)
        o2.state[q] = {"step": st["step"], "exp_avg": st["exp_avg"].clone(), "exp_avg_sq": st["exp_avg_sq"].clone()}
        o2.step()
        dense[grp["name"]] = q.data
    opt.step(visible_radii=radii)
    torch.cuda.synchronize()
    vis = radii > 0
    for grp in opt.param_groups:
        p = grp["params"][0]
        assert p.shape[0] == P2 and int(opt.state[p]["step"]) == 2
        assert torch.equal(p.data[~vis], before[grp["name"]][0][~vis]), grp["name"]
        assert torch.equal(opt.state[p]["exp_avg"][~vis], before[grp["name"]][1][~vis]), grp["name"]
        assert torch.equal(p.data[vis], dense[grp["name"]][vis]), grp["name"]


def test_leaf_mode_at_the_headline_size_is_bit_identical_in_the_forward():
    """C3 (1M Gaussians, 1980x1080, degree 3): leaf mode vs PyTorch activations in front of the rasterizer --
    identical radii and image, gradients within the bar, both runs bitwise reproducible."""
    _need_gpu()
    from diff_gaussian_rasterization import GaussianRasterizer
    from fused_params import rasterize_leaf_gaussians
    dev = torch.device("cuda:0")
    P, W, H, D, mu = gsr_scene.CONFIGS["C3"]
    scene = gsr_scene.make_scene(P, mu, D, seed=0)
    cam = gsr_scene.make_camera(W, H)
    st = util.hip_settings(scene, cam, D, dev)
    dpix = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)

    def run(leaf):
        lv = _leaves(scene, dev, 3)
        m2 = torch.zeros_like(lv["xyz"], requires_grad=True)
        if leaf:
            color, radii = rasterize_leaf_gaussians(lv["xyz"], m2, lv["features_dc"], lv["features_rest"], lv["opacity"], lv["scaling"],
                                                    lv["rotation"], st)
        else:
            color, radii = GaussianRasterizer(st)(means2D=m2, **_activated(lv))
        color.backward(dpix)
        return color.detach(), radii, {k: v.grad for k, v in lv.items()}, m2.grad

    c1, r1, g1, s1 = run(True)
    c2, r2, g2, s2 = run(True)
    assert torch.equal(c1, c2) and torch.equal(s1, s2) and all(torch.equal(g1[k], g2[k]) for k in g1)   # deterministic
    c0, r0, g0, s0 = run(False)
    assert torch.equal(r1, r0) and torch.equal(c1, c0)
    assert torch.equal(s1, s0) and torch.equal(g1["xyz"], g0["xyz"])      # no activation in between: same bits
    assert torch.equal(g1["features_dc"], g0["features_dc"]) and torch.equal(g1["features_rest"], g0["features_rest"])
    for k in ("opacity", "scaling", "rotation"):
        bar = GRAD_RTOL * float(g0[k].abs().max())
        assert float((g1[k] - g0[k]).abs().max()) <= bar, (k, float((g1[k] - g0[k]).abs().max()), bar)
