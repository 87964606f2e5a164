"""CPU tests (no GPU): pin the oracle against the golden vectors generated from the reference's own
importable Python (tests/golden/make_golden.py), and check the invariants of SURVEY.md 8c pin 5."""
import os

import numpy as np
import torch

import gsr_scene
import util
from oracle import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sh_colour_matches_reference_eval_sh():
    """forward.cu:21-81 restatement == clamp_min(eval_sh(...) + 0.5, 0) (utils/sh_utils.py:57-112)."""
    g = np.load(os.path.join(GOLD, "sh_golden.npz"))
    for deg in range(4):
        rgb, clamped = oracle.sh_forward(deg, g["pos"], g["campos"], g["shs"])
        np.testing.assert_allclose(rgb, g[f"rgb_deg{deg}"], rtol=0, atol=2e-6)
        raw = g[f"raw_deg{deg}"]
        sure = np.abs(raw) > 1e-5
        assert np.array_equal((clamped != 0)[sure], (raw < 0)[sure])
        assert (clamped != 0).any(), "fixture must exercise the clamp"


def test_sh_backward_matches_autograd_of_reference_eval_sh():
    """backward.cu:20-139 restatement == torch.autograd through the reference's eval_sh."""
    g = np.load(os.path.join(GOLD, "sh_golden.npz"))
    for deg in range(4):
        _, clamped = oracle.sh_forward(deg, g["pos"], g["campos"], g["shs"])
        dmean, dsh = oracle.sh_backward(deg, g["pos"], g["campos"], g["shs"], clamped, g["dL_dcolor"])
        np.testing.assert_allclose(dsh, g[f"dL_dsh_deg{deg}"], rtol=1e-5, atol=2e-6)
        ref = g[f"dL_dpos_deg{deg}"]
        np.testing.assert_allclose(dmean, ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()))


def test_camera_matrices_match_reference_graphics_utils():
    g = np.load(os.path.join(GOLD, "camera_golden.npz"))
    for k in range(g["R"].shape[0]):
        np.testing.assert_array_equal(gsr_scene.world_to_view(g["R"][k], g["T"][k]), g["w2v"][k])
        p = gsr_scene.projection_matrix(0.01, 100.0, float(g["fovx"][k]), float(g["fovy"][k])).numpy()
        np.testing.assert_array_equal(p, g["proj"][k])


def test_camera_composition_matches_reference_camera_class():
    """scene/cameras.py:57-61 (world_view_transform, projection, full_proj_transform, camera_center), produced by the
    reference's own Camera class: gsr_scene.make_camera composes the same four tensors bit for bit."""
    g = np.load(os.path.join(GOLD, "camera_class_golden.npz"))
    for k in range(g["R"].shape[0]):
        fovx, fovy = float(g["fovx"][k]), float(g["fovy"][k])
        wvt = torch.tensor(gsr_scene.world_to_view(g["R"][k], g["T"][k])).transpose(0, 1)
        proj = gsr_scene.projection_matrix(0.01, 100.0, fovx, fovy).transpose(0, 1)
        np.testing.assert_array_equal(wvt.numpy(), g["world_view_transform"][k])
        np.testing.assert_array_equal(proj.numpy(), g["projection_matrix"][k])
        full = wvt.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
        np.testing.assert_array_equal(full.numpy(), g["full_proj_transform"][k])
        np.testing.assert_array_equal(wvt.inverse()[3, :3].numpy(), g["camera_center"][k])
    # make_camera itself (tanfovy is tied to the aspect ratio there): same composition on its own FoV pair
    cam = gsr_scene.make_camera(64, 48, fovx=float(g["fovx"][0]), R=g["R"][0], T=g["T"][0])
    np.testing.assert_array_equal(cam.world_view_transform.numpy(), g["world_view_transform"][0])
    proj = gsr_scene.projection_matrix(0.01, 100.0, cam.FoVx, cam.FoVy).transpose(0, 1)
    np.testing.assert_array_equal(cam.full_proj_transform.numpy(), cam.world_view_transform.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0).numpy())
    np.testing.assert_array_equal(cam.camera_center.numpy(), g["camera_center"][0])


def test_cov3d_matches_reference_general_utils():
    """computeCov3D (forward.cu:146-180) as restated by the oracle, and the product's Python alternate
    (gsr_model.build_covariance_from_scaling_rotation, the compute_cov3D_python branch), against the covariance the
    reference's own utils/general_utils.py functions produce (scene/gaussian_model.py:32-37), unit quaternions."""
    import gsr_model
    g = np.load(os.path.join(GOLD, "cov3d_golden.npz"))
    scaling, q = torch.from_numpy(g["scaling"]), torch.from_numpy(g["rotation_unit"])
    n = scaling.shape[0]
    R = gsr_model.build_rotation(torch.from_numpy(g["rotation_raw"])).numpy()
    np.testing.assert_allclose(R, g["R_of_raw"], rtol=0, atol=1e-6)
    for mod in (1.0, 1.7):
        want = g[f"cov_mod{mod}"]
        scale_of = np.abs(want).max(axis=1, keepdims=True)
        got = gsr_model.build_covariance_from_scaling_rotation(scaling, mod, q).numpy()
        assert np.abs(got - want).max() <= 0 or (np.abs(got - want) / scale_of).max() < 2e-6
        # the oracle's computeCov3D through its preprocess (every Gaussian in front of the camera)
        means = np.zeros((n, 3), np.float32)
        means[:, 0] = np.linspace(-0.5, 0.5, n)
        cam = gsr_scene.make_camera(64, 48)
        o = oracle.forward(means, np.full((n, 1), 0.5, np.float32), cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(),
                           cam.camera_center.numpy(), np.zeros(3, np.float32), 64, 48, cam.tanfovx, cam.tanfovy, 0,
                           shs=np.zeros((n, 1, 3), np.float32), scales=g["scaling"], rotations=g["rotation_unit"], scale_modifier=mod)
        assert (np.abs(o["cov3D"] - want) / scale_of).max() < 2e-6


def test_product_eval_sh_matches_reference_eval_sh():
    """gsr_model.eval_sh (the convert_SHs_python alternate of the product) against the reference's eval_sh outputs."""
    import gsr_model
    g = np.load(os.path.join(GOLD, "sh_golden.npz"))
    pos, campos, shs = torch.from_numpy(g["pos"]), torch.from_numpy(g["campos"]), torch.from_numpy(g["shs"])
    d = pos - campos
    d = d / d.norm(dim=1, keepdim=True)
    for deg in range(4):
        raw = gsr_model.eval_sh(deg, shs.transpose(1, 2), d) + 0.5
        np.testing.assert_allclose(raw.numpy(), g[f"raw_deg{deg}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(torch.clamp_min(raw, 0).numpy(), g[f"rgb_deg{deg}"], rtol=0, atol=2e-6)


def test_get_higher_msb():
    # rasterizer_impl.cu:37-52; values quoted in SURVEY.md 8a-10: bit = 9 / 14 / 14 / 15 for C1 / C2 / C3 / C5
    assert oracle.get_higher_msb(256) == 9
    assert oracle.get_higher_msb(8432) == 14
    assert oracle.get_higher_msb(32400) == 15
    for n in (1, 2, 3, 255, 257, 65535, 65536):
        b = oracle.get_higher_msb(n)
        assert (n >> b) == 0 and (b == 0 or (n >> (b - 1)) != 0)


def _small():
    scene = gsr_scene.make_scene(3000, -3.0, sh_degree=2, seed=4)
    cam = gsr_scene.make_camera(200, 120)
    return scene, cam, util.oracle_forward(scene, cam, 2)


def test_forward_invariants():
    scene, cam, o = _small()
    R = o["num_rendered"]
    assert R == int(o["tiles_touched"].sum()) == int(o["point_offsets"][-1])
    keys = o["keys"]
    assert np.all(keys[1:] >= keys[:-1]), "keys sorted"
    # stable: equal keys keep ascending Gaussian index
    same = keys[1:] == keys[:-1]
    assert np.all(o["point_list"][1:][same] > o["point_list"][:-1][same])
    rng = o["ranges"].astype(np.int64)
    nonempty = rng[:, 1] > rng[:, 0]
    assert int((rng[:, 1] - rng[:, 0]).sum()) == R
    tiles_of_keys = (keys >> np.uint64(32)).astype(np.int64)
    for t in np.nonzero(nonempty)[0][:50]:
        assert np.all(tiles_of_keys[rng[t, 0]:rng[t, 1]] == t)
    assert np.all(o["final_T"] >= 1e-4 * (1 - 1e-6)) and np.all(o["final_T"] <= 1.0)
    lens = (rng[:, 1] - rng[:, 0])
    gx = (cam.image_width + 15) // 16
    ys, xs = np.divmod(np.arange(cam.image_width * cam.image_height), cam.image_width)
    assert np.all(o["n_contrib"] <= lens[(ys // 16) * gx + xs // 16])
    vis = o["radii"] > 0
    assert np.all(o["depths"][vis] > 0.2)
    assert np.all(o["tiles_touched"][~vis] == 0)


def test_statistics_match_survey_measurement_of_reference_preprocess():
    """SURVEY.md 8d records V and R measured with the reference's own preprocessCUDA for the C1
    workload definition (C++ mt19937 draws): V = 9 994, R = 58 855.  Same distribution, different
    generator: agree within sampling noise."""
    scene, cam, D = gsr_scene.make_config("C1")
    o = util.oracle_forward(scene, cam, D)
    V = int((o["radii"] > 0).sum())
    assert abs(V - 9994) <= 10
    assert abs(o["num_rendered"] - 58855) / 58855 < 0.04


def test_backward_is_gradient_of_forward_where_smooth():
    """Central finite differences of the oracle's own forward in float64-free form are not valid at
    the alpha cut-off (SURVEY.md 8c pin 3), so use the one place the analytic backward is an exact
    derivative of a smooth function of the inputs: colours.  dL/dcolors and dL/dopacity-through-
    colour are linear in the colour, so image = sum_i colour_i * w_i + T*bg with w independent of
    colour: check linearity and dL/dcolour = sum_pix w * dL/dpix."""
    scene = gsr_scene.make_scene(400, -2.5, sh_degree=0, seed=9)
    cam = gsr_scene.make_camera(96, 64)
    g = torch.Generator().manual_seed(3)
    c0 = torch.rand(400, 3, generator=g)
    c1 = torch.rand(400, 3, generator=g)
    dpix = torch.randn(3, 64, 96, generator=g)
    o0 = util.oracle_forward(scene, cam, 0, colors_precomp=c0, use_sh=False)
    o1 = util.oracle_forward(scene, cam, 0, colors_precomp=c1, use_sh=False)
    g0 = oracle.backward(o0, dpix.numpy())
    # linear in colour: L(c1) - L(c0) == <dL/dc, c1 - c0>
    L0 = float((o0["color"].astype(np.float64) * dpix.numpy()).sum())
    L1 = float((o1["color"].astype(np.float64) * dpix.numpy()).sum())
    lin = float((g0["blend64"]["colors"] * (c1 - c0).numpy().astype(np.float64)).sum())
    assert abs((L1 - L0) - lin) <= 1e-4 * max(1.0, abs(L1 - L0))


def test_edge_cases_cpu():
    cam = gsr_scene.make_camera(64, 48)
    scene = gsr_scene.make_scene(10, -3.0, sh_degree=0, seed=1)
    e = gsr_scene.Scene(scene.means3D[:0], scene.scales[:0], scene.rotations[:0], scene.opacities[:0], scene.shs[:0], scene.bg)
    o = util.oracle_forward(e, cam, 0)
    assert o["num_rendered"] == 0 and float(np.abs(o["color"]).max()) == 0.0  # rasterize_points.cu:94,129
    behind = gsr_scene.Scene(scene.means3D - torch.tensor([0.0, 0.0, 100.0]), scene.scales, scene.rotations,
                             scene.opacities, scene.shs, scene.bg)
    o = util.oracle_forward(behind, cam, 0)
    assert o["num_rendered"] == 0
    np.testing.assert_array_equal(o["color"], np.broadcast_to(scene.bg.numpy()[:, None, None], (3, 48, 64)))
