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
