"""GPU: randomised configurations against the oracle, same bars as test_parity_gpu.py.  Each seed draws
the image size (never a multiple of the tile), field of view, camera pose -- including cameras INSIDE
the cloud, where Gaussians sit behind the camera, cross the near plane, trip the 1.3*tan(fov) frustum
clamp of computeCov2D (forward.cu:94-99, gradient masks backward.cu:181-186) and reach screen radii of
thousands of pixels -- SH degree, scale spread, scale_modifier and background."""
import math

import numpy as np
import pytest
import torch

import gsr_scene
import util
from test_parity_gpu import check_forward, check_grads

pytestmark = pytest.mark.gpu


def _look_at(eye, target):
    fwd = target - eye
    fwd = fwd / np.linalg.norm(fwd)
    up = np.array([0.0, 1.0, 0.0])
    if abs(fwd @ up) > 0.95:
        up = np.array([1.0, 0.0, 0.0])
    right = np.cross(up, fwd)
    right /= np.linalg.norm(right)
    up2 = np.cross(fwd, right)
    R = np.stack([right, up2, fwd], axis=1)   # camera-to-world rotation, as Camera.R in scene/cameras.py
    T = -R.T @ eye
    return R, T


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_configuration_matches_oracle(seed):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = np.random.default_rng(1000 + seed)
    W = int(r.integers(40, 230))
    H = int(r.integers(30, 150))
    if W % 16 == 0:
        W += 3
    if H % 16 == 0:
        H += 5
    D = int(r.integers(0, 4))
    inside = seed % 2 == 1
    P = int(r.integers(1000, 3000)) if inside else int(r.integers(2000, 8000))
    mu = float(r.uniform(-4.0, -2.0))
    scene = gsr_scene.make_scene(P, mu, sh_degree=D, seed=seed + 500)
    bg = torch.tensor(r.uniform(0, 1, 3), dtype=torch.float32)
    scene = scene._replace(bg=bg)
    eye = r.normal(size=3)
    eye = eye / np.linalg.norm(eye) * (float(r.uniform(0.2, 1.4)) if inside else float(r.uniform(2.5, 6.0)))
    R, T = _look_at(eye, r.uniform(-0.5, 0.5, 3))
    cam = gsr_scene.make_camera(W, H, fovx=float(r.uniform(0.4, 1.9)), R=R, T=T)
    scale_modifier = float(r.choice([1.0, 1.0, 0.6, 1.7]))
    o = util.oracle_forward(scene, cam, D, scale_modifier=scale_modifier)
    assert o["num_rendered"] > 0
    if inside:   # the configuration must really exercise what it is for
        assert int((o["radii"] == 0).sum()) > 0, "no Gaussian culled behind / near the camera"
    dpix = util.fragile_free_dpix(o, cam, seed=seed)
    h = util.hip_forward_backward(scene, cam, D, dpix, scale_modifier=scale_modifier)
    check_forward(h, o, cam)
    check_grads(h, o, dpix, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"])
    print(f"seed {seed}: {W}x{H} D{D} P{P} inside={inside} R={o['num_rendered']} max radius {int(o['radii'].max())} "
          f"fov {math.degrees(2 * math.atan(cam.tanfovx)):.0f} deg")


@pytest.mark.parametrize("P,W,H,D", [(1, 1, 1, 0), (1, 33, 17, 3), (5, 1, 40, 1), (64, 16, 16, 2), (65, 17, 1, 0), (257, 31, 47, 3)])
def test_tiny_and_degenerate_sizes_match_oracle(P, W, H, D):
    """One Gaussian, one pixel, one-pixel-wide images, exactly one tile, P just over a wave / a workgroup."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    scene = gsr_scene.make_scene(P, -1.5, sh_degree=D, seed=P + W)
    scene = scene._replace(means3D=scene.means3D * 0.3)   # keep them in front of the camera and on screen
    cam = gsr_scene.make_camera(W, H)
    o = util.oracle_forward(scene, cam, D)
    dpix = util.fragile_free_dpix(o, cam, seed=3)
    h = util.hip_forward_backward(scene, cam, D, dpix)
    check_forward(h, o, cam)
    if o["num_rendered"] > 0:
        check_grads(h, o, dpix, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"])


@pytest.mark.parametrize("seed", list(range(6)))
def test_random_heavy_tile_scenes_match_oracle_split_or_not(seed):
    """Randomised scenes with a dense, mostly low-opacity clump: a few tiles carry lists many times the mean and are walked
    thousands of instances deep -- the scenes on which heavy tiles are dispatched as band waves (forward) and depth segments
    (backward).  Image sizes include partial edge tiles and images of fewer than 64 tiles (no split possible); the clump's
    size and opacity decide how many tiles are split, how deep they are cut and how coarse the segments get.  Against the
    oracle with the usual bars; the forward must also be bit-identical to the run with GSR_DEBUG_NO_SPLIT."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from diff_gaussian_rasterization import GaussianRasterizer
    from diff_gaussian_rasterization import _C
    r = np.random.default_rng(7000 + seed)
    W, H = [(330, 210), (97, 61), (500, 140), (257, 259), (640, 360), (120, 300)][seed]
    D = int(r.integers(0, 4))
    P = int(r.integers(20_000, 70_000))
    scene = gsr_scene.make_scene(P, float(r.uniform(-4.2, -3.2)), sh_degree=D, seed=900 + seed)
    g = torch.Generator().manual_seed(50 + seed)
    nb = int(P * float(r.uniform(0.5, 0.85)))
    means = scene.means3D.clone()
    centre = torch.tensor(r.uniform(-0.6, 0.6, 3), dtype=torch.float32)
    means[:nb] = centre + torch.randn(nb, 3, generator=g) * torch.tensor([float(r.uniform(0.05, 0.3)), float(r.uniform(0.05, 0.2)), 0.3])
    opac = scene.opacities.clone()
    opac[:nb] = torch.sigmoid(torch.randn(nb, 1, generator=g) + float(r.uniform(-4.5, -2.0)))
    scene = scene._replace(means3D=means.contiguous(), opacities=opac.contiguous())
    cam = gsr_scene.make_camera(W, H, fovx=float(r.uniform(0.7, 1.3)))
    o = util.oracle_forward(scene, cam, D)
    dpix = util.fragile_free_dpix(o, cam, seed=seed)
    h = util.hip_forward_backward(scene, cam, D, dpix)
    check_forward(h, o, cam)
    check_grads(h, o, dpix, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"], label=f"heavy_fuzz_{seed}")
    # the unsplit run: same image and state bit for bit
    dev = torch.device("cuda:0")
    st = util.hip_settings(scene, cam, D, dev)._replace(debug=_C.DEBUG_NO_SPLIT)
    t = {k: getattr(scene, k).to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    with torch.no_grad():
        color, radii = GaussianRasterizer(st)(means2D=torch.zeros_like(t["means3D"]), **t)
    assert np.array_equal(color.cpu().numpy(), h["color"]) and np.array_equal(radii.cpu().numpy(), h["radii"])
    T = ((W + 15) // 16) * ((H + 15) // 16)
    lens = (h["ranges"][:, 1].astype(np.int64) - h["ranges"][:, 0])
    print(f"seed {seed}: {W}x{H} ({T} tiles) D{D} P{P} R={o['num_rendered']} longest list {int(lens.max())} mean {int(lens.mean())}")


def _binning_pair(scene, cam, D, dpix, label):
    """One scene through both binning paths -- column pairs (csrc/tilebin.hip, the default up to 256 x 256 tiles) and instance
    emission + tile sort (GSR_DEBUG_TILE_SORT, the path of larger images): both against the oracle, and bit-identical to each
    other in every output and in the sorted instance list and the ranges."""
    from diff_gaussian_rasterization import _C
    o = util.oracle_forward(scene, cam, D)
    if dpix is None:
        dpix = util.fragile_free_dpix(o, cam, seed=5)
    d = util.hip_forward_backward(scene, cam, D, dpix)   # the default: column pairs, tiles a splat provably misses left out (gsr_rect_trim.h)
    a = util.hip_forward_backward(scene, cam, D, dpix, debug=_C.DEBUG_NO_TRIM)   # column pairs, every tile of every rectangle
    b = util.hip_forward_backward(scene, cam, D, dpix, debug=_C.DEBUG_TILE_SORT)
    check_forward(d, o, cam)
    check_forward(a, o, cam)
    check_forward(b, o, cam)
    for k in ("color", "radii", "final_T"):   # what is left out never contributed: the image cannot tell
        assert np.array_equal(d[k], a[k]), (label, "trimmed", k)
    for k in ("color", "radii", "final_T", "n_contrib", "ranges"):
        assert np.array_equal(a[k], b[k]), (label, k)
    vis = a["tiles_touched"] > 0   # (slot_base is written for Gaussians with tiles only)
    assert np.array_equal(a["slot_base"][vis], b["slot_base"][vis]), (label, "slot_base")
    assert np.array_equal(d["slot_base"][vis], b["slot_base"][vis]), (label, "slot_base, trimmed")
    if o["num_rendered"] > 0:
        for k in ("point_list", "keys"):
            assert np.array_equal(a[k], b[k]), (label, k)
        for k in a["grads"]:
            assert np.array_equal(a["grads"][k], b["grads"][k]), (label, k)
    return o


@pytest.mark.parametrize("seed", [0, 3, 5, 8])
def test_column_pair_binning_and_tile_sort_agree_bit_for_bit(seed):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = np.random.default_rng(4000 + seed)
    W, H = int(r.integers(40, 700)) | 1, int(r.integers(30, 400)) | 1
    D = int(r.integers(0, 4))
    P = int(r.integers(1500, 30000))
    scene = gsr_scene.make_scene(P, float(r.uniform(-4.5, -2.0)), sh_degree=D, seed=seed + 70)
    inside = seed % 2 == 1   # cameras inside the cloud: splats with rectangles of hundreds of tiles, Gaussians culled
    eye = r.normal(size=3)
    eye = eye / np.linalg.norm(eye) * (float(r.uniform(0.2, 1.2)) if inside else float(r.uniform(2.5, 5.0)))
    R, T = _look_at(eye, r.uniform(-0.4, 0.4, 3))
    cam = gsr_scene.make_camera(W, H, fovx=float(r.uniform(0.5, 1.7)), R=R, T=T)
    o = _binning_pair(scene, cam, D, None, f"seed {seed}")
    print(f"seed {seed}: {W}x{H} D{D} P{P} inside={inside} R={o['num_rendered']} max radius {int(o['radii'].max())}")


@pytest.mark.parametrize("W,H", [(4096, 20), (4097, 20), (20, 4096), (20, 4100), (4090, 37), (1, 1), (16, 16)])
def test_binning_at_the_256_tile_limit_of_the_column_pairs(W, H):
    """Exactly 256 tile columns / rows (the widest digit of the column-pair passes), one tile more (the image takes the tile
    sort by itself), and single-tile images."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from diff_gaussian_rasterization import _C
    P, D = 3000, 1
    scene = gsr_scene.make_scene(P, -2.5, sh_degree=D, seed=W + H)
    # spread the Gaussians over the whole strip: a wide field of view along the long side
    cam = gsr_scene.make_camera(W, H, fovx=2.6 if W > 4 * H else (0.02 if H > 4 * W else 1.0))
    bl = _C.binning_layout(P, 1000, W, H)
    assert int(bl.column_pairs) == (1 if (W + 15) // 16 <= 256 and (H + 15) // 16 <= 256 else 0)
    _binning_pair(scene, cam, D, None, f"{W}x{H}")


def test_column_pairs_with_splats_that_cover_256_tile_columns_and_rows():
    """The longest runs the two passes can meet: a 4096 x 4096 image (256 x 256 tiles) with a few splats whose rectangles span
    every tile column and every tile row (runs of 256 in both passes, 65 536 instances per splat), among small ones."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    P, D, W, H = 40, 0, 4096, 4096
    scene = gsr_scene.make_scene(P, -3.0, sh_degree=D, seed=77)
    scales = scene.scales.clone()
    scales[:3] = torch.tensor([[2.5, 2.5, 0.3], [3.0, 0.8, 0.5], [0.4, 3.5, 0.4]])   # screen-filling, a wide and a tall one
    means = scene.means3D.clone()
    means[:3] = torch.tensor([[0.0, 0.0, 0.5], [0.1, -0.2, 1.0], [-0.3, 0.1, 0.0]])
    opac = scene.opacities.clone()
    opac[:3] = 0.05
    scene = scene._replace(scales=scales.contiguous(), means3D=means.contiguous(), opacities=opac.contiguous())
    cam = gsr_scene.make_camera(W, H, fovx=1.2)
    o = _binning_pair(scene, cam, D, None, "giant")
    rect_tiles = o["tiles_touched"].max()
    assert int(rect_tiles) == 256 * 256, int(rect_tiles)
