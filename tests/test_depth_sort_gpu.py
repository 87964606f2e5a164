"""GPU: the depth order of the Gaussians (the depth half of the reference's 64-bit SortPairs, rasterizer_impl.cu:357-374).

Two implementations produce it -- top-digit buckets sorted inside LDS (csrc/depthsort.hip, up to 2 Mi Gaussians) and global
LSD radix passes (csrc/sort.hip; GSR_DEBUG_RADIX_DEPTH) -- and both must give exactly numpy's stable argsort of the depth bits,
culled Gaussians last, whatever the depth distribution: buckets that overflow LDS and are shared by several workgroups, slabs
of thousands of Gaussians inside a few float steps (next-digit levels down to identical keys), depth ranges beyond 2^24 float
steps, nothing visible at all."""
import numpy as np
import pytest
import torch

import gsr_scene
import util

pytestmark = pytest.mark.gpu


def _run(scene, cam, D, debug):
    from diff_gaussian_rasterization import _C
    dev = torch.device("cuda:0")
    st = util.hip_settings(scene, cam, D, dev)
    e = torch.empty(0, device=dev)
    t = {k: getattr(scene, k).to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(
        st.bg, t["means3D"], e, t["opacities"], t["scales"], t["rotations"], 1.0, e, st.viewmatrix, st.projmatrix, st.tanfovx,
        st.tanfovy, st.image_height, st.image_width, t["shs"], D, st.campos, False, debug)
    torch.cuda.synchronize()
    o = util.unpack_state(dict(R=R, geom=geom, binning=binning, img=img), scene.means3D.shape[0], cam.image_width, cam.image_height,
                          tile_sort=bool(debug & _C.DEBUG_TILE_SORT))
    o["color"], o["radii"], o["R"] = color.cpu().numpy(), radii.cpu().numpy(), R
    return o


def _scene_with_depths(P, z, seed, spread=1.2):
    """P small Gaussians at view depths z (camera at (0, 0, -4) looking down +z: depth = z_world + 4)."""
    scene = gsr_scene.make_scene(P, -5.0, sh_degree=0, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    xy = (torch.rand(P, 2, generator=g) * 2.0 - 1.0) * spread
    zt = torch.as_tensor(z, dtype=torch.float32)
    # keep them inside the frustum whatever their depth: x, y scale with the depth
    means = torch.cat([xy * (zt[:, None] * 0.12), (zt - 4.0)[:, None]], dim=1)
    return scene._replace(means3D=means.contiguous())


def _depth_cases():
    r = np.random.default_rng(7)
    cases = {}
    cases["uniform_5k"] = r.uniform(2.5, 5.5, 5000)
    cases["uniform_70k"] = r.uniform(2.5, 5.5, 70_000)                 # buckets of ~270: one workgroup each
    cases["uniform_700k"] = r.uniform(2.5, 5.5, 700_000)               # buckets beyond DS_CAP: shared by several workgroups
    cases["identical_20k"] = np.full(20_000, 4.0)                      # one key: nothing to sort, ids stay in order
    z = r.uniform(2.5, 5.5, 100_000)
    z[:60_000] = 4.0 + r.integers(0, 20, 60_000) * 4.7683716e-07        # a slab of 60 000 Gaussians on 20 float steps
    cases["slab_100k"] = r.permutation(z)
    z = np.exp(r.uniform(np.log(0.3), np.log(3000.0), 150_000))          # 13 binades: more than 2^24 float steps
    z[:40_000] = 7.0 + r.integers(0, 3, 40_000) * 4.7683716e-07
    z[40_000:50_000] = 7.0 + r.uniform(0, 3e-3, 10_000)
    cases["wide_150k"] = r.permutation(z)
    z = r.uniform(2.5, 5.5, 50_000)
    z[r.random(50_000) < 0.9] = -1.0                                   # nine in ten behind the camera
    cases["mostly_culled_50k"] = z
    cases["none_visible_3k"] = np.full(3000, -2.0)
    z = np.repeat(r.uniform(2.5, 5.5, 15_000), 4)                      # every depth four times (clones keep index order)
    cases["clones_60k"] = r.permutation(z)
    cases["two_values_9k"] = np.where(r.random(9000) < 0.5, 3.0, 3.0000002)
    cases["one_visible"] = np.concatenate([np.full(999, -1.0), [3.0]])
    return cases


@pytest.mark.parametrize("name", list(_depth_cases().keys()))
def test_bucket_sort_and_radix_sort_give_the_stable_argsort(name):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from diff_gaussian_rasterization import _C
    z = _depth_cases()[name]
    P = len(z)
    scene = _scene_with_depths(P, z, seed=len(name))
    cam = gsr_scene.make_camera(203, 117)
    a = _run(scene, cam, 0, 0)
    b = _run(scene, cam, 0, _C.DEBUG_RADIX_DEPTH)
    assert a["depth_sort_result_in_alt"] == 0   # the bucket sort always ends in (depth_keys, perm)
    bits = b["depth_bits"]
    np.testing.assert_array_equal(a["depth_bits"], bits)
    want = np.argsort(bits, kind="stable").astype(np.uint32)
    for o, path in ((a, "bucket"), (b, "radix")):
        np.testing.assert_array_equal(o["perm"], want, err_msg=f"{name}: {path} order")
        np.testing.assert_array_equal(o["sorted_depth_keys"], bits[want], err_msg=f"{name}: {path} keys")
    visible = int((bits != 0xFFFFFFFF).sum())
    if name.startswith("none"):
        assert visible == 0 and a["R"] == 0
    else:
        assert visible > 0 and a["R"] > 0
        for k in ("point_list", "ranges", "n_contrib", "color"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"{name}: {k}")
        # the gradient slots: numbered in index order on both paths (the bucket sort's first kernel does it on the way, the radix passes
        # have a kernel of their own in front)
        assert a["slots_in_index_order"] == 1 and b["slots_in_index_order"] == 1
        tt = a["tiles_touched"].astype(np.int64)
        for o in (a, b):
            np.testing.assert_array_equal(o["slot_base"][tt > 0], (np.cumsum(tt) - tt)[tt > 0].astype(np.uint32))
    # ... and the same through the instance emission + tile sort, which reads the depth order through perm and the block sums
    # (it bins every tile of every rectangle, like the reference: compared with the column pairs doing the same, GSR_DEBUG_NO_TRIM;
    # the image is the default run's bit for bit either way)
    c = _run(scene, cam, 0, _C.DEBUG_TILE_SORT)
    a0 = _run(scene, cam, 0, _C.DEBUG_NO_TRIM)
    np.testing.assert_array_equal(c["perm"], want)
    np.testing.assert_array_equal(c["color"], a["color"])
    np.testing.assert_array_equal(a0["color"], a["color"])
    if a["R"] > 0:
        np.testing.assert_array_equal(c["point_list"], a0["point_list"])
        np.testing.assert_array_equal(c["ranges"], a0["ranges"])
        assert len(a["point_list"]) <= len(a0["point_list"]) == a["R"]
    print(f"{name}: P {P}, visible {visible}, distinct keys {len(np.unique(bits))}, R {a['R']}")


@pytest.mark.parametrize("P", [(2 << 20), (2 << 20) + 1])
def test_both_sides_of_the_bucket_sort_limit(P):
    """GSR_BUCKET_SORT_MAX_P = 2 Mi Gaussians: the largest list the bucket sort takes (buckets of ~14 000: beyond the registers, shared by
    next-digit histogram) and the smallest one that goes to the global radix passes."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = np.random.default_rng(P & 0xFF)
    z = r.uniform(2.5, 5.5, P)
    z[: P // 8] = 3.0 + r.integers(0, 300, P // 8) * 2.3841858e-07   # a slab on 300 float steps
    scene = _scene_with_depths(P, r.permutation(z), seed=9)
    cam = gsr_scene.make_camera(203, 117)
    a = _run(scene, cam, 0, 0)
    bits = a["depth_bits"]
    want = np.argsort(bits, kind="stable").astype(np.uint32)
    np.testing.assert_array_equal(a["perm"], want)
    np.testing.assert_array_equal(a["sorted_depth_keys"], bits[want])
    assert a["depth_sort_result_in_alt"] == (0 if P <= (2 << 20) else 1)
    assert a["R"] > 0


def test_few_distinct_depths_sort_the_same_every_time():
    """Round-4 regression (tools/depth_sort_stress.py case 156): 318 498 Gaussians on seven distinct depths -- buckets of 45 000
    identical keys, which the bucket sort takes down its next-digit levels to a plain copy.  One run in ten left three of a
    workgroup's four waves with a stale level state (the compiler had dropped the LDS wait in front of the level loop's barrier:
    gsr_depth_key.h, gsr_sync(); DESIGN.md "The stale level state") and their part of the bucket unwritten.  The order must be the
    stable argsort, and the same in every one of 40 runs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = np.random.default_rng(500000 + 156)
    P = int(np.exp(r.uniform(np.log(1), np.log(600_000))))
    r.integers(0, 7)
    z = r.choice(r.uniform(0.5, 50.0, int(r.integers(1, 9))), P)
    assert P == 318498 and len(np.unique(z)) == 7
    scene = _scene_with_depths(P, z.copy(), seed=156)
    cam = gsr_scene.make_camera(203, 117)
    first = None
    for it in range(40):
        a = _run(scene, cam, 0, 0)
        if first is None:
            bits = a["depth_bits"]
            want = np.argsort(bits, kind="stable").astype(np.uint32)
            first = a
            np.testing.assert_array_equal(a["perm"], want)
            np.testing.assert_array_equal(a["sorted_depth_keys"], bits[want])
            assert a["R"] > 0
            continue
        assert a["R"] == first["R"], f"run {it}"
        for k in ("perm", "sorted_depth_keys", "point_list", "ranges"):
            np.testing.assert_array_equal(a[k], first[k], err_msg=f"run {it}: {k}")
