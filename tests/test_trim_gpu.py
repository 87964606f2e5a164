"""GPU: the tiles the column-pair binning leaves out (csrc/gsr_rect_trim.h).

The reference lists a Gaussian in every tile of the square of 3 sigma around it (auxiliary.h:42-58, rasterizer_impl.cu:78-126); the
blend loops then skip it pixel by pixel wherever alpha < 1 / 255 (forward.cu:449-456, backward.cu:527-534).  The library's default
binning leaves a (Gaussian, tile) instance out of the sorted list when the preprocess kernel can prove that no pixel centre of the
tile reaches alpha = 1 / 255 -- per tile column of the rectangle, up to three tile rows off either end -- and bins every tile like
the reference with GSR_DEBUG_NO_TRIM.  What has to hold, and is checked here:
  * nothing that could contribute is left out: against a float64 evaluation of every pixel centre of every rectangle tile;
  * no output can tell: image, radii, final T bit-identical, and (with the heavy tiles' depth segments off, whose cuts fall on list
    positions) every gradient bit-identical;
  * the lists are the untrimmed lists less exactly the instances the trim words name, in the same order.
"""
import numpy as np
import pytest
import torch

import gsr_scene
import util

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _scenes():
    out = []
    for seed in range(8):
        r = np.random.default_rng(900 + seed)
        P = int(r.integers(800, 5000))
        W, H = int(r.integers(60, 420)), int(r.integers(40, 300))
        D = int(r.integers(0, 4))
        scene = gsr_scene.make_scene(P, float(r.uniform(-4.0, -1.2)), sh_degree=D, seed=seed)   # up to splats of a third of the image
        op = scene.opacities.clone()
        g = torch.Generator().manual_seed(seed)
        if seed % 2:   # low opacities: the alpha = 1 / 255 ellipse is small, most of the rectangle goes
            op = torch.sigmoid(torch.randn(P, 1, generator=g) * 1.5 - 3.0)
        op[::17] = 0.0
        op[5::17] = 1.0 / 255.0
        scene = scene._replace(opacities=op.contiguous())
        cam = gsr_scene.ring_camera(W, H, int(r.integers(0, 8)), 8, radius=float(r.uniform(0.5, 4.5)))
        out.append((f"seed{seed}", scene, cam, D))
    return out


def test_no_tile_that_could_contribute_is_left_out():
    """Brute force in float64: for every Gaussian and every tile of its rectangle, the largest alpha over the tile's pixel centres
    inside the image.  A tile with a pixel at alpha >= (1 - 1e-4) / 255 and power <= 0 must be kept by the trim words."""
    _need_gpu()
    tot = kept_n = need_n = 0
    for name, scene, cam, D in _scenes():
        W, H = cam.image_width, cam.image_height
        gx = (W + 15) // 16
        h = util.hip_forward_backward(scene, cam, D, None)
        rs, m2, co = h["rshape"], h["means2D"].astype(np.float64), h["conic_opacity"].astype(np.float64)
        vis = np.nonzero(h["tiles_touched"] > 0)[0]
        packed = rs[vis, 0].astype(np.int64)
        x0, y0, w, hh = packed & 255, (packed >> 8) & 255, ((packed >> 16) & 255) + 1, (packed >> 24) + 1
        assert np.array_equal(w * hh, h["tiles_touched"][vis].astype(np.int64))
        gi, ti = [], []
        for k, g in enumerate(vis):   # every (Gaussian, tile) pair of the rectangles
            tx, ty = np.meshgrid(np.arange(x0[k], x0[k] + w[k]), np.arange(y0[k], y0[k] + hh[k]))
            t = (ty * gx + tx).reshape(-1)
            gi.append(np.full(t.shape, g)); ti.append(t)
        gi, ti = np.concatenate(gi), np.concatenate(ti)
        kept = util.trim_kept(rs, gi, ti, gx)
        # largest alpha over the tile's 256 pixel centres
        px = (ti % gx)[:, None] * 16 + np.arange(16)[None, :]          # (n, 16)
        py = (ti // gx)[:, None] * 16 + np.arange(16)[None, :]
        dx = m2[gi, 0][:, None, None] - px[:, None, :]                 # (n, 1, 16): the reference's d = mean - pixel
        dy = m2[gi, 1][:, None, None] - py[:, :, None]                 # (n, 16, 1)
        a, b, c, o = (co[gi, j][:, None, None] for j in range(4))
        power = -0.5 * (a * dx * dx + c * dy * dy) - b * dx * dy
        alpha = np.minimum(0.99, o * np.exp(np.minimum(power, 50.0)))
        inside = (px[:, None, :] < W) & (py[:, :, None] < H)
        can = ((alpha >= (1.0 - 1e-4) / 255.0) & (power <= 0.0) & inside).any(axis=(1, 2))
        lost = can & ~kept
        assert not lost.any(), f"{name}: {int(lost.sum())} instances that can contribute are left out, e.g. Gaussian {gi[lost][0]} tile {ti[lost][0]}"
        tot += len(gi); kept_n += int(kept.sum()); need_n += int(can.sum())
        assert len(h["point_list"]) == int(kept.sum()), f"{name}: the list holds {len(h['point_list'])} instances, the trim words keep {int(kept.sum())}"
    line = f"[trim] {tot} (Gaussian, tile) pairs in 8 random scenes: {need_n} hold a pixel that can reach alpha = 1/255 ({need_n / tot:.3f}), the trim words keep {kept_n} ({kept_n / tot:.3f}); none that can contribute is left out"
    print(line)
    util.parity_log(line)
    assert kept_n < tot


@pytest.mark.parametrize("case", ["fuzz", "C2"])
def test_outputs_cannot_tell_and_lists_are_the_whole_lists_less_the_named_instances(case):
    _need_gpu()
    from diff_gaussian_rasterization import _C
    runs = _scenes() if case == "fuzz" else [("C2",) + gsr_scene.make_config("C2")]
    for name, scene, cam, D in runs:
        W, H = cam.image_width, cam.image_height
        gx = (W + 15) // 16
        dpix = torch.randn(3, H, W, generator=torch.Generator().manual_seed(3))
        ns = _C.DEBUG_NO_SPLIT   # (no depth segments: their cuts fall on list positions, which differ between the two lists)
        d = util.hip_forward_backward(scene, cam, D, dpix, debug=ns)
        a = util.hip_forward_backward(scene, cam, D, dpix, debug=ns | _C.DEBUG_NO_TRIM)
        for k in ("color", "radii", "final_T", "tiles_touched", "slot_base"):
            assert np.array_equal(d[k], a[k]), (name, k)
        assert d["num_rendered"] == a["num_rendered"] == int(a["tiles_touched"].astype(np.int64).sum())
        for k in a["grads"]:
            assert np.array_equal(d["grads"][k], a["grads"][k]), (name, k)
        if a["num_rendered"] == 0:
            continue
        assert not a["rshape"][:, 1].any() and len(a["point_list"]) == a["num_rendered"]
        # the default lists = the whole lists less the instances the trim words name
        T = gx * ((H + 15) // 16)
        lens = (a["ranges"][:, 1].astype(np.int64) - a["ranges"][:, 0])
        tile_of = np.repeat(np.arange(T, dtype=np.int64), lens)
        kept = util.trim_kept(d["rshape"], a["point_list"], tile_of, gx)
        assert np.array_equal(d["point_list"], a["point_list"][kept]), name
        assert np.array_equal(d["keys"], a["keys"][kept]), name
        kcum = np.concatenate([[0], np.cumsum(kept)]).astype(np.int64)
        r0 = a["ranges"].astype(np.int64)
        new_len = kcum[r0[:, 1]] - kcum[r0[:, 0]]
        want = np.where((new_len > 0)[:, None], np.stack([kcum[r0[:, 0]], kcum[r0[:, 0]] + new_len], 1), 0)
        assert np.array_equal(d["ranges"].astype(np.int64), want), name
        ys, xs = np.divmod(np.arange(W * H, dtype=np.int64), W)
        tp = (ys // 16) * gx + xs // 16
        n = a["n_contrib"].astype(np.int64)
        assert np.array_equal(d["n_contrib"].astype(np.int64), kcum[r0[tp, 0] + n] - kcum[r0[tp, 0]]), name
        print(f"{name}: {len(d['point_list'])} of {len(a['point_list'])} instances listed")
