"""simple-knn replacement: oracle pinned on the CPU against an independent exact
nearest-neighbour search (scipy cKDTree, float64); HIP path bit-exact against the oracle on the GPU."""
import numpy as np
import pytest
import torch

from oracle import oracle


def _cloud(P, seed, kind="uniform"):
    r = np.random.default_rng(seed)
    if kind == "uniform":
        return r.uniform(-3, 3, size=(P, 3)).astype(np.float32)
    if kind == "clustered":   # COLMAP-like: dense surfaces + sparse outliers, spanning orders of magnitude in density
        c = r.normal(size=(max(P // 200, 1), 3)) * 4
        pts = c[r.integers(0, len(c), P)] + r.normal(size=(P, 3)) * r.choice([0.01, 0.1, 1.0], size=(P, 1))
        return pts.astype(np.float32)
    if kind == "planar":      # degenerate axis: z identical for all points, and x quantised (many exact ties)
        p = r.uniform(-1, 1, size=(P, 3)).astype(np.float32)
        p[:, 2] = 0.5
        p[:, 0] = np.round(p[:, 0] * 16) / 16
        return p
    raise ValueError(kind)


def _kdtree_mean_dist2(pts):
    from scipy.spatial import cKDTree
    p = pts.astype(np.float64)
    d, _ = cKDTree(p).query(p, k=4, workers=-1)   # the point itself (distance 0) + 3 others
    return (d[:, 1:] ** 2).sum(axis=1) / 3.0


@pytest.mark.parametrize("kind", ["uniform", "clustered", "planar"])
def test_oracle_is_the_exact_three_nearest_mean(kind):
    pts = _cloud(3000, 1, kind)
    got = oracle.knn_mean_dist2(pts)
    want = _kdtree_mean_dist2(pts)
    # fp32 evaluation of a sum of three squared differences: a few ulp of the largest coordinate difference
    assert np.all(np.abs(got - want) <= 1e-5 * want + 1e-9), float(np.max(np.abs(got - want) / (want + 1e-12)))


def test_oracle_edge_cases():
    # coincident points count with distance 0 (the self is excluded by position, simple_knn.cu:161-167)
    pts = np.array([[0, 0, 0], [0, 0, 0], [1, 0, 0], [0, 2, 0], [5, 5, 5]], np.float32)
    got = oracle.knn_mean_dist2(pts)
    assert got[0] == np.float32((0 + 1 + 4) / 3.0) and got[1] == got[0]
    assert got[2] == np.float32((1 + 1 + 5) / 3.0)
    # fewer than 4 points: missing neighbours stay FLT_MAX in (best[0]+best[1]+best[2])/3.0f:
    # one missing -> ~FLT_MAX/3, two or three missing -> the sum overflows to +inf
    assert np.all(oracle.knn_mean_dist2(pts[:3]) == np.float32(np.finfo(np.float32).max) / np.float32(3))
    assert np.all(np.isinf(oracle.knn_mean_dist2(pts[:2]))) and np.all(np.isinf(oracle.knn_mean_dist2(pts[:1])))
    assert oracle.knn_mean_dist2(np.zeros((0, 3), np.float32)).shape == (0,)


@pytest.mark.gpu
@pytest.mark.parametrize("P,kind", [(1, "uniform"), (2, "uniform"), (3, "uniform"), (4, "uniform"), (257, "uniform"), (5000, "planar"),
                                     (20011, "uniform"), (20000, "clustered")])
def test_hip_distCUDA2_is_bit_exact_against_the_oracle(P, kind):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from simple_knn._C import distCUDA2
    pts = _cloud(P, 7, kind)
    got = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    want = oracle.knn_mean_dist2(pts)
    assert got.dtype == np.float32 and got.shape == (P,)
    np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
def test_hip_distCUDA2_full_size_and_create_from_pcd_recipe():
    """1M points (the benchmark scene's size): exact against an independent float64 k-d tree, then the
    create_from_pcd recipe (gaussian_model.py:215-223): scales = log(sqrt(clamp_min(dist2, 1e-7)))."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from simple_knn._C import distCUDA2
    pts = _cloud(1_000_000, 11, "clustered")
    t = torch.from_numpy(pts).cuda()
    got = distCUDA2(t)
    again = distCUDA2(t)
    assert torch.equal(got, again)
    want = _kdtree_mean_dist2(pts)
    g = got.cpu().numpy().astype(np.float64)
    assert np.all(np.abs(g - want) <= 2e-5 * want + 1e-9), float(np.max(np.abs(g - want) / (want + 1e-12)))
    scales = torch.log(torch.sqrt(torch.clamp_min(got, 0.0000001)))[..., None].repeat(1, 3)
    assert scales.shape == (1_000_000, 3) and bool(torch.isfinite(scales).all())
    # duplicates of the whole cloud: every point has a coincident partner -> one zero term
    both = torch.cat([t[:5000], t[:5000]])
    d2 = distCUDA2(both)
    np.testing.assert_array_equal(d2.cpu().numpy(), oracle.knn_mean_dist2(both.cpu().numpy()))
