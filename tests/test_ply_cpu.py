"""CPU: PLY checkpoint / input-cloud formats (gsr_ply.py; scene/gaussian_model.py:262-364,
scene/dataset_readers.py:123-146).  The reference tree holds no PLY file and plyfile is not installed,
so byte identity with plyfile's writer is unpinned; these tests pin the layout the reference's code
defines: property names and order, float32 everywhere, channel-major SH features, raw leaves."""
import numpy as np
import pytest

import gsr_ply


def _leaves(P=37, M=16, seed=0):
    r = np.random.default_rng(seed)
    f = lambda *s: r.normal(size=s).astype(np.float32)
    return dict(xyz=f(P, 3), features_dc=f(P, 1, 3), features_rest=f(P, M - 1, 3), opacity=f(P, 1), scaling=f(P, 3), rotation=f(P, 4))


def test_checkpoint_layout_and_round_trip(tmp_path):
    lv = _leaves()
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")   # directories are created (mkdir_p)
    gsr_ply.save_gaussians_ply(path, **lv)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [l.split()[2] for l in lines[3:]]
    assert all(l.split()[:2] == ["property", "float"] for l in lines[3:])
    want = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + \
           ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert names == want and len(names) == 62
    assert len(body) == 37 * 62 * 4
    table = np.frombuffer(body, "<f4").reshape(37, 62)
    assert np.array_equal(table[:, 0:3], lv["xyz"]) and not table[:, 3:6].any()           # normals are zeros
    # channel-major: f_rest_{c*15+k} = features_rest[:, k, c]   (.transpose(1, 2).flatten(start_dim=1))
    for c in range(3):
        assert np.array_equal(table[:, 6 + c], lv["features_dc"][:, 0, c])
        for k in (0, 7, 14):
            assert np.array_equal(table[:, 9 + c * 15 + k], lv["features_rest"][:, k, c])
    assert np.array_equal(table[:, 54], lv["opacity"][:, 0])
    assert np.array_equal(table[:, 55:58], lv["scaling"]) and np.array_equal(table[:, 58:62], lv["rotation"])
    back = gsr_ply.load_gaussians_ply(path, max_sh_degree=3)
    for k, v in lv.items():
        assert back[k].dtype == np.float32 and back[k].shape == v.shape and np.array_equal(back[k], v), k
    with pytest.raises(ValueError, match="f_rest"):
        gsr_ply.load_gaussians_ply(path, max_sh_degree=2)


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_checkpoint_lower_degrees_and_torch_tensors(tmp_path, deg):
    import torch
    M = (deg + 1) ** 2
    lv = {k: torch.from_numpy(v).requires_grad_(True) for k, v in _leaves(11, M, seed=deg).items()}
    path = str(tmp_path / "pc.ply")
    gsr_ply.save_gaussians_ply(path, **lv)
    back = gsr_ply.load_gaussians_ply(path, max_sh_degree=deg)
    for k, v in lv.items():
        assert np.array_equal(back[k], v.detach().numpy()), k
    assert back["features_rest"].shape == (11, M - 1, 3)


def test_reader_accepts_ascii_big_endian_aliases_and_skips_other_elements(tmp_path):
    # big endian, float32/uint8 spellings, a comment, and a face element after the vertices
    xyz = np.array([[1.5, -2.0, 3.25], [0.0, 4.0, -1.0]], np.float32)
    rgb = np.array([[255, 0, 7], [1, 2, 3]], np.uint8)
    rec = np.empty(2, dtype=[("x", ">f4"), ("y", ">f4"), ("z", ">f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    for j, n in enumerate("xyz"):
        rec[n] = xyz[:, j]
    for j, n in enumerate(("red", "green", "blue")):
        rec[n] = rgb[:, j]
    head = ("ply\nformat binary_big_endian 1.0\ncomment made by hand\nelement vertex 2\nproperty float32 x\nproperty float32 y\n"
            "property float32 z\nproperty uint8 red\nproperty uint8 green\nproperty uint8 blue\nelement face 0\n"
            "property list uchar int vertex_indices\nend_header\n")
    p = tmp_path / "be.ply"
    p.write_bytes(head.encode() + rec.tobytes())
    v = gsr_ply.read_ply(str(p))
    assert list(v) == ["x", "y", "z", "red", "green", "blue"]
    assert np.array_equal(np.stack([v["x"], v["y"], v["z"]], 1), xyz) and v["x"].dtype == np.float32 and v["x"].dtype.isnative
    assert np.array_equal(np.stack([v["red"], v["green"], v["blue"]], 1), rgb)
    # ascii, with a scalar-only element stored BEFORE the vertices
    p2 = tmp_path / "ascii.ply"
    p2.write_text("ply\nformat ascii 1.0\nelement camera 1\nproperty float fx\nproperty int id\nelement vertex 2\nproperty float x\n"
                  "property double y\nproperty short z\nend_header\n500.5 3\n1.5 2.25 -7\n-0.5 1e-3 12\n")
    v = gsr_ply.read_ply(str(p2))
    assert np.array_equal(v["x"], np.float32([1.5, -0.5])) and v["y"].dtype == np.float64 and np.array_equal(v["z"], np.int16([-7, 12]))
    assert gsr_ply.read_ply(str(p2), element="camera")["id"][0] == 3
    # same with a binary file: the element in front is skipped by its byte size
    p3 = tmp_path / "skip.ply"
    p3.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement camera 1\nproperty float fx\nproperty int id\nelement vertex 1\n"
                   b"property float x\nend_header\n" + np.float32(9).tobytes() + np.int32(4).tobytes() + np.float32(2.5).tobytes())
    assert gsr_ply.read_ply(str(p3))["x"][0] == 2.5
    with pytest.raises(ValueError, match="not a PLY"):
        p4 = tmp_path / "bad.ply"
        p4.write_bytes(b"plx\n")
        gsr_ply.read_ply(str(p4))
    with pytest.raises(ValueError, match="truncated"):
        p5 = tmp_path / "short.ply"
        p5.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty float x\nend_header\n" + b"\0" * 8)
        gsr_ply.read_ply(str(p5))


def test_input_cloud_store_and_fetch(tmp_path):
    r = np.random.default_rng(3)
    xyz = r.normal(size=(100, 3))
    rgb = r.integers(0, 256, size=(100, 3))
    path = str(tmp_path / "points3D.ply")
    gsr_ply.store_ply(path, xyz, rgb)
    head = open(path, "rb").read().split(b"end_header\n")[0].decode().splitlines()
    assert head[3:] == ["property float x", "property float y", "property float z", "property float nx", "property float ny",
                        "property float nz", "property uchar red", "property uchar green", "property uchar blue"]
    pts, col, nrm = gsr_ply.fetch_ply(path)
    assert np.array_equal(pts, xyz.astype(np.float32)) and not nrm.any()
    assert np.array_equal(col, rgb / 255.0)
    # empty cloud
    gsr_ply.store_ply(path, np.zeros((0, 3)), np.zeros((0, 3)))
    pts, col, nrm = gsr_ply.fetch_ply(path)
    assert pts.shape == (0, 3) and col.shape == (0, 3)
