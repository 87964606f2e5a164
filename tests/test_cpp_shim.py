"""The boundary without Python: include/gsr.h as a C header (gcc -std=c99 -pedantic), and a C++ caller on the
reference's own library surface, CudaRasterizer::Rasterizer (cuda_rasterizer/rasterizer.h:20-85), through
include/gsr_rasterizer.hpp.  Recipe: tests/cpp/Makefile (built by __graft_entry__.build(); rebuilt here when missing)."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

import gsr_scene
import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _built(name):
    """make is incremental: a binary that is older than include/gsr.h or the library is rebuilt (a stale one would read the
    layout structs of the previous header)."""
    subprocess.check_call(["make", "-C", CPP, name], stdout=subprocess.DEVNULL)
    return os.path.join(CPP, name)


def test_header_compiles_as_c_and_every_entry_point_links():
    """abi_check.c includes gsr.h under -std=c99 -pedantic -Werror, takes the address of every declared entry point
    (a missing export fails the link) and runs the host-only calls; no device work."""
    r = subprocess.run([_built("abi_check")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi_check ok" in r.stdout and "gfx950" in r.stdout
    # the list in abi_check.c is complete: every gsr_* function the header declares is named there
    import re
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gsr.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", hdr))
    src = open(os.path.join(CPP, "abi_check.c")).read()
    missing = [n for n in sorted(declared) if f"(any_fn){n}" not in src]
    assert not missing, missing


def test_cpp_shim_header_builds():
    assert os.path.exists(_built("shim_demo"))


@pytest.mark.gpu
def test_cpp_caller_through_the_reference_signature_matches_the_python_binding(tmp_path):
    """shim_demo (C++, hipMalloc/hipMemcpy only) renders a 2 000-Gaussian scene through
    CudaRasterizer::Rasterizer::{markVisible, forward, backward} with resize callbacks, like rasterize_points.cu does;
    every output must equal the Python binding's bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from diff_gaussian_rasterization import _C
    dev = torch.device("cuda:0")
    P, D, M, W, H = 2000, 3, 16, 200, 120
    scene = gsr_scene.make_scene(P, -3.0, sh_degree=D, seed=3)
    cam = gsr_scene.ring_camera(W, H, 3, 8, radius=1.0)   # inside the cloud: part of the scene is behind the camera
    dpix = torch.randn(3, H, W, generator=torch.Generator().manual_seed(7))
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<5i3f", P, D, M, W, H, cam.tanfovx, cam.tanfovy, 1.0))
        for t in (scene.bg, scene.means3D, scene.shs, scene.opacities, scene.scales, scene.rotations, cam.world_view_transform,
                  cam.full_proj_transform, cam.camera_center, dpix):
            f.write(t.contiguous().numpy().astype("<f4").tobytes())
    r = subprocess.run([_built("shim_demo"), str(inp), str(outp)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    # the same through the Python binding
    st = util.hip_settings(scene, cam, D, dev)
    e = torch.empty(0, device=dev)
    t = {k: getattr(scene, k).to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(
        st.bg, t["means3D"], e, t["opacities"], t["scales"], t["rotations"], 1.0, e, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy,
        H, W, t["shs"], D, st.campos, False, False)
    dbg = {}
    grads = _C.rasterize_gaussians_backward(st.bg, t["means3D"], radii, e, t["scales"], t["rotations"], 1.0, e, st.viewmatrix, st.projmatrix,
                                            st.tanfovx, st.tanfovy, dpix.to(dev), t["shs"], D, st.campos, geom, R, binning, img, False,
                                            debug_out=dbg)
    present = _C.mark_visible(t["means3D"], st.viewmatrix, st.projmatrix)
    raw = open(outp, "rb").read()
    off = [0]

    def take(dtype, *shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        a = np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape)), offset=off[0]).reshape(shape)
        off[0] += n
        return a
    assert int(take("<i4", 1)[0]) == R and R > 0
    np.testing.assert_array_equal(take(np.uint8, P).astype(bool), present.cpu().numpy())
    assert 0 < int(present.sum()) < P
    np.testing.assert_array_equal(take("<f4", 3, H, W), color.cpu().numpy())
    np.testing.assert_array_equal(take("<i4", P), radii.cpu().numpy())
    shapes = [(P, 3), (P, 3), (P, 1), (P, 3), (P, 6), (P, M, 3), (P, 3), (P, 4)]
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]
    for name, shape, g in zip(names, shapes, grads):
        got = take("<f4", *shape)
        assert np.array_equal(got, g.cpu().numpy()), name
        assert np.abs(got).max() > 0, name
    np.testing.assert_array_equal(take("<f4", P, 2, 2), dbg["dL_dconic"].cpu().numpy())
    assert off[0] == len(raw)
