"""ctypes front-end of the CPU oracle (oracle/gsr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.  "parity unpinned" apart from
the SH / camera golden vectors -- see the header of gsr_oracle.c.

The call sequence mirrors CudaRasterizer::Rasterizer::forward / backward
(cuda_rasterizer/rasterizer_impl.cu:227-411, 416-518) and the tensor shapes of
rasterize_points.cu:38-215.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f = ctypes.POINTER(ctypes.c_float)
_d = ctypes.POINTER(ctypes.c_double)
_i = ctypes.POINTER(ctypes.c_int)
_u = ctypes.POINTER(ctypes.c_uint32)
_u64 = ctypes.POINTER(ctypes.c_uint64)
_b = ctypes.POINTER(ctypes.c_ubyte)


def build(force=False):
    so = os.path.join(_HERE, "libgsr_oracle.so")
    src = os.path.join(_HERE, "gsr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgsr_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.gsro_preprocess.restype = ctypes.c_longlong
        _LIB.gsro_get_higher_msb.restype = ctypes.c_uint32
    return _LIB


def _p(a, t):
    if a is None:
        return ctypes.cast(None, t)
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return a.ctypes.data_as(t)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def get_higher_msb(n):
    return int(lib().gsro_get_higher_msb(ctypes.c_uint32(n)))


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D, viewmatrix, projmatrix = _f32(means3D), _f32(viewmatrix), _f32(projmatrix)
    P = means3D.shape[0]
    out = np.zeros(P, np.uint8)
    lib().gsro_mark_visible(P, _p(means3D, _f), _p(viewmatrix, _f), _p(projmatrix, _f), _p(out, _b))
    return out.astype(bool)


def forward(means3D, opacities, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, sh_degree=0,
            shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            scale_modifier=1.0, prefiltered=False, margin=1e-4):
    """Full forward; returns a dict with the outputs and every intermediate."""
    means3D, opacities = _f32(means3D), _f32(opacities).reshape(-1)
    viewmatrix, projmatrix, campos, bg = _f32(viewmatrix), _f32(projmatrix), _f32(campos), _f32(bg)
    shs, colors_precomp, scales, rotations, cov3D_precomp = map(_f32, (shs, colors_precomp, scales, rotations, cov3D_precomp))
    P = means3D.shape[0]
    M = 0 if shs is None else shs.shape[1]
    N = W * H
    gx, gy = (W + 15) // 16, (H + 15) // 16
    s = dict(P=P, M=M, D=sh_degree, W=W, H=H, tanfovx=tanfovx, tanfovy=tanfovy, scale_modifier=scale_modifier,
             means3D=means3D, opacities=opacities, viewmatrix=viewmatrix, projmatrix=projmatrix, campos=campos,
             bg=bg, shs=shs, colors_precomp=colors_precomp, scales=scales, rotations=rotations,
             cov3D_precomp=cov3D_precomp)
    s["radii"] = np.zeros(P, np.int32)
    s["means2D"] = np.zeros((P, 2), np.float32)
    s["depths"] = np.zeros(P, np.float32)
    s["cov3D"] = np.zeros((P, 6), np.float32)
    s["rgb"] = np.zeros((P, 3), np.float32)
    s["conic_opacity"] = np.zeros((P, 4), np.float32)
    s["clamped"] = np.zeros((P, 3), np.uint8)
    s["tiles_touched"] = np.zeros(P, np.uint32)
    s["point_offsets"] = np.zeros(P, np.uint32)
    s["color"] = np.zeros((3, H, W), np.float32)
    s["final_T"] = np.zeros(N, np.float32)
    s["n_contrib"] = np.zeros(N, np.uint32)
    s["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    s["fragile"] = np.zeros(N, np.uint8)
    s["num_rendered"] = 0
    s["keys"] = np.zeros(0, np.uint64)
    s["point_list"] = np.zeros(0, np.uint32)
    if P == 0:  # rasterize_points.cu:94 -- no launches, image stays zero-filled
        return s
    R = lib().gsro_preprocess(
        P, sh_degree, M, _p(means3D, _f), _p(scales, _f), ctypes.c_float(scale_modifier), _p(rotations, _f),
        _p(opacities, _f), _p(shs, _f), _p(cov3D_precomp, _f), _p(colors_precomp, _f), _p(viewmatrix, _f),
        _p(projmatrix, _f), _p(campos, _f), W, H, ctypes.c_float(tanfovx), ctypes.c_float(tanfovy),
        int(prefiltered), _p(s["radii"], _i), _p(s["means2D"], _f), _p(s["depths"], _f), _p(s["cov3D"], _f),
        _p(s["rgb"], _f), _p(s["conic_opacity"], _f), _p(s["clamped"], _b), _p(s["tiles_touched"], _u),
        _p(s["point_offsets"], _u))
    if R < 0:
        raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    s["num_rendered"] = int(R)
    s["keys"] = np.zeros(R, np.uint64)
    s["point_list"] = np.zeros(R, np.uint32)
    rc = lib().gsro_bin(P, ctypes.c_longlong(R), W, H, _p(s["radii"], _i), _p(s["means2D"], _f), _p(s["depths"], _f),
                        _p(s["point_offsets"], _u), _p(s["keys"], _u64), _p(s["point_list"], _u), _p(s["ranges"], _u))
    assert rc == 0
    colors = colors_precomp if colors_precomp is not None else s["rgb"]
    lib().gsro_render_forward(W, H, _p(s["ranges"], _u), _p(s["point_list"], _u), _p(s["means2D"], _f),
                              _p(colors, _f), _p(s["conic_opacity"], _f), _p(bg, _f), _p(s["final_T"], _f),
                              _p(s["n_contrib"], _u), _p(s["color"], _f), _p(s["fragile"], _b),
                              ctypes.c_float(margin))
    return s


def backward(s, dL_dpix, accum_mode=0):
    """Backward for a state returned by forward(); returns the 8 gradients of
    rasterize_points.cu:215 (as float32) plus the double-precision per-Gaussian blend sums.
    accum_mode 0: blend sums in double (the checker).  1 / 2: fp32 accumulation like the reference's
    atomics, tiles visited in ascending / descending order -- two members of the family of results
    the reference itself can return; their spread is the reference's own reproducibility band."""
    dL_dpix = _f32(dL_dpix)
    P, M, W, H = s["P"], s["M"], s["W"], s["H"]
    g = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
             dL_dopacity=np.zeros((P, 1), np.float32), dL_dmeans3D=np.zeros((P, 3), np.float32),
             dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
             dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32),
             dL_dconic=np.zeros((P, 2, 2), np.float32))
    if P == 0:
        return g
    m2 = np.zeros((P, 2), np.float64)
    con = np.zeros((P, 3), np.float64)
    op = np.zeros(P, np.float64)
    col = np.zeros((P, 3), np.float64)
    colors = s["colors_precomp"] if s["colors_precomp"] is not None else s["rgb"]
    lib().gsro_render_backward(W, H, _p(s["ranges"], _u), _p(s["point_list"], _u), _p(s["bg"], _f),
                               _p(s["means2D"], _f), _p(s["conic_opacity"], _f), _p(colors, _f),
                               _p(s["final_T"], _f), _p(s["n_contrib"], _u), _p(dL_dpix, _f), _p(m2, _d),
                               _p(con, _d), _p(op, _d), _p(col, _d), int(accum_mode))
    g["dL_dmeans2D"][:, :2] = m2
    g["dL_dcolors"][:] = col
    g["dL_dopacity"][:, 0] = op
    conic4 = g["dL_dconic"].reshape(P, 4)
    conic4[:, 0] = con[:, 0]
    conic4[:, 1] = con[:, 1]
    conic4[:, 3] = con[:, 2]
    cov3D = s["cov3D_precomp"] if s["cov3D_precomp"] is not None else s["cov3D"]
    lib().gsro_preprocess_backward(
        P, s["D"], M, _p(s["means3D"], _f), _p(s["radii"], _i), _p(s["shs"], _f), _p(s["clamped"], _b),
        _p(s["scales"], _f), _p(s["rotations"], _f), ctypes.c_float(s["scale_modifier"]), _p(cov3D, _f),
        _p(s["viewmatrix"], _f), _p(s["projmatrix"], _f), W, H, ctypes.c_float(s["tanfovx"]),
        ctypes.c_float(s["tanfovy"]), _p(s["campos"], _f), _p(g["dL_dmeans2D"], _f), _p(conic4, _f),
        _p(g["dL_dmeans3D"], _f), _p(g["dL_dcolors"], _f), _p(g["dL_dcov3D"], _f), _p(g["dL_dsh"], _f),
        _p(g["dL_dscales"], _f), _p(g["dL_drotations"], _f))
    g["blend64"] = dict(mean2D=m2, conic=con, opacity=op, colors=col)
    return g


def blend_backward_exact(s, dL_dpix):
    """The backward blend evaluated in float64 with the fp32 path's discrete decisions
    (gsro_render_backward_exact).  Returns float64 dicts shaped like backward()'s blend outputs:
    dL_dmeans2D (P,3), dL_dconic (P,2,2), dL_dopacity (P,1), dL_dcolors (P,3)."""
    dL_dpix = _f32(dL_dpix)
    P, W, H = s["P"], s["W"], s["H"]
    m2 = np.zeros((P, 2), np.float64)
    con = np.zeros((P, 3), np.float64)
    op = np.zeros(P, np.float64)
    col = np.zeros((P, 3), np.float64)
    colors = s["colors_precomp"] if s["colors_precomp"] is not None else s["rgb"]
    if P and s["num_rendered"]:
        lib().gsro_render_backward_exact(W, H, _p(s["ranges"], _u), _p(s["point_list"], _u), _p(s["bg"], _f),
                                         _p(s["means2D"], _f), _p(s["conic_opacity"], _f), _p(colors, _f),
                                         _p(s["n_contrib"], _u), _p(dL_dpix, _f), _p(m2, _d), _p(con, _d), _p(op, _d),
                                         _p(col, _d))
    g = dict(dL_dmeans2D=np.zeros((P, 3)), dL_dconic=np.zeros((P, 2, 2)), dL_dopacity=op.reshape(P, 1), dL_dcolors=col)
    g["dL_dmeans2D"][:, :2] = m2
    c4 = g["dL_dconic"].reshape(P, 4)
    c4[:, 0], c4[:, 1], c4[:, 3] = con[:, 0], con[:, 1], con[:, 2]
    return g


def gaussian_backward(s, dL_dmeans2D, dL_dconic, dL_dcolors):
    """Only BACKWARD::preprocess (backward.cu:144-277, 349-399) on caller-supplied blend sums
    (dL_dmeans2D (P,3), dL_dconic (P,2,2), dL_dcolors (P,3), float32): lets a test feed the HIP
    blend kernel's own sums to the oracle's per-Gaussian chain, so that stage is compared on
    identical inputs."""
    P, M, W, H = s["P"], s["M"], s["W"], s["H"]
    m2 = _f32(dL_dmeans2D).reshape(P, 3)
    conic4 = _f32(dL_dconic).reshape(P, 4)
    g = dict(dL_dcolors=_f32(dL_dcolors).reshape(P, 3).copy(), dL_dmeans3D=np.zeros((P, 3), np.float32),
             dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
             dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32))
    cov3D = s["cov3D_precomp"] if s["cov3D_precomp"] is not None else s["cov3D"]
    lib().gsro_preprocess_backward(
        P, s["D"], M, _p(s["means3D"], _f), _p(s["radii"], _i), _p(s["shs"], _f), _p(s["clamped"], _b),
        _p(s["scales"], _f), _p(s["rotations"], _f), ctypes.c_float(s["scale_modifier"]), _p(cov3D, _f),
        _p(s["viewmatrix"], _f), _p(s["projmatrix"], _f), W, H, ctypes.c_float(s["tanfovx"]),
        ctypes.c_float(s["tanfovy"]), _p(s["campos"], _f), _p(m2, _f), _p(conic4, _f),
        _p(g["dL_dmeans3D"], _f), _p(g["dL_dcolors"], _f), _p(g["dL_dcov3D"], _f), _p(g["dL_dsh"], _f),
        _p(g["dL_dscales"], _f), _p(g["dL_drotations"], _f))
    return g


def sh_forward(deg, pos, campos, shs):
    """forward.cu:21-81 on its own -> (rgb (n,3), clamped (n,3) uint8)."""
    pos, campos, shs = _f32(pos), _f32(campos), _f32(shs)
    n, M = pos.shape[0], shs.shape[1]
    rgb = np.zeros((n, 3), np.float32)
    clamped = np.zeros((n, 3), np.uint8)
    lib().gsro_sh_forward(n, deg, M, _p(pos, _f), _p(campos, _f), _p(shs, _f), _p(rgb, _f), _p(clamped, _b))
    return rgb, clamped


def sh_backward(deg, pos, campos, shs, clamped, dL_dcolor):
    """backward.cu:20-139 on its own -> (dL_dmean (n,3) view-direction term only, dL_dsh (n,M,3))."""
    pos, campos, shs, dL_dcolor = _f32(pos), _f32(campos), _f32(shs), _f32(dL_dcolor)
    clamped = np.ascontiguousarray(clamped, dtype=np.uint8)
    n, M = pos.shape[0], shs.shape[1]
    dmean = np.zeros((n, 3), np.float32)
    dsh = np.zeros((n, M, 3), np.float32)
    lib().gsro_sh_backward(n, deg, M, _p(pos, _f), _p(campos, _f), _p(shs, _f), _p(clamped, _b), _p(dL_dcolor, _f),
                           _p(dmean, _f), _p(dsh, _f))
    return dmean, dsh


def l1_ssim(img, gt, lambda_dssim=0.2, want_grad=True):
    """(1-lambda)*L1 + lambda*(1-SSIM) of train.py:126-127 / utils/loss_utils.py, in double.
    -> (loss, l1, ssim, dloss/dimg (C,H,W) float32 or None)"""
    img, gt = _f32(img), _f32(gt)
    C, H, W = img.shape
    out = np.zeros(3, np.float64)
    grad = np.zeros((C, H, W), np.float32) if want_grad else None
    rc = lib().gsro_l1_ssim(C, H, W, _p(img, _f), _p(gt, _f), ctypes.c_float(lambda_dssim), _p(out, _d), _p(grad, _f))
    assert rc == 0
    return float(out[0]), float(out[1]), float(out[2]), grad


def knn_mean_dist2(points):
    """simple_knn distCUDA2: mean squared distance to the 3 nearest other points, (P,3) f32 -> (P,) f32."""
    pts = _f32(points)
    P = pts.shape[0]
    out = np.empty(P, np.float32)
    L = lib()
    L.gsro_knn_mean_dist2.restype = ctypes.c_int
    L.gsro_knn_mean_dist2.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.gsro_knn_mean_dist2(P, pts.ctypes.data, out.ctypes.data)
    assert rc == 0
    return out
