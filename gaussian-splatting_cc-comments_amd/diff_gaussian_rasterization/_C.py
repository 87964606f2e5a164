"""ctypes binding of libgsr_hip.so with the surface of the reference's torch extension
`diff_gaussian_rasterization._C` (submodules/diff-gaussian-rasterization/ext.cpp:18-22):

    rasterize_gaussians(...)          -> rasterize_points.cu:38-130  RasterizeGaussiansCUDA
    rasterize_gaussians_backward(...) -> rasterize_points.cu:132-216 RasterizeGaussiansBackwardCUDA
    mark_visible(...)                 -> rasterize_points.cu:218-237 markVisible

Same positional arguments, same return tuples.  PyTorch is only the allocator and the stream
provider here: every tensor is handed to the C ABI (include/gsr.h) as a raw device pointer.
There is no CPU fallback: a missing library or a non-HIP tensor raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(os.path.dirname(_HERE), "libgsr_hip.so")   # nothing in the environment changes this: see use_library()
_lib = None

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_f = ctypes.c_float
_sz = ctypes.c_size_t


class GeometryLayout(ctypes.Structure):
    _fields_ = [(n, _sz) for n in ("splat", "depth_keys", "depth_keys_alt", "perm", "perm_alt", "tiles_touched", "rect",
                                   "slot_base", "clamped", "sh_ddir", "status", "scan_temp", "sort_table", "col_table", "rshape", "total")]


class ImageLayout(ctypes.Structure):
    _fields_ = [(n, _sz) for n in ("final_C", "final_T", "n_contrib", "ranges", "tile_max_contrib", "tile_order", "total")]


class BinningLayout(ctypes.Structure):
    _fields_ = [(n, _sz) for n in ("point_list", "point_list_alt", "tile_keys", "tile_keys_alt", "sort_table",
                                   "checkpoints", "total", "tile_key_bytes", "column_pairs")]


class KernelTime(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("ms", _f)]


class BackwardArgs(ctypes.Structure):
    """include/gsr.h gsr_backward_args (the two-stage backward: gsr_backward_blend / gsr_backward_gaussians)"""
    _fields_ = ([("P", _i), ("D", _i), ("M", _i), ("num_rendered", _i64), ("width", _i), ("height", _i), ("leaf", _i)] +
                [(n, _vp) for n in ("background", "means3D", "shs", "shs_rest", "colors_precomp", "scales")] +
                [("scale_modifier", _f)] +
                [(n, _vp) for n in ("rotations", "cov3D_precomp", "viewmatrix", "projmatrix", "cam_pos")] +
                [("tan_fovx", _f), ("tan_fovy", _f)] +
                [(n, _vp) for n in ("radii", "geometry", "binning", "image", "scratch", "dL_dpix", "dL_dmean2D", "dL_dconic",
                                    "dL_dopacity", "dL_dcolor", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dsh_rest", "dL_dscale",
                                    "dL_drot", "stat_xyz_gradient_accum", "stat_denom", "stat_max_radii2D", "stream")] +
                [("debug", _i)])


# bits of the C ABI's `debug` mask (include/gsr.h GSR_DEBUG_*).  The reference's bool `debug` is DEBUG_SYNC; tests pass the
# diagnostic bits as an int in the same argument, per call -- nothing is read from the environment.
DEBUG_SYNC, DEBUG_NO_CULL, DEBUG_SERIAL, DEBUG_NO_SPLIT, DEBUG_TILE_SORT, DEBUG_RADIX_DEPTH, DEBUG_NO_TRIM = 1, 2, 4, 8, 16, 32, 64


def _dbg(debug):
    return int(debug) if (isinstance(debug, int) and not isinstance(debug, bool)) else int(bool(debug))


def library_path():
    return _LIB_PATH


def use_library(path):
    """Diagnostics (tools/tile_clock.py, bench.py --library): bind another build of the same C ABI -- the per-tile-clock twin, an
    A/B variant of one translation unit -- instead of the product library.  Must be called before the first lib(); the product
    never calls it."""
    global _LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_library() must be called before the library is first loaded")
    _LIB_PATH = os.path.abspath(path)


def lib():
    """Load libgsr_hip.so (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the rasterizer.")
    L = ctypes.CDLL(_LIB_PATH)
    L.gsr_last_error.restype = ctypes.c_char_p
    L.gsr_version.restype = ctypes.c_char_p
    L.gsr_geometry_bytes.restype = _sz
    L.gsr_geometry_bytes.argtypes = [_i]
    L.gsr_image_bytes.restype = _sz
    L.gsr_image_bytes.argtypes = [_i, _i]
    L.gsr_binning_bytes.restype = _sz
    L.gsr_binning_bytes.argtypes = [_i, _i64, _i, _i]
    L.gsr_backward_scratch_bytes.restype = _sz
    L.gsr_backward_scratch_bytes.argtypes = [_i, _i64]
    L.gsr_geometry_layout_of.argtypes = [_i, ctypes.POINTER(GeometryLayout)]
    L.gsr_image_layout_of.argtypes = [_i, _i, ctypes.POINTER(ImageLayout)]
    L.gsr_binning_layout_of.argtypes = [_i, _i64, _i, _i, ctypes.POINTER(BinningLayout)]
    L.gsr_get_higher_msb.restype = ctypes.c_uint32
    L.gsr_get_higher_msb.argtypes = [ctypes.c_uint32]
    L.gsr_forward_preprocess.restype = _i
    L.gsr_forward_preprocess.argtypes = [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp,
                                         _f, _f, _i, _vp, _vp, ctypes.POINTER(_i64), _vp, _i]
    L.gsr_forward_render.restype = _i
    L.gsr_forward_render.argtypes = [_i, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i]
    L.gsr_backward.restype = _i
    L.gsr_backward.argtypes = [_i, _i, _i, _i64, _i, _i] + [_vp] * 5 + [_f] + [_vp] * 5 + [_f, _f] + [_vp] * 16 + [_i]
    L.gsr_mark_visible.restype = _i
    L.gsr_mark_visible.argtypes = [_i, _vp, _vp, _vp, _vp, _vp]
    L.gsr_sh_grad_from_views.restype = _i
    L.gsr_sh_grad_from_views.argtypes = [_i, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp]
    L.gsr_backward_blend.restype = _i
    L.gsr_backward_blend.argtypes = [ctypes.POINTER(BackwardArgs)]
    L.gsr_backward_gaussians.restype = _i
    L.gsr_backward_gaussians.argtypes = [ctypes.POINTER(BackwardArgs), _i, _i, _i]
    L.gsr_thread_release.restype = _i
    L.gsr_thread_release.argtypes = []
    L.gsr_profile_begin.restype = _i
    L.gsr_profile_begin.argtypes = [_vp]
    L.gsr_profile_begin_only.restype = _i
    L.gsr_profile_begin_only.argtypes = [_vp, ctypes.c_char_p]
    L.gsr_profile_end.restype = _i
    L.gsr_profile_end.argtypes = [_vp, ctypes.POINTER(KernelTime), _i]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"gsr error {rc}: {lib().gsr_last_error().decode()}")


def _ptr(t):
    """Raw device pointer; an empty tensor is the reference's "not provided" and becomes NULL
    (rasterize_points.cu:108-125)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _dev_f32(t, device, what):
    if t.numel() == 0:
        return t
    if t.device != device:
        raise RuntimeError(f"{what} must be on {device} (got {t.device}); the HIP rasterizer has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{what} must be float32 (got {t.dtype})")
    return t.contiguous()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug):
    """-> (num_rendered, out_color (3,H,W) f32, radii (P,) i32, geomBuffer, binningBuffer, imgBuffer)"""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:60-63
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a HIP (cuda) tensor; the HIP rasterizer has no CPU path")
    L = lib()
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    means3D = _dev_f32(means3D, dev, "means3D")
    background = _dev_f32(background, dev, "bg")
    colors, opacity, scales, rotations, cov3D_precomp, sh = (
        _dev_f32(t, dev, n) for t, n in ((colors, "colors_precomp"), (opacity, "opacities"), (scales, "scales"),
                                         (rotations, "rotations"), (cov3D_precomp, "cov3D_precomp"), (sh, "shs")))
    viewmatrix, projmatrix, campos = (_dev_f32(t, dev, n) for t, n in ((viewmatrix, "viewmatrix"),
                                                                        (projmatrix, "projmatrix"), (campos, "campos")))
    byte = dict(dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        out_color = torch.zeros((3, H, W), dtype=torch.float32, device=dev) if P == 0 else \
            torch.empty((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        if P == 0:  # rasterize_points.cu:94
            e = torch.empty((0,), **byte)
            return 0, out_color, radii, e, e.clone(), e.clone()
        M = int(sh.size(1)) if sh.numel() != 0 else 0
        geom = torch.empty((L.gsr_geometry_bytes(P),), **byte)
        img = torch.empty((L.gsr_image_bytes(W, H),), **byte)
        R = _i64(0)
        stream = _stream(dev)
        _check(L.gsr_forward_preprocess(P, int(degree), M, W, H, _ptr(means3D), _ptr(sh), _ptr(colors), _ptr(opacity),
                                        _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
                                        _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), float(tan_fovx),
                                        float(tan_fovy), int(bool(prefiltered)), _ptr(radii), _ptr(geom),
                                        ctypes.byref(R), stream, _dbg(debug)))
        R = int(R.value)
        binning = torch.empty((L.gsr_binning_bytes(P, R, W, H),), **byte)
        _check(L.gsr_forward_render(P, R, W, H, _ptr(background), _ptr(radii), _ptr(geom), _ptr(binning), _ptr(img),
                                    _ptr(out_color), stream, _dbg(debug)))
    return R, out_color, radii, geom, binning, img


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree,
                                 campos, geomBuffer, R, binningBuffer, imageBuffer, debug, *, lean=False, skip_sh=False,
                                 debug_out=None, stats=None):
    """-> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)

    The 21 positional arguments and the tuple are the reference extension's.  Keyword-only extras (all per call,
    nothing is kept between calls):
      lean      outputs that autograd would drop anyway (dL_dcolors when the colours came from SH, dL_dcov3D when the
                covariances came from scales/rotations, the internal dL_dconic) are not computed: empty tensors
      skip_sh   view-parallel mode: dL_dsh is not produced (None) and dL_dcolors carries the clamp-masked dL/dRGB
                of the view, the input of sh_grad_from_views()
      debug_out dict that receives the internal "dL_dconic" tensor (tests)
      stats     (xyz_gradient_accum, denom, max_radii2D) float32 [P] tensors updated in place for the Gaussians
                visible in this view (train.py:157-159, gaussian_model.py:599-602); any of them may be None"""
    L = lib()
    dev = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if sh.numel() != 0 else 0
    f32 = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        # every element is written by gsr_backward (no zero-fill pass, unlike rasterize_points.cu:168-178)
        alloc = torch.zeros if P == 0 else torch.empty
        dL_dmeans3D = alloc((P, 3), **f32)
        dL_dmeans2D = alloc((P, 3), **f32)
        skip_sh = bool(skip_sh) and M > 0
        lean = bool(lean) and debug_out is None
        none = torch.empty((0,), **f32)
        dL_dcolors = alloc((P, 3), **f32) if (not lean or colors.numel() != 0 or skip_sh) else none
        dL_dconic = alloc((P, 2, 2), **f32) if not lean else none
        dL_dopacity = alloc((P, 1), **f32)
        dL_dcov3D = alloc((P, 6), **f32) if (not lean or cov3D_precomp.numel() != 0) else none
        dL_dsh = None if skip_sh else alloc((P, M, 3), **f32)
        dL_dscales = alloc((P, 3), **f32)
        dL_drotations = alloc((P, 4), **f32)
        if P != 0:
            means3D = _dev_f32(means3D, dev, "means3D")
            dL_dout_color = _dev_f32(dL_dout_color, dev, "dL_dout_color")
            background, colors, scales, rotations, cov3D_precomp, sh, viewmatrix, projmatrix, campos = (
                _dev_f32(t, dev, "input") for t in (background, colors, scales, rotations, cov3D_precomp, sh, viewmatrix,
                                                    projmatrix, campos))
            scratch = torch.empty((L.gsr_backward_scratch_bytes(P, int(R)),), dtype=torch.uint8, device=dev)
            if stats is None:
                _check(L.gsr_backward(P, int(degree), M, int(R), W, H, _ptr(background), _ptr(means3D), _ptr(sh), _ptr(colors),
                                      _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
                                      _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy),
                                      _ptr(radii), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(scratch),
                                      _ptr(dL_dout_color), _ptr(dL_dmeans2D), _ptr(dL_dconic), _ptr(dL_dopacity),
                                      _ptr(dL_dcolors), _ptr(dL_dmeans3D), _ptr(dL_dcov3D), _ptr(dL_dsh), _ptr(dL_dscales),
                                      _ptr(dL_drotations), _stream(dev), _dbg(debug)))
            else:
                a = backward_args(P=P, D=int(degree), M=M, R=int(R), W=W, H=H, leaf=0, background=background, means3D=means3D,
                                  shs=sh, colors_precomp=colors, scales=scales, scale_modifier=scale_modifier,
                                  rotations=rotations, cov3D_precomp=cov3D_precomp, viewmatrix=viewmatrix,
                                  projmatrix=projmatrix, cam_pos=campos, tan_fovx=tan_fovx, tan_fovy=tan_fovy, radii=radii,
                                  geometry=geomBuffer, binning=binningBuffer, image=imageBuffer, scratch=scratch,
                                  dL_dpix=dL_dout_color, debug=debug, device=dev)
                set_backward_outputs(a, dL_dmean2D=dL_dmeans2D, dL_dconic=dL_dconic, dL_dopacity=dL_dopacity,
                                     dL_dcolor=dL_dcolors, dL_dmean3D=dL_dmeans3D, dL_dcov3D=dL_dcov3D, dL_dsh=dL_dsh,
                                     dL_dscale=dL_dscales, dL_drot=dL_drotations)
                set_backward_stats(a, stats, P, dev)
                backward_blend(a)
                backward_gaussians(a, 0, P, 0)
            scratch.record_stream(torch.cuda.current_stream(dev))
    if debug_out is not None:
        debug_out["dL_dconic"] = dL_dconic
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


# ---- the backward in two stages (include/gsr.h gsr_backward_blend / gsr_backward_gaussians) ------------------
def backward_args(*, P, D, M, R, W, H, leaf, background, means3D, shs, scales, scale_modifier, rotations, viewmatrix,
                  projmatrix, cam_pos, tan_fovx, tan_fovy, radii, geometry, binning, image, scratch, dL_dpix, debug, device,
                  shs_rest=None, colors_precomp=None, cov3D_precomp=None):
    """Input side of a gsr_backward_args; the gradient pointers are set with set_backward_outputs().  The tensors must
    stay alive (and contiguous float32 on `device`) until the calls that use the struct have been enqueued."""
    a = BackwardArgs()
    a.P, a.D, a.M, a.num_rendered, a.width, a.height, a.leaf = int(P), int(D), int(M), int(R), int(W), int(H), int(leaf)
    for name, t in (("background", background), ("means3D", means3D), ("shs", shs), ("shs_rest", shs_rest),
                    ("colors_precomp", colors_precomp), ("scales", scales), ("rotations", rotations),
                    ("cov3D_precomp", cov3D_precomp), ("viewmatrix", viewmatrix), ("projmatrix", projmatrix),
                    ("cam_pos", cam_pos), ("radii", radii), ("geometry", geometry), ("binning", binning), ("image", image),
                    ("scratch", scratch), ("dL_dpix", dL_dpix)):
        setattr(a, name, _ptr(t))
    a.scale_modifier, a.tan_fovx, a.tan_fovy = float(scale_modifier), float(tan_fovx), float(tan_fovy)
    a.stream = _stream(device)
    a.debug = _dbg(debug)
    return a


def set_backward_outputs(a, **ptrs):
    """ptrs: field name -> tensor (its data_ptr), int address, or None.  With out_row0 = first, an address is where
    the row of Gaussian `first` goes."""
    for name, t in ptrs.items():
        setattr(a, name, t if (t is None or isinstance(t, int)) else _ptr(t))


def set_backward_stats(a, stats, P, device):
    if stats is None:
        return
    for name, t in zip(("stat_xyz_gradient_accum", "stat_denom", "stat_max_radii2D"), stats):
        if t is None:
            continue
        if t.device != device or t.dtype != torch.float32 or t.numel() != P or not t.is_contiguous():
            raise RuntimeError(f"{name} must be a contiguous float32 tensor with P = {P} elements on {device}")
        setattr(a, name, t.data_ptr())


def backward_blend(a):
    _check(lib().gsr_backward_blend(ctypes.byref(a)))


def backward_gaussians(a, first, count, out_row0=0):
    _check(lib().gsr_backward_gaussians(ctypes.byref(a), int(first), int(count), int(out_row0)))


def sh_grad_from_views(means3D, cam_pos, dL_dRGB, degree, M, out=None):
    """dL_dsh (P,M,3) summed over V views from their clamp-masked dL/dRGB (V,P,3) and camera
    positions (V,3): include/gsr.h gsr_sh_grad_from_views.  out: optional contiguous (P,M,3) destination (a row
    range of a larger tensor when the caller works part by part)."""
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a HIP (cuda) tensor; the HIP rasterizer has no CPU path")
    L = lib()
    dev = means3D.device
    P, V = int(means3D.size(0)), int(dL_dRGB.size(0))
    assert dL_dRGB.shape == (V, P, 3) and cam_pos.shape == (V, 3)
    # views may be strided along dim 0 (blocks of an all-gather with a trailer row): consumed in place
    if P and not (dL_dRGB.stride(2) == 1 and dL_dRGB.stride(1) == 3 and (V <= 1 or dL_dRGB.stride(0) >= 3 * P)):
        dL_dRGB = dL_dRGB.contiguous()
    view_stride = int(dL_dRGB.stride(0)) if (P and V > 1) else 0
    if dL_dRGB.device != dev or dL_dRGB.dtype != torch.float32:
        raise RuntimeError("dL_dRGB must be a float32 tensor on the device of means3D")
    means3D, cam_pos = (_dev_f32(t, dev, n) for t, n in ((means3D, "means3D"), (cam_pos, "cam_pos")))
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty((P, M, 3), dtype=torch.float32, device=dev)
        elif out.shape != (P, M, 3) or out.dtype != torch.float32 or out.device != dev or not out.is_contiguous():
            raise RuntimeError("sh_grad_from_views: out must be a contiguous float32 (P,M,3) tensor on the device of means3D")
        _check(L.gsr_sh_grad_from_views(P, int(degree), int(M), V, _ptr(means3D), _ptr(cam_pos), _ptr(dL_dRGB), view_stride,
                                        _ptr(out), _stream(dev)))
    return out


def mark_visible(means3D, viewmatrix, projmatrix):
    """-> bool (P,)"""
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a HIP (cuda) tensor; the HIP rasterizer has no CPU path")
    L = lib()
    dev = means3D.device
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        means3D = _dev_f32(means3D, dev, "means3D")
        viewmatrix = _dev_f32(viewmatrix, dev, "viewmatrix")
        projmatrix = _dev_f32(projmatrix, dev, "projmatrix")
        with torch.cuda.device(dev):
            _check(L.gsr_mark_visible(P, _ptr(means3D), _ptr(viewmatrix), _ptr(projmatrix), _ptr(present), _stream(dev)))
    return present


def thread_release():
    """include/gsr.h gsr_thread_release: frees the calling thread's helper stream, events and pinned buffer."""
    _check(lib().gsr_thread_release())


# ---- introspection helpers (tests / bench; not part of the reference surface) ----------------------
def geometry_layout(P):
    o = GeometryLayout()
    _check(lib().gsr_geometry_layout_of(P, ctypes.byref(o)))
    return o


def image_layout(W, H):
    o = ImageLayout()
    _check(lib().gsr_image_layout_of(W, H, ctypes.byref(o)))
    return o


def binning_layout(P, R, W, H):
    o = BinningLayout()
    _check(lib().gsr_binning_layout_of(P, R, W, H, ctypes.byref(o)))
    return o


def profile_begin(only=None, device=None):
    """Start recording per-stage HIP events for calls made on `device`'s current stream; `only` = name of the
    single stage to record."""
    stream = _stream(torch.device("cuda", torch.cuda.current_device()) if device is None else device)
    _check(lib().gsr_profile_begin_only(stream, None if only is None else only.encode()))


def profile_end(capacity=256, device=None):
    stream = _stream(torch.device("cuda", torch.cuda.current_device()) if device is None else device)
    arr = (KernelTime * capacity)()
    n = lib().gsr_profile_end(stream, arr, capacity)
    return [(arr[k].name.decode(), float(arr[k].ms)) for k in range(n)]
