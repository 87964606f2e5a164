"""Leaf-parameter rasterization and fused Adam.

The reference renders from ACTIVATED copies of its optimiser leaves -- `pc.get_scaling = exp(_scaling)`,
`pc.get_rotation = normalize(_rotation)`, `pc.get_opacity = sigmoid(_opacity)`, `pc.get_features =
cat(_features_dc, _features_rest)` (scene/gaussian_model.py:114-135, gaussian_renderer/__init__.py:59-86)
-- and autograd walks those four ops back after the rasterizer's backward: ~0.9 KB of HBM traffic per
Gaussian per step around a rasterizer that itself moves ~1.3 KB per Gaussian.  Here the per-Gaussian
kernels read the leaves directly and write gradients w.r.t. the leaves (include/gsr.h
gsr_forward_preprocess_leaf / gsr_backward_leaf); no activated tensor is ever materialised.

`FusedAdam` is torch.optim.Adam as gaussian_model.py:243-252 configures it (eps 1e-15, per-group lr,
one tensor per group) with `step()` done by ONE kernel over all groups (include/gsr.h gsr_adam_step).
It keeps torch.optim.Adam's state layout (`state[p] = {"step", "exp_avg", "exp_avg_sq"}`), so the
reference's densification code that edits the optimiser state (gaussian_model.py:412-460) works on it
unchanged.  HIP tensors only -- no CPU fallback.
"""
import ctypes

import torch

from diff_gaussian_rasterization import _C

_vp, _i, _i64, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double
ADAM_MAX_GROUPS = 8


class AdamGroup(ctypes.Structure):
    """include/gsr.h gsr_adam_group"""
    _fields_ = [("param", _vp), ("grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("numel", _i64), ("step", _i64),
                ("lr", _d), ("row", ctypes.c_int32)]


def _lib():
    L = _C.lib()
    if not getattr(L, "_gsr_leaf_bound", False):
        L.gsr_forward_preprocess_leaf.restype = _i
        L.gsr_forward_preprocess_leaf.argtypes = [_i] * 5 + [_vp] * 5 + [_f] + [_vp] * 4 + [_f, _f, _i, _vp, _vp,
                                                                                           ctypes.POINTER(_i64), _vp, _i]
        L.gsr_backward_leaf.restype = _i
        L.gsr_backward_leaf.argtypes = [_i, _i, _i, _i64, _i, _i] + [_vp] * 5 + [_f] + [_vp] * 4 + [_f, _f] + [_vp] * 15 + [_i]
        L.gsr_adam_step.restype = _i
        L.gsr_adam_step.argtypes = [_i, ctypes.POINTER(AdamGroup), _d, _d, _d, _vp, _vp]
        L._gsr_leaf_bound = True
    return L


def _f32(t, dev, what):
    if not t.is_cuda or t.device != dev:
        raise RuntimeError(f"{what} must be a HIP (cuda) tensor on {dev}; leaf mode has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{what} must be float32 (got {t.dtype})")
    return t.contiguous()


def leaf_forward(xyz, features_dc, features_rest, opacity, scaling, rotation, raster_settings):
    """Forward from the raw leaves -> (num_rendered, color, radii, geom, binning, img, M, contiguous inputs)."""
    if xyz.ndimension() != 2 or xyz.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    L = _lib()
    dev = xyz.device
    st = raster_settings
    P, H, W = int(xyz.size(0)), int(st.image_height), int(st.image_width)
    xyz, features_dc, opacity, scaling, rotation = (_f32(t, dev, n) for t, n in (
        (xyz, "xyz"), (features_dc, "features_dc"), (opacity, "opacity"), (scaling, "scaling"), (rotation, "rotation")))
    M = 1 + (int(features_rest.size(1)) if features_rest.numel() else 0)
    if features_dc.shape != (P, 1, 3) or (M > 1 and features_rest.shape != (P, M - 1, 3)):
        raise RuntimeError(f"features_dc must be (P,1,3) and features_rest (P,M-1,3); got {tuple(features_dc.shape)}, "
                           f"{tuple(features_rest.shape)}")
    features_rest = _f32(features_rest, dev, "features_rest") if M > 1 else features_rest
    bg, view, proj, campos = (_f32(t, dev, n) for t, n in ((st.bg, "bg"), (st.viewmatrix, "viewmatrix"),
                                                           (st.projmatrix, "projmatrix"), (st.campos, "campos")))
    byte = dict(dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        color = (torch.zeros if P == 0 else torch.empty)((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        geom = torch.empty((L.gsr_geometry_bytes(P) if P else 0,), **byte)
        img = torch.empty((L.gsr_image_bytes(W, H) if P else 0,), **byte)
        binning = torch.empty((0,), **byte)
        R = 0
        if P:
            Rv = _i64(0)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _C._check(L.gsr_forward_preprocess_leaf(
                P, int(st.sh_degree), M, W, H, xyz.data_ptr(), features_dc.data_ptr(), _C._ptr(features_rest),
                opacity.data_ptr(), scaling.data_ptr(), float(st.scale_modifier), rotation.data_ptr(), view.data_ptr(),
                proj.data_ptr(), campos.data_ptr(), float(st.tanfovx), float(st.tanfovy), int(bool(st.prefiltered)),
                radii.data_ptr(), geom.data_ptr(), ctypes.byref(Rv), stream, _C._dbg(st.debug)))
            R = int(Rv.value)
            binning = torch.empty((L.gsr_binning_bytes(P, R, W, H),), **byte)
            _C._check(L.gsr_forward_render(P, R, W, H, bg.data_ptr(), radii.data_ptr(), geom.data_ptr(), _C._ptr(binning),
                                           img.data_ptr(), color.data_ptr(), stream, _C._dbg(st.debug)))
    return R, color, radii, geom, binning, img, M, (xyz, features_dc, features_rest, scaling, rotation)


def leaf_backward_args(st, R, M, xyz, features_dc, features_rest, scaling, rotation, radii, geom, binning, img, scratch, grad_color):
    """gsr_backward_args (leaf = 1) for a state produced by leaf_forward(); gradient pointers still unset."""
    dev = xyz.device
    bg, view, proj, campos = (_f32(t, dev, "settings") for t in (st.bg, st.viewmatrix, st.projmatrix, st.campos))
    a = _C.backward_args(P=int(xyz.size(0)), D=int(st.sh_degree), M=M, R=R, W=int(st.image_width), H=int(st.image_height), leaf=1,
                         background=bg, means3D=xyz, shs=features_dc, shs_rest=features_rest, scales=scaling,
                         scale_modifier=st.scale_modifier, rotations=rotation, viewmatrix=view, projmatrix=proj, cam_pos=campos,
                         tan_fovx=st.tanfovx, tan_fovy=st.tanfovy, radii=radii, geometry=geom, binning=binning, image=img,
                         scratch=scratch, dL_dpix=grad_color, debug=st.debug, device=dev)
    a._keep = (bg, view, proj, campos)  # alive as long as the struct
    return a


class _RasterizeLeafGaussians(torch.autograd.Function):
    """(xyz, means2D, _features_dc, _features_rest, _opacity, _scaling, _rotation) -> (color, radii); the same
    contract as diff_gaussian_rasterization._RasterizeGaussians with the activations folded in.
    `stats` (optional): (xyz_gradient_accum, denom, max_radii2D) float32 [P] tensors updated in place by the
    backward for the Gaussians visible in this view (train.py:157-159, gaussian_model.py:599-602)."""

    @staticmethod
    def forward(ctx, xyz, means2D, features_dc, features_rest, opacity, scaling, rotation, raster_settings, stats=None):
        R, color, radii, geom, binning, img, M, (xyz, features_dc, features_rest, scaling, rotation) = leaf_forward(
            xyz, features_dc, features_rest, opacity, scaling, rotation, raster_settings)
        ctx.raster_settings, ctx.num_rendered, ctx.M, ctx.stats = raster_settings, R, M, stats
        ctx.save_for_backward(xyz, features_dc, features_rest, scaling, rotation, radii, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)   # no zero tensor for the radii output on the way back
        return color, radii

    @staticmethod
    def backward(ctx, grad_color, _):
        L = _lib()
        st, R, M = ctx.raster_settings, ctx.num_rendered, ctx.M
        xyz, features_dc, features_rest, scaling, rotation, radii, geom, binning, img = ctx.saved_tensors
        dev = xyz.device
        if grad_color is None:
            grad_color = torch.zeros((3, int(st.image_height), int(st.image_width)), dtype=torch.float32, device=dev)
        P = int(xyz.size(0))
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            alloc = torch.zeros if P == 0 else torch.empty
            d_means2D, d_xyz = alloc((P, 3), **f32), alloc((P, 3), **f32)
            d_dc, d_rest = alloc((P, 1, 3), **f32), alloc((P, M - 1, 3), **f32)
            d_opacity, d_scaling, d_rotation = alloc((P, 1), **f32), alloc((P, 3), **f32), alloc((P, 4), **f32)
            if P:
                grad_color = _f32(grad_color, dev, "dL_dout_color")
                scratch = torch.empty((L.gsr_backward_scratch_bytes(P, R),), dtype=torch.uint8, device=dev)
                a = leaf_backward_args(st, R, M, xyz, features_dc, features_rest, scaling, rotation, radii, geom, binning, img,
                                       scratch, grad_color)
                _C.set_backward_outputs(a, dL_dmean2D=d_means2D, dL_dmean3D=d_xyz, dL_dsh=d_dc, dL_dsh_rest=d_rest,
                                        dL_dopacity=d_opacity, dL_dscale=d_scaling, dL_drot=d_rotation)
                _C.set_backward_stats(a, ctx.stats, P, dev)
                _C.backward_blend(a)
                _C.backward_gaussians(a, 0, P, 0)
                scratch.record_stream(torch.cuda.current_stream(dev))
        return d_xyz, d_means2D, d_dc, d_rest, d_opacity, d_scaling, d_rotation, None, None


def rasterize_leaf_gaussians(xyz, means2D, features_dc, features_rest, opacity, scaling, rotation, raster_settings, stats=None):
    """Equivalent to
        GaussianRasterizer(raster_settings)(means3D=xyz, means2D=means2D, shs=cat(features_dc, features_rest, 1),
            opacities=sigmoid(opacity), scales=exp(scaling), rotations=normalize(rotation))
    -> (color (3,H,W), radii (P,) int32)."""
    return _RasterizeLeafGaussians.apply(xyz, means2D, features_dc, features_rest, opacity, scaling, rotation, raster_settings, stats)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (no weight decay / amsgrad / maximize) whose step() is one HIP kernel over all groups.

    Built like the reference builds its optimiser: `FusedAdam(l, lr=0.0, eps=1e-15)` with
    `l = [{'params': [t], 'lr': ..., 'name': ...}, ...]` (gaussian_model.py:243-252).
    `step(visible_radii=radii)` is the opt-in visible-only variant (NOT the reference's update rule):
    Gaussians with radii <= 0 keep parameters and moments untouched; it needs every tensor's dim 0 = P."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None, visible_radii=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        batches = {}  # (device, betas, eps) -> [AdamGroup]
        keep = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous float32 HIP (cuda) parameters; there is no CPU path")
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                state = self.state[p]
                if len(state) == 0 or "exp_avg" not in state:
                    state["step"] = state.get("step", 0)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                step = int(state["step"]) + 1  # an int or a 0-dim tensor (states edited by densification code)
                state["step"] = step
                g = p.grad.contiguous()
                m, v = state["exp_avg"], state["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous()):
                    raise RuntimeError("FusedAdam: optimizer state tensors must be contiguous")
                keep.append(g)
                row = p.numel() // p.size(0) if (p.dim() > 0 and p.size(0) > 0) else 1
                key = (p.device, tuple(float(b) for b in group["betas"]), float(group["eps"]))
                batches.setdefault(key, []).append(AdamGroup(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), step,
                                                             float(group["lr"]), row))
        L = _lib()
        for (dev, betas, eps), groups in batches.items():
            radii_ptr = None
            if visible_radii is not None:
                if visible_radii.dtype != torch.int32 or visible_radii.device != dev or not visible_radii.is_contiguous():
                    raise RuntimeError("visible_radii must be a contiguous int32 tensor on the parameters' device")
                radii_ptr = visible_radii.data_ptr()
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                for k in range(0, len(groups), ADAM_MAX_GROUPS):
                    chunk = groups[k:k + ADAM_MAX_GROUPS]
                    arr = (AdamGroup * len(chunk))(*chunk)
                    _C._check(L.gsr_adam_step(len(chunk), arr, betas[0], betas[1], eps, radii_ptr, stream))
        return loss
