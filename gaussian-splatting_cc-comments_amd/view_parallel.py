"""View-parallel data parallelism for the rasterizer hot path.

The reference is single-GPU, one view per step (train.py:105-119).  A rasterizer call depends on
one camera and the full, replicated Gaussian set, so views shard across ranks with no data-path
collective; only the parameter gradients (59 floats per Gaussian: xyz 3 + SH 48 + opacity 1 +
scale 3 + rotation 4) are summed once per step (`torch.distributed` backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests).  Summing G views before one optimiser step is the same maths as
running the reference G times with `optimizer.step()` deferred.

`rasterize_view_parallel()` is the rasterizer call of that mode: the forward of the drop-in
`GaussianRasterizer`, and a backward that runs the per-Gaussian stage part by part
(include/gsr.h gsr_backward_blend / gsr_backward_gaussians) and starts the exchange of a finished
part while the next part computes (`GradientExchange`).  Nothing here is process-global: the mode,
the buffers and the process group travel in the `GradientExchange` object the caller passes.
"""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def shard_views(num_views: int, rank: int, world_size: int) -> List[int]:
    """Round-robin view assignment: rank r renders views r, r + world, r + 2*world, ..."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of size {world_size}")
    return list(range(rank, num_views, world_size))


def _world(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


class GradientBucket:
    """One flat fp32 buffer holding the gradients of all parameter tensors, all-reduced in ONE
    collective (xGMI is point-to-point: few large messages, not many small ones)."""

    def __init__(self, params: Iterable[torch.Tensor]):
        self.params = list(params)
        if not self.params:
            raise ValueError("no parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, device=dev, dtype=dt)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def pack(self):
        grads = [p.grad for p in self.params]
        if all(g is not None and g.data_ptr() != v.data_ptr() for g, v in zip(grads, self.views)):
            torch.cat([g.reshape(-1) for g in grads], out=self.flat)  # one launch for all tensors
            return
        for g, v in zip(grads, self.views):
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():  # already aliasing its slice (all_reduce() leaves p.grad = view)
                v.copy_(g)

    def all_reduce(self, group: Optional[dist.ProcessGroup] = None, average: bool = False, async_op: bool = False):
        """Sum (or average) the packed gradients over all ranks; afterwards every p.grad aliases its
        slice of the flat buffer, so no unpack copy is needed.  With async_op the caller must wait() on the
        returned work before reading the gradients; averaging is then done up front (the bucket is
        pre-scaled by 1/world, which commutes with the sum)."""
        self.pack()
        work = None
        world = _world(group)
        if world > 1:
            if average:
                self.flat.div_(world)
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        for p, v in zip(self.params, self.views):
            p.grad = v
        return work


# ---- the per-step gradient exchange, pipelined with the per-Gaussian backward ------------------------------
_SMALL = (("dL_dmean3D", 3), ("dL_dopacity", 1), ("dL_dscale", 3), ("dL_drot", 4))  # 11 floats per Gaussian


class GradientExchange:
    """Buffers and collectives of one rank's view-parallel step for P Gaussians with M SH coefficients.

    The Gaussians are cut into `parts` row ranges (starts are multiples of 256).  For part k the per-Gaussian
    backward kernel writes its outputs straight into that part's buffers (no pack copy):
      * `bucket[k]`: one flat tensor [11 * count_k] = dL_dmean3D | dL_dopacity | dL_dscale | dL_drot rows of the part
                     -> ONE all-reduce;
      * sh_mode "compact":   `rgb[k]` (count_k + 1, 3) = the view's clamp-masked dL/dRGB rows, the camera position as
                             trailer row -> ONE all-gather; the summed SH gradient of the part is rebuilt locally as
                             sum_v basis(dir_v) x dL/dRGB_v (include/gsr.h gsr_sh_grad_from_views): 12 B instead of
                             192 B per Gaussian per view on the wire;
      * sh_mode "allreduce": the (count_k, M, 3) rows of the SH gradient itself -> a second all-reduce.
    submit(k) starts part k's collectives asynchronously (they run on the backend's own stream while the next
    part's kernel computes); finish() waits and returns the summed gradients as whole-scene tensors."""

    def __init__(self, P: int, M: int, device, group: Optional[dist.ProcessGroup] = None, sh_mode: str = "compact",
                 parts: int = 2):
        if sh_mode not in ("compact", "allreduce"):
            raise ValueError(f"sh_mode {sh_mode!r}")
        if int(P) < 0:
            raise ValueError(f"P = {P}")
        self.P, self.M, self.group, self.sh_mode = int(P), int(M), group, sh_mode
        self.world = _world(group)
        self.device = torch.device(device)
        nparts = max(1, min(int(parts), max(1, self.P // 256)))
        step = -(-self.P // nparts)
        step = max(256, -(-step // 256) * 256)   # P == 0 (an empty scene, rasterize_points.cu:94) gives the one empty part (0, 0)
        self.ranges = [(f, min(step, self.P - f)) for f in range(0, self.P, step)] or [(0, 0)]
        f32 = dict(dtype=torch.float32, device=self.device)
        self.bucket = [torch.zeros(11 * c, **f32) for _, c in self.ranges]
        self.rgb = [torch.zeros((c + 1, 3), **f32) for _, c in self.ranges] if sh_mode == "compact" else None
        self.rgb_all = [torch.zeros((self.world * (c + 1), 3), **f32) for _, c in self.ranges] \
            if (sh_mode == "compact" and self.world > 1) else None
        self.dsh = None
        self._works = []

    # -- where part k's kernel writes ------------------------------------------------------------------
    def begin_step(self):
        f32 = dict(dtype=torch.float32, device=self.device)
        self._works = []
        # a fresh tensor per step: it is handed to autograd as the gradient of the SH input
        self.dsh = torch.empty((self.P, self.M, 3), **f32)

    def sections(self, k):
        """name -> tensor view of bucket[k] (each contiguous, rows of the part)."""
        c = self.ranges[k][1]
        out, off = {}, 0
        for name, w in _SMALL:
            out[name] = self.bucket[k][off:off + w * c].view(c, w)
            off += w * c
        return out

    def output_pointers(self, k):
        """field -> address for gsr_backward_gaussians(first_k, count_k, out_row0 = first_k)."""
        first, c = self.ranges[k]
        ptrs = {n: t.data_ptr() for n, t in self.sections(k).items()}
        if self.sh_mode == "compact":
            ptrs["dL_dcolor"] = self.rgb[k].data_ptr()   # clamp-masked dL/dRGB of the part
            ptrs["dL_dsh"] = None                         # not produced
        else:
            ptrs["dL_dsh"] = self.dsh.data_ptr() + first * self.M * 3 * 4
        return ptrs

    # -- collectives ---------------------------------------------------------------------------------------
    def submit(self, k, campos):
        first, c = self.ranges[k]
        if self.sh_mode == "compact":
            self.rgb[k][c] = campos.reshape(3)
        if self.world == 1 or c == 0:
            return
        if self.sh_mode == "compact":
            # output in the concatenated layout (world * n along dim 0): accepted by both RCCL and gloo
            self._works.append(dist.all_gather_into_tensor(self.rgb_all[k], self.rgb[k], group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(self.dsh[first:first + c], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._works.append(dist.all_reduce(self.bucket[k], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self, means3D, sh_degree, rebuild_sh=True):
        """-> dict of summed whole-scene gradients: dL_dmean3D (P,3), dL_dopacity (P,1), dL_dscale (P,3),
        dL_drot (P,4), dL_dsh (P,M,3).  rebuild_sh=False (plumbing rehearsals without the HIP library) leaves dL_dsh
        unset in "compact" mode."""
        from diff_gaussian_rasterization import _C
        for w in self._works:
            w.wait()
        self._works = []
        if self.sh_mode == "compact" and rebuild_sh:
            for k, (first, c) in enumerate(self.ranges):
                if c == 0:
                    continue
                if self.world > 1:
                    blocks = self.rgb_all[k].view(self.world, c + 1, 3)
                else:
                    blocks = self.rgb[k].view(1, c + 1, 3)
                # strided views of the gathered blocks: consumed in place by the kernel (view_stride)
                _C.sh_grad_from_views(means3D[first:first + c], blocks[:, c, :].contiguous(), blocks[:, :c, :], sh_degree, self.M,
                                      out=self.dsh[first:first + c])
        out = {"dL_dsh": self.dsh}
        secs = [self.sections(k) for k in range(len(self.ranges))]
        for name, _ in _SMALL:
            out[name] = secs[0][name].clone() if len(secs) == 1 else torch.cat([s[name] for s in secs], dim=0)
        return out


class _RasterizeViewParallel(torch.autograd.Function):
    """Forward of the drop-in rasterizer (SH colours, scales + rotations); backward pipelined with the exchange.
    The gradients it returns for means3D / shs / opacities / scales / rotations are already summed over the ranks'
    views; dL_dmeans2D (the densification carrier) stays this view's own."""

    @staticmethod
    def forward(ctx, means3D, means2D, shs, opacities, scales, rotations, raster_settings, exchange, stats):
        from diff_gaussian_rasterization import _C
        st = raster_settings
        e = torch.empty(0, device=means3D.device)
        R, color, radii, geom, binning, img = _C.rasterize_gaussians(
            st.bg, means3D, e, opacities, scales, rotations, st.scale_modifier, e, st.viewmatrix, st.projmatrix, st.tanfovx,
            st.tanfovy, st.image_height, st.image_width, shs, st.sh_degree, st.campos, st.prefiltered, st.debug)
        ctx.st, ctx.R, ctx.exchange, ctx.stats = st, R, exchange, stats
        ctx.save_for_backward(means3D, shs, scales, rotations, radii, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)   # no zero tensor for the radii output on the way back
        return color, radii

    @staticmethod
    def backward(ctx, grad_color, _):
        from diff_gaussian_rasterization import _C
        st, R, ex = ctx.st, ctx.R, ctx.exchange
        means3D, shs, scales, rotations, radii, geom, binning, img = ctx.saved_tensors
        dev = means3D.device
        if grad_color is None:
            grad_color = torch.zeros((3, int(st.image_height), int(st.image_width)), dtype=torch.float32, device=dev)
        P, M = int(means3D.size(0)), int(shs.size(1))
        if (P, M) != (ex.P, ex.M):
            raise RuntimeError(f"GradientExchange was built for P={ex.P}, M={ex.M}; got P={P}, M={M}")
        with torch.cuda.device(dev):
            d_means2D = torch.empty((P, 3), dtype=torch.float32, device=dev)
            ex.begin_step()
            tensors = [_C._dev_f32(t, dev, "input") for t in (means3D, shs, scales, rotations, grad_color, st.bg, st.viewmatrix,
                                                             st.projmatrix, st.campos)]
            means3D_c, shs_c, scales_c, rot_c, dpix, bg, view, proj, campos = tensors
            scratch = torch.empty((_C.lib().gsr_backward_scratch_bytes(P, int(R)),), dtype=torch.uint8, device=dev)
            a = _C.backward_args(P=P, D=int(st.sh_degree), M=M, R=int(R), W=int(st.image_width), H=int(st.image_height), leaf=0,
                                 background=bg, means3D=means3D_c, shs=shs_c, scales=scales_c, scale_modifier=st.scale_modifier,
                                 rotations=rot_c, viewmatrix=view, projmatrix=proj, cam_pos=campos, tan_fovx=st.tanfovx,
                                 tan_fovy=st.tanfovy, radii=radii, geometry=geom, binning=binning, image=img, scratch=scratch,
                                 dL_dpix=dpix, debug=st.debug, device=dev)
            _C.set_backward_stats(a, ctx.stats, P, dev)
            # required by the C ABI even when the part buffers are set below (validation happens per call)
            _C.set_backward_outputs(a, dL_dmean2D=d_means2D, **ex.output_pointers(0))
            _C.backward_blend(a)
            for k, (first, count) in enumerate(ex.ranges):
                _C.set_backward_outputs(a, dL_dmean2D=d_means2D.data_ptr() + first * 12, **ex.output_pointers(k))
                _C.backward_gaussians(a, first, count, first)
                ex.submit(k, campos)   # part k's collectives run while part k+1 computes
            scratch.record_stream(torch.cuda.current_stream(dev))
            g = ex.finish(means3D_c.detach(), int(st.sh_degree))
        return g["dL_dmean3D"], d_means2D, g["dL_dsh"], g["dL_dopacity"], g["dL_dscale"], g["dL_drot"], None, None, None


def rasterize_view_parallel(means3D, means2D, shs, opacities, scales, rotations, raster_settings, exchange, stats=None):
    """GaussianRasterizer(raster_settings)(means3D=..., means2D=..., shs=..., opacities=..., scales=..., rotations=...)
    for ONE view of a view-parallel step: same (color, radii); after backward the parameter gradients are the
    sums over all ranks' views (exchange: GradientExchange; stats: optional densification tensors, see
    DensificationStats.kernel_tensors())."""
    return _RasterizeViewParallel.apply(means3D, means2D, shs, opacities, scales, rotations, raster_settings, exchange, stats)


def exchange_sh_gradient(means3D: torch.Tensor, campos: torch.Tensor, dL_dRGB: torch.Tensor, sh_degree: int,
                         num_coeffs: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """SH gradient summed over the views of all ranks, from a 12-bytes-per-Gaussian exchange (synchronous,
    whole-scene form of GradientExchange's "compact" mode).

    dL/dsh of one view is basis(view direction) x dL/dRGB per Gaussian (backward.cu:45-96), so ranks
    all-gather their view's clamp-masked dL/dRGB (P,3) with the camera position as trailer row and every rank
    rebuilds  sum_v basis(dir_v) x dL/dRGB_v  locally with one kernel -- (7/8)*12*P bytes received per
    rank instead of the 2*(7/8)*192*P bytes of an all-reduce of the (P,16,3) gradient.  xGMI is
    point-to-point, so bytes per link, not launches, set the time.  Returns (P, num_coeffs, 3)."""
    from diff_gaussian_rasterization import _C
    world = _world(group)
    P = dL_dRGB.numel() // 3
    if world == 1:
        return _C.sh_grad_from_views(means3D.detach(), campos.reshape(1, 3), dL_dRGB.reshape(1, P, 3), sh_degree, num_coeffs)
    mine = torch.empty((P + 1, 3), dtype=dL_dRGB.dtype, device=dL_dRGB.device)
    mine[:P] = dL_dRGB.reshape(P, 3)
    mine[P] = campos.reshape(3)
    gathered = torch.empty((world * (P + 1), 3), dtype=dL_dRGB.dtype, device=dL_dRGB.device)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    blocks = gathered.view(world, P + 1, 3)
    return _C.sh_grad_from_views(means3D.detach(), blocks[:, P, :].contiguous(), blocks[:, :P, :], sh_degree, num_coeffs)


def all_reduce_max_radii(radii: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Per-Gaussian MAX of the int32 screen radii over the views of one step (what
    train.py:157 `max_radii2D` consumes in view-parallel mode)."""
    out = radii.clone()
    if _world(group) > 1:
        dist.all_reduce(out, op=dist.ReduceOp.MAX, group=group)
    return out


class DensificationStats:
    """The training-loop statistics that consume the rasterizer's side outputs
    (train.py:157-159, scene/gaussian_model.py:599-602):

        max_radii2D[vis] = max(max_radii2D[vis], radii[vis])                      per view
        xyz_gradient_accum[vis] += || viewspace_points.grad[vis, :2] ||           per view
        denom[vis] += 1                                                           per view

    Per view they are updated ON THE DEVICE by the epilogue of the per-Gaussian backward kernel
    (`stats=self.kernel_tensors()` of the rasterizer calls; include/gsr.h gsr_backward_args.stat_*), into this rank's
    LOCAL accumulators -- no extra kernel, no extra pass over (P,3) tensors.  update() is the same bookkeeping in
    PyTorch ops for callers that only hold (viewspace_grad, radii).  The reference reads the statistics only every
    100 iterations (densification), so ranks fold their local accumulators together lazily: sync() does ONE float MAX
    and ONE float SUM all-reduce and leaves in max_radii2D / xyz_gradient_accum / denom what the reference would
    hold after applying its update once per view of every rank, in any order."""

    def __init__(self, num_points: int, device=None):
        self.P = int(num_points)
        self.max_radii2D = torch.zeros(self.P, device=device)
        self.xyz_gradient_accum = torch.zeros(self.P, 1, device=device)
        self.denom = torch.zeros(self.P, 1, device=device)
        self._local_sum = torch.zeros(2, self.P, device=device)   # row 0: gradient norms, row 1: visibility counts
        self._local_max = torch.zeros(self.P, device=device)

    def kernel_tensors(self):
        """(xyz_gradient_accum, denom, max_radii2D) views of the local accumulators for the kernel epilogue."""
        return self._local_sum[0], self._local_sum[1], self._local_max

    @torch.no_grad()
    def update(self, viewspace_grad: torch.Tensor, radii: torch.Tensor):
        vis = radii > 0
        self._local_sum[0] += torch.norm(viewspace_grad[:, :2], dim=-1) * vis
        self._local_sum[1] += vis.to(self._local_sum.dtype)
        # in place: kernel_tensors() hands out views of these accumulators, and rasterizer objects / autograd contexts keep
        # them across steps -- a rebound tensor would orphan what the kernel epilogue writes afterwards
        torch.maximum(self._local_max, torch.where(vis, radii, torch.zeros_like(radii)).to(self._local_max.dtype), out=self._local_max)

    @torch.no_grad()
    def sync(self, group: Optional[dist.ProcessGroup] = None):
        if _world(group) > 1:
            dist.all_reduce(self._local_max, op=dist.ReduceOp.MAX, group=group)
            dist.all_reduce(self._local_sum, op=dist.ReduceOp.SUM, group=group)
        self.max_radii2D = torch.max(self.max_radii2D, self._local_max)
        self.xyz_gradient_accum += self._local_sum[0].unsqueeze(1)
        self.denom += self._local_sum[1].unsqueeze(1)
        self._local_sum.zero_()
        self._local_max.zero_()

    def reset(self):
        """after densify_and_prune (gaussian_model.py:396-398 zeroes the accumulators)"""
        for t in (self.max_radii2D, self.xyz_gradient_accum, self.denom, self._local_sum, self._local_max):
            t.zero_()


class ViewsInFlight:
    """Several of a rank's views in flight on one GPU: forward + backward of `in_flight` consecutive views on as many streams.

    One view leaves the chip mostly idle for a fifth of its time -- the depth sort and the binning between the per-Gaussian kernel
    and the forward blend are chains of short, latency-bound launches -- and the two blend kernels end in tails of half-empty
    SIMDs.  The rasterizer keeps no state between calls and enqueues everything on the caller's stream, so a second view on
    another stream fills both: its stage 1 and binning run under the first view's blend kernels.  Views are independent until
    their gradients are added (train.py:105-119 renders one view per step), and PyTorch's autograd adds them into
    the parameters' .grad in the order of the backward calls -- the order of the sequential loop -- so the accumulated gradients
    are bit for bit those of rendering the views one after the other (tests/test_boundary_gpu.py).

        vif = ViewsInFlight(device, in_flight=2)
        vif.forward_backward(render_fns, upstream_grads)      # render_fns[v]() -> image of view v (calls the rasterizer)
    """

    def __init__(self, device, in_flight: int = 2, staggered: bool = False):
        if in_flight < 1:
            raise ValueError("in_flight must be at least 1")
        self.device = torch.device(device)
        # (equal priorities: with one of two streams at high priority the staggered step took 3.39 ms instead of 2.35, round 4)
        self.streams = [torch.cuda.Stream(self.device) for _ in range(in_flight)]
        # staggered: every view's forward AND backward are issued before the next view's forward, views alternating over the
        # streams -- a view's stage 1 and binning then run under the previous view's backward blend and its forward blend under
        # the previous view's per-Gaussian backward, instead of like phases of `in_flight` views running side by side
        self.staggered = staggered

    def forward_backward(self, render_fns, upstream_grads):
        """Forward and backward of every view, `in_flight` at a time; returns the images (detached).  Work issued before the call on
        the current stream is waited for by the side streams, and the current stream waits for them at the end."""
        cur = torch.cuda.current_stream(self.device)
        n = len(self.streams)
        images = []
        if self.staggered:
            for st in self.streams:
                st.wait_stream(cur)
            for v, f in enumerate(render_fns):
                st = self.streams[v % n]
                g = upstream_grads[v]
                g.record_stream(st)
                with torch.cuda.stream(st):
                    img = f()
                    img.backward(g)
                images.append(img.detach())
            for st in self.streams:
                cur.wait_stream(st)
            return images
        for first in range(0, len(render_fns), n):
            group = list(range(first, min(first + n, len(render_fns))))
            live = []
            for v, st in zip(group, self.streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    live.append(render_fns[v]())
            # the backward calls in view order: autograd adds a view's gradients into .grad when its backward runs
            for v, st, img in zip(group, self.streams, live):
                g = upstream_grads[v]
                g.record_stream(st)
                with torch.cuda.stream(st):
                    img.backward(g)
                images.append(img.detach())
            for st in self.streams[:len(group)]:
                cur.wait_stream(st)
        return images
