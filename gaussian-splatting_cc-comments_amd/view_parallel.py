"""View-parallel data parallelism for the rasterizer hot path.

The reference is single-GPU, one view per step (train.py:105-119).  A rasterizer call depends on
one camera and the full, replicated Gaussian set, so views shard across ranks with no data-path
collective; only the parameter gradients (59 floats per Gaussian: xyz 3 + SH 48 + opacity 1 +
scale 3 + rotation 4) are summed, once per step, with one all-reduce over a single flat bucket
(`torch.distributed` backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).
Summing G views before one optimiser step is the same maths as running the reference G times
with `optimizer.step()` deferred.
"""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def shard_views(num_views: int, rank: int, world_size: int) -> List[int]:
    """Round-robin view assignment: rank r renders views r, r + world, r + 2*world, ..."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of size {world_size}")
    return list(range(rank, num_views, world_size))


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
        slice of the flat buffer, so no unpack copy is needed."""
        self.pack()
        work = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
            if average and not async_op:
                self.flat.div_(dist.get_world_size(group))
        for p, v in zip(self.params, self.views):
            p.grad = v
        return work


class skip_sh_gradient:
    """Context manager for the backward of ONE view in view-parallel mode: the rasterizer does not
    produce dL_dsh (48 of the 59 gradient floats per Gaussian); instead the clamp-masked dL/dRGB of
    the view (3 floats per Gaussian) is collected in `.dL_dRGB` for exchange_sh_gradient()."""

    def __enter__(self):
        from diff_gaussian_rasterization import _C
        self._C = _C
        self._prev = _C.SKIP_SH_GRAD
        _C.SKIP_SH_GRAD = True
        _C.view_parallel_last.pop("dL_dRGB", None)
        self.dL_dRGB = None
        return self

    def __exit__(self, *exc):
        self._C.SKIP_SH_GRAD = self._prev
        self.dL_dRGB = self._C.view_parallel_last.pop("dL_dRGB", None)
        return False


class ShExchange:
    """In-flight exchange of one step's SH-gradient inputs (see exchange_sh_gradient).  start() issues ONE
    asynchronous all-gather of a (P+1, 3) block per rank -- the view's clamp-masked dL/dRGB with the camera
    position as its last row -- so other work (packing and all-reducing the remaining gradients) can be
    enqueued behind it; finish() waits for it and rebuilds the summed (P, M, 3) gradient with one kernel,
    which then overlaps the all-reduce running on the collective's own stream."""

    def __init__(self, campos: torch.Tensor, dL_dRGB: torch.Tensor, group: Optional[dist.ProcessGroup] = None):
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.P = P = dL_dRGB.numel() // 3
        self.work = None
        if self.world > 1:
            mine = torch.empty((P + 1, 3), dtype=dL_dRGB.dtype, device=dL_dRGB.device)
            mine[:P] = dL_dRGB.reshape(P, 3)
            mine[P] = campos.reshape(3)
            # output in the concatenated layout (world * n along dim 0): accepted by both RCCL and gloo
            self.all = torch.empty((self.world * (P + 1), 3), dtype=dL_dRGB.dtype, device=dL_dRGB.device)
            self.mine = mine  # kept alive until finish()
            self.work = dist.all_gather_into_tensor(self.all, mine, group=group, async_op=True)
        else:
            self.rgb_all, self.cam_all = dL_dRGB.reshape(1, P, 3), campos.reshape(1, 3)

    def finish(self, means3D: torch.Tensor, sh_degree: int, num_coeffs: int) -> torch.Tensor:
        from diff_gaussian_rasterization import _C
        if self.work is not None:
            self.work.wait()
            blocks = self.all.view(self.world, self.P + 1, 3)
            self.rgb_all = blocks[:, :self.P, :]          # (world, P, 3): strided view, made contiguous by the binding
            self.cam_all = blocks[:, self.P, :].contiguous()
            self.work = None
        return _C.sh_grad_from_views(means3D.detach(), self.cam_all, self.rgb_all, sh_degree, num_coeffs)


def exchange_sh_gradient(means3D: torch.Tensor, campos: torch.Tensor, dL_dRGB: torch.Tensor, sh_degree: int,
                         num_coeffs: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """SH gradient summed over the views of all ranks, from a 12-bytes-per-Gaussian exchange.

    dL/dsh of one view is basis(view direction) x dL/dRGB per Gaussian (backward.cu:45-96), so ranks
    all-gather their view's clamp-masked dL/dRGB (P,3) and camera position (3,) and every rank
    rebuilds  sum_v basis(dir_v) x dL/dRGB_v  locally with one kernel -- (7/8)*12*P bytes received per
    rank instead of the 2*(7/8)*192*P bytes of an all-reduce of the (P,16,3) gradient.  xGMI is
    point-to-point, so bytes per link, not launches, set the time.  Returns (P, num_coeffs, 3).
    (Synchronous form of ShExchange.)"""
    return ShExchange(campos, dL_dRGB, group).finish(means3D, sh_degree, num_coeffs)


def all_reduce_max_radii(radii: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Per-Gaussian MAX of the int32 screen radii over the views of one step (what
    train.py:157 `max_radii2D` consumes in view-parallel mode)."""
    out = radii.clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(out, op=dist.ReduceOp.MAX, group=group)
    return out


class DensificationStats:
    """The training-loop statistics that consume the rasterizer's side outputs
    (train.py:157-159, scene/gaussian_model.py:599-602), view-parallel:

        max_radii2D[vis] = max(max_radii2D[vis], radii[vis])                      per view
        xyz_gradient_accum[vis] += || viewspace_points.grad[vis, :2] ||           per view
        denom[vis] += 1                                                           per view

    With one view per rank, the N views of a step are folded with ONE int32 MAX all-reduce (radii,
    0 where culled) and ONE float SUM all-reduce of a (P,2) tensor [norm * vis, vis]; the result is
    what the reference would hold after applying its update once per view, in any order.
    """

    def __init__(self, num_points: int, device=None):
        self.max_radii2D = torch.zeros(num_points, device=device)
        self.xyz_gradient_accum = torch.zeros(num_points, 1, device=device)
        self.denom = torch.zeros(num_points, 1, device=device)

    @torch.no_grad()
    def update(self, viewspace_grad: torch.Tensor, radii: torch.Tensor, group: Optional[dist.ProcessGroup] = None):
        vis = radii > 0
        r = torch.where(vis, radii, torch.zeros_like(radii))
        s = torch.zeros(radii.shape[0], 2, device=radii.device, dtype=viewspace_grad.dtype)
        s[:, 0] = torch.norm(viewspace_grad[:, :2], dim=-1) * vis
        s[:, 1] = vis.to(s.dtype)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(r, op=dist.ReduceOp.MAX, group=group)
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
        self.max_radii2D = torch.max(self.max_radii2D, r.to(self.max_radii2D.dtype))
        self.xyz_gradient_accum += s[:, 0:1]
        self.denom += s[:, 1:2]
