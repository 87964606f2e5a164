"""CPU: the N>1 path (view sharding + one flat-bucket gradient all-reduce) with gloo, world_size 2."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import view_parallel


def test_shard_views_partition():
    for world in (1, 2, 3, 8):
        seen = sorted(v for r in range(world) for v in view_parallel.shard_views(19, r, world))
        assert seen == list(range(19))
    assert view_parallel.shard_views(8, 3, 8) == [3]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, P):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        shapes = [(P, 3), (P, 16, 3), (P, 1), (P, 3), (P, 4)]  # xyz, SH, opacity, scale, rotation = 59 floats
        params = [torch.zeros(s, requires_grad=True) for s in shapes]
        base = [torch.randn(s, generator=g) for s in shapes]
        for p, b in zip(params, base):
            p.grad = b * (rank + 1)  # what "this rank's view" contributed
        bucket = view_parallel.GradientBucket(params)
        assert bucket.numel == 59 * P
        bucket.all_reduce()
        tot = sum(range(1, world + 1))
        for p, b in zip(params, base):
            assert torch.allclose(p.grad, b * tot)
            assert p.grad.data_ptr() >= bucket.flat.data_ptr()  # aliases the bucket, no unpack copy
        radii = torch.tensor([rank, 5 - rank, 7], dtype=torch.int32)
        m = view_parallel.all_reduce_max_radii(radii)
        assert m.tolist() == [world - 1, 5, 7]
    finally:
        dist.destroy_process_group()


def test_gradient_bucket_all_reduce_gloo_world2():
    mp.spawn(_worker, args=(2, _free_port(), 257), nprocs=2, join=True)


def _stats_worker(rank, world, port, P):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def view(r):  # what rank r's view produced (same on every rank, seeded by r)
            g = torch.Generator().manual_seed(100 + r)
            radii = (torch.rand(P, generator=g) * 40).to(torch.int32) * (torch.rand(P, generator=g) > 0.3)
            grad = torch.randn(P, 3, generator=g)
            return grad, radii.to(torch.int32)
        stats = view_parallel.DensificationStats(P)
        for step in range(2):
            grad, radii = view(rank + world * step)
            stats.update(grad, radii)
        # the reference's bookkeeping applied once per view, sequentially (train.py:157-159, gaussian_model.py:599-602)
        max_radii2D, accum, denom = torch.zeros(P), torch.zeros(P, 1), torch.zeros(P, 1)
        for v in range(2 * world):
            grad, radii = view(v)
            vis = radii > 0
            max_radii2D[vis] = torch.max(max_radii2D[vis], radii[vis].float())
            accum[vis] += torch.norm(grad[vis, :2], dim=-1, keepdim=True)
            denom[vis] += 1
        assert torch.equal(stats.max_radii2D, max_radii2D)
        assert torch.equal(stats.denom, denom)
        assert torch.allclose(stats.xyz_gradient_accum, accum, rtol=1e-6, atol=1e-6)
    finally:
        dist.destroy_process_group()


def test_densification_stats_match_per_view_updates_gloo_world2():
    mp.spawn(_stats_worker, args=(2, _free_port(), 513), nprocs=2, join=True)


def _exchange_worker(rank, world, port, P):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diff_gaussian_rasterization import _C
        seen = {}

        def stub(means3D, cam_all, rgb_all, degree, M):  # stands in for the HIP kernel: records what it is handed
            seen["cam"], seen["rgb"], seen["D"], seen["M"] = cam_all.clone(), rgb_all.clone(), degree, M
            return rgb_all.sum(0)[:, None, :].expand(-1, M, -1).contiguous()
        orig = _C.sh_grad_from_views
        _C.sh_grad_from_views = stub
        try:
            means = torch.zeros(P, 3)
            rgb = torch.full((P, 3), float(rank + 1)) + torch.arange(P)[:, None]
            cam = torch.tensor([rank, 10.0 * rank, -4.0])
            out = view_parallel.exchange_sh_gradient(means, cam, rgb, 2, 9)
        finally:
            _C.sh_grad_from_views = orig
        assert seen["rgb"].shape == (world, P, 3) and seen["cam"].shape == (world, 3) and (seen["D"], seen["M"]) == (2, 9)
        for r in range(world):  # every rank sees every view, in rank order
            assert torch.equal(seen["rgb"][r], torch.full((P, 3), float(r + 1)) + torch.arange(P)[:, None])
            assert torch.equal(seen["cam"][r], torch.tensor([r, 10.0 * r, -4.0]))
        assert out.shape == (P, 9, 3)
    finally:
        dist.destroy_process_group()


def test_sh_gradient_exchange_wiring_gloo_world2():
    mp.spawn(_exchange_worker, args=(2, _free_port(), 37), nprocs=2, join=True)
