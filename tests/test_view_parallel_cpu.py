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


def test_gradient_exchange_ranges_cover_every_size_including_the_empty_scene():
    """Part ranges: starts are multiples of 256, the parts cover [0, P) exactly, and P == 0 (an empty scene is legal at the
    boundary, rasterize_points.cu:94) gives one empty part instead of raising."""
    for P, parts in ((0, 2), (1, 2), (255, 3), (256, 2), (257, 2), (1500, 3), (10_000, 4), (1_000_000, 2)):
        ex = view_parallel.GradientExchange(P, 16, "cpu", parts=parts)
        assert all(f % 256 == 0 for f, _ in ex.ranges) and sum(c for _, c in ex.ranges) == P, (P, ex.ranges)
        assert 1 <= len(ex.ranges) <= max(1, parts)
        if P == 0:
            assert ex.ranges == [(0, 0)]
            ex.begin_step()
            ex.submit(0, torch.zeros(3))
            out = ex.finish(torch.zeros(0, 3), 3, rebuild_sh=False)
            assert out["dL_dmean3D"].shape == (0, 3) and out["dL_dsh"].shape == (0, 16, 3)


def test_densification_stats_update_keeps_the_tensors_the_kernel_was_given():
    """kernel_tensors() hands views of the local accumulators to rasterizer objects and autograd contexts that keep them;
    update() (the PyTorch form of the same bookkeeping) must write INTO them, or later kernel-epilogue maxima are lost."""
    P = 6
    stats = view_parallel.DensificationStats(P)
    accum, denom, maxr = stats.kernel_tensors()          # captured before any update, like GaussianRasterizer(densify_stats=...)
    ptrs = [t.data_ptr() for t in (accum, denom, maxr)]
    stats.update(torch.ones(P, 3), torch.tensor([0, 3, 0, 9, 1, 0], dtype=torch.int32))
    assert [t.data_ptr() for t in stats.kernel_tensors()] == ptrs
    maxr[2] = 40.0                                        # what the kernel epilogue of a later view would write
    accum[2] += 2.0
    denom[2] += 1.0
    stats.update(torch.ones(P, 3), torch.tensor([5, 1, 0, 2, 0, 0], dtype=torch.int32))
    stats.sync()
    assert stats.max_radii2D.tolist() == [5.0, 3.0, 40.0, 9.0, 1.0, 0.0]
    assert stats.denom.flatten().tolist() == [1.0, 2.0, 1.0, 2.0, 1.0, 0.0]
    assert abs(float(stats.xyz_gradient_accum[2]) - 2.0) < 1e-6


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
            stats.update(grad, radii)   # local accumulation (on the GPU: the backward kernel's epilogue does this)
        stats.sync()                    # ONE MAX + ONE SUM all-reduce, whenever the statistics are read
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

        def stub(means3D, cam_all, rgb_all, degree, M, out=None):  # stands in for the HIP kernel: records what it is handed
            seen["cam"], seen["rgb"], seen["D"], seen["M"] = cam_all.clone(), rgb_all.clone(), degree, M
            r = rgb_all.sum(0)[:, None, :].expand(-1, M, -1).contiguous()
            if out is not None:
                out.copy_(r)
                return out
            return r
        orig = _C.sh_grad_from_views
        _C.sh_grad_from_views = stub
        try:
            means = torch.zeros(P, 3)
            rgb = torch.full((P, 3), float(rank + 1)) + torch.arange(P)[:, None]
            cam = torch.tensor([rank, 10.0 * rank, -4.0])
            out = view_parallel.exchange_sh_gradient(means, cam, rgb, 2, 9)
            assert seen["rgb"].shape == (world, P, 3) and seen["cam"].shape == (world, 3) and (seen["D"], seen["M"]) == (2, 9)
            for r in range(world):  # every rank sees every view, in rank order
                assert torch.equal(seen["rgb"][r], torch.full((P, 3), float(r + 1)) + torch.arange(P)[:, None])
                assert torch.equal(seen["cam"][r], torch.tensor([r, 10.0 * r, -4.0]))
            assert out.shape == (P, 9, 3)

            # the pipelined per-part exchange: what a rank's backward kernel would write into the part buffers
            P2, M = 1500, 16
            g = torch.Generator().manual_seed(7)
            full = {n: torch.randn(P2, w, generator=g) for n, w in view_parallel._SMALL}   # same on every rank
            rgb_full = torch.randn(P2, 3, generator=g)
            sh_full = torch.randn(P2, M, 3, generator=g)
            tot = sum(range(1, world + 1))
            for mode in ("compact", "allreduce"):
                ex = view_parallel.GradientExchange(P2, M, "cpu", sh_mode=mode, parts=3)
                assert [f % 256 for f, _ in ex.ranges] == [0] * len(ex.ranges) and sum(c for _, c in ex.ranges) == P2
                assert len(ex.ranges) == 3
                for _ in range(2):   # buffers are reused step after step
                    ex.begin_step()
                    for k, (first, c) in enumerate(ex.ranges):
                        for n, sec in ex.sections(k).items():
                            sec.copy_(full[n][first:first + c] * (rank + 1))
                        if mode == "compact":
                            ex.rgb[k][:c] = rgb_full[first:first + c] * (rank + 1)
                        else:
                            ex.dsh[first:first + c] = sh_full[first:first + c] * (rank + 1)
                        ex.submit(k, torch.tensor([float(rank), 0.0, 1.0]))
                    res = ex.finish(torch.zeros(P2, 3), 3)
                    for n, _ in view_parallel._SMALL:
                        assert res[n].shape == full[n].shape and torch.allclose(res[n], full[n] * tot), n
                    if mode == "compact":   # the stub summed the gathered views: every rank's rows, rank order, camera trailer
                        want = (rgb_full * tot)[:, None, :].expand(-1, M, -1)
                        assert torch.allclose(res["dL_dsh"], want)
                        assert torch.equal(seen["cam"], torch.tensor([[float(r), 0.0, 1.0] for r in range(world)]))
                    else:
                        assert torch.allclose(res["dL_dsh"], sh_full * tot)
        finally:
            _C.sh_grad_from_views = orig

        # GradientBucket: average + async is pre-scaled, not silently skipped
        p = torch.zeros(5, requires_grad=True)
        p.grad = torch.full((5,), float(rank + 1))
        b = view_parallel.GradientBucket([p])
        w = b.all_reduce(average=True, async_op=True)
        w.wait()
        assert torch.allclose(p.grad, torch.full((5,), sum(range(1, world + 1)) / world))
    finally:
        dist.destroy_process_group()


def test_sh_gradient_exchange_wiring_gloo_world2():
    mp.spawn(_exchange_worker, args=(2, _free_port(), 37), nprocs=2, join=True)


def test_bench_gpus_2_starts_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` (no torchrun, no RANK in the environment) must start two ranks itself.  Here, without
    a GPU, through the plumbing rehearsal mode: process group over gloo, both exchange modes, the JSON contract."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GSR_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "C1", "--steps", "3", "--warmup", "1",
                        "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["alt_exchange"]["mode"] == "allreduce" and d["step_ms"]["n"] == 3
    assert "dry-run" in d["data"]
    # a launcher that started a different number of ranks than --gpus says is an error, not a silent n_gpus
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env2, capture_output=True,
                        text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in (r2.stderr + r2.stdout)


def test_bench_rank_whose_peer_never_arrives_exits_nonzero_instead_of_hanging():
    """The 8-GPU run is this code's first RCCL run: a rank that cannot form the group (or whose collective never completes)
    must leave with a message and a non-zero code within --collective-timeout-s, so that launch_ranks stops the others."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               GSR_BENCH_BACKEND="gloo")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--collective-timeout-s", "4"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    assert "could not be formed" in r.stderr and time.time() - t0 < 60
