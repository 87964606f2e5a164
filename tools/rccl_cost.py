"""Fixed cost of the view-parallel step's collectives over RCCL, measured with the ONE-rank communicator a one-GPU box allows
(RCCL refuses two ranks on one device): per exchange mode and number of parts, the collectives of one step on exactly the
tensors of the N-rank run at P Gaussians -- issue them asynchronously, wait, repeat -- timed on the host (issue + wait) and on
the GPU (events on the compute stream around the whole sequence, i.e. including the stream hand-offs to RCCL's stream and back).
With one rank nothing crosses a link: what is measured is the launch / synchronisation cost a collective adds to a step
whatever its size, the term that decides --parts 1 against 2.

    python tools/rccl_cost.py [out.json]        (GPU box)
"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

P, M = 1_000_000, 16
ITERS = 50


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29613")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    res = {"P": P, "M": M, "world_size": 1, "iters": ITERS, "cases": []}

    def timed(label, issue):
        for _ in range(5):
            for w in issue():
                w.wait()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(ITERS):
            for w in issue():
                w.wait()
        e1.record()
        host_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        gpu_ms = e0.elapsed_time(e1) / ITERS
        res["cases"].append(dict(case=label, gpu_ms_per_step=round(gpu_ms, 4), host_ms_per_step=round(host_issue / ITERS * 1e3, 4)))
        print(res["cases"][-1], flush=True)

    # single calls on the step's tensor sizes and on a tiny tensor (the pure fixed cost)
    tiny = torch.zeros(256, **f32)
    timed("all_reduce, 1 KB", lambda: [dist.all_reduce(tiny, async_op=True)])
    for parts in (1, 2):
        c = P // parts
        bucket = [torch.zeros(11 * c, **f32) for _ in range(parts)]
        rgb = [torch.zeros((c + 1, 3), **f32) for _ in range(parts)]
        rgb_all = [torch.zeros((c + 1, 3), **f32) for _ in range(parts)]
        dsh = torch.zeros((P, M, 3), **f32)
        timed(f"compact, {parts} part(s): {parts} x (all_gather {4 * 3 * (c + 1) / 1e6:.1f} MB + all_reduce {4 * 11 * c / 1e6:.1f} MB)",
              lambda: [w for k in range(parts) for w in (dist.all_gather_into_tensor(rgb_all[k], rgb[k], async_op=True),
                                                         dist.all_reduce(bucket[k], async_op=True))])
        timed(f"allreduce, {parts} part(s): {parts} x (all_reduce {4 * 48 * c / 1e6:.1f} MB + all_reduce {4 * 11 * c / 1e6:.1f} MB)",
              lambda: [w for k in range(parts) for w in (dist.all_reduce(dsh[k * c:(k + 1) * c], async_op=True),
                                                         dist.all_reduce(bucket[k], async_op=True))])
    flat = torch.zeros(59 * P, **f32)
    timed("one flat all_reduce of 59 floats per Gaussian (236 MB): --views-per-rank k", lambda: [dist.all_reduce(flat, async_op=True)])
    res["note"] = ("ONE rank: no byte crosses a link; gpu_ms_per_step = what the step's collectives cost on the compute stream when "
                   "nothing overlaps them (issue, RCCL's own kernel / copy on its stream, the two stream hand-offs per collective)")
    dist.destroy_process_group()
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
