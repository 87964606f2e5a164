"""One `all_fused` training iteration in a loop (leaf-parameter rasterizer + fused L1+SSIM loss + one-launch Adam: bench.py's
train_iteration.all_fused) for profilers:  rocprofv3 --kernel-trace --stats -- python3 tools/train_iter.py [--config C3] [--iters 20]
tools/timeline.py --all turns the trace into the per-iteration GPU timeline (every kernel of the process, PyTorch's included)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_cc-comments_amd"))

import torch  # noqa: E402

import bench  # noqa: E402
import gsr_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
from diff_gaussian_rasterization import GaussianRasterizationSettings  # noqa: E402
dev = torch.device("cuda:0")
P, W, H, D, mu = gsr_scene.CONFIGS[a.config]
scene = gsr_scene.make_scene(P, mu, D, seed=0)
c = gsr_scene.make_camera(W, H)
to = lambda t: t.to(dev)
settings = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=to(scene.bg), scale_modifier=1.0,
                                         viewmatrix=to(c.world_view_transform), projmatrix=to(c.full_proj_transform), sh_degree=D,
                                         campos=to(c.camera_center), prefiltered=False, debug=False)
print(json.dumps(bench.bench_train_step(scene, settings, D, dev, iters=a.iters, modes=("all_fused",))))
