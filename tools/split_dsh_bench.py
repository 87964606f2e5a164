"""Per-Gaussian backward at C3: the fused kernel (every gradient, dL/dsh included) against the split the view-parallel "compact"
mode uses -- the kernel without the SH gradient (it emits the clamp-masked dL/dRGB) followed by gsr_sh_grad_from_views for one view.
usage (GPU box, repo root): python tools/split_dsh_bench.py"""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R_, os.path.join(R_, "gaussian-splatting_cc-comments_amd")):
    sys.path.insert(0, p)
import torch
import gsr_scene
from diff_gaussian_rasterization import _C

dev = torch.device("cuda:0")
P, W, H, D, mu = gsr_scene.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
scene = gsr_scene.make_scene(P, mu, D, seed=0)
cam = gsr_scene.make_camera(W, H)
to = lambda t: t.to(dev)
bg, m, op, sc, ro, sh = (to(t) for t in (scene.bg, scene.means3D, scene.opacities, scene.scales, scene.rotations, scene.shs))
vm, pm, cp = to(cam.world_view_transform), to(cam.full_proj_transform), to(cam.camera_center)
e = torch.empty(0, device=dev)
Rn, color, radii, geom, binning, img = _C.rasterize_gaussians(bg, m, e, op, sc, ro, 1.0, e, vm, pm, cam.tanfovx, cam.tanfovy, H, W, sh, D, cp, False, False)
dpix = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
M = sh.shape[1]

def bwd(skip):
    return _C.rasterize_gaussians_backward(bg, m, radii, e, sc, ro, 1.0, e, vm, pm, cam.tanfovx, cam.tanfovy, dpix, sh, D, cp, geom, Rn,
                                           binning, img, False, lean=True, skip_sh=skip)

def times(skip, n=30):
    out = {}
    for _ in range(5):
        bwd(skip)
    _C.profile_begin()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n)]
    tot = 0.0
    for i in range(n):
        r = bwd(skip)
        if skip:
            ev[2 * i].record()
            dsh = _C.sh_grad_from_views(m, cp.reshape(1, 3), r[1].reshape(1, P, 3), D, M)
            ev[2 * i + 1].record()
    torch.cuda.synchronize()
    for name, ms in _C.profile_end(capacity=64 * n):
        out.setdefault(name, []).append(ms)
    res = {k: round(sum(v) / n, 4) for k, v in out.items()}
    if skip:
        res["sh_grad_from_views"] = round(sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(n)) / n, 4)
        full = bwd(False)
        assert torch.equal(full[5], dsh), "the split's dL/dsh differs from the fused kernel's"
    return res

print("fused:", times(False))
print("split:", times(True))
