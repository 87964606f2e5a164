"""Host-side timing of the two forward calls at C3 (where does the host spend the time between the count read-back and the launch
of forward stage 2?).  usage (GPU box, repo root): python tools/host_timing.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "gaussian-splatting_cc-comments_amd")):
    sys.path.insert(0, p)
import ctypes
import torch
import gsr_scene
from diff_gaussian_rasterization import _C

dev = torch.device("cuda:0")
P, W, H, D, mu = gsr_scene.CONFIGS["C3"]
scene = gsr_scene.make_scene(P, mu, D, seed=0)
cam = gsr_scene.make_camera(W, H)
to = lambda t: t.to(dev)
a = dict(bg=to(scene.bg), m=to(scene.means3D), op=to(scene.opacities), sc=to(scene.scales), ro=to(scene.rotations), sh=to(scene.shs),
         vm=to(cam.world_view_transform), pm=to(cam.full_proj_transform), cp=to(cam.camera_center))
L = _C.lib()
e = torch.empty(0, device=dev)
orig_pre, orig_ren = L.gsr_forward_preprocess, L.gsr_forward_render
t = {"pre": [], "mid": [], "ren": []}
last = [0.0]
def pre(*args):
    t0 = time.perf_counter(); r = orig_pre(*args); t1 = time.perf_counter(); t["pre"].append(t1 - t0); last[0] = t1; return r
def ren(*args):
    t0 = time.perf_counter(); t["mid"].append(t0 - last[0]); r = orig_ren(*args); t["ren"].append(time.perf_counter() - t0); return r
class Wrap:
    def __getattr__(self, n):
        return pre if n == "gsr_forward_preprocess" else ren if n == "gsr_forward_render" else getattr(L, n)
_C._lib = Wrap()
for i in range(200):
    _C.rasterize_gaussians(a["bg"], a["m"], e, a["op"], a["sc"], a["ro"], 1.0, e, a["vm"], a["pm"], cam.tanfovx, cam.tanfovy, H, W, a["sh"], D, a["cp"], False, False)
torch.cuda.synchronize()
med = lambda v: sorted(v[50:])[len(v[50:]) // 2] * 1e6
print({k: round(med(v), 1) for k, v in t.items()}, "us (median of the last 150 calls): pre = host time inside gsr_forward_preprocess "
      "(until the count is back), mid = Python between the two calls, ren = host time inside gsr_forward_render (five launches)")
