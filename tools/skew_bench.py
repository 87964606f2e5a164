"""Kernel times on a deliberately skewed scene: most Gaussians clustered so that a few tiles carry
very long instance lists (real captures look like this, the uniform benchmark scene does not)."""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "gaussian-splatting_cc-comments_amd")): sys.path.insert(0, p)
import torch, gsr_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
dev = torch.device("cuda:0")
P, W, H, D = 1_000_000, 1980, 1080, 3
sc = gsr_scene.make_scene(P, -5.0, D, seed=0)
g = torch.Generator().manual_seed(1)
means = sc.means3D.clone()
means[: P // 2] = torch.randn(P // 2, 3, generator=g) * torch.tensor([0.25, 0.15, 0.4])   # dense blob in the centre
cam = gsr_scene.make_camera(W, H)
to = lambda t: t.to(dev)
params = dict(means3D=to(means).requires_grad_(True), shs=to(sc.shs).requires_grad_(True), opacities=to(sc.opacities).requires_grad_(True),
              scales=to(sc.scales).requires_grad_(True), rotations=to(sc.rotations).requires_grad_(True))
st = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, to(sc.bg), 1.0, to(cam.world_view_transform), to(cam.full_proj_transform), D, to(cam.camera_center), False, False)
rast = GaussianRasterizer(st)
dpix = torch.randn(3, H, W, device=dev)
def step():
    for p in params.values(): p.grad = None
    c, r = rast(means2D=torch.zeros_like(params["means3D"], requires_grad=True), **params)
    c.backward(dpix)
    return r
for _ in range(3): step()
torch.cuda.synchronize(); _C.profile_begin()
import time; t0 = time.perf_counter()
for _ in range(10): r = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
k = {}
for n, ms in _C.profile_end(4096): k.setdefault(n, []).append(ms)
cap = {}
orig = _C.rasterize_gaussians
def spy(*a):
    o = orig(*a); cap["R"], cap["img"] = o[0], o[5]; return o
_C.rasterize_gaussians = spy
with torch.no_grad(): rast(means2D=torch.zeros_like(params["means3D"]), **params)
T = ((W + 15) // 16) * ((H + 15) // 16)
il = _C.image_layout(W, H)
rng = cap["img"][il.ranges:il.ranges + 8 * T].view(torch.int32).view(T, 2)
lens = (rng[:, 1] - rng[:, 0]).float()
tmc = cap["img"][il.tile_max_contrib:il.tile_max_contrib + 4 * T].view(torch.int32).float()
print(json.dumps(dict(ms_per_step=round(dt, 3), R=cap["R"], tile_len_mean=float(lens.mean()), tile_len_max=float(lens.max()),
                      max_contrib_mean=float(tmc.mean()), max_contrib_max=float(tmc.max()),
                      kernels={n: round(sum(v) / len(v), 4) for n, v in k.items()})))
