"""Per-kernel times of the leaf-parameter path next to the standard path on the benchmark scene."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "gaussian-splatting_cc-comments_amd")): sys.path.insert(0, p)
import torch, gsr_scene, gsr_model
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
from fused_params import rasterize_leaf_gaussians
dev = torch.device("cuda:0")
P, W, H, D, mu = gsr_scene.CONFIGS["C3"]
sc = gsr_scene.make_scene(P, mu, D, seed=0); cam = gsr_scene.make_camera(W, H)
to = lambda t: t.to(dev)
st = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, to(sc.bg), 1.0, to(cam.world_view_transform), to(cam.full_proj_transform), D, to(cam.camera_center), False, False)
pc = gsr_model.GaussianParams.from_activated(sc.means3D, sc.shs, sc.scales, sc.rotations, sc.opacities, device=dev)
dpix = torch.randn(3, H, W, device=dev)
def leaf():
    for p in pc.parameters(): p.grad = None
    c, _ = rasterize_leaf_gaussians(pc._xyz, torch.zeros_like(pc._xyz, requires_grad=True), pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation, st)
    c.backward(dpix)
def std():
    for p in pc.parameters(): p.grad = None
    c, _ = GaussianRasterizer(st)(means3D=pc.get_xyz, means2D=torch.zeros_like(pc._xyz, requires_grad=True), shs=pc.get_features, opacities=pc.get_opacity, scales=pc.get_scaling, rotations=pc.get_rotation)
    c.backward(dpix)
for name, fn in (("leaf", leaf), ("standard + torch activations", std)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); _C.profile_begin()
    for _ in range(8): fn()
    torch.cuda.synchronize()
    k = {}
    for n, ms in _C.profile_end(2048): k.setdefault(n, []).append(ms)
    print(name, {n: round(sum(v) / len(v), 4) for n, v in k.items() if n in ("preprocess", "gaussian_backward")})
