import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "gaussian-splatting_cc-comments_amd"), os.path.join(R, "tests")): sys.path.insert(0, p)
import numpy as np, torch
import gsr_scene, util
g = torch.Generator().manual_seed(77)
P = 300
base = gsr_scene.make_scene(P, -3.0, sh_degree=0, seed=77)
scales = base.scales.clone()
scales[:, 0] = torch.exp(torch.randn(P, generator=g) * 0.5 + 0.3)
scales[:, 1:] = torch.exp(torch.randn(P, 2, generator=g) * 0.3 - 5.0)
scene = base._replace(scales=scales.contiguous(), opacities=torch.full((P, 1), 0.95))
cam = gsr_scene.make_camera(640, 360)
o = util.oracle_forward(scene, cam, 0)
h = util.hip_forward_backward(scene, cam, 0, None)
W, H = 640, 360
ok = o["fragile"] == 0
d = np.abs(h["color"].reshape(3, -1) - o["color"].reshape(3, -1)).max(0)
d[~ok] = 0
pix = int(d.argmax()); py, px = divmod(pix, W)
print("pixel", px, py, "err", d[pix], "gpu", h["color"].reshape(3,-1)[:,pix], "oracle", o["color"].reshape(3,-1)[:,pix])
print("final_T gpu/oracle", h["final_T"][pix], o["final_T"][pix], "n_contrib", h["n_contrib"][pix], o["n_contrib"][pix])
tile = (py // 16) * ((W + 15) // 16) + px // 16
r0, r1 = o["ranges"][tile]
print("tile", tile, "range", r0, r1)
T = np.float32(1.0); C = np.zeros(3, np.float32)
f = np.float32
for k, i in enumerate(range(r0, r1)):
    gid = o["point_list"][i]
    mx, my = o["means2D"][gid]; a, b, c, op = o["conic_opacity"][gid]
    dx = f(mx - f(px)); dy = f(my - f(py))
    power = f(f(-0.5) * f(f(f(a * dx) * dx) + f(f(c * dy) * dy))) - f(f(b * dx) * dy)
    power = f(power)
    if power > 0: 
        print(k, gid, "power>0", power); continue
    G = f(np.exp(np.float64(power))); alpha = min(f(0.99), f(op * G))
    if alpha < f(1/255): continue
    tT = f(T * f(1 - alpha))
    print(k, "g", gid, "power", power, "alpha", alpha, "T", T, "terms", f(a*dx)*dx, f(c*dy)*dy, f(b*dx)*dy, "radius", o["radii"][gid])
    if tT < 1e-4: print("stop"); break
    C += o["rgb"][gid] * alpha * T; T = tT
print("emulated", C + T * o["bg"], "T", T)
