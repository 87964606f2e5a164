import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "gaussian-splatting_cc-comments_amd"), os.path.join(R, "tests")): sys.path.insert(0, p)
import numpy as np, torch
import gsr_scene, util
scene = gsr_scene.make_scene(2_500, -3.0, sh_degree=0, seed=21)
cam = gsr_scene.make_camera(180, 100)
g = torch.Generator().manual_seed(5)
colors = torch.rand(2_500, 3, generator=g)
o0 = util.oracle_forward(scene, cam, 0, scale_modifier=1.3)
cov = torch.from_numpy(o0["cov3D"].copy()); vis = o0["radii"] > 0
cov[~torch.from_numpy(vis)] = torch.eye(3)[[0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]] * 1e-4
o = util.oracle_forward(scene, cam, 0, colors_precomp=colors, cov3D_precomp=cov, use_sh=False, use_scale_rot=False)
dpix = util.fragile_free_dpix(o, cam)
h = util.hip_forward_backward(scene, cam, 0, dpix, colors_precomp=colors, cov3D_precomp=cov, use_sh=False, use_scale_rot=False)
og = util.oracle.backward(o, dpix.numpy()); g1 = util.oracle.backward(o, dpix.numpy(), accum_mode=1); ex = util.oracle.blend_backward_exact(o, dpix.numpy())
print("FLAGS", os.environ.get("GSR_DEBUG_FLAGS"), "G1249 exact", ex["dL_dconic"].reshape(-1,4)[1249], "gpu", h["raw_grads"]["dL_dconic"].reshape(-1,4)[1249], "oracle", og["dL_dconic"].reshape(-1,4)[1249])
a = h["raw_grads"]["dL_dconic"].reshape(-1, 4).astype(np.float64); b = og["dL_dconic"].reshape(-1, 4).astype(np.float64); c = g1["dL_dconic"].reshape(-1,4).astype(np.float64)
err = np.abs(a - b); i = np.unravel_index(err.argmax(), err.shape)
print("max|g|", np.abs(b).max(), "argmax err", i, "gpu", a[i], "oracle64", b[i], "oracle32", c[i], "abs err", err[i])
gi = i[0]
print("gaussian", gi, "radius", o["radii"][gi], "tiles", o["tiles_touched"][gi], "opacity", o["conic_opacity"][gi], "mean2D", o["means2D"][gi], "depth", o["depths"][gi])
print("row gpu", a[gi], "\nrow o64", b[gi], "\nrow o32", c[gi])
top = np.argsort(-err.max(1))[:5]
for t in top: print(t, err[t].max(), o["radii"][t], o["tiles_touched"][t], a[t], b[t])
# image diff + n_contrib diffs including fragile pixels
ok = o["fragile"] == 0
d = np.abs(h["color"].reshape(3,-1) - o["color"].reshape(3,-1)).max(0)
print("image maxdiff nonfragile", d[ok].max(), "fragile", d[~ok].max() if (~ok).any() else None, "n fragile", (~ok).sum())
print("n_contrib mismatches (all pixels)", (h["n_contrib"] != o["n_contrib"]).sum())

ok = o["fragile"] == 0
rel = np.abs(h["final_T"] - o["final_T"]) / np.maximum(o["final_T"], 1e-30)
print("final_T rel err max (non fragile)", rel[ok].max(), "mean", rel[ok].mean())
for k in ("dL_dmeans2D","dL_dconic","dL_dopacity","dL_dcolors"):
    P = 2500
    a = h["raw_grads"][k].reshape(P,-1).astype(np.float64); b = ex[k].reshape(P,-1); c = og[k].reshape(P,-1).astype(np.float64)
    print(k, "gpu-vs-exact", np.abs(a-b).max()/np.abs(b).max(), "oracle-vs-exact", np.abs(c-b).max()/np.abs(b).max())
