"""A differentiable float64 PyTorch restatement of the rasterizer forward ("PyTorch-CPU autograd
splat", BASELINE.json configs[0]).  TEST INFRASTRUCTURE: it gives gradients by torch.autograd,
i.e. derived independently of the hand-written backward formulas that the reference, the oracle
and the HIP kernels share, so agreement with the oracle's analytic backward checks those formulas
(and their transcription) rather than a copy of them.

Dense (pixels x Gaussians), for small scenes only.  Discrete decisions (tile membership, depth
order) are taken from the oracle state; the accept/reject masks are recomputed here from detached
values.  The reference's deliberate deviations from the true gradient are encoded as in
SURVEY.md Appendix A item 14: (i) straight-through 0.99 clamp; (ii) inside the frustum clamp of the EWA Jacobian
(forward.cu:102-107) the clamped t.x / t.y are constants for the gradient -- no gradient into t.x (x_grad_mul,
backward.cu:177-178, 265-266) and dJ02/dt.z taken with the clamped value held fixed (backward.cu:174-176); (iv) dL/dscale
without the scale_modifier factor (backward.cu:281-345); (v) masks carry no gradient.
"""
import numpy as np
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def sh_color(deg, sh, d):  # sh (P,M,3), d (P,3)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = C0 * sh[:, 0]
    if deg > 0:
        r = r - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6] +
             C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        r = (r + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10] +
             C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12] +
             C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14] +
             C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp_min(r + 0.5, 0.0)


def render(o, means3D, scales, rotations, opacities, shs, scale_modifier=1.0):
    """o: oracle forward state (numpy) for the same inputs.  Returns image (3,H,W) float64."""
    W, H, D = o["W"], o["H"], o["D"]
    dt = torch.float64
    V = torch.from_numpy(o["viewmatrix"]).to(dt).reshape(4, 4)
    PM = torch.from_numpy(o["projmatrix"]).to(dt).reshape(4, 4)
    campos = torch.from_numpy(o["campos"]).to(dt)
    bg = torch.from_numpy(o["bg"]).to(dt)
    tanx, tany = o["tanfovx"], o["tanfovy"]
    fx, fy = W / (2.0 * tanx), H / (2.0 * tany)
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    hom = torch.cat([means3D, ones], 1)
    t = (hom @ V)[:, :3]                       # auxiliary.h:60-69 with the flat index 4*c+r convention
    ph = hom @ PM
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None]
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], 1)
    vis = torch.from_numpy(o["radii"] > 0)
    tz = t[:, 2]
    # Appendix A 14 (ii): where the clamp is active the reference uses lim * t.z as a CONSTANT in J
    limx, limy = 1.3 * tanx, 1.3 * tany
    in_x = ((t[:, 0] / tz).detach().abs() <= limx)
    in_y = ((t[:, 1] / tz).detach().abs() <= limy)
    tx = torch.where(in_x, t[:, 0], (torch.sign(t[:, 0]) * limx * tz).detach())
    ty = torch.where(in_y, t[:, 1], (torch.sign(t[:, 1]) * limy * tz).detach())
    clamp_active = int(((~in_x | ~in_y) & vis).sum())
    # Sigma = R S^2 R^T with the standard rotation of the (unnormalised) quaternion (forward.cu:146-180)
    r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
    Rm = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                      2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                      2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
    # Appendix A 14 (iv): value mod * s, derivative w.r.t. s taken as 1 (the reference omits the factor)
    S = torch.diag_embed(scales + (scale_modifier - 1.0) * scales.detach())
    Mm = Rm @ S
    Sigma = Mm @ Mm.transpose(1, 2)
    # EWA: cov2D = (J W) Sigma (J W)^T + 0.3 I (forward.cu:84-140)
    Wm = V[:3, :3].t()                          # world -> view rotation
    J = torch.zeros(P, 2, 3, dtype=dt)
    J[:, 0, 0] = fx / tz
    J[:, 0, 2] = -fx * tx / (tz * tz)
    J[:, 1, 1] = fy / tz
    J[:, 1, 2] = -fy * ty / (tz * tz)
    JW = J @ Wm
    cov = JW @ Sigma @ JW.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    det = torch.where(vis, det, torch.ones_like(det))
    ca, cb, cc = c / det, -b / det, a / det
    d = means3D - campos
    d = d / d.norm(dim=1, keepdim=True)
    rgb = sh_color(D, shs, d)

    # per-pixel compositing in global (depth, index) order = the per-tile order restricted to the tile
    order = np.lexsort((np.arange(P), o["depths"]))
    order = order[o["radii"][order] > 0]
    ot = torch.from_numpy(order)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pxs, pys = xs.reshape(-1, 1), ys.reshape(-1, 1)
    # tile membership from the oracle's rectangles (auxiliary.h:48-58)
    m2, rad = o["means2D"][order], o["radii"][order]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    f2i = lambda v: np.trunc(v).astype(np.int64)
    minx = np.clip(f2i((m2[:, 0] - rad) / np.float32(16)), 0, gx)
    miny = np.clip(f2i((m2[:, 1] - rad) / np.float32(16)), 0, gy)
    maxx = np.clip(f2i((m2[:, 0] + rad + 15) / np.float32(16)), 0, gx)
    maxy = np.clip(f2i((m2[:, 1] + rad + 15) / np.float32(16)), 0, gy)
    txs, tys = (xs.reshape(-1).numpy() // 16).astype(np.int64)[:, None], (ys.reshape(-1).numpy() // 16).astype(np.int64)[:, None]
    member = torch.from_numpy((txs >= minx) & (txs < maxx) & (tys >= miny) & (tys < maxy))
    dx = pix[ot, 0][None, :] - pxs
    dy = pix[ot, 1][None, :] - pys
    power = -0.5 * (ca[ot][None] * dx * dx + cc[ot][None] * dy * dy) - cb[ot][None] * dx * dy
    G = torch.exp(power)
    oG = opacities.reshape(-1)[ot][None] * G
    alpha = oG + (torch.clamp(oG, max=0.99) - oG).detach()      # straight-through clamp (backward.cu:528-529)
    live = member & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
    one_m = torch.where(live, 1.0 - alpha, torch.ones_like(alpha))
    Tincl = torch.cumprod(one_m, dim=1)
    Texcl = torch.cat([torch.ones(Tincl.shape[0], 1, dtype=dt), Tincl[:, :-1]], 1)
    stop = live & (Tincl.detach() < 1e-4)                       # forward.cu:451-456: this pair does not contribute
    stopped = torch.cumsum(stop.to(torch.int64), dim=1) > 0
    valid = live & ~stopped
    w = torch.where(valid, alpha * Texcl, torch.zeros_like(alpha))
    img = w @ rgb[ot]                                           # (N,3)
    T_final = torch.prod(torch.where(valid, 1.0 - alpha, torch.ones_like(alpha)), dim=1)
    img = img + T_final[:, None] * bg[None]
    n_contrib = torch.where(valid.any(1), (valid.to(torch.int64) * torch.arange(1, valid.shape[1] + 1)).max(1).values, 0)
    render.clamp_active = clamp_active   # how many visible Gaussians had the frustum clamp active (for the tests)
    return img.t().reshape(3, H, W), T_final, n_contrib
