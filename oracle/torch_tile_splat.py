"""A tiled float32 PyTorch-CPU autograd splat: the "PyTorch autograd splat timed on the same box's host cores" that
BASELINE.json's north_star names as CPU baseline (configs[0]: 10k Gaussians, SH degree 0, 256x256, one view).

TEST / BENCH INFRASTRUCTURE (lives under oracle/): imported by tests/ and by bench.py's cpu_baseline leg only.  The
reference has no CPU path of its own; this is a restatement of its forward
(cuda_rasterizer/forward.cu:84-140, 192-324, 331-485; rasterizer_impl.cu:78-159) in plain PyTorch ops, with the
gradients left to torch.autograd and the reference's deliberate deviations encoded (SURVEY.md Appendix A item 14:
straight-through 0.99 clamp, masks without gradient).  Per-Gaussian preprocessing is vectorised over P; binning
builds the (tile, depth)-sorted instance list like duplicateWithKeys + SortPairs; the blend processes the tiles in
batches of similar length as dense (tiles, 256 pixels, instances) tensors with a cumulative product over the
instances.
"""
import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def _sh_color(deg, sh, d):  # forward.cu:21-81; sh (V,M,3), d (V,3)
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


def render(means3D, scales, rotations, opacities, shs, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, sh_degree,
           scale_modifier=1.0, tile_batch_elems=48_000_000):
    """-> (image (3,H,W), radii (P,) int32, num_rendered).  All inputs CPU float32 tensors (matrices in the
    reference's transposed layout, scene/cameras.py:57-61); differentiable w.r.t. the five parameter tensors."""
    dt = means3D.dtype
    P = means3D.shape[0]
    V4, PM = viewmatrix.reshape(4, 4).to(dt), projmatrix.reshape(4, 4).to(dt)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    hom = torch.cat([means3D, torch.ones(P, 1, dtype=dt)], 1)
    t = (hom @ V4)[:, :3]
    keep = t[:, 2] > 0.2                                           # auxiliary.h:165
    idx = torch.nonzero(keep).flatten()
    t, hom_v = t[idx], hom[idx]
    ph = hom_v @ PM
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None]
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], 1)
    # computeCov3D (forward.cu:146-180), quaternion used as given
    q = rotations[idx]
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rm = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                      2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                      2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    sv = scales[idx]
    Mm = Rm * (sv + (scale_modifier - 1.0) * sv.detach())[:, None, :]   # dL/dscale without the modifier (backward.cu:281-345)
    Sigma = Mm @ Mm.transpose(1, 2)
    # computeCov2D (forward.cu:84-140): where the clamp is active the clamped value is a constant for the gradient
    # (backward.cu:174-178, 265-266; SURVEY.md Appendix A 14 ii)
    tz = t[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    in_x, in_y = (t[:, 0] / tz).detach().abs() <= limx, (t[:, 1] / tz).detach().abs() <= limy
    tx = torch.where(in_x, t[:, 0], (torch.sign(t[:, 0]) * limx * tz).detach())
    ty = torch.where(in_y, t[:, 1], (torch.sign(t[:, 1]) * limy * tz).detach())
    Vn = idx.shape[0]
    J = torch.zeros(Vn, 2, 3, dtype=dt)
    J[:, 0, 0] = fx / tz
    J[:, 0, 2] = -fx * tx / (tz * tz)
    J[:, 1, 1] = fy / tz
    J[:, 1, 2] = -fy * ty / (tz * tz)
    JW = J @ V4[:3, :3].t()
    cov = JW @ Sigma @ JW.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    with torch.no_grad():
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
        radius = torch.ceil(3.0 * torch.sqrt(lam))
        f2i = lambda v: torch.trunc(v).to(torch.int64)
        minx = f2i((pix[:, 0] - radius) / 16).clamp(0, gx)
        miny = f2i((pix[:, 1] - radius) / 16).clamp(0, gy)
        maxx = f2i((pix[:, 0] + radius + 15) / 16).clamp(0, gx)
        maxy = f2i((pix[:, 1] + radius + 15) / 16).clamp(0, gy)
        tw, th = maxx - minx, maxy - miny
        tiles = tw * th
        ok = (det != 0) & (tiles > 0)
        radii = torch.zeros(P, dtype=torch.int32)
        radii[idx[ok]] = radius[ok].to(torch.int32)
    det_s = torch.where(ok, det, torch.ones_like(det))
    ca, cb, cc = c / det_s, -b / det_s, a / det_s
    d = means3D[idx] - campos.to(dt)
    d = d / d.norm(dim=1, keepdim=True)
    rgb = _sh_color(sh_degree, shs[idx], d)
    op = opacities.reshape(-1)[idx]

    # ---- binning: (tile, depth, index)-sorted instance list (rasterizer_impl.cu:78-159) ----
    with torch.no_grad():
        vsel = torch.nonzero(ok).flatten()
        cnt = tiles[vsel]
        R = int(cnt.sum())
        owner = torch.repeat_interleave(vsel, cnt)                         # index into the visible arrays
        first = torch.cumsum(cnt, 0) - cnt
        k = torch.arange(R) - torch.repeat_interleave(first, cnt)
        w_o = tw[owner]
        tile_id = (miny[owner] + k // w_o) * gx + (minx[owner] + k % w_o)
        depth_bits = tz.detach().to(torch.float32).view(torch.int32).to(torch.int64)[owner]
        order = torch.argsort((tile_id << 32) | depth_bits, stable=True)
        owner, tile_id = owner[order], tile_id[order]
        T = gx * gy
        start = torch.searchsorted(tile_id, torch.arange(T))
        end = torch.searchsorted(tile_id, torch.arange(T), right=True)
        length = end - start
    image = torch.zeros(H * W, 3, dtype=dt) + bg.to(dt)[None, :]
    nz = torch.nonzero(length > 0).flatten()
    if nz.numel():
        by_len = nz[torch.argsort(length[nz])]
        # pixel coordinates of a tile, row-major
        ly, lx = torch.meshgrid(torch.arange(16), torch.arange(16), indexing="ij")
        lx, ly = lx.reshape(-1), ly.reshape(-1)
        pieces_idx, pieces_val = [], []
        pos = 0
        while pos < by_len.numel():
            Lmax_probe = int(length[by_len[min(pos + 63, by_len.numel() - 1)]])
            nb = max(1, min(by_len.numel() - pos, tile_batch_elems // (256 * max(Lmax_probe, 1))))
            tl = by_len[pos:pos + nb]
            pos += nb
            L = int(length[tl].max())
            j = torch.arange(L)[None, :]
            valid_slot = j < length[tl][:, None]                                     # (B,L)
            inst = owner[(start[tl][:, None] + j).clamp(max=R - 1)]                  # (B,L) visible-array index
            px = ((tl % gx) * 16)[:, None] + lx[None, :]                              # (B,256)
            py = ((tl // gx) * 16)[:, None] + ly[None, :]
            inside = (px < W) & (py < H)
            dx = pix[inst, 0][:, None, :] - px.to(dt)[:, :, None]                    # (B,256,L)
            dy = pix[inst, 1][:, None, :] - py.to(dt)[:, :, None]
            power = -0.5 * (ca[inst][:, None, :] * dx * dx + cc[inst][:, None, :] * dy * dy) - cb[inst][:, None, :] * dx * dy
            oG = op[inst][:, None, :] * torch.exp(power)
            alpha = oG + (torch.clamp(oG, max=0.99) - oG).detach()                   # straight-through (backward.cu:528-529)
            live = valid_slot[:, None, :] & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
            one_m = torch.where(live, 1.0 - alpha, torch.ones_like(alpha))
            Tincl = torch.cumprod(one_m, dim=2)
            Texcl = torch.cat([torch.ones_like(Tincl[:, :, :1]), Tincl[:, :, :-1]], 2)
            stop = live & (Tincl.detach() < 1e-4)                                    # forward.cu:451-456
            contributes = live & ~(torch.cumsum(stop.to(torch.int8), dim=2) > 0)
            wgt = torch.where(contributes, alpha * Texcl, torch.zeros_like(alpha))  # (B,256,L)
            col = torch.einsum("bpl,blc->bpc", wgt, rgb[inst])                       # (B,256,3)
            Tfin = torch.prod(torch.where(contributes, 1.0 - alpha, torch.ones_like(alpha)), dim=2)
            col = col + Tfin[:, :, None] * bg.to(dt)[None, None, :]
            pid = (py * W + px)[inside]
            pieces_idx.append(pid)
            pieces_val.append(col[inside])
        image = image.index_put((torch.cat(pieces_idx),), torch.cat(pieces_val), accumulate=False)
    return image.t().reshape(3, H, W), radii, R


def time_forward_backward(scene, cam, sh_degree, repeats=5, threads=None):
    """Median wall time of forward + backward on the host cores -> (seconds, num_rendered, image)."""
    import time
    if threads:
        torch.set_num_threads(int(threads))
    g = torch.Generator().manual_seed(1)
    dpix = torch.randn(3, cam.image_height, cam.image_width, generator=g)
    times, img, R = [], None, 0
    for _ in range(repeats):
        leaves = [t.clone().requires_grad_(True) for t in (scene.means3D, scene.scales, scene.rotations, scene.opacities, scene.shs)]
        t0 = time.perf_counter()
        img, radii, R = render(*leaves, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, scene.bg,
                               cam.image_width, cam.image_height, cam.tanfovx, cam.tanfovy, sh_degree)
        (img * dpix).sum().backward()
        times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2], R, img.detach()
