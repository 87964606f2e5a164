"""Shared helpers for the parity tests: run the HIP path (through the drop-in autograd surface /
the ctypes `_C` binding, i.e. through the C ABI) and the CPU oracle on the same inputs."""
import numpy as np
import torch

import gsr_scene
from oracle import oracle


# error-versus-bar lines of the parity checks, kept so that tools/parity_report.py can write them to a file (pytest -q
# drops what the tests print)
PARITY_LOG = []


def parity_log(text):
    PARITY_LOG.append(text)


def oracle_forward(scene, cam, D, margin=2e-5, colors_precomp=None, cov3D_precomp=None, scale_modifier=1.0,
                   use_sh=True, use_scale_rot=True):
    return oracle.forward(
        scene.means3D.numpy(), scene.opacities.numpy(), cam.world_view_transform.numpy(),
        cam.full_proj_transform.numpy(), cam.camera_center.numpy(), scene.bg.numpy(), cam.image_width,
        cam.image_height, cam.tanfovx, cam.tanfovy, D,
        shs=scene.shs.numpy() if use_sh else None,
        colors_precomp=None if colors_precomp is None else colors_precomp.numpy(),
        scales=scene.scales.numpy() if use_scale_rot else None,
        rotations=scene.rotations.numpy() if use_scale_rot else None,
        cov3D_precomp=None if cov3D_precomp is None else cov3D_precomp.numpy(),
        scale_modifier=scale_modifier, margin=margin)


def hip_settings(scene, cam, D, dev, scale_modifier=1.0, debug=False, prefiltered=False):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=scene.bg.to(dev), scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform.to(dev),
        projmatrix=cam.full_proj_transform.to(dev), sh_degree=D, campos=cam.camera_center.to(dev),
        prefiltered=prefiltered, debug=debug)


def hip_forward_backward(scene, cam, D, dpix=None, dev="cuda:0", colors_precomp=None, cov3D_precomp=None,
                         scale_modifier=1.0, use_sh=True, use_scale_rot=True, debug=False):
    """Runs GaussianRasterizer fwd (+bwd when dpix is given) and returns numpy outputs, including
    the intermediates read out of the opaque state buffers through the published layouts."""
    from diff_gaussian_rasterization import GaussianRasterizer, _C, _RasterizeGaussians
    dev = torch.device(dev)
    leaf = lambda t: t.to(dev).clone().requires_grad_(True)
    means = leaf(scene.means3D)
    opac = leaf(scene.opacities)
    means2D = torch.zeros_like(means, requires_grad=True)
    kw = {}
    if use_sh:
        kw["shs"] = leaf(scene.shs)
    else:
        kw["colors_precomp"] = leaf(colors_precomp)
    if use_scale_rot:
        kw["scales"] = leaf(scene.scales)
        kw["rotations"] = leaf(scene.rotations)
    else:
        kw["cov3D_precomp"] = leaf(cov3D_precomp)
    settings = hip_settings(scene, cam, D, dev, scale_modifier, debug)
    # capture the saved state buffers
    captured = {}
    orig = _C.rasterize_gaussians

    def spy(*a):
        r = orig(*a)
        captured["R"], captured["geom"], captured["binning"], captured["img"] = r[0], r[3], r[4], r[5]
        return r
    _C.rasterize_gaussians = spy
    try:
        color, radii = GaussianRasterizer(settings)(means3D=means, means2D=means2D, opacities=opac, **kw)
    finally:
        _C.rasterize_gaussians = orig
    out = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy(), num_rendered=captured["R"])
    tile_sort = isinstance(debug, int) and not isinstance(debug, bool) and bool(debug & _C.DEBUG_TILE_SORT)
    out.update(unpack_state(captured, means.shape[0], cam.image_width, cam.image_height, tile_sort=tile_sort))
    if dpix is not None:
        raw = {}
        orig_b = _C.rasterize_gaussians_backward

        def spy_b(*a, **kw):   # all eight outputs plus the internal dL_dconic (debug_out switches the lean mode off)
            dbg = {}
            r = orig_b(*a, **dict(kw, debug_out=dbg))
            for k, t in zip(("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh",
                             "dL_dscales", "dL_drotations"), r):
                raw[k] = t
            raw["dL_dconic"] = dbg["dL_dconic"]
            return r
        _C.rasterize_gaussians_backward = spy_b
        try:
            color.backward(dpix.to(dev))
            torch.cuda.synchronize()
        finally:
            _C.rasterize_gaussians_backward = orig_b
        out["raw_grads"] = {k: v.cpu().numpy() for k, v in raw.items()}
        g = dict(dL_dmeans3D=means.grad, dL_dmeans2D=means2D.grad, dL_dopacity=opac.grad)
        if use_sh:
            g["dL_dsh"] = kw["shs"].grad
        else:
            g["dL_dcolors"] = kw["colors_precomp"].grad
        if use_scale_rot:
            g["dL_dscales"] = kw["scales"].grad
            g["dL_drotations"] = kw["rotations"].grad
        else:
            g["dL_dcov3D"] = kw["cov3D_precomp"].grad
        out["grads"] = {k: v.cpu().numpy() for k, v in g.items()}
    return out


def unpack_state(cap, P, W, H, tile_sort=False):
    from diff_gaussian_rasterization import _C
    R = cap["R"]
    o = {}
    if P == 0:
        return o
    geom, img, binning = cap["geom"].cpu().numpy(), cap["img"].cpu().numpy(), cap["binning"].cpu().numpy()
    gl, il = _C.geometry_layout(P), _C.image_layout(W, H)
    N, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    splat = geom[gl.splat:gl.splat + 48 * P].view(np.float32).reshape(P, 12)    # x y | conic a b c opacity | rect (2 words) | r g b -
    o["means2D"] = splat[:, 0:2]
    o["conic_opacity"] = splat[:, 2:6]
    o["rgb"] = splat[:, 8:11]
    o["tiles_touched"] = geom[gl.tiles_touched:gl.tiles_touched + 4 * P].view(np.uint32)
    o["clamped_bits"] = geom[gl.clamped:gl.clamped + P]
    # depth sort outputs: Gaussian ids in (depth, id) order and their sorted depth bits
    # status word 2: the depth sort ended in the ping-pong partners (three passes sufficed for the depth range)
    in_alt = int(geom[gl.status:gl.status + 16].view(np.uint32)[2])
    o["depth_sort_result_in_alt"] = in_alt
    # status word 3: the gradient slots' numbering (index order: the exclusive prefix of tiles_touched) is final -- set by forward stage 1 on
    # both depth-sort paths since the end of round 4 (rounds 1-3, and round 4's radix path at first, numbered them in depth order: 0)
    o["slots_in_index_order"] = int(geom[gl.status:gl.status + 16].view(np.uint32)[3])
    p_off, k_off = (gl.perm_alt, gl.depth_keys_alt) if in_alt else (gl.perm, gl.depth_keys)
    perm = geom[p_off:p_off + 4 * P].view(np.uint32)
    skeys = geom[k_off:k_off + 4 * P].view(np.uint32)
    o["perm"], o["sorted_depth_keys"] = perm, skeys
    depth_bits = np.empty(P, np.uint32)
    depth_bits[perm] = skeys
    o["depth_bits"] = depth_bits                      # per Gaussian; 0xFFFFFFFF = culled
    o["depths"] = depth_bits.view(np.float32)
    o["slot_base"] = geom[gl.slot_base:gl.slot_base + 4 * P].view(np.uint32)
    o["rect"] = geom[gl.rect:gl.rect + 8 * P].view(np.uint32).reshape(P, 2)
    o["rshape"] = geom[gl.rshape:gl.rshape + 8 * P].view(np.uint32).reshape(P, 2)   # {rectangle in one word, trim word}: csrc/gsr_rect_trim.h
    o["final_T"] = img[il.final_T:il.final_T + 4 * N].view(np.float32)
    o["n_contrib"] = img[il.n_contrib:il.n_contrib + 4 * N].view(np.uint32)
    o["ranges"] = img[il.ranges:il.ranges + 8 * T].view(np.uint32).reshape(T, 2)
    if R > 0:
        bl = _C.binning_layout(P, R, W, H)
        o["point_list"] = binning[bl.point_list:bl.point_list + 4 * R].view(np.uint32)
        if int(bl.column_pairs) and not tile_sort:
            # column-pair binning (csrc/tilebin.hip): no per-instance tile key is ever stored; the tile of sorted instance i is
            # the tile whose range holds i.  check_forward compares `ranges` and `point_list` with the oracle's directly; the
            # keys below then restate that (and fail loudly if the ranges do not partition [0, R))
            # The list holds the instances the binning kept: all R of them with GSR_DEBUG_NO_TRIM, fewer otherwise (tiles a splat
            # provably misses are left out, csrc/gsr_rect_trim.h) -- the ranges partition [0, listed) either way.
            lens = (o["ranges"][:, 1] - o["ranges"][:, 0]).astype(np.int64)
            ne = lens > 0
            starts = o["ranges"][ne, 0].astype(np.int64)
            listed = int(lens.sum())
            assert listed <= R and np.array_equal(starts, np.concatenate([[0], np.cumsum(lens[ne])[:-1]])), "ranges do not partition [0, listed) in tile order"
            assert listed == R or bool((o["rshape"][:, 1] != 0).any()), "instances are missing from the list although nothing was trimmed"
            o["point_list"] = o["point_list"][:listed]
            o["tile_keys"] = np.repeat(np.arange(T, dtype=np.uint32), lens)
        else:
            kb = int(bl.tile_key_bytes)   # 2: uint16 tile ids (every id of the image < 65 536), 4: uint32
            o["tile_keys"] = binning[bl.tile_keys:bl.tile_keys + kb * R].view(np.uint16 if kb == 2 else np.uint32).astype(np.uint32)
        # the reference's 64-bit key of every sorted instance: tile id << 32 | depth bits
        o["keys"] = (o["tile_keys"].astype(np.uint64) << np.uint64(32)) | depth_bits[o["point_list"]].astype(np.uint64)
    return o


def trim_kept(rshape, gauss, tile, gx):
    """numpy restatement of the binning's reading of the trim words (csrc/gsr_rect_trim.h: gsr_rect_unpack, gsr_trim_col_shift,
    gsr_trim_row_shift, gsr_trim_of, gsr_trim_columns): is instance (Gaussian gauss[i], tile tile[i]) -- a tile of that Gaussian's
    rectangle -- kept in the list?"""
    P = rshape.shape[0]
    packed, trim = rshape[:, 0].astype(np.int64), rshape[:, 1].astype(np.int64)
    x0, y0, w, h = packed & 255, (packed >> 8) & 255, ((packed >> 16) & 255) + 1, (packed >> 24) + 1
    cs, rs = np.zeros(P, np.int64), np.zeros(P, np.int64)   # log2 of the columns per nibble / rows per unit
    for _ in range(6):
        cs = np.where(((w + (1 << cs) - 1) >> cs) > 8, cs + 1, cs)
        rs = np.where(((h + (1 << rs) - 1) >> rs) > 16, rs + 1, rs)
    groups = (w + (1 << cs) - 1) >> cs
    active = (trim != 0) & (rshape[:, 0] != 0xFFFFFFFF)
    first, last = np.full(P, 99, np.int64), np.full(P, -1, np.int64)   # first and last column GROUP that keeps a row
    for g_ in range(8):
        nib = (trim >> (4 * g_)) & 15
        keeps = (g_ < groups) & ((((nib & 3) + (nib >> 2)) << rs) < h)
        first = np.where(keeps & (first == 99), g_, first)
        last = np.where(keeps, g_, last)
    g = gauss.astype(np.int64)
    t64 = tile.astype(np.int64)
    c, r = t64 % gx - x0[g], t64 // gx - y0[g]
    assert bool(((c >= 0) & (c < w[g]) & (r >= 0) & (r < h[g])).all()), "an instance lies outside its Gaussian's rectangle"
    grp = c >> cs[g]
    nib = (trim[g] >> (4 * (grp & 7))) & 15
    t, b = (nib & 3) << rs[g], (nib >> 2) << rs[g]
    whole = t + b >= h[g]   # (an empty group between kept ones: the producer makes none; the binning would take it whole)
    rows_ok = whole | ((r >= t) & (r < h[g] - b))
    return ~active[g] | ((grp >= first[g]) & (grp <= last[g]) & rows_ok)


def trimmed_expectation(o, rshape, W, H):
    """The oracle's sorted instance list, keys, tile ranges and n_contrib as they must look after the binning has left out the
    instances the trim words name: same order, positions counted in the shorter lists (n_contrib = the position of the last
    contributor in ITS tile's list -- a contributor is never left out)."""
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    rng = o["ranges"].astype(np.int64)
    lens = rng[:, 1] - rng[:, 0]
    tile_of = np.repeat(np.arange(T, dtype=np.int64), lens)   # (the oracle's ranges partition [0, R) in tile order)
    kept = trim_kept(rshape, o["point_list"], tile_of, gx)
    kcum = np.concatenate([[0], np.cumsum(kept)]).astype(np.int64)
    new_len = kcum[rng[:, 1]] - kcum[rng[:, 0]]
    new_start = kcum[rng[:, 0]]
    ranges = np.where((new_len > 0)[:, None], np.stack([new_start, new_start + new_len], 1), 0).astype(np.uint32)
    ys, xs = np.divmod(np.arange(W * H, dtype=np.int64), W)
    tile_pix = (ys // 16) * gx + xs // 16
    n = o["n_contrib"].astype(np.int64)
    n_contrib = (kcum[rng[tile_pix, 0] + n] - kcum[rng[tile_pix, 0]]).astype(np.uint32)
    # the last contributor itself must have been kept: position n - 1 of the tile's list
    last = rng[tile_pix, 0] + n - 1
    assert bool(kept[last[n > 0]].all()), "the binning left out an instance the oracle blends"
    return dict(point_list=o["point_list"][kept], keys=o["keys"][kept], ranges=ranges, n_contrib=n_contrib, kept=kept, tile_of=tile_of)


def fragile_free_dpix(o, cam, seed=1):
    """Upstream gradient that is zero on pixels whose accept/reject decisions sit within the
    oracle's margin of a threshold: a legitimate 1-ulp exp() difference there flips a pixel by up
    to alpha*T and would leak into every gradient."""
    g = torch.Generator().manual_seed(seed)
    dpix = torch.randn(3, cam.image_height, cam.image_width, generator=g)
    ok = torch.from_numpy((o["fragile"] == 0).reshape(cam.image_height, cam.image_width))
    return dpix * ok
