"""ORACLE — CPU restatement of CoPoNeRF's per-ray render path.   *** TEST INFRASTRUCTURE ***

This file is the checker, never the product: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The product path
(coponerf_amd/) must not, and fails loudly when the HIP library is missing.

Parity status: the upstream repository has no tests or golden vectors
 — the oracle is pinned instead against outputs of the upstream
code itself, imported in the build container by tests/golden/make_golden.py and
committed as tests/golden/*.npz (tests/test_oracle_golden.py replays them).

What it restates (file:line relative to /root/reference):
  pose algebra                models/CoPoNeRF.py:239-244,325-332; utils_training/utils.py:111-138
  Plücker embedding           utils_training/geometry.py:236-245,426-433,409-419,353-371
  epipolar segment clipping   models/epipolar.py:175-253 (+ :74-162, :23-43)
  sample generation           models/CoPoNeRF.py:259-309
  primary/secondary gather    models/CoPoNeRF.py:312,361-370 (F.grid_sample border / zeros)
  closest point (float64)     utils_training/geometry.py:98-162
  relative points, reproject  utils_training/utils.py:99-108,242-245; utils_training/geometry.py:374-393
  per-sample MLPs, attention  models/CoPoNeRF.py:375-485
  light-field decoder phi     models/lightfield.py:9-61,131-167; models/CoPoNeRF.py:542-566
  auxiliary outputs           models/CoPoNeRF.py:230-236,493-541; utils_training/utils.py:52-69,140-170,260-276,576-602,642-671

Numerical contract: every geometric quantity that decides a *sample index*
(pixel_val, secondary sample coordinates, bilinear tap indices) is written as
an explicit sequence of IEEE-754 add/sub/mul/div/sqrt on scalars-per-element —
no BLAS, no einsum, and float32 sqrt pinned to the correctly rounded value (_sqrt_cr) — so that the HIP
kernels, compiled with -ffp-contract=off, reproduce it bit for bit on any host CPU.  Where upstream calls einsum/bmm for 3- or 4-term dot
products the summation here is left-to-right; tests/golden pins how close that
is to upstream (bit-identical on the fixtures, see tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

EPS_BOUNDS = 1e-6        # epipolar.py:30,40,180
EPS_PROJECT = 1e-8       # epipolar.py:23
ATT_SCALE = 11.31        # CoPoNeRF.py:450,475 (literal, not sqrt(128))


# ----------------------------------------------------------------------------
# (a1) pose algebra — host side, 4x4 only
# ----------------------------------------------------------------------------
def rigid_inverse(m: torch.Tensor) -> torch.Tensor:
    """[R t; 0 1]^-1 = [R^T, -R^T t; 0 1] without a general inverse (utils.py:111-138)."""
    out = torch.zeros_like(m)
    rt = m[..., :3, :3].transpose(-1, -2)
    out[..., :3, :3] = rt
    out[..., :3, 3] = (-(rt @ m[..., :3, 3:]))[..., 0]
    out[..., 3, 3] = 1
    return out


def pose_algebra(ctx_c2w: torch.Tensor, qry_c2w: torch.Tensor, rel_pose: Optional[torch.Tensor], val: bool):
    """Returns (Tq (B,V,4,4) query->context-frame, M (B,V,4,4) ~identity,
    A1, A2 (B,V,4,4): view v -> frame of view 0 / view 1)."""
    inv_ctx = torch.inverse(ctx_c2w)
    M = inv_ctx @ ctx_c2w                                            # CoPoNeRF.py:239
    if val:
        q0 = inv_ctx[:, 0].unsqueeze(1) @ qry_c2w                    # :241
        q1 = rigid_inverse(rel_pose).unsqueeze(1) @ q0               # :242
        Tq = torch.cat((q0, q1), dim=1)
        a1_0 = torch.inverse(ctx_c2w[:, 0:1]) @ ctx_c2w[:, 0].unsqueeze(1)     # :326
        A1 = torch.cat((a1_0, rel_pose.unsqueeze(1)), dim=1)                   # :327
        a2_1 = torch.inverse(ctx_c2w[:, 1:2]) @ ctx_c2w[:, -1].unsqueeze(1)    # :328
        A2 = torch.cat((rigid_inverse(rel_pose).unsqueeze(1), a2_1), dim=1)    # :329
    else:
        Tq = inv_ctx @ qry_c2w                                       # :244
        A1 = torch.inverse(ctx_c2w[:, 0:1]) @ ctx_c2w                # :331
        A2 = torch.inverse(ctx_c2w[:, 1:2]) @ ctx_c2w                # :332
    return Tq, M, A1, A2


# ----------------------------------------------------------------------------
# small explicit-order helpers (scalar-per-element IEEE arithmetic only)
# ----------------------------------------------------------------------------
def _dot3(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _sqrt_cr(x):
    """Correctly rounded sqrt.  torch.sqrt on float32 CPU tensors is a vectorised approximation whose last bit
    depends on the CPU (0.7 % of results off by 1 ulp on a Xeon, 19.6 % on an EPYC 9575F — measured); the
    float64 route is exact after rounding and identical everywhere, and equals gfx950's IEEE sqrtf."""
    return torch.sqrt(x.double()).to(x.dtype) if x.dtype == torch.float32 else torch.sqrt(x)


def _norm3(v):
    return _sqrt_cr((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])


def _normalize3(v, eps=1e-12):
    n = torch.clamp_min(_norm3(v), eps)          # F.normalize: v / max(||v||, eps)
    return (v[0] / n, v[1] / n, v[2] / n)


def _affine_rows(T, p):
    """rows 0..2 of T(4x4) applied to homogeneous p=(x,y,z,w), left-to-right sum."""
    return tuple(((T[..., i, 0] * p[0] + T[..., i, 1] * p[1]) + T[..., i, 2] * p[2]) + T[..., i, 3] * p[3]
                 for i in range(3))


def _scrub(x, value):
    """NaN/±Inf -> value."""
    return torch.where(torch.isfinite(x), x, torch.full_like(x, value))


# ----------------------------------------------------------------------------
# (a2) Plücker coordinates of the query rays in each context frame
# ----------------------------------------------------------------------------
def plucker_rays(T: torch.Tensor, uv: torch.Tensor, K: torch.Tensor):
    """T (N,4,4) camera->frame, uv (N,R,2) pixels, K (N,4,4) -> dir (N,R,3), moment (N,R,3), origin (N,3).

    geometry.py:236-245 (plucker_embedding) -> :426-433 -> :409-419 -> :353-371."""
    fx, fy, cx, cy = (K[:, 0, 0, None], K[:, 1, 1, None], K[:, 0, 2, None], K[:, 1, 2, None])
    one = torch.ones_like(uv[..., 0])
    xl = (uv[..., 0] - cx) / fx * one
    yl = (uv[..., 1] - cy) / fy * one
    Tb = T[:, None]
    w = _affine_rows(Tb, (xl, yl, one, one))
    o = (T[:, None, 0, 3], T[:, None, 1, 3], T[:, None, 2, 3])
    d = _normalize3((w[0] - o[0], w[1] - o[1], w[2] - o[2]))
    ob = tuple(x.expand_as(d[0]) for x in o)
    m = _cross(ob, d)
    return torch.stack(d, -1), torch.stack(m, -1), T[:, :3, 3]


# ----------------------------------------------------------------------------
# (a3) epipolar segment: clip the projection of o + t d, t in [0, inf) to the unit square
# ----------------------------------------------------------------------------
def _in_bounds(x, y):
    lo, hi = -EPS_BOUNDS, 1 + EPS_BOUNDS
    return (x >= lo) & (y >= lo) & (x <= hi) & (y <= hi)


def _frame_hit(Kn, o, d, dim: int, value: float):
    """Intersection of the projected ray with the image-frame line x=value or y=value
    (epipolar.py:74-122).  Returns t, x, y, valid."""
    other = 1 - dim
    fs, fo = Kn[:, dim, dim, None], Kn[:, other, other, None]
    cs, co = Kn[:, dim, 2, None], Kn[:, other, 2, None]
    os_, oo, ds, do = o[dim], o[other], d[dim], d[other]
    oz, dz = o[2], d[2]
    c = (value - cs) / fs
    t = (c * oz - os_) / (ds - c * dz)
    num = fo * (oo * (c * dz - ds) + do * (os_ - c * oz))
    den = dz * os_ - ds * oz
    coord_other = co + num / den
    coord_same = torch.ones_like(coord_other) * value
    x, y = (coord_same, coord_other) if dim == 0 else (coord_other, coord_same)
    z = oz + t * dz
    valid = _in_bounds(x, y) & (z > -EPS_BOUNDS)
    return t, x, y, valid


def _pick(cands, take_max: bool):
    """First-index arg-min/max over candidates whose invalid t was set to ±inf (epipolar.py:125-149)."""
    fill = -math.inf if take_max else math.inf
    ts = [torch.where(v, t, torch.full_like(t, fill)) for (t, _, _, v) in cands]
    bt, bx, by, bv = ts[0], cands[0][1], cands[0][2], cands[0][3]
    for i in range(1, len(cands)):
        better = (ts[i] > bt) if take_max else (ts[i] < bt)
        bt = torch.where(better, ts[i], bt)
        bx = torch.where(better, cands[i][1], bx)
        by = torch.where(better, cands[i][2], by)
        bv = torch.where(better, cands[i][3], bv)
    return bt, bx, by, bv


def _project_point(Kn, p):
    """epipolar.py:23-26: p/(p_z+1e-8) then K(3x3) @ p, left-to-right."""
    q = p[2] + EPS_PROJECT
    h = (p[0] / q, p[1] / q, p[2] / q)
    k = lambda i, j: Kn[:, i, j, None]
    x = (k(0, 0) * h[0] + k(0, 1) * h[1]) + k(0, 2) * h[2]
    y = (k(1, 0) * h[0] + k(1, 1) * h[1]) + k(1, 2) * h[2]
    return x, y


def project_rays(origin: torch.Tensor, direction: torch.Tensor, Kn: torch.Tensor):
    """origin (N,3) (one per camera), direction (N,R,3), Kn (N,3,3) normalised intrinsics.
    -> xy_min (N,R,2), xy_max (N,R,2), overlaps (N,R) bool, t_min, t_max.   epipolar.py:175-253
    (extrinsics = identity at the only call site, CoPoNeRF.py:265)."""
    d = (direction[..., 0], direction[..., 1], direction[..., 2])
    o = tuple(origin[:, i, None].expand_as(d[0]) for i in range(3))
    cands = [_frame_hit(Kn, o, d, 0, 0.0), _frame_hit(Kn, o, d, 0, 1.0),
             _frame_hit(Kn, o, d, 1, 0.0), _frame_hit(Kn, o, d, 1, 1.0)]
    fmin = _pick(cands, take_max=False)
    fmax = _pick(cands, take_max=True)

    depth_zero = o[2] < EPS_BOUNDS
    at_camera = _norm3(o) < EPS_BOUNDS
    p0 = tuple(torch.where(at_camera, d[i], o[i]) for i in range(3))
    x0, y0 = _project_point(Kn, p0)
    v0 = _in_bounds(x0, y0) & (p0[2] > -EPS_BOUNDS) & ~(depth_zero & ~at_camera)
    xi, yi = _project_point(Kn, d)
    vi = _in_bounds(xi, yi) & (d[2] > -EPS_BOUNDS)

    zero = torch.zeros_like(x0)
    t_min = torch.where(v0, zero, fmin[0])
    x_min = torch.where(v0, x0, fmin[1])
    y_min = torch.where(v0, y0, fmin[2])
    ok_min = torch.where(v0, v0, fmin[3])
    t_max = torch.where(vi, torch.full_like(x0, math.inf), fmax[0])
    x_max = torch.where(vi, xi, fmax[1])
    y_max = torch.where(vi, yi, fmax[2])
    ok_max = torch.where(vi, vi, fmax[3])
    return (torch.stack((x_min, y_min), -1), torch.stack((x_max, y_max), -1), ok_min & ok_max, t_min, t_max)


# ----------------------------------------------------------------------------
# (a4) sample coordinates on the segment
# ----------------------------------------------------------------------------
def sample_coords(xy_min, xy_max, S: int):
    """-> pixel_val (N,R,S,2) in [-1,1].  CoPoNeRF.py:279-309."""
    start = _scrub((xy_min - 0.5) * 2, 0.0)
    end = _scrub((xy_max - 0.5) * 2, 0.0)
    interval = torch.linspace(0, 1, S)
    diff = end[:, :, None, :] - start[:, :, None, :]
    return start[:, :, None, :] + diff * interval[None, None, :, None]


# ----------------------------------------------------------------------------
# bilinear tap indices / weights exactly as grid_sample(align_corners=False) forms them
# ----------------------------------------------------------------------------
def bilinear_taps(g: torch.Tensor, Wl: int, Hl: int, border: bool):
    """g (...,2) normalised coords -> (ix0, iy0 int64 floor indices, fx, fy fractional weights).
    border=True clamps the unnormalised coordinate to [0, size-1] first (padding_mode='border');
    border=False leaves it (padding_mode='zeros': taps outside contribute 0).
    Unnormalisation ((g+1)*size-1)/2 as ATen's grid_sampler_unnormalize."""
    x = ((g[..., 0] + 1) * Wl - 1) / 2
    y = ((g[..., 1] + 1) * Hl - 1) / 2
    if border:
        x = torch.clamp(x, 0, Wl - 1)
        y = torch.clamp(y, 0, Hl - 1)
    x0, y0 = torch.floor(x), torch.floor(y)
    return x0.to(torch.int64), y0.to(torch.int64), x - x0, y - y0


# ----------------------------------------------------------------------------
# (a7) closest point on the query line to each context pixel ray — float64 island
# ----------------------------------------------------------------------------
def context_pixel_rays(pixel_val, M, Kc, H: int, W: int):
    """Plücker coords of the context-camera ray through every sample (geometry.py:100-109)."""
    px = (pixel_val[..., 0] + 1) / 2 * (W - 1)
    py = (pixel_val[..., 1] + 1) / 2 * (H - 1)
    fx, fy, cx, cy = (Kc[:, 0, 0, None, None], Kc[:, 1, 1, None, None], Kc[:, 0, 2, None, None], Kc[:, 1, 2, None, None])
    one = torch.ones_like(px)
    xl = (px - cx) / fx * one
    yl = (py - cy) / fy * one
    Mb = M[:, None, None]
    w = _affine_rows(Mb, (xl, yl, one, one))
    o = (M[:, None, None, 0, 3], M[:, None, None, 1, 3], M[:, None, None, 2, 3])
    dc = _normalize3((w[0] - o[0], w[1] - o[1], w[2] - o[2]))
    ob = tuple(x.expand_as(dc[0]) for x in o)
    mc = _cross(ob, dc)
    return dc, mc


def closest_point_on_query(dir_q, mom_q, dc, mc):
    """p1 of geometry.py:132-162 in float64, NaN/Inf -> 0, cast to float32 (geometry.py:126-129)."""
    l1 = tuple(dir_q[..., i, None].double() for i in range(3))
    m1 = tuple(mom_q[..., i, None].double() for i in range(3))
    l2 = tuple(x.double() for x in dc)
    m2 = tuple(x.double() for x in mc)
    l1 = tuple(x.expand_as(l2[0]) for x in l1)
    m1 = tuple(x.expand_as(l2[0]) for x in m1)
    c12 = _cross(l1, l2)
    l2c = _cross(l2, c12)
    mt = _cross(m1, l2c)
    s = _dot3(m2, c12)
    n = torch.sqrt((c12[0] * c12[0] + c12[1] * c12[1]) + c12[2] * c12[2])
    cd = n * n + 1e-12
    p = tuple(_scrub((-mt[i] + s * l1[i]) / cd, 0.0) for i in range(3))
    return torch.stack(p, -1).float()


# ----------------------------------------------------------------------------
# (a8, a9) points in the other view's frame, reprojection
# ----------------------------------------------------------------------------
def transform_points(pt, A):
    """pt (N,R,S,3), A (N,4,4): rows 0..2 of A @ [pt,1], left-to-right (utils.py:99-108)."""
    p = (pt[..., 0], pt[..., 1], pt[..., 2], torch.ones_like(pt[..., 0]))
    Ab = A[:, None, None]
    r = tuple(((p[0] * Ab[..., i, 0] + p[1] * Ab[..., i, 1]) + p[2] * Ab[..., i, 2]) + p[3] * Ab[..., i, 3]
              for i in range(3))
    return torch.stack(r, -1)


def reproject(pt, K, H: int, W: int):
    """pinhole projection with NaN/Inf -> 1e10, then to grid_sample units
    (geometry.py:374-393; utils.py:242-245).  pt (N,R,S,3), K (N,4,4)."""
    fx, fy, cx, cy = (K[:, 0, 0, None, None], K[:, 1, 1, None, None], K[:, 0, 2, None, None], K[:, 1, 2, None, None])
    z = pt[..., 2] + 1e-12
    xp = _scrub(fx * pt[..., 0] / z + cx, 1e10)
    yp = _scrub(fy * pt[..., 1] / z + cy, 1e10)
    return torch.stack(((xp / (W - 1)) * 2 - 1, (yp / (H - 1)) * 2 - 1), -1)


# ----------------------------------------------------------------------------
# (a14) per-sample geometric "query" features, 16 channels
# ----------------------------------------------------------------------------
def local_coords(pixel_val, Kc, H, W, dir_q, origin_q, pt):
    """[ctx ray dir 3 | zeros 3 | query dir 3 | tanh(depth*{1,.1,.01,.001}) 4 | query origin 3]
    CoPoNeRF.py:411-445; geometry.py:313-324."""
    px = (pixel_val[..., 0] + 1) / 2 * (W - 1)
    py = (pixel_val[..., 1] + 1) / 2 * (H - 1)
    fx, fy, cx, cy = (Kc[:, 0, 0, None, None], Kc[:, 1, 1, None, None], Kc[:, 0, 2, None, None], Kc[:, 1, 2, None, None])
    one = torch.ones_like(px)
    cam = _normalize3(((px - cx) / fx * one, (py - cy) / fy * one, one))
    oq = origin_q[:, None, None, :]
    dlt = pt - oq
    depth = _scrub(_sqrt_cr((dlt[..., 0] * dlt[..., 0] + dlt[..., 1] * dlt[..., 1]) + dlt[..., 2] * dlt[..., 2]), 1e6)
    enc = [torch.tanh(depth), torch.tanh(depth / 10.), torch.tanh(depth / 100.), torch.tanh(depth / 1000.)]
    zero = torch.zeros_like(px)
    dq = dir_q[:, :, None, :].expand(-1, -1, px.shape[2], -1)
    chans = [cam[0], cam[1], cam[2], zero, zero, zero, dq[..., 0], dq[..., 1], dq[..., 2], *enc,
             oq[..., 0].expand_as(px), oq[..., 1].expand_as(px), oq[..., 2].expand_as(px)]
    return torch.stack(chans, -1), depth


# ----------------------------------------------------------------------------
# feature gathers (ATen's own grid_sampler, i.e. exactly the op upstream calls)
# ----------------------------------------------------------------------------
def gather_levels(z: Sequence[torch.Tensor], grid: torch.Tensor, padding: str):
    """z[l] (N,C_l,h_l,w_l), grid (N,R,S,2) -> (N,R,S,sum C_l) channels-last."""
    outs = [F.grid_sample(lat, grid, mode="bilinear", padding_mode=padding, align_corners=False) for lat in z]
    return torch.cat(outs, dim=1).permute(0, 2, 3, 1)


def _lin(x, w, b):
    return F.linear(x, w.reshape(w.shape[0], -1), b)


# ----------------------------------------------------------------------------
# the whole render path
# ----------------------------------------------------------------------------
def render_core(inp: Dict, z: Sequence[torch.Tensor], rel_pose, val: bool, w: Dict[str, torch.Tensor],
                S: int, H: int, W: int, keep: bool = False) -> Dict:
    """Everything between the input dict and rgb, minus the flow-based auxiliaries.
    Returns a dict with the outputs and (keep=True) every intermediate the kernel tests compare."""
    ctx, qry = inp["context"], inp["query"]
    B, V = ctx["rgb"].shape[:2]
    R = qry["uv"].shape[2]
    N = B * V
    Tq, M, A1, A2 = pose_algebra(ctx["cam2world"], qry["cam2world"], rel_pose, val)
    Tq_f, M_f = Tq.reshape(N, 4, 4), M.reshape(N, 4, 4)
    Kc = ctx["intrinsics"].reshape(N, 4, 4)
    Kq = qry["intrinsics"].expand(-1, V, -1, -1).reshape(N, 4, 4)
    uv = qry["uv"].expand(-1, V, -1, -1).reshape(N, R, 2)

    dir_q, mom_q, origin_q = plucker_rays(Tq_f, uv, Kq)
    Kn = Kc[:, :3, :3].clone()
    Kn[:, :2, :] = Kn[:, :2, :] / H                                   # CoPoNeRF.py:259-261 (both rows by H)
    xy_min, xy_max, overlaps, _, _ = project_rays(origin_q, dir_q, Kn)
    pixel_val = sample_coords(xy_min, xy_max, S)                      # (N,R,S,2)

    prim = gather_levels(z, pixel_val, "border")                      # (N,R,S,832)
    dc, mc = context_pixel_rays(pixel_val, M_f, Kc, H, W)
    pt = closest_point_on_query(dir_q, mom_q, dc, mc)                 # (N,R,S,3)
    pt_in1 = transform_points(pt, A1.reshape(N, 4, 4)).view(B, V, R, S, 3)
    pt_in2 = transform_points(pt, A2.reshape(N, 4, 4)).view(B, V, R, S, 3)
    # secondary coordinates: view-1 samples seen from image 0 (K of view 0), view-0 samples from image 1
    g_img0 = reproject(pt_in1[:, 1], ctx["intrinsics"][:, 0], H, W)
    g_img1 = reproject(pt_in2[:, 0], ctx["intrinsics"][:, 1], H, W)
    sec_grid = torch.stack([g_img0, g_img1], dim=1).reshape(N, R, S, 2)
    sec = gather_levels(z, sec_grid, "zeros").view(B, V, R, S, -1)    # [:,0]=image-0 feats @ view-1 samples
    prim5 = prim.view(B, V, R, S, -1)

    nn0 = lambda t: torch.nan_to_num(t, 0)
    pe = lambda t: torch.tanh(nn0(t) / 5.)
    # the four encoder inputs (CoPoNeRF.py:384-394); index [v][j]: j=0 own image, j=1 other image
    x_in = torch.stack([
        torch.stack([torch.cat([prim5[:, 0], pe(pt_in1[:, 0])], -1), torch.cat([sec[:, 1], pe(pt_in2[:, 0])], -1)], 3),
        torch.stack([torch.cat([prim5[:, 1], pe(pt_in2[:, 1])], -1), torch.cat([sec[:, 0], pe(pt_in1[:, 1])], -1)], 3),
    ], dim=1)                                                         # (B,V,R,S,2,835)
    hid = F.relu(_lin(x_in, w["query_encode_latent.weight"], w["query_encode_latent.bias"]))
    enc = _lin(hid, w["query_encode_latent_2.weight"], w["query_encode_latent_2.bias"])   # (B,V,R,S,2,416)
    X = enc.reshape(B, V, R, S, 832)
    value = _lin(X, w["latent_value.weight"], w["latent_value.bias"])                     # (B,V,R,S,416)
    key = _lin(F.relu(_lin(X, w["key_map.weight"], w["key_map.bias"])), w["key_map_2.weight"], w["key_map_2.bias"])

    L, depth = local_coords(pixel_val, Kc, H, W, dir_q, origin_q, pt)                     # (N,R,S,16)
    L5 = L.view(B, V, R, S, 16)
    ce = _lin(F.relu(_lin(L5, w["query_embed.weight"], w["query_embed.bias"])),
              w["query_embed_2.weight"], w["query_embed_2.bias"])                         # (B,V,R,S,128)

    def joint_softmax(logit):                                                             # (B,V,R,S)
        flat = logit.permute(0, 2, 1, 3).reshape(B, R, V * S)
        return F.softmax(flat, dim=-1).view(B, R, V, S).permute(0, 2, 1, 3)

    w1 = joint_softmax((key * ce).sum(-1) / ATT_SCALE)                                    # (B,V,R,S)
    z1 = (value * w1[..., None]).sum(dim=3).sum(dim=1)                                    # (B,R,416)
    ze = _lin(z1, w["encode_latent.weight"], w["encode_latent.bias"])                     # (B,R,128)
    q2_in = torch.cat([ze[:, None, :, None, :].expand(-1, V, -1, S, -1), L5], -1)         # (B,V,R,S,144)
    q2 = _lin(F.relu(_lin(q2_in, w["query_repeat_embed.weight"], w["query_repeat_embed.bias"])),
              w["query_repeat_embed_2.weight"], w["query_repeat_embed_2.bias"])
    w2 = joint_softmax((q2 * ce).sum(-1) / ATT_SCALE)
    # quirk kept: the round-1 vector sits in BOTH view slots when the per-view sums are added up
    # (CoPoNeRF.py:481-485), so it enters the result n_context = 2 times.
    zl = ((value * w2[..., None]).sum(dim=3) + z1[:, None]).sum(dim=1)                    # (B,R,416)

    coords9 = torch.cat([dir_q, mom_q, origin_q[:, None, :].expand(-1, R, -1)], -1).view(B, V, R, 9)
    phi_in = torch.cat([zl, zl, coords9[:, 0], coords9[:, 1]], -1)                        # (B,R,850)
    rgb_raw = phi_forward(phi_in, w)
    valid = overlaps.view(B, V, R).any(dim=1).float()                                     # (B,R)
    rgb = rgb_raw * valid[..., None] + 1 * (1 - valid[..., None])

    out = {
        "rgb": rgb.view(B, 1, R, 3), "valid_mask": valid[..., None], "pixel_val": pixel_val,
        "at_wt": w1.reshape(N, R, S), "z_local": zl, "coords": coords9.reshape(N, R, 9),
        "pt": pt, "Tq": Tq, "M": M, "A1": A1, "A2": A2,
    }
    if keep:
        out.update({
            "dir_q": dir_q, "mom_q": mom_q, "origin_q": origin_q, "xy_min": xy_min, "xy_max": xy_max,
            "overlaps": overlaps, "sec_grid": sec_grid, "prim": prim, "sec": sec.reshape(N, R, S, -1),
            "pt_in1": pt_in1, "pt_in2": pt_in2, "x_in": x_in, "hid": hid, "X": X, "value": value, "key": key,
            "local": L, "depth": depth, "ce": ce, "w2": w2.reshape(N, R, S), "z1": z1, "rgb_raw": rgb_raw,
        })
    return out


def phi_forward(zx: torch.Tensor, w: Dict[str, torch.Tensor]) -> torch.Tensor:
    """lightfield.ResnetFC with d_in=18, d_latent=832, 3 blocks (lightfield.py:131-167, :52-61)."""
    zlat, x = zx[..., :832], zx[..., 832:]
    x = _lin(x, w["phi.lin_in.weight"], w["phi.lin_in.bias"])
    for k in range(3):
        x = x + _lin(zlat, w[f"phi.lin_z.{k}.weight"], w[f"phi.lin_z.{k}.bias"])
        net = _lin(F.relu(x), w[f"phi.blocks.{k}.fc_0.weight"], w[f"phi.blocks.{k}.fc_0.bias"])
        x = x + _lin(F.relu(net), w[f"phi.blocks.{k}.fc_1.weight"], w[f"phi.blocks.{k}.fc_1.bias"])
    return _lin(F.relu(x), w["phi.lin_out.weight"], w["phi.lin_out.bias"])


# ----------------------------------------------------------------------------
# (a20) flow-based auxiliaries — consumed by the cycle/ssim losses and the summaries only
# ----------------------------------------------------------------------------
def _pixel_grid(Bn, Hn, Wn, device):
    xx = torch.arange(0, Wn, device=device).view(1, 1, 1, Wn).expand(Bn, 1, Hn, Wn)
    yy = torch.arange(0, Hn, device=device).view(1, 1, Hn, 1).expand(Bn, 1, Hn, Wn)
    return torch.cat((xx, yy), 1).float()


def warp_by_flow(x, flo):
    """sample x at (pixel + flow); utils.py:642-671 (default grid_sample: bilinear, zeros, align_corners=False)."""
    Bn, _, Hn, Wn = x.shape
    vgrid = _pixel_grid(Bn, Hn, Wn, x.device) + flo
    gx = 2.0 * vgrid[:, 0] / max(Wn - 1, 1) - 1.0
    gy = 2.0 * vgrid[:, 1] / max(Hn - 1, 1) - 1.0
    return F.grid_sample(x, torch.stack((gx, gy), -1), align_corners=False)


def in_image_mask(flow):
    """utils.py:576-602 on a (B,2,H,W) torch flow."""
    Bn, _, Hn, Wn = flow.shape
    m = flow + _pixel_grid(Bn, Hn, Wn, flow.device)
    return m[:, 0].ge(0) & m[:, 0].le(Wn - 1) & m[:, 1].ge(0) & m[:, 1].le(Hn - 1)


def cycle_masks(flow, img_w: int):
    """CoPoNeRF.py:230-236 (scale is 256 / rgb.shape[-2], i.e. the image WIDTH of the (B,V,H,W,3) tensor)."""
    up1 = F.interpolate(flow[0], 256, mode="bilinear") * (256 / img_w)
    up2 = F.interpolate(flow[1], 256, mode="bilinear") * (256 / img_w)
    e1 = torch.norm(up1 + warp_by_flow(up2, up1), dim=1).le(10)
    e2 = torch.norm(up2 + warp_by_flow(up1, up2), dim=1).le(10)
    return e1 * in_image_mask(up1), e2 * in_image_mask(up2)


def project_to_other(kp, depth, Ki, Kj, T):
    """utils.py:140-170 (+ to/from_homogeneous :71-96; from_homogeneous divides by (w + 1e-6))."""
    ones = kp.new_ones(kp.shape[:-1] + (1,))
    p = torch.cat([kp, ones], -1) @ torch.inverse(Ki).transpose(-1, -2)
    p = p * depth[..., None]
    q = torch.cat([p, ones], -1) @ T.transpose(-1, -2)
    q = q[..., :-1] / (q[..., -1:] + 1e-6)
    r = q @ Kj.transpose(-1, -2)
    return r[..., :-1] / (r[..., -1:] + 1e-6)


def aux_outputs(inp, flow, core: Dict, Tq, S: int):
    """depth_ray, reprojections, masks, argmax (CoPoNeRF.py:493-541)."""
    ctx, qry = inp["context"], inp["query"]
    B, V = ctx["rgb"].shape[:2]
    R = qry["uv"].shape[2]
    at_wt, pt = core["at_wt"], core["pt"]
    mask1, mask2 = cycle_masks(flow, ctx["rgb"].shape[-2])
    at_max = at_wt.argmax(dim=-1)[..., None]                                   # (N,R,1) int64
    exp_pt = (at_wt[..., None] * torch.clamp(pt, -100, 100)).sum(dim=-2)       # (N,R,3)
    exp_pt = exp_pt.view(B, V, R, 3).sum(dim=1)
    hom = torch.cat((exp_pt, torch.ones(B, R, 1, device=exp_pt.device)), dim=2).permute(0, 2, 1)
    cam = torch.inverse(qry["cam2world"][:, 0]).bmm(hom).permute(0, 2, 1)[..., :3]   # geometry.py:395-406
    depth_ray = cam[:, :, 2]
    uvq = qry["uv"].squeeze(1)
    t1 = project_to_other(uvq, depth_ray, qry["intrinsics"][:, 0, :3, :3], ctx["intrinsics"][:, 0, :3, :3], Tq[:, 0])
    t2 = project_to_other(uvq, depth_ray, qry["intrinsics"][:, 0, :3, :3], ctx["intrinsics"][:, 1, :3, :3], Tq[:, 1])
    n_pts = depth_ray.shape[-1]
    # utils.py:260-276: confidence lookup at truncated, clamped pixel positions
    kp = torch.clamp(t2.long().transpose(1, 2)[:, :, :n_pts], 0, 255)          # (B,2,R)
    bidx = torch.arange(B, device=kp.device)[:, None]
    match_mask = mask2[bidx, kp[:, 1], kp[:, 0]]
    # utils.py:52-69
    hf = flow[1].shape[2]
    flow_up = F.interpolate(flow[1], (256, 256), mode="bilinear") * (256 / hf)
    tl = t2.long()
    inb = ((0 <= tl) & (tl < 256))
    mask_c2 = inb[..., 0] & inb[..., 1]
    cidx = torch.arange(2, device=kp.device)[None, :, None]
    src = kp + flow_up[bidx[:, :, None], cidx, kp[:, 1:2], kp[:, 0:1]]                 # (B,2,R)
    return {
        "matchability_cycle_mask": match_mask, "T_to_C1_pts": t1, "T_to_C2_pts": t2, "mask_c2": mask_c2,
        "C2_pts_to_C1": src.transpose(1, 2), "at_wt_max": at_max, "depth_ray": torch.clamp(depth_ray, 0, 10)[..., None],
    }


def forward(inp: Dict, z: Sequence[torch.Tensor], rel_pose: torch.Tensor, flow, val: bool,
            weights: Dict[str, torch.Tensor], npoints: int = 64, H: Optional[int] = None, W: Optional[int] = None,
            keep: bool = False) -> Dict:
    """Same contract as CoPoNeRF.forward(input, z=z, rel_pose=rel_pose, val=val, flow=flow) with the
    render-path weights given as a state_dict-style mapping (CoPoNeRF.py:208-576)."""
    ctx = inp["context"]
    if H is None:
        H, W = ctx["rgb"].shape[2], ctx["rgb"].shape[3]
    core = render_core(inp, z, rel_pose, val, weights, npoints, H, W, keep=keep)
    out = dict(core) if keep else {}
    aux = aux_outputs(inp, flow, core, core["Tq"], npoints)
    out.update(aux)
    out.update({
        "flow": flow, "uv": inp["query"]["uv"], "coords": core["coords"], "pixel_val": core["pixel_val"],
        "at_wts": [core["at_wt"]], "at_wt": core["at_wt"], "valid_mask": core["valid_mask"], "rgb": core["rgb"],
        "z": z, "rel_pose": rel_pose, "rel_pose_flip": rigid_inverse(rel_pose),
        "gt_rel_pose": torch.inverse(ctx["cam2world"][:, 0]) @ ctx["cam2world"][:, 1],
        "gt_rel_pose_flip": torch.inverse(torch.inverse(ctx["cam2world"][:, -1]) @ ctx["cam2world"][:, 0]),
    })
    return out
