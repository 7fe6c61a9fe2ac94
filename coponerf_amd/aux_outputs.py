"""Auxiliary outputs of CoPoNeRF.forward that only feed the cycle/ssim losses and the summaries.

<0.5 % of the reference's time: cycle-consistency masks from the flows, the
attention-weighted expected 3-D point and its reprojections.  Inference (`ray_outputs`): the per-ray part is ONE
kernel (cpn_ray_outputs, csrc/rayout.hip) — as ~30 stock launches it cost the host more than a 3 641-ray forward()
call of the reference's evaluation loop takes on the GPU; the per-image part (flow upsampling, cycle masks) stays stock
ops, computed once per flow pair.  Training (`aux_outputs`): stock differentiable PyTorch ops on device tensors — no
per-batch Python loops and no host syncs, unlike /root/reference utils_training/utils.py:52-69,260-276.
Cites: models/CoPoNeRF.py:230-236,493-541; utils_training/utils.py:140-170,576-602,642-671;
utils_training/geometry.py:395-406.
"""
from __future__ import annotations

import weakref
from typing import Dict, Sequence

import torch
import torch.nn.functional as F

from ._hip import stream_handle as _stream_handle


_GRIDS: Dict = {}


def _grid(B: int, H: int, W: int, device) -> torch.Tensor:
    key = (H, W, str(device))
    g = _GRIDS.get(key)
    if g is None:                      # a constant: built once per (size, device), not with 8 launches per use
        ys, xs = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device), indexing="ij")
        g = _GRIDS[key] = torch.stack((xs, ys), 0).float()[None]
    return g.expand(B, -1, -1, -1)


def _warp(x: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    B, _, H, W = x.shape
    g = _grid(B, H, W, x.device) + flow
    gx = 2.0 * g[:, 0] / max(W - 1, 1) - 1.0
    gy = 2.0 * g[:, 1] / max(H - 1, 1) - 1.0
    return F.grid_sample(x, torch.stack((gx, gy), -1), align_corners=False)


def _inside(flow: torch.Tensor) -> torch.Tensor:
    B, _, H, W = flow.shape
    m = flow + _grid(B, H, W, flow.device)
    return m[:, 0].ge(0) & m[:, 0].le(W - 1) & m[:, 1].ge(0) & m[:, 1].le(H - 1)


def cycle_masks(flow: Sequence[torch.Tensor], width: int):
    """Forward/backward flow consistency (<= 10 px) and in-image masks at 256x256 (CoPoNeRF.py:230-236)."""
    up1 = F.interpolate(flow[0], 256, mode="bilinear") * (256 / width)
    up2 = F.interpolate(flow[1], 256, mode="bilinear") * (256 / width)
    m1 = torch.norm(up1 + _warp(up2, up1), dim=1).le(10) * _inside(up1)
    m2 = torch.norm(up2 + _warp(up1, up2), dim=1).le(10) * _inside(up2)
    return m1, m2


def _reproject(kp, depth, Ki_inv, Kj, T):
    ones = kp.new_ones(kp.shape[:-1] + (1,))
    p = torch.cat([kp, ones], -1) @ Ki_inv.transpose(-1, -2)
    p = p * depth[..., None]
    q = torch.cat([p, ones], -1) @ T.transpose(-1, -2)
    q = q[..., :-1] / (q[..., -1:] + 1e-6)
    r = q @ Kj.transpose(-1, -2)
    return r[..., :-1] / (r[..., -1:] + 1e-6)


_FLOW_CACHE: Dict[str, list] = {}


def _flow_products(flow: Sequence[torch.Tensor], width: int):
    return flow_products(flow, width)[0]


def flow_products(flow: Sequence[torch.Tensor], width: int):
    """((cycle mask of view 2, flow upsampled to 256x256), cache hit?): functions of the flows alone, so a full-image
    render that calls forward() once per ray chunk with the same `flow` tensors computes them once.  The cache holds the
    three most recently used pairs: the one being rendered and the (up to two) pairs RenderEngine.prepare_next() built
    beside it on its own stream - such an entry carries the event of its last kernel; its first use makes the current
    stream wait for it and is reported as a miss (the caller orders its other streams after a miss)."""
    key = tuple((f._version, tuple(f.shape)) for f in flow[:2]) + (width,)
    entries = _FLOW_CACHE.setdefault("entries", [])
    # an entry is matched on the IDENTITY (+ version) of the flow tensors, held as weak references: a new pair's flows that
    # happen to be allocated at a freed pair's addresses can never hit it, and this module-level cache never keeps a
    # tensor of a captured get_z graph's private pool alive past the graph (freeing one after the graph is gone crashed
    # at interpreter exit)
    for i, e in enumerate(entries):
        if e[0] == key and e[3][0]() is flow[0] and e[3][1]() is flow[1]:
            hit = True
            if e[4] is not None:
                # no record_stream: when the entry is dropped its blocks return to the preparation stream's pool, and that
                # stream starts every preparation by waiting for what the caller's stream holds at that moment
                torch.cuda.current_stream(e[1].device).wait_event(e[4])
                e, hit = e[:4] + (None,), False
            if i or not hit:
                entries.pop(i)
                entries.insert(0, e)
            return (e[1], e[2]), hit
    _, mask2 = cycle_masks(flow, width)
    mask2 = mask2.contiguous()
    flow_up = (F.interpolate(flow[1], (256, 256), mode="bilinear") * (256 / flow[1].shape[2])).contiguous()
    if not any(f.requires_grad for f in flow[:2]):
        entries.insert(0, (key, mask2, flow_up, (weakref.ref(flow[0]), weakref.ref(flow[1])), None))
        del entries[3:]
    return (mask2, flow_up), False


def prepare_flow_products(flow: Sequence[torch.Tensor], width: int, stream: "torch.cuda.Stream") -> None:
    """flow_products() of a pair that is rendered LATER, computed on `stream` (which the caller has ordered after the
    flows' producer) and parked in the cache (least recently used entry out)."""
    if any(f.requires_grad for f in flow[:2]):
        return
    key = tuple((f._version, tuple(f.shape)) for f in flow[:2]) + (width,)
    entries = _FLOW_CACHE.setdefault("entries", [])
    if any(e[0] == key and e[3][0]() is flow[0] and e[3][1]() is flow[1] for e in entries):
        return
    with torch.cuda.stream(stream):
        _, mask2 = cycle_masks(flow, width)
        mask2 = mask2.contiguous()
        flow_up = (F.interpolate(flow[1], (256, 256), mode="bilinear") * (256 / flow[1].shape[2])).contiguous()
        ready = torch.cuda.Event()
        ready.record(stream)
    entries.insert(0, (key, mask2, flow_up, (weakref.ref(flow[0]), weakref.ref(flow[1])), ready))
    del entries[3:]


def aux_outputs(inp: Dict, flow: Sequence[torch.Tensor], at_wt: torch.Tensor, pt: torch.Tensor,
                Tq: torch.Tensor, inv_Kq: torch.Tensor = None, inv_qc2w: torch.Tensor = None) -> Dict[str, torch.Tensor]:
    """inv_Kq (B,3,3) / inv_qc2w (B,4,4): inverses of the query intrinsics / pose computed with the host pose algebra
    (a GPU torch.inverse is a host synchronisation); computed here if not given."""
    ctx, qry = inp["context"], inp["query"]
    B, V = ctx["rgb"].shape[:2]
    R = qry["uv"].shape[2]
    dev = at_wt.device
    mask2, flow_up = _flow_products(flow, ctx["rgb"].shape[-2])
    at_max = at_wt.argmax(dim=-1)[..., None]
    expected = (at_wt[..., None] * torch.clamp(pt, -100, 100)).sum(dim=-2).view(B, V, R, 3).sum(dim=1)
    hom = torch.cat((expected, torch.ones(B, R, 1, device=dev)), dim=2).permute(0, 2, 1)
    if inv_qc2w is None:
        inv_qc2w = torch.inverse(qry["cam2world"][:, 0])
    if inv_Kq is None:
        inv_Kq = torch.inverse(qry["intrinsics"][:, 0, :3, :3])
    depth_ray = inv_qc2w.bmm(hom).permute(0, 2, 1)[..., 2]
    uvq = qry["uv"].squeeze(1)
    t1 = _reproject(uvq, depth_ray, inv_Kq, ctx["intrinsics"][:, 0, :3, :3], Tq[:, 0])
    t2 = _reproject(uvq, depth_ray, inv_Kq, ctx["intrinsics"][:, 1, :3, :3], Tq[:, 1])
    tl = t2.long()
    kp = torch.clamp(tl.transpose(1, 2), 0, 255)                          # (B,2,R): x row 0, y row 1
    bidx = torch.arange(B, device=dev)[:, None]
    match_mask = mask2[bidx, kp[:, 1], kp[:, 0]]
    cidx = torch.arange(2, device=dev)[None, :, None]
    src = kp + flow_up[bidx[:, :, None], cidx, kp[:, 1:2], kp[:, 0:1]]
    inb = (0 <= tl) & (tl < 256)
    return {
        "matchability_cycle_mask": match_mask, "T_to_C1_pts": t1, "T_to_C2_pts": t2,
        "mask_c2": inb[..., 0] & inb[..., 1], "C2_pts_to_C1": src.transpose(1, 2), "at_wt_max": at_max,
        "depth_ray": torch.clamp(depth_ray, 0, 10)[..., None],
    }


@torch.no_grad()
def ray_outputs(inp: Dict, flow_prods, at_wt: torch.Tensor, pt: torch.Tensor, rayc: torch.Tensor,
                uv_rows) -> Dict[str, torch.Tensor]:
    """Same outputs as aux_outputs() from one cpn_ray_outputs launch (inference path).  flow_prods: flow_products(...)[0];
    rayc: the (B, RAYC_STRIDE) block of render.build_ray_constants; uv_rows: (float32 uv storage, batch stride) as the
    geometry kernels read it."""
    from ._hip import call
    ctx = inp["context"]
    B, V = ctx["rgb"].shape[:2]
    N, R, S = at_wt.shape
    dev = at_wt.device
    mask2, flow_up = flow_prods
    if mask2.dtype != torch.bool or flow_up.dtype != torch.float32 or tuple(flow_up.shape) != (B, 2, 256, 256):
        raise ValueError("ray_outputs: unexpected flow products")
    f32 = torch.float32
    at_max = torch.empty(N, R, 1, dtype=torch.int64, device=dev)
    depth = torch.empty(B, R, 1, dtype=f32, device=dev)
    t1 = torch.empty(B, R, 2, dtype=f32, device=dev)
    t2 = torch.empty(B, R, 2, dtype=f32, device=dev)
    c21 = torch.empty(B, R, 2, dtype=f32, device=dev)
    mc2 = torch.empty(B, R, dtype=torch.bool, device=dev)
    mm = torch.empty(B, R, dtype=torch.bool, device=dev)
    uvc, uvs = uv_rows
    call("cpn_ray_outputs", at_wt.data_ptr(), pt.data_ptr(), uvc.data_ptr(), uvs, rayc.data_ptr(), mask2.data_ptr(),
         flow_up.data_ptr(), B, V, R, S, at_max.data_ptr(), depth.data_ptr(), t1.data_ptr(), t2.data_ptr(), mc2.data_ptr(),
         mm.data_ptr(), c21.data_ptr(), _stream_handle())
    return {"matchability_cycle_mask": mm, "T_to_C1_pts": t1, "T_to_C2_pts": t2, "mask_c2": mc2, "C2_pts_to_C1": c21,
            "at_wt_max": at_max, "depth_ray": depth}
