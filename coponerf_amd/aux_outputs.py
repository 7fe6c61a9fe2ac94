"""Auxiliary outputs of CoPoNeRF.forward that only feed the cycle/ssim losses and the summaries.

<0.5 % of the reference's time: cycle-consistency masks from the flows, the
attention-weighted expected 3-D point and its reprojections.  They stay stock PyTorch ops on device tensors —
no per-batch Python loops and no host syncs, unlike /root/reference utils_training/utils.py:52-69,260-276.
Cites: models/CoPoNeRF.py:230-236,493-541; utils_training/utils.py:140-170,576-602,642-671;
utils_training/geometry.py:395-406.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def _grid(B: int, H: int, W: int, device) -> torch.Tensor:
    ys, xs = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device), indexing="ij")
    return torch.stack((xs, ys), 0).float()[None].expand(B, -1, -1, -1)


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


_FLOW_CACHE: Dict[str, tuple] = {}


def _flow_products(flow: Sequence[torch.Tensor], width: int):
    """(cycle mask of view 2, flow upsampled to 256x256): functions of the flows alone, so a full-image render that
    calls forward() once per ray chunk with the same `flow` tensors computes them once."""
    key = tuple((f._version, tuple(f.shape)) for f in flow[:2]) + (width,)
    hit = _FLOW_CACHE.get("entry")
    # the entry holds the flow tensors themselves and is matched on IDENTITY (+ version): a new pair's flows that
    # happen to be allocated at a freed pair's addresses can never hit it
    if hit is not None and hit[0] == key and hit[3][0] is flow[0] and hit[3][1] is flow[1]:
        return hit[1], hit[2]
    _, mask2 = cycle_masks(flow, width)
    flow_up = F.interpolate(flow[1], (256, 256), mode="bilinear") * (256 / flow[1].shape[2])
    if not any(f.requires_grad for f in flow[:2]):
        _FLOW_CACHE["entry"] = (key, mask2, flow_up, (flow[0], flow[1]))
    return mask2, flow_up


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
