"""The full-image evaluation loop the reference's callers wrap around `CoPoNeRF.forward`.

/root/reference test.py:164-212 (18 ray chunks per image, batch 2) and wrapper.py:176-211 (`nrays // 512 + 1` chunks):
`get_z` once per batch of pairs, then one `forward(val=True)` per chunk of query rays with the SAME camera tensors,
latents and flows; three keys are dropped from each chunk's dict, `pixel_val` goes through `.cpu()`, and the chunks are
joined key by key — along the ray axis, which is dim -3 for `pixel_val`, -1 for the two per-ray boolean masks and -2 for
everything else.  This module restates that loop so that it can be benchmarked (`bench.py --ref-loop`) and tested
against a single full-image call (tests/test_gpu_refloop.py); it contains no compute of its own.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

DROPPED = ("z", "coords", "at_wts")                                   # test.py:190-192
NOT_JOINED = ("rel_pose", "gt_rel_pose", "flow", "cyclic_consistency_error")     # test.py:203-204
LAST_CHUNK = ("rel_pose", "gt_rel_pose", "flow")                       # test.py:216-218


def ray_axis(key: str) -> int:
    """Axis the caller concatenates `key` along (test.py:206-211)."""
    if key == "pixel_val":
        return -3
    if key in ("mask_c2", "matchability_cycle_mask"):
        return -1
    return -2


def join_chunks(chunks) -> Dict[str, object]:
    full = {}
    for k in chunks[0].keys():
        if k in NOT_JOINED:
            continue
        full[k] = torch.cat([c[k] for c in chunks], dim=ray_axis(k))
    for k in LAST_CHUNK:
        full[k] = chunks[-1][k]
    return full


@torch.no_grad()
def render_in_chunks(model, model_input: Dict, nchunks: int = 18, latents: Optional[tuple] = None,
                     join: bool = True):
    """`model_input['query']['rgb' / 'uv']` hold the full image's rays (B,1,R,·).  Returns the joined output dict
    (or the list of per-chunk dicts with join=False).  `latents` = (z, rel_pose, flow) skips the get_z call."""
    qry = model_input["query"]
    rgb_full, uv_full = qry["rgb"], qry["uv"]
    z, rel_pose, flow = latents if latents is not None else model.get_z(model_input)
    chunks = []
    try:
        for rgb_c, uv_c in zip(torch.chunk(rgb_full, nchunks, dim=2), torch.chunk(uv_full, nchunks, dim=2)):
            qry["rgb"], qry["uv"] = rgb_c, uv_c
            out = model(model_input, z=z, rel_pose=rel_pose, val=True, flow=flow)
            for k in DROPPED:
                del out[k]
            out["pixel_val"] = out["pixel_val"].cpu()
            chunks.append(out)
    finally:
        qry["rgb"], qry["uv"] = rgb_full, uv_full
    return join_chunks(chunks) if join else chunks
