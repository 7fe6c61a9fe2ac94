"""Image pipeline over a sequence of stereo pairs: get_z of pair i+1 runs under the render pass of pair i.

The reference's evaluation loop (/root/reference test.py:164-212, wrapper.py:176-211) is strictly serial: `get_z`, then
the chunked `forward(val=True)` calls, pair after pair.  The two halves are complementary on an MI355X — `get_z` is
~550 small kernels (launch / latency bound, 8 ms alone), the render pass is a handful of HBM-bound kernels that fill
the chip (25.5 ms) — and stereo pairs are independent, so the next pair's features can be produced while the current
image renders.  Same kernels, same inputs as the serial order; only the schedule changes, and since get_z's GroupNorm
statistics are reduced in a fixed order the results are bit-identical to the serial ones.

Two ordinary streams do NOT overlap the halves (measured: 35.3-38.9 ms per image against 35.3-38.0 serial): the persistent grids
of the render pass hold every CU, so each small kernel of get_z waits for a chip-filling launch to drain.  `cu_split`
partitions the chip instead (coponerf_amd/streams.py): the render pass keeps `cu_split[0]` CUs, get_z the other
`cu_split[1]`, an equal share of every XCD each.
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, Optional, Tuple

import torch


def _record(obj, stream) -> None:
    """Tensors produced on the side stream are consumed (and later freed) on the main one."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record(o, stream)
    elif isinstance(obj, dict):
        for o in obj.values():
            _record(o, stream)


def _collate(items):
    """Batch a list of model_input dicts (each B = 1, or any B) along dim 0, leaf by leaf."""
    first = items[0]
    if torch.is_tensor(first):
        return torch.cat(items, dim=0)
    if isinstance(first, dict):
        return {k: _collate([it[k] for it in items]) for k in first}
    if isinstance(first, (list, tuple)):
        return type(first)(_collate([it[i] for it in items]) for i in range(len(first)))
    return first


def _pair_slice(obj, lo: int, hi: int, per: int):
    """Rows [lo*per, hi*per) of every tensor in a (nested) get_z result: per = 2 for the (2B, ...) feature maps."""
    if torch.is_tensor(obj):
        return obj[lo * per:hi * per]
    if isinstance(obj, (list, tuple)):
        return type(obj)(_pair_slice(o, lo, hi, per) for o in obj)
    return obj


def render_images(model, inputs: Iterable[Dict], getz_batch: int = 1, graph: bool = False,
                  cu_split: Optional[Tuple[int, int]] = None, nchunks: Optional[int] = None,
                  overlap=False) -> Iterator[Tuple[Dict, Dict]]:
    """For every model_input dict (on the device) yield (model_input, forward(model_input, z, rel_pose, val=True, flow)),
    with `get_z` of the next inputs overlapped with the render of the current ones.  Call under torch.no_grad().

    cu_split=(render_cus, getz_cus) runs the two halves on CU-masked streams over disjoint shares of the chip
    (multiples of 32, sum <= the device's CU count; e.g. (192, 64)); the yielded outputs are ordered after the
    caller's current stream as usual.  None: ONE stream — the GPU runs render(i), get_z(i+1), render(i+1), ... in that
    order while the host issues get_z(i+1) under render(i) (the host half of get_z, ~6 ms of Python and launches, is what
    this order hides); `overlap=True` puts get_z(i+1) on a second, high-priority stream, which measured SLOWER
    than the serial GPU order ever since the render kernels became persistent (34.9 against 31.6 ms per image in round 4:
    each small kernel of get_z waits for a chip-filling launch to drain and the render kernels lose their cache state).

    nchunks renders every input the way the reference's callers do (test.py:176-212: that many forward() calls on
    torch.chunk(uv, nchunks), joined key by key — coponerf_amd/evalloop.render_in_chunks) instead of in one call.

    getz_batch > 1 runs `get_z` ONCE for that many consecutive inputs (batched along dim 0: its kernels are launch /
    latency bound, 8 ms for one pair, ~5.6 ms per pair at four) and renders them one after the other from slices of
    the batched features; per-pair results are those of the serial order up to the rounding differences of batched
    library GEMMs / convolutions (tests/test_gpu_getz.py).

    graph=True replays `get_z` as a captured HIP graph (coponerf_amd/graphs.py: one hipGraphLaunch instead of ~550
    eager launches; same kernels, same results)."""
    it = iter(inputs)
    getz = model.get_z
    if graph:
        from .graphs import GraphedGetZ
        getz = getattr(model, "_graphed_getz", None)
        if getz is None:
            getz = model._graphed_getz = GraphedGetZ(model)

    def take():
        grp = []
        for x in it:
            grp.append(x)
            if len(grp) == max(1, getz_batch):
                break
        return grp

    def features(grp):
        feats = getz(grp[0] if len(grp) == 1 else _collate(grp))
        return feats, getattr(model._engine, "_l3_hint", None), [int(g["context"]["rgb"].shape[0]) for g in grp]

    cur = take()
    if not cur:
        return
    outer = torch.cuda.current_stream()
    engine = model._engine
    if cu_split is not None:
        from .streams import CUPartition
        # ONE partition at a time: every CU-masked stream is a hardware queue of its own, and more of them than the
        # device has queues are time-sliced (measured: a second cached partition turned 34 ms per image into 48)
        key = (int(cu_split[0]), int(cu_split[1]), outer.device.index)
        part = engine.__dict__.get("_cu_partition")
        if part is None or part.key != key:
            if part is not None:
                part.close()
            part = engine._cu_partition = CUPartition(key[0], key[1], outer.device)
            part.key = key
        main, side = part.render, part.getz
        main.wait_stream(outer)                 # the inputs were produced on the caller's stream
        # consecutive calls alternate over lanes INSIDE the render share, not over the engine's own unmasked streams
        engine.set_call_streams(part.render_lanes(engine.call_lanes) if engine.call_lanes > 1 else None)
    else:
        main = outer
        # overlap: high priority, so that the small kernels of get_z are dispatched ahead of the render's queued workgroups
        # instead of waiting behind each of the render's chip-filling launches
        side = torch.cuda.Stream(device=main.device, priority=-1) if overlap else main
    try:
        with torch.cuda.stream(main):
            state = features(cur)                                  # first group: nothing to hide it under
        while cur:
            nxt = take()
            (z, rel_pose, flow), hint, sizes = state
            H, W = model.H, model.W
            if nxt:
                side.wait_stream(main)      # BEFORE the renders are enqueued: the side stream then only waits for what
                                            # is already on the main stream (the previous group), not for these renders
            nstate = None
            lo = 0
            # the pairs of a batched group as (z, rel_pose, flow, full-resolution NHWC copy) slices, made ONCE: the engine
            # matches what prepare_next() built to the later forward() by tensor identity
            parts = []
            if len(cur) > 1:
                for n in sizes:
                    l3 = hint[2][2 * lo:2 * (lo + n)] if hint is not None and hint[0] is z[3] else None
                    parts.append((_pair_slice(z, lo, lo + n, 2), rel_pose[lo:lo + n], _pair_slice(flow, lo, lo + n, 1), l3))
                    lo += n
            lo = 0
            for i, inp in enumerate(cur):
                hi = lo + sizes[i]
                with torch.cuda.stream(main):                      # closed again before the yield below
                    if len(cur) == 1:
                        zi, ri, fi = z, rel_pose, flow
                    else:
                        if i + 1 < len(cur):
                            # the next pair of the group is known already: its per-pair preparation runs on the engine's
                            # own stream under this pair's kernels (CoPoNeRF.prepare_next)
                            zn, rn, fn, l3n = parts[i + 1]
                            if l3n is not None:
                                engine.adopt_level3(zn[3], l3n)
                            model.prepare_next(cur[i + 1], zn, rn, fn)
                        zi, ri, fi, l3 = parts[i]
                        if l3 is not None:
                            engine.adopt_level3(zi[3], l3)
                    model.H, model.W = H, W
                    # 1. the render of this pair: a few dozen launches, asynchronous
                    if nchunks:
                        from .evalloop import join_chunks, render_in_chunks
                        chunks = render_in_chunks(model, inp, nchunks, latents=(zi, ri, fi), join=False)
                    else:
                        out = model(inp, z=zi, rel_pose=ri, val=True, flow=fi)
                    if i == 0 and nxt:
                        # 2. the ~550 launches of the next group's get_z on the side stream: the host issues them while
                        #    the GPU renders, the small kernels run beside the HBM-bound render kernels
                        with torch.cuda.stream(side):
                            nstate = features(nxt)
                    if i == len(cur) - 1 and nxt:
                        main.wait_stream(side)  # the NEXT renders (and whatever the caller enqueues) follow get_z(next)
                        _record(nstate[0], main)
                        if nstate[1] is not None:
                            _record(nstate[1][2], main)
                    if nchunks:
                        # the callers' per-key concatenation waits for the host copies of `pixel_val`, i.e. for this
                        # image's render: only now, with get_z of the next inputs already issued
                        out = join_chunks(chunks)
                if main is not outer:
                    outer.wait_stream(main)
                    _record(out, outer)
                lo = hi
                yield inp, out
            cur, state = nxt, nstate
    finally:
        if main is not outer:
            outer.wait_stream(main)             # the lanes were joined into `main` by every call's completion event
            engine.set_call_streams(None)
