"""Data-parallel gradient exchange for the training caller, RCCL-over-xGMI friendly.

The reference averages gradients with one blocking `dist.all_reduce` per parameter (<= 636 collectives per step,
4 B ... 275 MB each) after a per-rank NaN guard that can deadlock: a rank that sees a NaN skips the collectives the
others are blocked in (/root/reference wrapper.py:21-28, 44-58, 139-151; train.py:58-60).

Here the same semantics (SUM then divide by world size, parameters without a gradient skipped) run as a few large
flat buckets — xGMI is point-to-point (7 links x ~153 GB/s per GPU), so per-collective latency, not bandwidth,
is what 636 small all-reduces pay — and the finite check is one MIN-all-reduced flag, so every rank takes the same
branch.  Backend "nccl" on PyTorch-ROCm is RCCL; the CPU tests use gloo.
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


def _buckets(items: List, bucket_bytes: int, key=lambda t: t) -> List[List]:
    """Greedy split of `items` (tensors, or records whose tensor is key(item)) into same-dtype buckets."""
    out, cur, size = [], [], 0
    for it in items:
        t = key(it)
        nb = t.numel() * t.element_size()
        if cur and (size + nb > bucket_bytes or t.dtype != key(cur[0]).dtype):
            out.append(cur)
            cur, size = [], 0
        cur.append(it)
        size += nb
    if cur:
        out.append(cur)
    return out


def grads_finite(params: Iterable[torch.nn.Parameter], group=None) -> bool:
    """True iff every gradient on EVERY rank is finite (one fused check + one MIN all-reduce)."""
    grads = [p.grad for p in params if p.grad is not None]
    ok = torch.ones((), dtype=torch.float32, device=grads[0].device if grads else "cpu")
    if grads:
        # isfinite per tensor (the reference tests isnan/isinf, wrapper.py:47-57): a sum of squares would overflow
        # to inf on large-but-finite gradients and skip the step on every rank
        ok = torch.stack([torch.isfinite(g.detach()).all() for g in grads]).all().float()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    return bool(ok.item() > 0)


def _exchanging(group, force: bool) -> bool:
    """Is there an initialised process group whose collectives should run?  `force`: also on a one-rank group — a
    one-rank RCCL communicator still initialises, launches and completes every collective, which is how the exchange
    path is exercised on a single MI355X (tests/test_gpu_dist.py, bench.py `exchange_probe`)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return force or dist.get_world_size(group) > 1


def guard_and_clip(params: Iterable[torch.nn.Parameter], max_norm: float = 0.0, group=None, force: bool = False):
    """The reference's invalid-gradient guard (wrapper.py:44-58) and `clip_grad_norm_` (wrapper.py:142-146) from ONE
    pass over the gradients: per-tensor 2-norms (`torch._foreach_norm`, a handful of launches for the 570 tensors),
    combined in float64.  NaN / Inf entries make their tensor's norm non-finite, so `finite` means "no gradient holds
    a NaN or an Inf and no TENSOR's fp32 sum of squares overflows" (a tensor whose 2-norm exceeds ~1.8e19 also skips
    the step; the reference would clip it to norm 1 instead — such a step is lost either way).  The flag is
    MIN-all-reduced, by every rank whether or not it has gradients: every rank takes the same branch.  Clipping (per rank, BEFORE the
    exchange, like the reference) scales the gradients in place by min(1, max_norm / (norm + 1e-6)).
    Returns (finite: bool, total_norm: float64 tensor)."""
    finite, total, grads, coef = _guard(params, max_norm, group, force)
    if coef is not None:
        torch._foreach_mul_(grads, coef)
    return finite, total


def guard_and_clip_coefficient(params: Iterable[torch.nn.Parameter], max_norm: float = 0.0, group=None, force: bool = False):
    """`guard_and_clip` that leaves the gradients alone and hands the clip coefficient back as a device scalar (None: nothing
    to clip) for a caller that applies it inside its update kernel — only where no exchange follows the clip.
    Returns (finite, total_norm, coef)."""
    finite, total, _, coef = _guard(params, max_norm, group, force)
    return finite, total, coef


def guard_on_device(params: Iterable[torch.nn.Parameter], max_norm: float = 0.0):
    """The same guard and clip coefficient WITHOUT a host read, for a single rank whose update kernel takes them as device
    scalars (optim.OneLaunchAdam.step(gscale, gate)): returns (ok: fp32 device scalar 1 / 0, total_norm: float64 device
    scalar, coef: fp32 device scalar or None).  The caller's host never waits for the backward pass here."""
    plist = list(params)
    grads = [p.grad for p in plist if p.grad is not None]
    if not grads:                                             # nothing to guard or clip: an update would be a no-op anyway
        dev = plist[0].device if plist else "cpu"
        return torch.ones((), dtype=torch.float32, device=dev), torch.zeros((), dtype=torch.float64, device=dev), None
    norms = torch.stack(torch._foreach_norm(grads)).double()
    total = norms.square().sum().sqrt()
    ok = torch.isfinite(total).float()
    coef = None
    if max_norm and max_norm > 0:
        # a non-finite norm would make the coefficient NaN: the gate already skips that step, keep the scalar finite
        coef = torch.nan_to_num(torch.clamp(max_norm / (total + 1e-6), max=1.0), nan=0.0).float()
    return ok, total, coef


def _guard(params, max_norm, group, force):
    plist = list(params)
    grads = [p.grad for p in plist if p.grad is not None]
    if grads:
        norms = torch.stack(torch._foreach_norm(grads)).double()
        total = norms.square().sum().sqrt()
        ok = torch.isfinite(total).float()
    else:
        # a rank without gradients still takes part in the flag exchange: the others are blocked in it
        dev = plist[0].device if plist else "cpu"
        total = torch.zeros((), dtype=torch.float64, device=dev)
        ok = torch.ones((), dtype=torch.float32, device=dev)
    if _exchanging(group, force):
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if not grads:
        return bool(ok.item() > 0), total, grads, None
    finite = bool(ok.item() > 0)
    coef = None
    if finite and max_norm and max_norm > 0:
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0).float()
    return finite, total, grads, coef


def average_gradients(params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, group=None,
                      force: bool = False) -> int:
    """In-place gradient averaging across ranks; returns the number of data collectives issued.

    Parameters without a gradient are skipped like wrapper.py:26 does.  The flat buckets need the SAME set of
    gradients on every rank (the reference's per-parameter loop hangs just the same when one rank lacks a gradient
    another has): one MAX all-reduce of the has-gradient mask establishes the union, and a rank missing one of those
    gradients contributes zeros for it and RECEIVES the average as its `.grad`, so the optimizers of all ranks update
    the same set of parameters and the replicas stay identical."""
    if not _exchanging(group, force):
        return 0
    world = dist.get_world_size(group)
    plist = [p for p in params]
    if not plist:
        return 0
    dev = next((p.grad.device for p in plist if p.grad is not None), plist[0].device)
    mask = torch.tensor([p.grad is not None for p in plist], dtype=torch.int32, device=dev)
    dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=group)
    union = mask.bool().tolist()
    grads, owned = [], []
    for p, u in zip(plist, union):
        if not u:
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p.data)
        grads.append(p.grad.data)
        owned.append(True)
    handles = []
    for bucket in _buckets(list(zip(grads, owned)), bucket_bytes, key=lambda t: t[0]):
        flat = torch.cat([g.reshape(-1) for g, _ in bucket])
        handles.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, bucket))
    for work, flat, bucket in handles:
        work.wait()
        flat.div_(world)
        off = 0
        for g, mine in bucket:
            if mine:
                g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    return len(handles)


class DeviceExchange:
    """The gradient exchange of a training step WITHOUT a device -> host read (round 5; `average_gradients` + `guard_and_clip`
    above read the MIN-reduced flag with `.item()` and the has-gradient mask with `.tolist()` every step, i.e. the host waits
    for the end of the backward pass twice and the GPU idles while it catches up):

      * which parameters take part - the union over the ranks of "has a gradient" (wrapper.py:26 skips None) - is agreed
        HOST to host over a gloo side group (a 636-entry CPU all-reduce; with the main group on gloo, that group itself),
        and only when some rank's own mask differs from the one it had at the last agreement: one 1-entry CPU vote per step
        otherwise.  The mask is a property of the loss configuration: in practice it is exchanged once;
      * the finite flag is MIN-all-reduced as a device scalar and GATES the update kernel (optim.OneLaunchAdam.step(gate));
      * the gradients are packed (x the rank's own clip coefficient: the reference clips BEFORE it averages, wrapper.py:142-148)
        into persistent flat buckets - one `_foreach_copy_` + one multiply per bucket, no `torch.cat` - summed by one
        asynchronous all-reduce per bucket, and handed to the optimizer AS views of those buckets with 1 / world as its
        gradient scale: no copy back.
    A rank that lacks a gradient of the union contributes zeros and receives the average, so all replicas update the same
    parameters.  `host_reads` counts device -> host reads made here (none in steady state)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, group=None, force: bool = False):
        self.params = list(params)
        self.bucket_bytes, self.group, self.force = int(bucket_bytes), group, bool(force)
        self.active = _exchanging(group, force)
        self.host_reads = 0
        self.mask_exchanges = 0
        self._local = None                         # this rank's mask at the last agreement
        self._union: List[bool] = []
        self._flats = []                           # [(flat buffer, [(param index, offset, numel)])]
        self._views: List = [None] * len(self.params)
        self._side = None
        self._inv_world = None
        self.timing = None                         # set to [] to collect (event before, [event after each collective]) per exchange
        if self.active:
            if dist.get_backend(group) == "gloo":
                self._side = group
            else:
                ranks = dist.get_process_group_ranks(group) if group is not None else None
                try:
                    self._side = dist.new_group(ranks=ranks, backend="gloo")   # every rank of the group constructs one
                except Exception as e:                                          # e.g. no usable network interface for gloo
                    import warnings
                    warnings.warn(f"coponerf_amd.dist.DeviceExchange: no gloo side group ({e}); the gradient-mask agreement "
                                  "falls back to a device all-reduce with a host read per step")
                    self._side = None
            import logging
            logging.getLogger("coponerf_amd.dist").info(
                "DeviceExchange: %d rank(s) over %s, gradient-mask agreement %s", self.world, dist.get_backend(group),
                "host to host (gloo vote)" if self._side is not None else "on the device (one host read per step)")

    @property
    def mask_path(self) -> str:
        """How the ranks agree on the set of exchanged gradients: "host-vote" (gloo, no device read) or "device-fallback"."""
        return "host-vote" if (self._side is not None or not self.active) else "device-fallback"

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if self.active else 1

    def agree(self) -> List[bool]:
        """Host half, callable as soon as the backward pass is ENQUEUED (whether a parameter has a `.grad` is known then)."""
        local = [p.grad is not None for p in self.params]
        if self._side is None and dist.get_backend(self.group) != "gloo":
            # fallback without a host-side channel: the round-4 exchange of the mask on the device, one read per step
            dev = next((p.device for p in self.params), "cpu")
            mask = torch.tensor(local, dtype=torch.int32, device=dev)
            dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
            union = mask.bool().tolist()
            self.host_reads += 1
            if union != self._union or not self._flats:
                self._local, self._union = local, union
                self.mask_exchanges += 1
                self._build()
            return self._union
        vote = torch.tensor([0 if local == self._local else 1], dtype=torch.int32)
        dist.all_reduce(vote, op=dist.ReduceOp.MAX, group=self._side)
        if int(vote) != 0:
            mask = torch.tensor(local, dtype=torch.int32)
            dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self._side)
            self._local, self._union = local, mask.bool().tolist()
            self.mask_exchanges += 1
            self._build()
        return self._union

    def _build(self) -> None:
        members = [(i, self.params[i]) for i, u in enumerate(self._union) if u]
        self._flats, self._views = [], [None] * len(self.params)
        for bucket in _buckets(members, self.bucket_bytes, key=lambda t: t[1]):
            n = sum(p.numel() for _, p in bucket)
            flat = torch.zeros(n, dtype=bucket[0][1].dtype, device=bucket[0][1].device)
            entries, off = [], 0
            for i, p in bucket:
                entries.append((i, off, p.numel()))
                self._views[i] = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            self._flats.append((flat, entries))
        dev = members[0][1].device if members else "cpu"
        self._inv_world = torch.full((), 1.0 / self.world, dtype=torch.float32, device=dev)

    def timing_summary(self):
        """Mean milliseconds from the start of the exchange to the completion of [flag, bucket 0, bucket 1, ...] (cumulative:
        the collectives run in order on the communicator's stream) and the bytes of each bucket.  After a synchronize."""
        if not self.timing:
            return {}
        n = len(self.timing)
        k = len(self.timing[0][1])
        ms = [sum(t[0].elapsed_time(t[1][j]) for t in self.timing) / n for j in range(k)]
        return {"done_ms_cumulative": ms, "bucket_bytes": [f.numel() * f.element_size() for f, _ in self._flats]}

    def grad_views(self) -> List:
        """Per parameter: where its averaged gradient will be (a view of a persistent bucket; None outside the union)."""
        return self._views

    @property
    def nbytes(self) -> int:
        return sum(f.numel() * f.element_size() for f, _ in self._flats)

    def exchange(self, ok: torch.Tensor, coef=None):
        """ok: this rank's finite flag (fp32 scalar on the gradients' device), coef: its clip coefficient (device scalar) or
        None.  Returns (ok MIN-reduced over the ranks, 1 / world as a device scalar, collectives issued); afterwards every
        parameter of the union has `.grad` = its view of the SUM over the ranks of coef_r * grad_r.  Nothing here waits on
        the host for the device (on RCCL `wait()` orders the streams).
        After a step whose flag came back 0 the buckets (= the parameters' `.grad`) hold whatever the ranks summed - NaN where
        one of them had a non-finite gradient (inf x 0): gradients are UNDEFINED after a skipped step, the gated update does
        not read them and the next step overwrites them."""
        rec = None
        if self.timing is not None and torch.cuda.is_available():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            rec = (e, [])
        works = [dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group, async_op=True)]
        for flat, entries in self._flats:
            src, dst = [], []
            for i, _, _ in entries:
                g = self.params[i].grad
                if g is None:
                    self._views[i].zero_()                     # this rank lacks a gradient another rank has
                elif g.data_ptr() != self._views[i].data_ptr():
                    src.append(g.detach())
                    dst.append(self._views[i])
            if src:
                torch._foreach_copy_(dst, src)
            if coef is not None:
                flat.mul_(coef.to(flat.dtype))
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
            if rec is not None:                      # completion of this collective as the current stream sees it
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                rec[1].append(e)
        if rec is not None:
            self.timing.append(rec)
        for i, v in enumerate(self._views):
            if v is not None:
                self.params[i].grad = v
        return ok, self._inv_world, len(works)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, bucket_bytes: int = 64 << 20, group=None,
                         force: bool = False) -> int:
    """Initial weight sync (train.py:58-60), parameters AND floating-point buffers, in flat buckets."""
    if not _exchanging(group, force):
        return 0
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers() if b.is_floating_point()]
    n = 0
    for bucket in _buckets(tensors, bucket_bytes):
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.broadcast(flat, src, group=group)
        off = 0
        for t in bucket:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        n += 1
    engine = getattr(module, "_engine", None)
    if engine is not None:          # `.data` writes do not bump the version counters the weight cache is keyed on
        engine.invalidate()
    return n


def shard_pairs(num_pairs: int, rank: int, world: int) -> range:
    """Inference partitioning: stereo pairs round-robin over ranks, no collective."""
    return range(rank, num_pairs, world)
