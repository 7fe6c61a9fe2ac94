"""One data-parallel training step of the drop-in model, as the reference's loop performs it.

Mirrors /root/reference wrapper.py:104-151 for the default loss configuration (image loss only,
models/loss_function.py:65-71, 105-108): forward with `val=False` (get_z inside), `|gt - rgb|.mean()`, backward,
invalid-gradient guard, `clip_grad_norm_(max_norm=1)` BEFORE the exchange (wrapper.py:142-148), gradient averaging,
Adam step.  The exchange and the guard are the RCCL-friendly forms of coponerf_amd/dist.py: one MIN-all-reduced finite
flag (every rank takes the same branch: no deadlock) and a few flat buckets instead of <= 636 blocking per-parameter
all-reduces.  One process per GPU; ranks draw independent batches (train.py:84-97 uses no DistributedSampler).
"""
from __future__ import annotations

import collections
import os
from typing import Dict, Optional

import torch

from . import dist as cdist


class _LazyFlag:
    """`stepped` of a step whose guard ran on the device: truth value on demand (waits for a 4-byte copy that was started
    when the step was issued)."""

    def __init__(self, host: torch.Tensor, event):
        self._host, self._event, self._value = host, event, None

    def __bool__(self) -> bool:
        # read ONCE: the pinned word is one of a few that TrainStep rotates through (it resolves every flag itself before the
        # word comes round again), so a caller that keeps the flag of an old step must get that step's answer
        if self._value is None:
            self._event.synchronize()
            self._value = bool(self._host.item() > 0)
            self._host = self._event = None
        return self._value

    def __repr__(self) -> str:
        return f"_LazyFlag({bool(self)})"


class TrainStep:
    def __init__(self, model: torch.nn.Module, lr: float = 5e-5 * 4, clip_grad: float = 1.0,
                 bucket_bytes: int = 64 << 20, group=None, force_collectives: bool = False, exchange: bool = True):
        """exchange=False: never exchange gradients, even inside an initialised process group (every rank for itself: the
        single-GPU step measured beside the N-rank one, bench.py `ms_per_step_without_exchange`)."""
        self.model = model
        self.exchange_enabled = bool(exchange)
        self.params = [p for p in model.parameters()]
        # train.py:102-105 (both groups share lr); where the parameters live on the GPU the update of all 636 tensors is ONE
        # launch (optim.OneLaunchAdam, csrc/adam.hip: 1.7 ms for the library's fused multi-tensor form -> 0.3; same rule)
        on_gpu = len(self.params) > 0 and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in self.params)
        if on_gpu:
            from .optim import OneLaunchAdam
            self.opt = OneLaunchAdam(self.params, lr=lr)
        else:
            self.opt = torch.optim.Adam(self.params, lr=lr)
        self._one_launch = on_gpu
        self.clip_grad = clip_grad
        self.bucket_bytes = bucket_bytes
        self.group = group
        # run the flag exchange and the gradient buckets even on a one-rank group (dist._exchanging): exercises RCCL on a
        # single device; the update is unchanged (the mean over one rank is the rank's own gradient)
        self.force_collectives = bool(force_collectives)
        self.timing: Optional[Dict[str, list]] = None                    # set to {} to collect HIP-event timings
        # reorder every step's query rays by image tile (_rays_by_tile); COPONERF_SORT_RAYS=0: as given
        self.sort_rays = os.environ.get("COPONERF_SORT_RAYS", "1") != "0"
        # fp16 activation gradients carry a static per-pass scale (train_fns.GradScale).  If they overflow the guard
        # skips the step and the next pass would pick the same scale: back the target off (x 1/4 per skipped step, down
        # to 1) and restore it (x 2 every `growth_interval` good steps) like an AMP GradScaler does.  Where the guard runs on
        # the device the outcome of a step reaches this adaptation one or two steps LATE (its 4-byte copy is not waited for):
        # after an overflow the next one or two steps reuse the scale and are skipped as well, then the back-off applies.
        self.skipped_in_a_row = 0
        self.skipped_total = 0
        self.growth_interval = 200
        self._good = 0
        eng = getattr(model, "_engine", None)
        self._scale_ceiling = float(eng.grad_scale_target) if eng is not None else 0.0
        # dist.DeviceExchange is built HERE, not at the first step: its gloo side group is a collective over the whole default
        # world (dist.new_group), so every rank must reach it at the same point - constructing TrainStep is that point
        # (ADVICE r5).  Every rank of the job constructs its TrainStep, with the same `group`.
        self._exchange = None
        if self._exchanging():
            self._exchange = cdist.DeviceExchange(self.params, self.bucket_bytes, self.group, self.force_collectives)
        self._ex_reads_seen = 0
        self._pending = collections.deque()                              # `stepped` flags of steps guarded on the device
        self._flag_host = [torch.zeros(1).pin_memory() for _ in range(4)] if on_gpu else None
        self._flag_turn = 0

    def _exchanging(self) -> bool:
        return self.exchange_enabled and cdist._exchanging(self.group, self.force_collectives)

    def _ev(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    @staticmethod
    def _rays_by_tile(model_input: Dict, gt_rgb: torch.Tensor):
        """The step's query rays of every pair reordered by 8 x 8-pixel tile of the query image (the dataset draws them at
        random, dataio.py:385-390): rays that are neighbours in the image project to neighbouring epipolar lines, so the
        table taps of consecutive rays share cache lines and the backward's scatter tiles fill in bursts - 1.0-1.5 ms of the
        step.  Every ray is rendered exactly as before (rays do not interact) and the loss is a mean over them; the per-ray
        outputs come back in the new order, `order` (B, R) says which input ray each one is."""
        q = model_input["query"]
        uv = q["uv"]                                                   # (B, 1, R, 2) pixel coordinates
        R = uv.shape[2]
        x, y = uv[..., 0].floor().long(), uv[..., 1].floor().long()
        key = ((y >> 3) << 20) + ((x >> 3) << 6) + ((y & 7) << 3) + (x & 7)
        order = key.argsort(dim=-1)                                    # (B, 1, R)
        take = lambda t: torch.gather(t, 2, order[..., None].expand(-1, -1, -1, t.shape[-1]))
        q2 = {k: (take(v) if torch.is_tensor(v) and v.dim() == 4 and v.shape[1] == 1 and v.shape[2] == R else v)
              for k, v in q.items()}
        out = dict(model_input)
        out["query"] = q2
        gt = take(gt_rgb) if gt_rgb.dim() == 4 and gt_rgb.shape[2] == R else gt_rgb
        return out, gt, order[:, 0]

    def __call__(self, model_input: Dict, gt_rgb: torch.Tensor) -> Dict[str, object]:
        """model_input: the reference's input dict on the device; gt_rgb (B,1,R,3).  Returns loss / bookkeeping."""
        timed = self.timing is not None and torch.cuda.is_available()
        ray_order = None
        if self.sort_rays and gt_rgb.is_cuda:
            model_input, gt_rgb, ray_order = self._rays_by_tile(model_input, gt_rgb)
        # steps whose guard ran on the device: their outcome reaches the scale adaptation when its 4-byte copy has landed —
        # normally one step late, never by waiting (unless three are outstanding)
        if self._exchanging():
            # N > 1: `query()` depends on each rank's host timing, and the ranks must change their gradient scale at the SAME
            # step to stay reproducible step for step (the flag itself is MIN-reduced, so its value is the same everywhere):
            # always consume the flag of step k - 2, whose copy landed long ago
            while len(self._pending) > 2:
                self._adapt_grad_scale(bool(self._pending.popleft()))
        else:
            while self._pending and (self._pending[0]._value is not None or self._pending[0]._event.query()
                                     or len(self._pending) > 2):
                self._adapt_grad_scale(bool(self._pending.popleft()))
        e0 = self._ev() if timed else None
        out = self.model(model_input, val=False)
        zero = lambda t: torch.where(torch.isnan(t), torch.zeros_like(t), t)      # loss_function.py:66-69
        loss = (zero(gt_rgb) - zero(out["rgb"])).abs().mean()
        e1 = self._ev() if timed else None
        loss.backward()
        e2 = self._ev() if timed else None
        # guard + clip in one pass over the gradients; the flag is the same on every rank.  No exchange behind the clip (it
        # may hand a rank gradients it did not have): flag and coefficient STAY on the device — the update kernel is gated by
        # the flag and multiplies the coefficient in — so the host never waits for the backward pass (behind a `.item()` it
        # is no longer ahead of the GPU, and everything it does until the next step's first launch is GPU idle time: 0.4-0.9 ms
        # per step depending on the host)
        exchanging = self._exchanging()
        on_device = self._one_launch
        ncoll, nbytes = 0, 0
        if exchanging and self._exchange is None:
            self._exchange = cdist.DeviceExchange(self.params, self.bucket_bytes, self.group, self.force_collectives)
        ex = self._exchange if exchanging else None
        if on_device:
            # N > 1 is as host-free as N = 1 (round 5): the union of the ranks' gradient masks is agreed host to host (a CPU
            # vote over gloo), the finite flag is MIN-reduced ON THE DEVICE and gates the update, the buckets are persistent
            # and the optimizer reads the averaged gradients in place (dist.DeviceExchange)
            if ex is not None:
                ex.agree()
                self.opt.prepare(grads=ex.grad_views())
            else:
                self.opt.prepare()
            ok, _, coef = cdist.guard_on_device(self.params, float(self.clip_grad or 0.0))
            e3 = self._ev() if timed else None
            if ex is not None:
                ok, inv_world, ncoll = ex.exchange(ok, coef)
                nbytes, coef = ex.nbytes, inv_world
            e4 = self._ev() if timed else None
            self.opt.step(gscale=coef, gate=ok)
            host = self._flag_host[self._flag_turn & 3]
            self._flag_turn += 1
            host.copy_(ok.reshape(1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            stepped = _LazyFlag(host, ev)
            self._pending.append(stepped)
        else:
            # parameters on the CPU (the gloo tests) / a stock optimizer without a gate: the same exchange, then the flag is
            # read where reading it costs nothing
            ok, _, coef = cdist.guard_on_device(self.params, float(self.clip_grad or 0.0))
            e3 = self._ev() if timed else None
            grads = [p.grad for p in self.params if p.grad is not None]
            if ex is not None:
                ex.agree()
                ok, inv_world, ncoll = ex.exchange(ok, coef)
                nbytes, coef = ex.nbytes, inv_world
                grads = [p.grad for p in self.params if p.grad is not None]
            e4 = self._ev() if timed else None
            stepped = bool(ok.item() > 0)
            if stepped:
                if coef is not None and grads:
                    torch._foreach_mul_(grads, coef)
                self.opt.step()
            self._adapt_grad_scale(stepped)
        self.opt.zero_grad(set_to_none=True)
        reads_before, self._ex_reads_seen = self._ex_reads_seen, (0 if ex is None else ex.host_reads)
        e5 = self._ev() if timed else None
        if timed:
            self.timing.setdefault("events", []).append((e0, e1, e2, e3, e4, e5))
        return {"loss": loss.detach(), "stepped": stepped, "collectives": ncoll, "allreduce_bytes": nbytes,
                "host_reads": (0 if on_device else 1) + (0 if ex is None else ex.host_reads - reads_before),
                "mask_exchanges": 0 if ex is None else ex.mask_exchanges,
                "at_wt": out["at_wt"].detach(), "ray_order": ray_order, "skipped_in_a_row": self.skipped_in_a_row}

    def _adapt_grad_scale(self, stepped: bool) -> None:
        eng = getattr(self.model, "_engine", None)
        from . import getz as _getz
        _getz.trunk_bwd_target_backoff(stepped)
        if stepped:
            self.skipped_in_a_row = 0
            self._good += 1
            if eng is not None and self._good >= self.growth_interval and eng.grad_scale_target < self._scale_ceiling:
                eng.grad_scale_target = min(self._scale_ceiling, eng.grad_scale_target * 2.0)
                self._good = 0
            return
        self.skipped_in_a_row += 1
        self.skipped_total += 1
        self._good = 0
        if eng is not None:
            eng.grad_scale_target = max(1.0, eng.grad_scale_target / 4.0)
        if self.skipped_in_a_row in (3, 10, 100):
            import warnings
            scale = f" (fp16 gradient scale target now {eng.grad_scale_target:g})" if eng is not None else ""
            warnings.warn(f"coponerf_amd.TrainStep: {self.skipped_in_a_row} consecutive steps skipped by the "
                          f"finite-gradient guard{scale}")

    def timing_summary(self) -> Dict[str, float]:
        """Mean milliseconds per phase over the recorded steps (call after torch.cuda.synchronize())."""
        ev = (self.timing or {}).get("events", [])
        if not ev:
            return {}
        names = ("forward_ms", "backward_ms", "guard_clip_ms", "allreduce_ms", "optimizer_ms")
        acc = [0.0] * 5
        for e in ev:
            for i in range(5):
                acc[i] += e[i].elapsed_time(e[i + 1])
        return {n: a / len(ev) for n, a in zip(names, acc)}
