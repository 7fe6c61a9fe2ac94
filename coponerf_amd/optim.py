"""torch.optim.Adam for the training step's parameters as ONE launch (csrc/adam.hip).

The reference's loop steps `torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8)` (/root/reference train.py:102-105,
wrapper.py:149-151).  The moments of all tensors live in two flat buffers; the kernel finds parameters and gradients through
a device table whose gradient addresses (new tensors every step) and bias corrections the host rewrites per step: 30 KB
through a pinned buffer, under the GPU's backward pass.
"""
from __future__ import annotations

from typing import Iterable, Optional

import numpy as np
import torch

from . import _hip
from ._hip import call

from ._hip import stream_handle as _stream_handle

_SEG = np.dtype([("p", "<u8"), ("g", "<u8"), ("off", "<i8"), ("n", "<i4"), ("step_size", "<f4"), ("inv_sqrt_bc2", "<f4"),
                 ("pad", "<i4", (3,))])
assert _SEG.itemsize == _hip.ADAM_SEG_BYTES


class OneLaunchAdam(torch.optim.Optimizer):
    """Adam (no weight decay, no amsgrad) over fp32 CUDA parameters.  `step(gscale)`: gscale = optional device scalar
    multiplied into every gradient first (the clip coefficient).  A parameter without a gradient is skipped and keeps its
    update count, like torch.optim.Adam.

    A torch.optim.Optimizer: `param_groups` (lr / betas / eps are read from them at every step, so `ExponentialLR` and the
    reference's MultiLR wrapper drive it - /root/reference train.py:106-108, wrapper.py:134-136), and `state_dict()` /
    `load_state_dict()` in torch.optim.Adam's layout (per parameter `step`, `exp_avg`, `exp_avg_sq`: the reference's
    checkpoints hold `optimizer.state_dict()`, wrapper.py:98).  All groups must share one learning rate (the reference's two
    do): it is a scalar argument of the single launch.

    Two differences from torch.optim.Adam a caller's tooling may see: (1) `optimizer.state` is EMPTY between calls - the
    moments live in two flat buffers (`exp_avg`, `exp_avg_sq`) and the update counts on the device; `state_dict()` builds
    the per-parameter layout from them (and synchronises the device to read the counts), so read state through
    `state_dict()`, not `opt.state[p]`.  (2) every parameter with a gradient gets its version counter bumped at every
    step(), also when the device-side gate skipped the update (non-finite gradients): the version-keyed caches (the
    render engine's weight packs, the trunk / UFC packs, captured get_z graphs) are rebuilt after a skipped step too -
    harmless, one repack."""

    def __init__(self, params: Iterable, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        # the keys of torch.optim.Adam's groups, at the values this kernel implements: a checkpoint written here loads into
        # the library's Adam and back
        ref = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=float(lr), betas=betas, eps=float(eps)).defaults
        defaults = dict(ref, lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=0.0,
                        amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self.params = [p for g in self.param_groups for p in g["params"]]
        assert self.params and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in self.params), \
            "OneLaunchAdam: fp32 contiguous CUDA parameters only"
        dev = self.params[0].device
        n = len(self.params)
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        chunk = _hip.lib().cpn_adam_chunk()
        blocks = np.array([(t, s) for t, p in enumerate(self.params) for s in range(0, p.numel(), chunk)], dtype=np.int32)
        self.nblocks = int(blocks.shape[0])
        self._blocks = torch.from_numpy(blocks).to(dev)
        # update counts per tensor: on the DEVICE (two arrays, read / written in turn) so that a step gated by a device flag
        # needs no host read; `steps` reads them back
        self._counts = torch.zeros(2, n, dtype=torch.int32, device=dev)
        self._count_turn = 0
        self._host = [torch.empty(n * _SEG.itemsize, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._host_np = [h.numpy().view(_SEG) for h in self._host]
        for seg in self._host_np:
            seg["p"] = [p.data_ptr() for p in self.params]
            seg["off"] = offs
            seg["n"] = [p.numel() for p in self.params]
            seg["pad"] = 0
        self._dev = [torch.empty(n * _SEG.itemsize, dtype=torch.uint8, device=dev) for _ in range(2)]
        self._sent = [None, None]
        self._turn = 0

    # ---- the hyper-parameters live in param_groups (schedulers write `lr` there)
    def _hyper(self):
        g0 = self.param_groups[0]
        for g in self.param_groups[1:]:
            if g["lr"] != g0["lr"] or tuple(g["betas"]) != tuple(g0["betas"]) or g["eps"] != g0["eps"]:
                raise NotImplementedError("OneLaunchAdam: every param group must share lr / betas / eps (one launch, scalar arguments)")
        if any(g.get("weight_decay") or g.get("amsgrad") or g.get("maximize") for g in self.param_groups):
            raise NotImplementedError("OneLaunchAdam: plain Adam only (no weight decay, amsgrad or maximize)")
        return float(g0["lr"]), (float(g0["betas"][0]), float(g0["betas"][1])), float(g0["eps"])

    @property
    def lr(self) -> float:
        return self._hyper()[0]

    @lr.setter
    def lr(self, value: float) -> None:
        for g in self.param_groups:
            g["lr"] = float(value)

    @property
    def betas(self):
        return self._hyper()[1]

    @property
    def eps(self) -> float:
        return self._hyper()[2]

    # ---- checkpoints: torch.optim.Adam's layout
    def state_dict(self):
        """{'state': {index: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} as torch.optim.Adam writes it
        (a device read of the update counts: synchronises).  Parameters that never received an update have no entry."""
        steps = self.steps
        self.state.clear()
        for i, p in enumerate(self.params):
            if steps[i] > 0:
                m, v = self.moments(i)
                self.state[p] = {"step": torch.tensor(float(steps[i])), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        try:
            return super().state_dict()
        finally:
            self.state.clear()

    @torch.no_grad()
    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)                   # param_groups + per-parameter state, cast to the parameters' device
        counts = np.zeros(len(self.params), dtype=np.int32)
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, p in enumerate(self.params):
            st = self.state.get(p)
            if not st:
                continue
            m, v = self.moments(i)
            m.copy_(st["exp_avg"])
            v.copy_(st["exp_avg_sq"])
            counts[i] = int(round(float(st["step"])))
        self._counts[self._count_turn & 1].copy_(torch.from_numpy(counts))
        self.state.clear()
        self._prepared = False

    def moments(self, i: int):
        """(exp_avg, exp_avg_sq) of parameter i as views shaped like it."""
        p = self.params[i]
        o = int(self._host_np[0]["off"][i])
        return self.exp_avg[o:o + p.numel()].view_as(p), self.exp_avg_sq[o:o + p.numel()].view_as(p)

    @torch.no_grad()
    def prepare(self, grads=None) -> None:
        """Host half of a step: collect the gradient addresses, write the table, start its upload.  Callable as soon as the
        backward pass is ENQUEUED (addresses exist then) — a caller that reads a device flag before it steps does this first,
        so the ~0.8 ms of Python run under the GPU's backward pass.  `grads`: per parameter the tensor the update will
        read instead of `p.grad` (None: no gradient) — the bucket views of dist.DeviceExchange, known before the exchange ran."""
        i = self._turn & 1
        if self._sent[i] is not None:
            self._sent[i].synchronize()                       # the copy two steps ago read this pinned block
        seg = self._host_np[i]
        n = len(self.params)
        pptr, gptr = np.empty(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        for j, p in enumerate(self.params):
            g = p.grad if grads is None else grads[j]
            pptr[j] = p.data_ptr()
            if g is None:
                continue
            if not (g.is_contiguous() and g.dtype == torch.float32):
                assert grads is None, "OneLaunchAdam.prepare: explicit gradient tensors must be fp32 and contiguous"
                g = p.grad = g.contiguous().float()
            gptr[j] = g.data_ptr()
        seg["p"], seg["g"] = pptr, gptr                       # (step size / bias corrections: formed on the device from its counts)
        self._no_grad_at_prepare = [p for p, g in zip(self.params, gptr) if g == 0]
        self._dev[i].copy_(self._host[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._sent[i] = ev
        self._prepared = True

    @torch.no_grad()
    def step(self, gscale: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None) -> None:
        """gate: optional device scalar (fp32); 0 turns the whole step into a no-op ON THE DEVICE (no moment, count or
        parameter changes) — the finite-gradient guard without the host reading its flag."""
        if not getattr(self, "_prepared", False) or any(p.grad is not None for p in self._no_grad_at_prepare):
            self.prepare()                                    # (a gradient exchange may hand a rank gradients it did not have)
        i = self._turn & 1
        self._turn += 1
        self._prepared = False
        cin, cout = self._counts[self._count_turn & 1], self._counts[(self._count_turn + 1) & 1]
        self._count_turn += 1
        lr, (b1, b2), eps = self._hyper()
        call("cpn_adam_step", self._dev[i].data_ptr(), self._blocks.data_ptr(), self.nblocks, self.exp_avg.data_ptr(),
             self.exp_avg_sq.data_ptr(), 0 if gscale is None else gscale.data_ptr(), 0 if gate is None else gate.data_ptr(),
             cin.data_ptr(), cout.data_ptr(), lr, b1, b2, eps, _stream_handle())
        # the kernel wrote the parameters through raw pointers: tell autograd (every derived-weight cache of the inference
        # path - RenderEngine._weights, the trunk / UFC packs, the captured get_z graphs - is keyed on `_version`, and
        # torch.optim.Adam, which this replaces, bumped it).  Host-only, no launch.
        torch.autograd.graph.increment_version([p for p in self.params if p.grad is not None])

    @property
    def steps(self) -> np.ndarray:
        """Updates every tensor has received (a device read: synchronises)."""
        return self._counts[self._count_turn & 1].cpu().numpy().astype(np.int64)

    def discard(self) -> None:
        """Drop a prepared table (the step is skipped); its slot is written again by the next `prepare`."""
        self._prepared = False

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def add_param_group(self, param_group) -> None:
        if getattr(self, "exp_avg", None) is not None:
            raise NotImplementedError("OneLaunchAdam: parameters are fixed at construction (flat moment buffers)")
        super().add_param_group(param_group)
