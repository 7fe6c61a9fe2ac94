"""get_z as a HIP graph (inference).

`get_z` of one 256x256 stereo pair is ~550 kernel launches of 5-80 us: issued eagerly the host needs about as long to
launch them (~12 us each through PyTorch + ctypes) as the GPU needs to run them, so neither the launch gaps nor a second
stream can help (tools/getz_graph.py).  Captured once per input signature and replayed, the host cost is one
hipGraphLaunch, and the source / target attention passes of every UFC layer — forked onto a second stream inside the
capture (coponerf_amd/getz.py) — really run side by side.  The reference has no counterpart: its evaluation loop calls
`model.get_z` eagerly per pair (/root/reference test.py:173, wrapper.py:178).

Contract: inference only (`torch.no_grad()`, `model.eval()`).  The graph reads the parameters in place, so in-place
parameter updates through torch ops bump the parameters' version counters and re-capture; `load_state_dict` /
`RenderEngine.invalidate()` (writes that replace or bypass the tensors) drop the captured graphs through
`CoPoNeRF._param_epoch`.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch


def _map(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(v, fn) for v in obj)
    return obj


def _copy_into(dst, src) -> None:
    if torch.is_tensor(dst):
        dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_into(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for a, b in zip(dst, src):
            _copy_into(a, b)


def _signature(obj) -> Tuple:
    if torch.is_tensor(obj):
        return (tuple(obj.shape), obj.dtype, obj.device)
    if isinstance(obj, dict):
        return tuple((k, _signature(v)) for k, v in sorted(obj.items()))
    if isinstance(obj, (list, tuple)):
        return tuple(_signature(v) for v in obj)
    return (type(obj).__name__,)


class GraphedGetZ:
    """callable(model_input) -> (z, rel_pose, flow), as `model.get_z(model_input)` returns them (fresh tensors)."""

    def __init__(self, model):
        self.model = model
        self._graphs: Dict[Tuple, tuple] = {}

    @staticmethod
    def _stream_cus(stream) -> int:
        from .streams import stream_cus
        return stream_cus(stream)

    @staticmethod
    def _device_cus() -> int:
        from .streams import device_cus
        return device_cus()

    def _capture(self, inp):
        model = self.model
        static_in = _map(inp["context"], lambda t: t.clone())
        static = {"context": static_in}
        cur = torch.cuda.current_stream()
        warm = torch.cuda.Stream(device=cur.device)
        warm.wait_stream(cur)
        with torch.cuda.stream(warm):                       # allocator / lazy-init warm-up outside the capture
            for _ in range(2):
                model.get_z(static)
        cur.wait_stream(warm)
        graph = torch.cuda.CUDAGraph()
        # Under a CU partition (coponerf_amd/streams.py) the caller's stream is CU-masked, and the persistent kernels of
        # get_z size their grids by the stream they are captured on (cpn_stream_cus): capture on THAT stream, not on
        # torch's own unmasked side stream, or the graph bakes in whole-device grids.  The replay runs on the
        # caller's stream again, i.e. on the masked hardware queue.
        masked = self._stream_cus(cur) < self._device_cus()
        with torch.cuda.graph(graph, stream=cur if masked else None):
            out = model.get_z(static)
            hint = getattr(model._engine, "_l3_hint", None)
        nhwc = hint[2] if hint is not None and hint[0] is out[0][3] else None
        return graph, static_in, out, nhwc, (model.H, model.W)

    def __call__(self, inp):
        model = self.model
        if torch.is_grad_enabled() or model.training:
            raise RuntimeError("GraphedGetZ is the inference path: call it under torch.no_grad() with model.eval()")
        # parameter epoch (load_state_dict / invalidate) + the sum of the parameters' version counters: an optimizer step or
        # any other in-place update re-captures, because derived tensors the capture baked in (e.g. the concatenated q/k
        # projection weights of UFCLayer._qk_weights) are rebuilt as NEW tensors when their sources change
        if self.__dict__.get("_params") is None:
            self._params = list(model.parameters())
        # ... and the CU share of the stream the call runs on: a graph captured for one share has that share's grids baked in
        key = (_signature(inp["context"]), getattr(model, "_param_epoch", 0), sum(p._version for p in self._params),
               self._stream_cus(torch.cuda.current_stream()))
        rec = self._graphs.get(key)
        if rec is None:
            self._graphs = {k: v for k, v in self._graphs.items() if k[1:3] == key[1:3]}    # stale parameter states go
            rec = self._graphs[key] = self._capture(inp)
        graph, static_in, out, nhwc, hw = rec
        _copy_into(static_in, inp["context"])
        graph.replay()
        z, rel_pose, flow = _map(out, lambda t: t.clone())     # the static outputs are overwritten by the next replay
        model.H, model.W = hw
        if nhwc is not None:
            model._engine.adopt_level3(z[3], nhwc.clone())
        return z, rel_pose, flow
