"""CU-partitioned HIP streams (csrc/streams.cpp): the render pass and get_z side by side on disjoint CUs.

The reference's callers run `get_z` and the chunked `forward(val=True)` calls strictly one after the other
(/root/reference test.py:164-212, wrapper.py:176-211).  On an MI355X the two halves want different things — the render
pass is a few HBM-bound launches whose persistent grids take every CU, `get_z` is ~550 small launches that rarely fill a
quarter of the chip — but on ordinary streams they cannot overlap: every small kernel waits for a chip-filling one to
drain.  A `CUPartition` gives each its own CUs: `render_cus` of the 256 for the render stream, `getz_cus` for the other,
both an equal share of each shader engine of each of the 8 XCDs, i.e. multiples of 32 (so the XCD-aware tile orders of
the kernels keep their meaning and a one-workgroup-per-CU grid still lands one workgroup on every CU).  The
persistent launchers of the library size their grids by the stream they are launched on (`cpn_stream_cu_count`).

What is and is not confined: eager launches on the two streams are, and so is a captured `get_z` graph
(coponerf_amd/graphs.py captures ON the masked stream when one is current and keys its graphs on the stream's share).
The intra-call chunk lanes of `RenderEngine(lanes > 1)` are ordinary unmasked streams of their own: do not combine
`lanes > 1` with a partition (`render_images` uses call lanes inside the render share instead, `CUPartition.render_lanes`).

Both streams are ordinary HIP streams to PyTorch (`torch.cuda.ExternalStream`): MIOpen / hipBLASLt work launched under
`torch.cuda.stream(part.getz)` is confined to that share as well.
"""
from __future__ import annotations

import ctypes
from typing import Tuple

import torch

from . import _hip


def device_cus() -> int:
    return int(_hip.lib().cpn_device_cu_count())


def stream_cus(stream: torch.cuda.Stream) -> int:
    """CUs a persistent launch on `stream` spreads over (the whole device unless the stream came from CUPartition)."""
    return int(_hip.lib().cpn_stream_cu_count(ctypes.c_void_p(stream.cuda_stream)))


class CUPartition:
    """Two streams over disjoint CU ranges: `render` on the first `render_cus`, `getz` on the last `getz_cus`."""

    def __init__(self, render_cus: int, getz_cus: int, device=None):
        total = device_cus()
        if render_cus <= 0 or getz_cus <= 0 or render_cus % 32 or getz_cus % 32 or render_cus + getz_cus > total:
            raise ValueError(f"CUPartition({render_cus}, {getz_cus}): both shares must be positive multiples of 32 "
                             f"(8 XCDs x 4 shader engines) and fit the device's {total} CUs")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.split: Tuple[int, int] = (render_cus, getz_cus)
        self._handles = []
        self._lanes = []
        with torch.cuda.device(self.device):
            self.render = self._make(0, render_cus)
            self.getz = self._make(total - getz_cus, getz_cus)

    def render_lanes(self, n: int):
        """n more streams over the render share (RenderEngine.set_call_streams: consecutive calls alternate over them)."""
        while len(self._lanes) < n:
            with torch.cuda.device(self.device):
                self._lanes.append(self._make(0, self.split[0]))
        return self._lanes[:n]

    def _make(self, first: int, n: int) -> torch.cuda.Stream:
        out = ctypes.c_void_p()
        _hip.call("cpn_stream_create_cu_range", first, n, ctypes.byref(out))
        self._handles.append(out.value)
        return torch.cuda.ExternalStream(out.value, device=self.device)

    def close(self) -> None:
        """Wait for both streams and destroy them (the torch wrappers do not own the HIP streams)."""
        handles, self._handles = self._handles, []
        if handles:
            for st in [self.render, self.getz] + self._lanes:
                st.synchronize()
        self._lanes = []
        for h in handles:
            _hip.call("cpn_stream_destroy", ctypes.c_void_p(h))

    def __del__(self):                                   # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass
