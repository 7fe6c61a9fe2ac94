"""Input pipeline next to the hot path: pre-decoded uint8 scene shards + GPU-side preparation.

The reference's dataset (/root/reference data/realestate10k_dataio.py:237-456) `np.load`s a whole compressed per-scene
`data.npz` for EVERY sample, crops / converts on the host and sends ~2.4 MB of float32 per sample through DataLoader
workers — at BASELINE config 3 (4 pairs per step per GPU, 8 GPUs, ~0.2 s steps) that starves the GPUs.  Here a scene
is one memory-mappable shard of raw uint8 frames at the working resolution (256 x 455, what the reference resizes to,
:340) plus its poses; a sample is three row-slices of the mmap (no decode, no float conversion on the host), batches
are assembled in pinned memory and `cpn_prepare_input` (csrc/input.hip) crops, normalises and gathers on the GPU.
The model-facing result is the reference's input dict: same keys, shapes, dtypes and values.

Shard file = 64-byte magic/header length + JSON header + arrays at 64-byte aligned offsets:
    frames (N, Hs, Ws, 3) uint8 | timestamps (N) int64 | c2w (N, 4, 4) float32 | intrinsics (N, 4) float64 (fx fy cx cy,
    normalised by image size as in the RealEstate10K pose files)
"""
from __future__ import annotations

import json
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from ._hip import stream_handle as _stream_handle

MAGIC = b"CPNSHARD1\n"


def write_shard(path: str, frames_u8: np.ndarray, timestamps: np.ndarray, c2w: np.ndarray, intrinsics_norm: np.ndarray) -> None:
    frames_u8 = np.ascontiguousarray(frames_u8, dtype=np.uint8)
    n = frames_u8.shape[0]
    assert frames_u8.ndim == 4 and frames_u8.shape[3] == 3
    arrays = {"frames": frames_u8, "timestamps": np.ascontiguousarray(timestamps, dtype=np.int64).reshape(n),
              "c2w": np.ascontiguousarray(c2w, dtype=np.float32).reshape(n, 4, 4),
              # float64 like the pose rows the reference parses (dataio.py:37-55): it un-normalises in float64 and
              # rounds to float32 once, at the end — a float32 copy here moves fx by an ulp (tests/golden/input.npz)
              "intrinsics": np.ascontiguousarray(intrinsics_norm, dtype=np.float64).reshape(n, 4)}
    order = np.argsort(arrays["timestamps"], kind="stable")          # frames sorted by time (dataio.py:264-268)
    arrays = {k: v[order] for k, v in arrays.items()}
    meta, off = {}, 0
    for k, a in arrays.items():
        off = (off + 63) // 64 * 64
        meta[k] = {"dtype": str(a.dtype), "shape": list(a.shape), "offset": off}
        off += a.nbytes
    header = json.dumps({"arrays": meta, "nbytes": off}).encode()
    data0 = (len(MAGIC) + 8 + len(header) + 63) // 64 * 64
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(len(header).to_bytes(8, "little"))
        f.write(header)
        f.write(b"\0" * (data0 - f.tell()))
        for k, a in arrays.items():
            f.seek(data0 + meta[k]["offset"])
            f.write(a.tobytes())


class Shard:
    """Memory-mapped view of one scene shard."""

    def __init__(self, path: str):
        with open(path, "rb") as f:
            if f.read(len(MAGIC)) != MAGIC:
                raise ValueError(f"{path}: not a CoPoNeRF scene shard")
            hlen = int.from_bytes(f.read(8), "little")
            header = json.loads(f.read(hlen))
        data0 = (len(MAGIC) + 8 + hlen + 63) // 64 * 64
        self.path = path
        for k, m in header["arrays"].items():
            setattr(self, k, np.memmap(path, mode="r", dtype=np.dtype(m["dtype"]), shape=tuple(m["shape"]),
                                       offset=data0 + m["offset"]))

    def __len__(self) -> int:
        return int(self.frames.shape[0])


def sample_pair(num_frames: int, rng: np.random.Generator, min_gap: int = 50, query_margin: int = 32) -> Optional[Tuple[int, int, int]]:
    """Frame ids (context 0, context 1, query) with the reference's constraints for two context views
    (data/realestate10k_dataio.py:276-313): contexts drawn from [0, N-1) more than `min_gap` frames apart, the query
    uniformly from [min(ctx) - margin, max(ctx) + margin) clipped to the scene.  None if the scene is too short."""
    cand = np.arange(0, num_frames - 1)
    ids = []
    for _ in range(2):
        if len(cand) == 0:
            return None
        c = int(rng.choice(cand))
        cand = cand[(cand < c - min_gap) | (cand > c + min_gap)]
        ids.append(c)
    low, high = max(min(ids) - query_margin, 0), min(max(ids) + query_margin, num_frames - 1)
    if high <= low:
        return None
    return ids[0], ids[1], int(rng.integers(low, high))


def sample_intrinsics(intr_norm: np.ndarray, Hs: int, Ws: int) -> np.ndarray:
    """(fx fy cx cy) normalised -> the 4x4 pixel-unit matrix the reference feeds the model after its centre square crop
    (dataio.py:37-55 Camera / unnormalize_intrinsics, :343-348: cx and cy are DIVIDED by W/min(H,W), H/min(H,W))."""
    fx, fy, cx, cy = (float(v) for v in intr_norm)
    K = np.array([[fx, 0, cx, 0], [0, fy, cy, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    K[0] *= Ws
    K[1] *= Hs
    m = min(Hs, Ws)
    K[0, 2] = K[0, 2] / (Ws / m)
    K[1, 2] = K[1, 2] / (Hs / m)
    return K.astype(np.float32)


def crop_window(Hs: int, Ws: int) -> Tuple[int, int, int]:
    """Centre square crop of utils_training/data_util.py:116-121 -> (y0, x0, side)."""
    m = min(Hs, Ws)
    return Hs // 2 - m // 2, Ws // 2 - m // 2, (m // 2) * 2


class BatchAssembler:
    """Pinned uint8 staging for `batch` samples + the device tensors of the model's input dict."""

    def __init__(self, batch: int, Hs: int, Ws: int, rays: int, device: torch.device):
        self.B, self.Hs, self.Ws, self.R, self.dev = batch, Hs, Ws, rays, device
        self.y0, self.x0, self.side = crop_window(Hs, Ws)
        pin = device.type == "cuda"
        self.frames = torch.empty(batch, 3, Hs, Ws, 3, dtype=torch.uint8, pin_memory=pin)
        self.ray_pix = torch.empty(batch, rays, dtype=torch.int32, pin_memory=pin)
        self.small = torch.empty(batch, 3, 32, dtype=torch.float32, pin_memory=pin)       # c2w (16) + K (16) per frame
        # the H2D copies of to_model_input() read the pinned staging asynchronously: the next fill() must not overwrite
        # it before they have run (the GPU queue is a whole training step deep)
        self._uploaded: Optional[torch.cuda.Event] = None

    def _wait_uploaded(self) -> None:
        if self._uploaded is not None:
            self._uploaded.synchronize()
            self._uploaded = None

    def fill(self, b: int, shard: Shard, ids: Sequence[int], rng: np.random.Generator) -> None:
        """Host side of one sample: three row-slices of the mmap, the ray selection, 3 x 32 floats."""
        S = self.side
        self._wait_uploaded()
        for j, fid in enumerate(ids):
            self.frames[b, j].numpy()[...] = shard.frames[fid]                  # mmap page cache -> pinned staging, one copy
            self.small[b, j, :16] = torch.from_numpy(np.asarray(shard.c2w[fid]).reshape(16).copy())
            self.small[b, j, 16:] = torch.from_numpy(sample_intrinsics(shard.intrinsics[fid], self.Hs, self.Ws).reshape(16))
        self.ray_pix[b] = torch.from_numpy(rng.permutation(S * S)[:self.R].astype(np.int32))          # dataio.py:385-390

    def to_model_input(self) -> Tuple[Dict, Dict]:
        """Asynchronous H2D of the staged bytes + cpn_prepare_input -> (model_input, gt) like the reference's loader."""
        from ._hip import call
        S, B, R = self.side, self.B, self.R
        frames = self.frames.to(self.dev, non_blocking=True)
        pix = self.ray_pix.to(self.dev, non_blocking=True)
        small = self.small.to(self.dev, non_blocking=True)
        if self.dev.type == "cuda":
            self._uploaded = torch.cuda.Event()
            self._uploaded.record()
        ctx = torch.empty(B, 2, S, S, 3, dtype=torch.float32, device=self.dev)
        qrgb = torch.empty(B, 1, R, 3, dtype=torch.float32, device=self.dev)
        call("cpn_prepare_input", frames.data_ptr(), B, self.Hs, self.Ws, self.y0, self.x0, S, S, R, pix.data_ptr(),
             ctx.data_ptr(), qrgb.data_ptr(), _stream_handle())
        uv = torch.stack((pix % S, pix // S), dim=-1).float().view(B, 1, R, 2)               # (x = column, y = row)
        mat = lambda j, k: small[:, j, 16 * k:16 * k + 16].reshape(B, 1, 4, 4)
        query = {"rgb": qrgb, "cam2world": mat(2, 0), "intrinsics": mat(2, 1), "uv": uv}
        context = {"rgb": ctx, "cam2world": torch.cat((mat(0, 0), mat(1, 0)), 1), "intrinsics": torch.cat((mat(0, 1), mat(1, 1)), 1)}
        return {"query": query, "context": context}, query
