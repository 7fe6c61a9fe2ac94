"""Deterministic synthetic stereo-pair inputs, latents and weights.

Nothing here depends on torch's RNG: every value is a pure function of
(seed, flat index) through a splitmix64 counter hash, so the container that
generates the golden fixtures and the GPU box that replays them build
bit-identical inputs.  The camera rig follows SURVEY.md §8(d).

Input dict schema = what the reference's data loaders hand to the model
(/root/reference/data/realestate10k_dataio.py:237-456, consumed at
/root/reference/models/CoPoNeRF.py:213-216).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 counters."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def _bits(n: int, seed: int, stream: int = 0) -> np.ndarray:
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed * 1000003 + stream], dtype=np.uint64))[0]
        return _splitmix64(idx ^ base)


def uniform(shape, seed: int, lo: float = 0.0, hi: float = 1.0, stream: int = 0) -> torch.Tensor:
    """U[lo, hi) float32, 24 random mantissa bits (exact in fp32)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_bits(n, seed, stream) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    out = (lo + (hi - lo) * u).astype(np.float32)
    return torch.from_numpy(out.reshape(shape))


def normal(shape, seed: int, std: float = 1.0, stream: int = 0) -> torch.Tensor:
    """N(0, std^2) float32 via Box-Muller evaluated in float64."""
    n = int(np.prod(shape)) if len(shape) else 1
    m = (n + 1) // 2
    b1 = (_bits(m, seed, stream * 2 + 1) >> np.uint64(11)).astype(np.float64)
    b2 = (_bits(m, seed, stream * 2 + 2) >> np.uint64(11)).astype(np.float64)
    u1 = (b1 + 1.0) * (1.0 / (1 << 53))          # (0, 1]
    u2 = b2 * (1.0 / (1 << 53))
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.concatenate([r * np.cos(2 * math.pi * u2), r * np.sin(2 * math.pi * u2)])[:n]
    return torch.from_numpy((std * z).astype(np.float32).reshape(shape))


def permutation(n: int, seed: int) -> np.ndarray:
    """Deterministic permutation of range(n) (argsort of hashed keys)."""
    return np.argsort(_bits(n, seed, 77), kind="stable")


# --------------------------------------------------------------------------
# cameras
# --------------------------------------------------------------------------
def _rot_y(a: float) -> np.ndarray:
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def _pose(rot: np.ndarray, t) -> np.ndarray:
    m = np.eye(4, dtype=np.float64)
    m[:3, :3] = rot
    m[:3, 3] = t
    return m


RIGS = {
    # name: (yaw of view 1 / view 2, translation of view 1 / view 2)
    "narrow": ((+0.05, -0.05), ((-0.15, 0.0, 0.0), (+0.15, 0.0, 0.0))),   # RealEstate10K-like
    "wide": ((+0.25, -0.25), ((-0.6, 0.0, 0.0), (+0.6, 0.0, 0.3))),       # ACID-like (config 4)
}


def make_inputs(B: int, H: int, W: int, R: int, seed: int = 0, rig: str = "narrow",
                full_image: bool = False, jitter: float = 0.01) -> Dict:
    """Model input dict with B stereo pairs and R query rays per pair.

    full_image=True uses all H*W pixels row-major (x = column, y = row) and
    ignores R; otherwise a seeded permutation prefix of the pixel grid.
    Per-pair pose jitter keeps the B pairs different from one another.
    """
    yaw, trans = RIGS[rig]
    K = np.eye(4, dtype=np.float64)
    K[0, 0] = K[1, 1] = 0.8 * W
    K[0, 2] = W / 2.0
    K[1, 2] = H / 2.0
    jit = uniform((B, 3, 4), seed, -jitter, jitter, stream=5).numpy().astype(np.float64)
    c2w_ctx = np.zeros((B, 2, 4, 4))
    c2w_q = np.zeros((B, 1, 4, 4))
    for b in range(B):
        for v in range(2):
            c2w_ctx[b, v] = _pose(_rot_y(yaw[v] + jit[b, v, 3]),
                                  np.asarray(trans[v]) + jit[b, v, :3])
        c2w_q[b, 0] = _pose(_rot_y(jit[b, 2, 3]), np.array([0.02, 0.0, 0.0]) + jit[b, 2, :3])
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    grid = np.stack([xs.reshape(-1), ys.reshape(-1)], -1).astype(np.float32)  # (H*W, 2) x=col,y=row
    if full_image:
        uv = np.broadcast_to(grid[None, None], (B, 1, H * W, 2)).copy()
    else:
        uv = np.stack([grid[permutation(H * W, seed + 31 * b)[:R]] for b in range(B)])[:, None]
    rgb_ctx = uniform((B, 2, H, W, 3), seed, -1.0, 1.0, stream=1)
    rgb_q = uniform((B, 1, uv.shape[2], 3), seed, -1.0, 1.0, stream=2)
    K32 = torch.from_numpy(K.astype(np.float32))
    return {
        "context": {
            "rgb": rgb_ctx,
            "intrinsics": K32[None, None].repeat(B, 2, 1, 1).contiguous(),
            "cam2world": torch.from_numpy(c2w_ctx.astype(np.float32)),
        },
        "query": {
            "uv": torch.from_numpy(uv.astype(np.float32)),
            "intrinsics": K32[None, None].repeat(B, 1, 1, 1).contiguous(),
            "cam2world": torch.from_numpy(c2w_q.astype(np.float32)),
            "rgb": rgb_q,
        },
    }


def make_latents(B: int, H: int, W: int, seed: int = 1) -> Tuple[List[torch.Tensor], torch.Tensor, Tuple]:
    """Synthetic stand-ins for get_z()'s outputs.

    z: [(2B,256,H/16,W/16),(2B,256,H/8,W/8),(2B,256,H/4,W/4),(2B,64,H,W)]
    rel_pose: (B,4,4) identity rotation, t_x = 0.3
    flow: 4 x (B,2,H/4,W/4): two pixel-unit flows N(0,2^2), two normalised U[-1,1]
    """
    z = [
        normal((2 * B, 256, H // 16, W // 16), seed, stream=10),
        normal((2 * B, 256, H // 8, W // 8), seed, stream=11),
        normal((2 * B, 256, H // 4, W // 4), seed, stream=12),
        normal((2 * B, 64, H, W), seed, stream=13),
    ]
    rel = torch.eye(4).repeat(B, 1, 1)
    rel[:, 0, 3] = 0.3
    rel[:, :3, :3] = torch.from_numpy(_rot_y(-0.1).astype(np.float32))
    hf, wf = H // 4, W // 4
    flow = (
        normal((B, 2, hf, wf), seed, std=2.0, stream=20),
        normal((B, 2, hf, wf), seed, std=2.0, stream=21),
        uniform((B, 2, hf, wf), seed, -1.0, 1.0, stream=22),
        uniform((B, 2, hf, wf), seed, -1.0, 1.0, stream=23),
    )
    return z, rel, flow


# --------------------------------------------------------------------------
# weights of the render path (names/shapes: SURVEY.md Appendix C.1,
# /root/reference/models/CoPoNeRF.py:69-104, /root/reference/models/lightfield.py:64-129)
# --------------------------------------------------------------------------
RENDER_PARAM_SHAPES = {
    "query_encode_latent.weight": (832, 835, 1, 1), "query_encode_latent.bias": (832,),
    "query_encode_latent_2.weight": (416, 832, 1, 1), "query_encode_latent_2.bias": (416,),
    "latent_value.weight": (416, 832, 1, 1), "latent_value.bias": (416,),
    "key_map.weight": (128, 832, 1, 1), "key_map.bias": (128,),
    "key_map_2.weight": (128, 128, 1, 1), "key_map_2.bias": (128,),
    "query_embed.weight": (128, 16, 1, 1), "query_embed.bias": (128,),
    "query_embed_2.weight": (128, 128, 1, 1), "query_embed_2.bias": (128,),
    "query_repeat_embed.weight": (128, 144, 1, 1), "query_repeat_embed.bias": (128,),
    "query_repeat_embed_2.weight": (128, 128, 1, 1), "query_repeat_embed_2.bias": (128,),
    "encode_latent.weight": (128, 416, 1), "encode_latent.bias": (128,),
    "phi.lin_in.weight": (128, 18), "phi.lin_in.bias": (128,),
    "phi.lin_out.weight": (3, 128), "phi.lin_out.bias": (3,),
}
for _k in range(3):
    RENDER_PARAM_SHAPES[f"phi.lin_z.{_k}.weight"] = (128, 832)
    RENDER_PARAM_SHAPES[f"phi.lin_z.{_k}.bias"] = (128,)
    for _f in ("fc_0", "fc_1"):
        RENDER_PARAM_SHAPES[f"phi.blocks.{_k}.{_f}.weight"] = (128, 128)
        RENDER_PARAM_SHAPES[f"phi.blocks.{_k}.{_f}.bias"] = (128,)


def make_render_weights(seed: int = 7, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic weights at default-init scale: U(-1/sqrt(fan_in), +1/sqrt(fan_in)).

    Unlike the module's default init (phi.*.fc_1.weight = 0, biases of phi = 0,
    lightfield.py:35-38,88-93) every tensor is non-zero so parity tests
    exercise every term.
    """
    out = {}
    for i, (name, shape) in enumerate(sorted(RENDER_PARAM_SHAPES.items())):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else None
        if fan_in is None:  # bias: fan-in of the matching weight
            wshape = RENDER_PARAM_SHAPES[name.replace(".bias", ".weight")]
            fan_in = int(np.prod(wshape[1:]))
        bound = gain / math.sqrt(fan_in)
        out[name] = uniform(shape, seed, -bound, bound, stream=100 + i)
    return out


# ---- a render case with PEAKED attention at latent statistics like get_z's --------------------------------
# Default-init weights give logits <key, coords_embed> / 11.31 of ~1e-2: the joint softmax over the 2 x S samples of a ray is
# flat (max weight ~ 1 / (2 S)) and rounding errors of single samples average out.  A trained model attends: scaling the two
# factors of each logit by PEAK_GAIN makes the logits 4096 x larger (median of the largest weight of a ray 0.8, > 0.5 on most
# rays).  The synthetic latents (unit variance, zero mean) are moved to the per-level statistics get_z produces on the
# make_full_weights state (std 1.14 / 1.67 / 1.87 / 0.75, mean 0.24 / 0.22 / 0.28 / 0.03; tests/golden/make_golden.py).
PEAK_GAIN = 64.0
GETZ_LEVEL_STD = (1.144, 1.667, 1.871, 0.752)
GETZ_LEVEL_MEAN = (0.241, 0.216, 0.282, 0.025)


def peaked_weights(weights: Dict[str, torch.Tensor], gain: float = PEAK_GAIN) -> Dict[str, torch.Tensor]:
    out = dict(weights)
    for k in ("key_map_2.weight", "query_embed_2.weight", "query_repeat_embed_2.weight"):
        out[k] = weights[k] * gain
    return out


def latents_at_getz_statistics(z):
    return [t * s + m for t, s, m in zip(z, GETZ_LEVEL_STD, GETZ_LEVEL_MEAN)]


def make_full_weights(shapes: Dict[str, Tuple], seed: int = 11) -> Dict[str, torch.Tensor]:
    """Deterministic values for EVERY state_dict entry (name -> shape), at scales that keep the 256x256 get_z stack
    numerically tame: matrices/kernels U(+-1/sqrt(fan_in)), norm scales 1 +- 0.1, biases +-0.05, BatchNorm running
    statistics (mean +-0.05, var in [0.8, 1.2])."""
    out = {}
    for i, name in enumerate(sorted(shapes)):
        shape = tuple(shapes[name])
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.zeros(shape, dtype=torch.int64)
        elif leaf == "running_var":
            out[name] = uniform(shape, seed, 0.8, 1.2, stream=1000 + i)
        elif leaf == "running_mean":
            out[name] = uniform(shape, seed, -0.05, 0.05, stream=1000 + i)
        elif leaf == "pos_embed":
            out[name] = uniform(shape, seed, -0.04, 0.04, stream=1000 + i)
        elif len(shape) >= 2:
            bound = 1.0 / math.sqrt(int(np.prod(shape[1:])))
            out[name] = uniform(shape, seed, -bound, bound, stream=1000 + i)
        elif leaf == "weight":                                   # 1-D weight = normalisation scale
            out[name] = uniform(shape, seed, 0.9, 1.1, stream=1000 + i)
        else:
            out[name] = uniform(shape, seed, -0.05, 0.05, stream=1000 + i)
    return out


# --------------------------------------------------------------------------
# a RealEstate10K-shaped scene for the input pipeline (coponerf_amd/shards.py): uint8 frames, timestamps, camera rows
# --------------------------------------------------------------------------
def make_scene(n: int = 140, Hs: int = 256, Ws: int = 455, seed: int = 5):
    """(frames (n,Hs,Ws,3) uint8, timestamps (n) int64 out of order, w2c (n,3,4) float64, intrinsics (n,4) float64
    normalised fx fy cx cy) — the contents of one scene of the reference's dataset (an .npz of frames named
    '<timestamp>.png' + the pose rows 'timestamp fx fy cx cy k1 k2 w2c[12]', data/realestate10k_dataio.py:37-80)."""
    frames = (_bits(n * Hs * Ws * 3, seed, 1) >> np.uint64(56)).astype(np.uint8).reshape(n, Hs, Ws, 3)
    ts = permutation(n, seed + 1).astype(np.int64) * 33366 + 1000
    ang = uniform((n,), seed, -0.3, 0.3, stream=2).numpy().astype(np.float64)
    t = uniform((n, 3), seed, -1.0, 1.0, stream=3).numpy().astype(np.float64)
    w2c = np.zeros((n, 3, 4))
    for i in range(n):
        w2c[i, :, :3] = _rot_y(float(ang[i]))
        w2c[i, :, 3] = t[i]
    intr = np.array([0.5, 0.89, 0.5, 0.5]) + uniform((n, 4), seed, -0.01, 0.01, stream=4).numpy().astype(np.float64)
    return frames, ts, w2c, intr


def scene_c2w(w2c: np.ndarray) -> np.ndarray:
    """cam2world (n,4,4) float32 from the dataset's w2c rows, as the reference's Camera class forms it (float64 inverse)."""
    m = np.tile(np.eye(4), (w2c.shape[0], 1, 1))
    m[:, :3, :] = w2c
    return np.linalg.inv(m).astype(np.float32)
