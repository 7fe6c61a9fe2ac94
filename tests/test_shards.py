"""Input pipeline: shard round trip, the frame sampler's constraints, and the GPU preparation kernel
against the numpy restatement of the reference's per-sample processing (oracle/input_ref.py)."""
import numpy as np
import pytest
import torch

from coponerf_amd import shards


def _scene(n=140, Hs=256, Ws=455, seed=5):
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, size=(n, Hs, Ws, 3), dtype=np.uint8)
    ts = rng.permutation(n).astype(np.int64) * 33366 + 1000                      # stored out of order on purpose
    c2w = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    c2w[:, :3, 3] = rng.normal(size=(n, 3)).astype(np.float32)
    intr = np.tile(np.array([0.5, 0.89, 0.5, 0.5], dtype=np.float32), (n, 1)) + rng.normal(size=(n, 4)).astype(np.float32) * 0.01
    return frames, ts, c2w, intr


def test_shard_round_trip_and_sampler(tmp_path):
    frames, ts, c2w, intr = _scene(n=140, Hs=32, Ws=56)
    path = str(tmp_path / "scene.cpnshard")
    shards.write_shard(path, frames, ts, c2w, intr)
    sh = shards.Shard(path)
    order = np.argsort(ts, kind="stable")
    assert len(sh) == 140 and (np.diff(sh.timestamps) > 0).all()
    assert np.array_equal(sh.frames, frames[order]) and np.array_equal(sh.c2w, c2w[order]) and np.array_equal(sh.intrinsics, intr[order])
    for name in ("frames", "timestamps", "c2w", "intrinsics"):
        assert getattr(sh, name).offset % 64 == 0
    rng = np.random.default_rng(0)
    for _ in range(200):
        a, b, q = shards.sample_pair(len(sh), rng)
        assert abs(a - b) > 50 and 0 <= a < 139 and 0 <= b < 139                  # dataio.py:276-296
        assert max(min(a, b) - 32, 0) <= q < min(max(a, b) + 32, 139)             # dataio.py:299-309
    assert shards.sample_pair(30, rng) is None                                    # scene too short for a 50-frame gap
    assert shards.crop_window(256, 455) == (0, 99, 256)


@pytest.mark.gpu
def test_prepare_input_matches_reference_processing(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from oracle import input_ref
    dev = torch.device("cuda:0")
    frames, ts, c2w, intr = _scene(n=140)
    path = str(tmp_path / "scene.cpnshard")
    shards.write_shard(path, frames, ts, c2w, intr)
    sh = shards.Shard(path)
    B, R = 3, 4096
    asm = shards.BatchAssembler(B, 256, 455, R, dev)
    rng = np.random.default_rng(1)
    picks = []
    for b in range(B):
        ids = shards.sample_pair(len(sh), rng)
        asm.fill(b, sh, ids, rng)
        picks.append(ids)
    inp, gt = asm.to_model_input()
    torch.cuda.synchronize()
    assert inp["context"]["rgb"].shape == (B, 2, 256, 256, 3) and inp["query"]["uv"].shape == (B, 1, R, 2)
    for b in range(B):
        want = input_ref.prepare_sample(np.asarray(sh.frames), np.asarray(sh.c2w), np.asarray(sh.intrinsics), picks[b],
                                        asm.ray_pix[b].numpy().astype(np.int64))
        for grp in ("context", "query"):
            for k, v in want[grp].items():
                got = inp[grp][k][b].cpu().numpy()
                assert got.shape == v.shape, (grp, k, got.shape, v.shape)
                assert np.array_equal(got, v), (grp, k)                           # bit-exact: same IEEE operations
    assert gt["rgb"] is inp["query"]["rgb"]
