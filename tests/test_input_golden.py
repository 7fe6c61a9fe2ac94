"""Row f4 pinned: oracle/input_ref.py, coponerf_amd/shards.py (host side) and cpn_prepare_input (GPU) against
tests/golden/input.npz — one sample as the reference's own `RealEstate10k.__getitem__` produced it from the synthetic
scene of coponerf_amd.synthetic.make_scene (tests/golden/make_golden_input.py)."""
import ast
import os

import numpy as np
import pytest
import torch

from coponerf_amd import shards, synthetic as syn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "input.npz")


def _fixture():
    g = dict(np.load(GOLD, allow_pickle=False))
    cfg = ast.literal_eval(str(g.pop("cfg")))
    frames, ts, w2c, intr = syn.make_scene(cfg["n"], cfg["Hs"], cfg["Ws"], cfg["scene_seed"])
    return g, cfg, frames, ts, syn.scene_c2w(w2c), intr


def _check(got_ctx_rgb, got, g):
    assert np.array_equal(got_ctx_rgb[:, ::5, ::5], g["ctx_rgb_1in5"])
    assert float(np.float64(got_ctx_rgb.astype(np.float64).sum())) == float(g["ctx_rgb_sum"])
    for k, v in got.items():
        assert v.shape == g[k].shape, (k, v.shape, g[k].shape)
        assert np.array_equal(v, g[k]), k


def test_input_oracle_reproduces_the_reference_dataset_sample():
    from oracle import input_ref
    g, cfg, frames, ts, c2w, intr = _fixture()
    order = np.argsort(ts, kind="stable")                       # dataio.py:258-262: frames in timestamp order
    want = input_ref.prepare_sample(frames[order], c2w[order], intr[order], [int(i) for i in g["ids"]], g["ray_pix"])
    _check(want["context"]["rgb"], {"ctx_c2w": want["context"]["cam2world"], "ctx_K": want["context"]["intrinsics"],
                                    "qry_rgb": want["query"]["rgb"], "qry_c2w": want["query"]["cam2world"],
                                    "qry_K": want["query"]["intrinsics"], "uv": want["query"]["uv"][0]}, g)
    # the reference's sampler constraints hold for the sample it drew (dataio.py:276-313)
    a, b, q = (int(i) for i in g["ids"])
    assert abs(a - b) > 50 and max(min(a, b) - 32, 0) <= q < min(max(a, b) + 32, cfg["n"] - 1)
    assert len(np.unique(g["ray_pix"])) == cfg["R"]            # a permutation prefix: no pixel twice


def test_shard_host_side_reproduces_the_reference_dataset_sample(tmp_path):
    g, cfg, frames, ts, c2w, intr = _fixture()
    path = str(tmp_path / "scene.cpnshard")
    shards.write_shard(path, frames, ts, c2w, intr)
    sh = shards.Shard(path)
    ids = [int(i) for i in g["ids"]]
    K = np.stack([shards.sample_intrinsics(sh.intrinsics[i], cfg["Hs"], cfg["Ws"]) for i in ids])
    assert np.array_equal(K[:2], g["ctx_K"]) and np.array_equal(K[2:], g["qry_K"])
    assert np.array_equal(np.asarray(sh.c2w)[ids[:2]], g["ctx_c2w"]) and np.array_equal(np.asarray(sh.c2w)[ids[2:]], g["qry_c2w"])
    y0, x0, side = shards.crop_window(cfg["Hs"], cfg["Ws"])
    crop = np.asarray(sh.frames)[ids[0], y0:y0 + side, x0:x0 + side].astype(np.float32) / 127.5 - 1
    assert np.array_equal(crop[::5, ::5], g["ctx_rgb_1in5"][0])


@pytest.mark.gpu
def test_prepare_input_kernel_reproduces_the_reference_dataset_sample(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    g, cfg, frames, ts, c2w, intr = _fixture()
    dev = torch.device("cuda:0")
    path = str(tmp_path / "scene.cpnshard")
    shards.write_shard(path, frames, ts, c2w, intr)
    sh = shards.Shard(path)
    asm = shards.BatchAssembler(1, cfg["Hs"], cfg["Ws"], cfg["R"], dev)
    asm.fill(0, sh, [int(i) for i in g["ids"]], np.random.default_rng(0))
    asm.ray_pix[0] = torch.from_numpy(g["ray_pix"].astype(np.int32))             # the rays the reference drew
    inp, gt = asm.to_model_input()
    torch.cuda.synchronize()
    ctx, qry = inp["context"], inp["query"]
    _check(ctx["rgb"][0].cpu().numpy(), {"ctx_c2w": ctx["cam2world"][0].cpu().numpy(), "ctx_K": ctx["intrinsics"][0].cpu().numpy(),
                                         "qry_rgb": qry["rgb"][0].cpu().numpy(), "qry_c2w": qry["cam2world"][0].cpu().numpy(),
                                         "qry_K": qry["intrinsics"][0].cpu().numpy(), "uv": qry["uv"][0, 0].cpu().numpy()}, g)
    # the staging buffers are reusable only after their upload has run (a second fill() waits for it)
    asm.fill(0, sh, [int(i) for i in g["ids"]], np.random.default_rng(1))
    assert asm._uploaded is None
