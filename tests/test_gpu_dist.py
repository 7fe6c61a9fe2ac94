"""2-rank RCCL test of the data-parallel training exchange (skipped on a box with fewer than 2 HIP devices).

One process per GPU over backend "nccl" (= RCCL on PyTorch-ROCm), as bench.py --gpus N launches it: after
broadcast_parameters + one TrainStep on rank-specific batches, every rank holds identical weights, and the
averaged gradient equals the mean of the per-rank gradients computed without the exchange
(/root/reference wrapper.py:21-28,139-151; train.py:58-60).
"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from coponerf_amd import CoPoNeRF, dist as cd, synthetic as syn
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    model = CoPoNeRF.CoPoNeRF(n_view=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes, seed=11 + rank), strict=True)
    model = model.to(dev).train()
    cd.broadcast_parameters(model)
    mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
    inp = mv(syn.make_inputs(1, 256, 256, 128, seed=61 + rank))
    out = model(inp, val=False)
    (out["rgb"] - inp["query"]["rgb"]).abs().mean().backward()
    params = [p for p in model.parameters()]
    local = [None if p.grad is None else p.grad.clone() for p in params]
    assert cd.grads_finite(params)
    ncoll = cd.average_gradients(params)
    ok = ncoll >= 1
    for p, g in zip(params, local):
        if g is None:
            continue
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        ok &= bool(torch.allclose(p.grad, g / world, rtol=1e-5, atol=1e-8))
    w0 = model.query_encode_latent.weight.detach().clone()
    gathered = [torch.zeros_like(w0) for _ in range(world)]
    dist.all_gather(gathered, w0)
    q.put((rank, bool(ok), ncoll, bool(torch.equal(gathered[0], gathered[1]))))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_gradient_exchange_world2():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 HIP devices")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok, ncoll, synced in res:
        assert ok and synced and 1 <= ncoll <= 8
