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
    # two whole TrainSteps through dist.DeviceExchange (device-side flag, persistent buckets, gradients read in place):
    # the replicas hold identical weights afterwards, no device -> host read was made
    from coponerf_amd.train_step import TrainStep
    model.zero_grad(set_to_none=True)
    step = TrainStep(model)
    r1 = step(inp, inp["query"]["rgb"])
    r2 = step(inp, inp["query"]["rgb"])
    ok &= bool(r1["stepped"]) and bool(r2["stepped"]) and r2["host_reads"] == 0 and r2["mask_exchanges"] == 1
    w0 = torch.cat([p.detach().reshape(-1) for p in (model.query_encode_latent.weight, model.phi.lin_out.weight,
                                                      model.conv_map.weight)])
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


def _worker_world1(port, q):
    """ONE rank over backend nccl: RCCL initialises, and with `force` every collective of the step (parameter broadcast,
    MIN-reduced guard flag, MAX-reduced gradient mask, the flat gradient buckets) is issued to it and completes on the
    MI355X — the data path of tests on a box with a single device."""
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from coponerf_amd import CoPoNeRF, dist as cd, synthetic as syn
    from coponerf_amd.train_step import TrainStep
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    model = CoPoNeRF.CoPoNeRF(n_view=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes), strict=True)
    model = model.to(dev).train()
    before = model.query_encode_latent.weight.detach().clone()
    nb = cd.broadcast_parameters(model, force=True)
    same_after_bcast = bool(torch.equal(before, model.query_encode_latent.weight.detach()))
    assert cd.broadcast_parameters(model) == 0 and cd.average_gradients(list(model.parameters())) == 0   # unforced: early-outs
    mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
    inp = mv(syn.make_inputs(1, 256, 256, 128, seed=61))
    # the exchange on one rank is the identity, bit for bit
    out = model(inp, val=False)
    (out["rgb"] - inp["query"]["rgb"]).abs().mean().backward()
    params = [p for p in model.parameters()]
    local = [None if p.grad is None else p.grad.clone() for p in params]
    finite, norm = cd.guard_and_clip(params, 0.0, force=True)
    ncoll = cd.average_gradients(params, bucket_bytes=16 << 20, force=True)
    ident = all((g is None and p.grad is None) or torch.equal(p.grad, g) for p, g in zip(params, local))
    model.zero_grad(set_to_none=True)
    # ... and the whole TrainStep with the forced exchange
    step = TrainStep(model, force_collectives=True)
    r1 = step(inp, inp["query"]["rgb"])
    r2 = step(inp, inp["query"]["rgb"])
    torch.cuda.synchronize()
    # the step with the exchange is as host-free as the step without it (dist.DeviceExchange): the flags come back lazily,
    # no device -> host read, the gradient-mask union agreed once
    from coponerf_amd.train_step import _LazyFlag
    lazy = isinstance(r1["stepped"], _LazyFlag)
    w_ex = model.query_encode_latent.weight.detach().clone()
    q.put(dict(nb=nb, same_after_bcast=same_after_bcast, finite=finite, ncoll=ncoll, ident=ident, lazy=lazy,
               host_reads=(r1["host_reads"], r2["host_reads"]), mask_exchanges=r2["mask_exchanges"],
               stepped=(bool(r1["stepped"]), bool(r2["stepped"])), collectives=r1["collectives"], nbytes=r1["allreduce_bytes"],
               losses=(float(r1["loss"]), float(r2["loss"])),
               moved=not torch.equal(before, model.query_encode_latent.weight.detach())))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_world1_forced_exchange_runs_on_device():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_world1, args=(_free_port(), q))
    p.start()
    import queue
    res = None
    for _ in range(900):
        try:
            res = q.get(timeout=1)
            break
        except queue.Empty:
            if not p.is_alive():
                break
    assert res is not None, f"worker exited with code {p.exitcode} without a result"
    p.join(timeout=120)
    assert p.exitcode == 0
    print("world-1 RCCL exchange:", res)
    assert res["nb"] >= 1 and res["same_after_bcast"] and res["finite"] and res["ident"]
    assert res["ncoll"] >= 5                             # ~143 MB of gradients in 16 MB buckets
    assert all(res["stepped"]) and res["collectives"] >= 2 and res["nbytes"] > 100e6 and res["moved"]
    assert res["lazy"] and res["host_reads"] == (0, 0) and res["mask_exchanges"] == 1
    assert all(l == l and l < 10 for l in res["losses"])


def _worker_two_ranks_one_device(rank, world, port, q):
    """Two ranks on ONE MI355X over backend gloo (it all-reduces device tensors through the host): everything of the N > 1 step
    except RCCL itself runs on the device - the MIN-reduced flag gating cpn_adam_step, the persistent buckets, the optimizer
    reading the summed gradients in place with 1 / world as its scale."""
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import torch.distributed as dist
    from coponerf_amd import CoPoNeRF, dist as cd, synthetic as syn
    from coponerf_amd.train_step import TrainStep, _LazyFlag
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    model = CoPoNeRF.CoPoNeRF(n_view=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes, seed=11 + rank), strict=True)          # ranks start different ...
    model = model.to(dev).train()
    cd.broadcast_parameters(model)                                                             # ... train.py:58-60
    mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
    inp = mv(syn.make_inputs(1, 256, 256, 128, seed=61 + rank))                                # independent batches
    probe = model.query_encode_latent.weight
    w0 = probe.detach().clone()
    # the local gradient of this rank's batch, for the hand-made reference of the first update
    out = model(inp, val=False)
    (out["rgb"] - inp["query"]["rgb"]).abs().mean().backward()
    g_local = probe.grad.detach().clone()
    total = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in model.parameters() if p.grad is not None))
    coef = float(torch.clamp(1.0 / (total + 1e-6), max=1.0))                                   # per-rank clip BEFORE the exchange
    model.zero_grad(set_to_none=True)
    g_sum = g_local * coef
    dist.all_reduce(g_sum, op=dist.ReduceOp.SUM)
    step = TrainStep(model, lr=1e-3)
    r1 = step(inp, inp["query"]["rgb"])
    # Adam's first update is -lr * sign-like g / (|g| + eps): compare with the averaged, clipped gradient
    g_avg = g_sum / world
    want = w0 - 1e-3 * g_avg / (g_avg.abs() + 1e-8)
    big = g_avg.abs() > 1e-3 * g_avg.abs().max()            # (where |g| is near Adam's eps the ratio is ill-conditioned)
    first_ok = bool(big.float().mean() > 0.2) and bool(torch.allclose(probe.detach()[big], want[big], rtol=0, atol=2e-5))
    r2 = step(inp, inp["query"]["rgb"])
    w = torch.cat([p.detach().reshape(-1) for p in (model.query_encode_latent.weight, model.phi.lin_out.weight, model.conv_map.weight)])
    both = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(both, w)
    q.put(dict(rank=rank, lazy=isinstance(r1["stepped"], _LazyFlag), stepped=(bool(r1["stepped"]), bool(r2["stepped"])),
               host_reads=(r1["host_reads"], r2["host_reads"]), mask_exchanges=r2["mask_exchanges"], collectives=r2["collectives"],
               equal=bool(torch.equal(both[0], both[1])), first_ok=first_ok, moved=not torch.equal(w0, probe.detach())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_one_device_over_gloo():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_two_ranks_one_device, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    import queue
    for _ in range(900):
        try:
            res.append(q.get(timeout=1))
        except queue.Empty:
            if not any(p.is_alive() for p in procs):
                break
        if len(res) == 2:
            break
    for p in procs:
        p.join(timeout=120)
    assert len(res) == 2 and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for r in res:
        print("two ranks, one device:", r)
        assert r["lazy"] and all(r["stepped"]) and r["host_reads"] == (0, 0) and r["mask_exchanges"] == 1 and r["collectives"] >= 2
        assert r["equal"] and r["first_ok"] and r["moved"]
