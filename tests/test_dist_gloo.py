"""world_size-2 gloo tests (CPU) of the data-parallel helpers in coponerf_amd/dist.py."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coponerf_amd import dist as cd
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.Linear(16, 4),
                                torch.nn.Linear(4, 4))
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    nb = cd.broadcast_parameters(model, bucket_bytes=256)
    x = torch.full((4, 8), float(rank + 1)) + torch.arange(4.0)[:, None]
    model[:3](x).sum().backward()                      # the last Linear gets no gradient (skip-None path)
    expected = [None if p.grad is None else p.grad.clone() for p in model.parameters()]
    ncoll = cd.average_gradients(model.parameters(), bucket_bytes=512)
    ok = True                                          # reference semantics: per-parameter all_reduce(SUM) / world
    for p, e in zip(model.parameters(), expected):
        if e is None:
            ok &= p.grad is None
            continue
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
        ok &= bool(torch.allclose(p.grad, e / world, atol=1e-6))
    finite_all = cd.grads_finite(model.parameters())
    if rank == 1:
        next(model.parameters()).grad[0, 0] = float("nan")
    finite_after_nan = cd.grads_finite(model.parameters())   # False on BOTH ranks: same branch, no deadlock
    # ---- ranks with DIFFERENT sets of gradients (rank 1 also trains the last Linear): no hang, the rank without the
    #      gradient contributes zeros and RECEIVES the average (both replicas then take the same optimizer step);
    #      large-but-finite gradients still count as finite
    model.zero_grad(set_to_none=True)
    (model(x) if rank == 1 else model[:3](x)).sum().backward()
    local = None if model[3].weight.grad is None else model[3].weight.grad.clone()
    cd.average_gradients(model.parameters(), bucket_bytes=512)
    src = local if rank == 1 else torch.zeros_like(model[3].weight)
    dist.broadcast(src, 1)
    uneven_ok = model[3].weight.grad is not None and bool(torch.allclose(model[3].weight.grad, src / world))
    # a rank WITHOUT any gradient still joins the guard's flag exchange (the others are blocked in it)
    saved = [p.grad for p in model.parameters()]
    if rank == 0:
        for p in model.parameters():
            p.grad = None
    fin0, _ = cd.guard_and_clip(model.parameters(), max_norm=0.0)
    uneven_ok = uneven_ok and fin0 is True
    for p, g in zip(model.parameters(), saved):
        p.grad = g
    model[0].weight.grad.fill_(1e30)                   # squares overflow fp32; the values themselves are finite
    big_finite = cd.grads_finite(model.parameters())
    # guard + clip in one pass: equals clip_grad_norm_ on finite gradients, reports the NaN on both ranks
    model.zero_grad(set_to_none=True)
    model[:3](x).sum().backward()
    ref = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    ref_norm = torch.sqrt(sum(g.double().pow(2).sum() for g in ref))            # what clip_grad_norm_ computes
    fin, tot = cd.guard_and_clip(model.parameters(), max_norm=0.5)
    got = [p.grad for p in model.parameters() if p.grad is not None]
    scale = min(1.0, 0.5 / (float(ref_norm) + 1e-6))
    clip_ok = fin and abs(float(tot) - float(ref_norm)) <= 1e-4 * float(ref_norm) and all(
        torch.allclose(g, r * scale, rtol=1e-5, atol=1e-7) for g, r in zip(got, ref))
    if rank == 0:
        got[0].view(-1)[0] = float("inf")
    clip_nan, _ = cd.guard_and_clip(model.parameters(), max_norm=0.5)
    w0 = model[0].weight.detach().clone()
    gathered = [torch.zeros_like(w0) for _ in range(world)]
    dist.all_gather(gathered, w0)
    q.put((rank, ok, ncoll, nb, finite_all, finite_after_nan, bool(torch.equal(gathered[0], gathered[1])),
           uneven_ok, big_finite, clip_ok, clip_nan, list(cd.shard_pairs(5, rank, world))))
    dist.destroy_process_group()


def test_bucketed_allreduce_and_finite_flag_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, ncoll, nb, finite_all, finite_after_nan, synced, uneven_ok, big_finite, clip_ok, clip_nan, shard in res:
        assert ok and synced and uneven_ok and big_finite and clip_ok and clip_nan is False
        assert 1 <= ncoll < 6 and nb >= 1            # fewer collectives than the 6 gradient tensors
        assert finite_all is True and finite_after_nan is False
    assert res[0][-1] == [0, 2, 4] and res[1][-1] == [1, 3]


class _TinyRenderer(torch.nn.Module):
    """Stand-in with the drop-in model's call contract (model(model_input, val=False) -> {'rgb', 'at_wt'}), small enough
    for two CPU ranks: coponerf_amd.train_step.TrainStep itself is what runs (wrapper.py:104-151)."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(2, 16)
        self.b = torch.nn.Linear(16, 3)
        self.unused = torch.nn.Linear(4, 4)                  # never reached by the loss: no gradient on any rank

    def forward(self, inp, val=False):
        uv = inp["query"]["uv"]
        h = torch.tanh(self.a(uv / 64.0))
        rgb = self.b(h) * inp["scale"]
        if inp.get("use_extra"):                             # a loss term only some ranks have (their gradient masks differ)
            rgb = rgb + 0.1 * self.unused(h[..., :4])[..., :3]
        return {"rgb": rgb, "at_wt": h.detach()[..., :1]}


def _train_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coponerf_amd import dist as cd
    from coponerf_amd.train_step import TrainStep
    torch.manual_seed(10 + rank)                             # ranks start from different weights ...
    model = _TinyRenderer()
    cd.broadcast_parameters(model)                           # ... train.py:58-60
    ref = _TinyRenderer()
    ref.load_state_dict(model.state_dict())
    step = TrainStep(model, lr=1e-2, clip_grad=0.05, bucket_bytes=256)
    g = torch.Generator().manual_seed(100 + rank)            # independent batches per rank (train.py:84-97)
    uv = torch.rand(2, 1, 32, 2, generator=g) * 64
    gt = torch.rand(2, 1, 32, 3, generator=g)
    batch = {"query": {"uv": uv}, "scale": torch.tensor(1.0)}
    info = step(batch, gt)
    # the same step by hand: per-rank clip BEFORE the exchange (wrapper.py:142-151), per-parameter SUM / world, Adam
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    loss = (gt - ref(batch)["rgb"]).abs().mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.05)
    for p in ref.parameters():
        if p.grad is not None:
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
            p.grad /= world
    opt.step()
    same = all(torch.allclose(a, b, rtol=1e-5, atol=1e-7) for a, b in zip(model.parameters(), ref.parameters()))
    w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    both = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(both, w)
    replicas_equal = bool(torch.equal(both[0], both[1]))
    # a NaN on ONE rank: every rank skips the update (same branch, no hang), nothing changes, the skip is counted
    before = [p.detach().clone() for p in model.parameters()]
    bad = {"query": {"uv": uv}, "scale": torch.tensor(float("nan") if rank == 1 else 1.0)}
    info_bad = step(bad, gt)
    unchanged = all(torch.equal(a, b) for a, b in zip(before, model.parameters()))
    info_ok = step(batch, gt)                                # and training goes on afterwards
    masks_so_far = int(info_ok["mask_exchanges"])            # the has-gradient union was agreed ONCE in three steps
    # rank 1 alone gains a gradient (`unused` enters its loss): its vote re-opens the agreement on BOTH ranks, rank 0
    # contributes zeros and receives the average - the replicas stay identical; afterwards the mask is cached again
    extra = dict(batch, use_extra=(rank == 1))
    before_unused = model.unused.weight.detach().clone()
    info_x = step(extra, gt)
    info_x2 = step(extra, gt)
    w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    both = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(both, w)
    mask_ok = (masks_so_far == 1 and int(info_x["mask_exchanges"]) == 2 and int(info_x2["mask_exchanges"]) == 2
               and bool(torch.equal(both[0], both[1])) and not torch.equal(before_unused, model.unused.weight.detach()))
    q.put((rank, float(info["loss"]), bool(info["stepped"]), int(info["collectives"]), same, replicas_equal,
           bool(info_bad["stepped"]), unchanged, int(info_bad["skipped_in_a_row"]), bool(info_ok["stepped"]),
           int(info_ok["skipped_in_a_row"]), all(p.grad is None for p in model.parameters()), mask_ok))
    dist.destroy_process_group()


def test_train_step_world2_matches_reference_semantics_and_skips_together():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, loss, stepped, ncoll, same, replicas_equal, bad_stepped, unchanged, skipped, ok_stepped, skipped_after, cleared, mask_ok in res:
        assert stepped and ncoll >= 1 and same and replicas_equal and mask_ok, res
        assert bad_stepped is False and unchanged and skipped == 1, res          # BOTH ranks, though only rank 1 saw the NaN
        assert ok_stepped and skipped_after == 0 and cleared, res
    assert res[0][1] != res[1][1]                                                # the ranks really trained on different data
