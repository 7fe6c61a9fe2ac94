#!/usr/bin/env python
"""Headline benchmark: rendered rays/sec of the CoPoNeRF render path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: spawns N ranks itself, one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

--mode render (default; BASELINE.json configs[1]): one RealEstate10K-shaped 256x256 stereo pair per GPU, full-image
render (65 536 query rays), 64 samples per epipolar line, synthetic latents/cameras/weights of the reference's shapes
(no dataset or checkpoint offline).  A "step" is one full-image render pass (forward with z/rel_pose/flow given,
val=True, no_grad — the test.py path) with the feature maps already resident in HBM.  Weak scaling: every rank renders
its own pair; no data-path collective (stereo pairs are independent).

--mode train (BASELINE.json configs[2]): batch = 4 pairs x 4096 rays x 64 samples per GPU; a step is get_z + render +
L1 loss + backward + fused finite guard + clip + bucketed RCCL gradient all-reduce + Adam (coponerf_amd/train_step.py,
mirroring /root/reference wrapper.py:104-151, train.py:58-60,141-147).  The default render run appends a short
training measurement as the secondary `train` block, so the all-reduce is exercised at every N the driver runs.

Prints ONE JSON line on rank 0 (driver contract) with `roofline` (dominant kernel, timed with HIP events on the launch
stream inside the timed region) and `cpu_baseline` (the CPU oracle on a bounded sample of the same workload).
"""
import argparse
import hashlib
import json
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F16_MFMA_PEAK_TFLOPS = 2500.0          # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0                  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def f_ray(S: int) -> float:
    """Algorithmic FLOPs per rendered ray, V = 2."""
    return S * 10419968.0 + 1053952.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=("render", "train"), default="render")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--pairs", type=int, default=None, help="stereo pairs per GPU (default 1 render, 4 train)")
    ap.add_argument("--pair-by-pair", action="store_true",
                    help="render the --pairs stereo pairs one forward() call each (different pairs: the feature tables "
                         "are rebuilt for every call inside the timed region) instead of one batched call")
    ap.add_argument("--train-rays", type=int, default=4096, help="query rays per pair in a training step")
    ap.add_argument("--chunk-rays", type=int, default=0, help="rays per chunk of the per-sample stages; 0 = the engine's automatic "
                    "choice (a 65 536-ray image is one chunk on a 288 GB device)")
    ap.add_argument("--lanes", type=int, default=1,
                    help="HIP streams the ray chunks are spread over (kernels of different chunks then share the GPU "
                         "and the per-kernel roofline timing is no longer clean)")
    ap.add_argument("--no-image", action="store_true", help="skip the secondary image-pipeline figures (get_z + render)")
    ap.add_argument("--no-f32", action="store_true", help="skip the reference-arithmetic (fp32-operand) pass of the same step")
    ap.add_argument("--cpu-rays", type=int, default=8192, help="upper bound on the rays of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-fresh-pair", action="store_true",
                    help="skip the two loops over NEW stereo pairs (rays_per_s_fresh_pair[_announced]); the profiled command of "
                         "tools/capture_profiles.sh uses it: the announced loop's preparation kernels run BESIDE the render "
                         "kernels on their own stream and would show up in the kernel statistics with their co-run durations")
    ap.add_argument("--no-two-stream-pass", action="store_true",
                    help="skip the second timed pass with consecutive calls on two streams (keeps a rocprofv3 kernel "
                         "trace of this command to the one-stream headline loop, whose kernels never overlap)")
    ap.add_argument("--no-ref-loop", action="store_true",
                    help="skip the secondary measurement of the reference callers' chunked loop (test.py:164-212)")
    ap.add_argument("--rig", choices=("narrow", "wide"), default="narrow",
                    help="camera rig of the synthetic pairs: RealEstate10K-like (configs[1]) or ACID-like wide baseline "
                         "(configs[3])")
    ap.add_argument("--train-steps", type=int, default=6,
                    help="steps of the secondary training measurement in render mode (0 = skip)")
    return ap.parse_args(argv)


def _to(o, dev):
    if torch.is_tensor(o):
        return o.to(dev)
    if isinstance(o, dict):
        return {k: _to(v, dev) for k, v in o.items()}
    return type(o)(_to(v, dev) for v in o)


def _max_over_ranks(x: float, dev, distributed: bool) -> float:
    if not distributed:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _fence(distributed: bool):
    torch.cuda.synchronize()
    if distributed:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


# --------------------------------------------------------------------------------------------------------------
def measure_train(args, dev, rank, world, steps, warmup):
    """configs[2] on this rank's GPU: returns the dict of the training measurement (all ranks must call it)."""
    from coponerf_amd import CoPoNeRF, dist as cdist, synthetic as syn
    from coponerf_amd.train_step import TrainStep
    distributed = world > 1
    B = args.pairs if (args.mode == "train" and args.pairs) else 4
    R, S = args.train_rays, args.samples
    import gc
    gc.collect()                                            # the render model of the earlier measurements sits in reference cycles
    torch.cuda.empty_cache()
    resident = torch.cuda.memory_allocated(dev)             # whatever those measurements still hold after that
    model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes, seed=11 + rank), strict=True)     # ranks start different ...
    model = model.to(dev).train()
    nbcast = cdist.broadcast_parameters(model)                                               # ... train.py:58-60
    inp = _to(syn.make_inputs(B, 256, 256, R, seed=61 + rank), dev)                          # independent batches
    gt = inp["query"]["rgb"]
    step = TrainStep(model)
    info = None
    torch.cuda.reset_peak_memory_stats(dev)                 # the render measurements before this one do not count
    for _ in range(warmup):
        info = step(inp, gt)
    _fence(distributed)
    step.timing = {}
    if step._exchange is not None:
        step._exchange.timing = []
    t0 = time.perf_counter()
    for _ in range(steps):
        info = step(inp, gt)
    _fence(distributed)
    elapsed = _max_over_ranks(time.perf_counter() - t0, dev, distributed)
    phases = step.timing_summary()
    rays = B * R * world * steps / elapsed
    # what the communicator itself reports (not the WORLD_SIZE this process was given): ranks, backend, the device every rank
    # sits on, the bytes the timed steps all-reduced and when each bucket's collective completed
    comm = {"world_size": 1, "backend": None, "devices": [f"{torch.cuda.get_device_name(dev)} #{dev.index}"]}
    if distributed:
        import socket
        import torch.distributed as dist
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": dist.get_rank(), "host": socket.gethostname(), "device": dev.index, "name": props.name,
                "pci_bus_id": getattr(props, "pci_bus_id", None)}
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, mine)
        comm = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "devices": gathered,
                "distinct_devices": len({(g["host"], g["device"]) for g in gathered})}
    ex_t = step._exchange.timing_summary() if step._exchange is not None else {}
    # weak-scaling efficiency of THIS run: the same ranks step again with the exchange switched off (every rank for itself -
    # what a 1-GPU run does); ms_per_step(1 rank's work) / ms_per_step(N ranks exchanging).  The replicas diverge here, so
    # this runs last.
    local_ms = None
    if distributed:
        solo = TrainStep(model, exchange=False)
        for _ in range(2):
            solo(inp, gt)
        _fence(distributed)
        t1 = time.perf_counter()
        for _ in range(steps):
            solo(inp, gt)
        _fence(distributed)
        local_ms = 1e3 * _max_over_ranks(time.perf_counter() - t1, dev, distributed) / steps
        del solo
    res = {"rays_per_s": rays, "ms_per_step": 1e3 * elapsed / steps, "steps": steps, "warmup": warmup,
           "pairs_per_gpu": B, "rays_per_pair": R, "samples": S, "n_gpus": world,
           "collectives_per_step": info["collectives"], "allreduce_bytes_per_step": info["allreduce_bytes"],
           # ranks whose gradients were exchanged over RCCL in the timed steps (1: no exchange ran) and device -> host reads
           # the step makes (the guard flag gates the update kernel on the device, with or without an exchange)
           "rccl_ranks": comm["world_size"], "comm": comm, "exchange_timing": ex_t,
           "gradient_mask_path": step._exchange.mask_path if step._exchange is not None else None,
           "ms_per_step_without_exchange": local_ms,
           "weak_scaling_efficiency": (local_ms / (1e3 * elapsed / steps)) if local_ms else None,
           "host_reads_per_step": info["host_reads"],
           "gradient_mask_exchanges": info["mask_exchanges"],
           "broadcast_collectives": nbcast, "stepped": bool(info["stepped"]), "loss": float(info["loss"]),
           "phases_ms": phases, "peak_mem_GB": max(0.0, torch.cuda.max_memory_allocated(dev) - resident) / 2 ** 30,
           # forward + backward ~ 3x the forward's algorithmic FLOPs; get_z 227.8 GFLOP per pair
           "algorithmic_tflops": rays * 3.0 * (f_ray(S) + 227.8e9 / R) / 1e12}
    del step, model
    torch.cuda.empty_cache()
    return res


# --------------------------------------------------------------------------------------------------------------
def run(args):
    # (round 2 set torch.backends.cudnn.benchmark = True here: MIOpen's search then won 0.8 ms per get_z.  With the
    # deterministic convolution choices get_z now asks for on the inference path the search LOSES 1.1 ms — 12.1 vs 11.0 ms,
    # tools/getz_bench_flags.py — so the library default stays; CPN_BENCH_CUDNN_BENCHMARK=1 turns the search on for A/B runs.)
    if os.environ.get("CPN_BENCH_CUDNN_BENCHMARK") == "1":
        torch.backends.cudnn.benchmark = True
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU")
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the render path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        # the host side of a step is a few O(B) 4x4 products: keep N ranks from oversubscribing the host cores
        torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)          # "nccl" on PyTorch-ROCm is RCCL

    if args.mode == "train":
        tr = measure_train(args, dev, rank, world, args.steps, args.warmup)
        if rank == 0:
            line = {
                "metric": "training rays/sec (RealEstate10K-shaped 256x256 pairs, 4096 rays/pair, 64 samples/ray, "
                          "forward + backward + RCCL gradient all-reduce + Adam)",
                "value": tr["rays_per_s"], "unit": "rays/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": tr["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None,
                "dtype": "f16 (fp16-input/fp32-accumulate MFMA per-sample MLPs, scaled fp16 activation gradients; "
                         "f32 get_z / decoder / weight gradients; f64 geometry island)",
                "data": "synthetic",
                "config": {"workload": f"configs[2]: batch {tr['pairs_per_gpu']} pairs x {tr['rays_per_pair']} rays x "
                                       f"{tr['samples']} samples per GPU, data-parallel over {world} GPU(s), one process "
                                       f"per GPU, bucketed RCCL all-reduce of the gradients", "pairs_per_gpu": tr["pairs_per_gpu"]},
                "train": tr,
                "roofline": {"bound": "mfma", "kernel": "whole step (forward + backward), algorithmic FLOPs",
                             "achieved": tr["algorithmic_tflops"] / world, "peak": F16_MFMA_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": tr["algorithmic_tflops"] / world / F16_MFMA_PEAK_TFLOPS,
                             "traffic": None},
            }
            print(json.dumps(line))
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---------------------------------------------------------------------------------------------- render mode
    from coponerf_amd import CoPoNeRF, synthetic as syn
    H, S, B = args.height, args.samples, (args.pairs or 1)
    torch.manual_seed(0)
    model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
    model.load_state_dict(syn.make_render_weights(), strict=False)
    model = model.to(dev).eval()
    model._engine.chunk_rays = args.chunk_rays
    model._engine.lanes = args.lanes

    P = B if args.pair_by_pair else 1                      # forward() calls per step
    Bc = 1 if args.pair_by_pair else B                     # pairs per call
    jobs = []
    for j in range(P):
        ic = syn.make_inputs(Bc, H, H, 0, seed=100 + rank + 1000 * j, full_image=True, rig=args.rig)
        zc, rc, fc = syn.make_latents(Bc, H, H, seed=200 + rank + 1000 * j)
        jobs.append((ic, zc, rc, fc, _to(ic, dev), _to(zc, dev), rc.to(dev), _to(fc, dev)))
    inp_cpu, z_cpu, rel_cpu, flow_cpu, inp, z, rel, flow = jobs[0]
    R = inp["query"]["uv"].shape[2]
    rays_per_step = B * R

    def step():
        with torch.no_grad():
            out0 = None
            for job in jobs:
                o = model(job[4], z=job[5], rel_pose=job[6], val=True, flow=job[7])
                out0 = o if out0 is None else out0
            return out0

    # the headline loop runs every call on ONE stream (call_lanes = 1): with consecutive forward() calls alternating over
    # two streams (the library default, made for the callers' 18-call loop) the tail of step i overlaps step i+1 and the
    # per-kernel HIP-event timings behind `roofline` would include the other stream's kernels
    lanes_default = model._engine.call_lanes
    model._engine.call_lanes = 1
    # the warm-up steps run with the per-kernel event timing of the timed region switched on as well (first-time creation of the
    # timing events then happens here); their readings are dropped.  (What made --warmup 1 read 8 % slower than --warmup 2 was
    # the SECOND pinned pixel_val buffer, pinned inside the second call: RenderEngine._start_host_copy now parks it with the first)
    model._engine.profile = {}
    for _ in range(args.warmup):
        out = step()
    _fence(distributed)
    model._engine.profile = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    _fence(distributed)
    elapsed = time.perf_counter() - t0
    prof, model._engine.profile = model._engine.profile, None
    elapsed = _max_over_ranks(elapsed, dev, distributed)
    model._engine.call_lanes = lanes_default
    overlapped = None
    if lanes_default > 1 and not args.pair_by_pair and not args.no_two_stream_pass:
        for _ in range(2):
            step()
        _fence(distributed)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        _fence(distributed)
        overlapped = _max_over_ranks(time.perf_counter() - t0, dev, distributed)

    # ---- the same call on a NEW stereo pair every step: what the headline loop finds cached after its warm-up — the NHWC fp16
    #      copies of the latent maps, the node tables (node features + a 97 GFLOP projection GEMM per pair), the camera
    #      block upload, the flow products — is rebuilt inside the timed region (four pairs in turn; the caches hold one)
    fresh = announced = None
    if not args.pair_by_pair and B == 1 and not args.no_fresh_pair:
        fjobs = []
        for j in range(4):
            ic = syn.make_inputs(1, H, H, 0, seed=500 + rank + 1000 * j, full_image=True, rig=args.rig)
            zc, rc, fc = syn.make_latents(1, H, H, seed=600 + rank + 1000 * j)
            fjobs.append((_to(ic, dev), _to(zc, dev), rc.to(dev), _to(fc, dev)))

        def fresh_step(i):
            a = fjobs[i % len(fjobs)]
            with torch.no_grad():
                return model(a[0], z=a[1], rel_pose=a[2], val=True, flow=a[3])
        model._engine.call_lanes = 1
        for i in range(2):
            fresh_step(i)
        _fence(distributed)
        t0 = time.perf_counter()
        for i in range(args.steps):
            fresh_step(2 + i)
        _fence(distributed)
        fresh = _max_over_ranks(time.perf_counter() - t0, dev, distributed)
        # the same loop with every pair ANNOUNCED one call ahead (CoPoNeRF.prepare_next): its camera copy to the host, its
        # maps / tables / flow products run on their own stream under the previous pair's kernels
        def announce(i):
            a = fjobs[i % len(fjobs)]
            model.prepare_next(a[0], a[1], a[2], a[3])
        for i in range(2):
            announce(i + 1)
            fresh_step(i)
        _fence(distributed)
        t0 = time.perf_counter()
        for i in range(args.steps):
            announce(2 + i + 1)
            fresh_step(2 + i)
        _fence(distributed)
        announced = _max_over_ranks(time.perf_counter() - t0, dev, distributed)
        model._engine.call_lanes = lanes_default
        del fjobs
        with torch.no_grad():
            step()                                              # back to the headline pair (re-primes its caches)

    # ---- the same step in the reference's arithmetic (RenderEngine.precision = "f32"): fp32 node tables, fp32 blends, hid as fp16
    #      (hi, lo) pairs with exact products, exact-fp32 small layers (csrc/encode_f32.hip) - the same-precision number beside
    #      the headline, how far the fp16-operand image is from it, and its roofline view.  One untimed + three timed steps.
    f32 = None
    if B == 1 and not args.pair_by_pair and not args.no_f32 and world == 1:
        eng = model._engine
        rgb16 = out["rgb"].clone()
        eng.precision = "f32"
        lanes_held, eng.call_lanes = eng.call_lanes, 1          # one stream, like the headline loop (two co-running calls of the
        try:                                                    # mode's LDS-resident persistent kernels take turns, badly)
            step()
            _fence(distributed)
            t0 = time.perf_counter()
            for _ in range(3):
                o32 = step()
            _fence(distributed)
            f32 = {"seconds_per_step": (time.perf_counter() - t0) / 3,
                   "rgb_max_abs_f16_vs_f32": float((rgb16 - o32["rgb"]).abs().max()),
                   "form": "fp32 tables, folded key / value, (hi, lo) fp16 hid (csrc/encode_f32.hip)"}
            f32["rays_per_s_f32"] = rays_per_step / f32["seconds_per_step"]
            # what this form executes per ray: the 4 table taps in fp32 on the vector unit, the K = 68 block of the first layer and
            # the 128-wide layers on the exact fp32 MFMA, the folded key layer as three fp16 MFMA products (exact)
            S_ = args.samples
            vec = 2.0 * 2 * S_ * 2 * 832 * 4
            mfma16 = 2.0 * 2 * S_ * 128 * (3328 + 1664)
            mfma32 = 2.0 * 2 * S_ * (2 * 832 * 68 + 3 * 128 * 128 + 2 * 16 * 128)
            f32["executed_tflops"] = {"fp32_vector": vec * f32["rays_per_s_f32"] / 1e12,
                                      "fp16_mfma_exact_products": mfma16 * f32["rays_per_s_f32"] / 1e12,
                                      "fp32_mfma": mfma32 * f32["rays_per_s_f32"] / 1e12}
            # HBM: hid written once and read by the key layer (twice: two GEMM launches) and by both attention rounds
            f32["roofline"] = {"bound": "hbm", "achieved": 5.0 * 2 * S_ * 6656 * f32["rays_per_s_f32"] / 1e9, "peak": 8000.0,
                               "unit": "GB/s", "frac": 5.0 * 2 * S_ * 6656 * f32["rays_per_s_f32"] / 1e9 / 8000.0,
                               "note": "5 passes over the (hi, lo) hidden activations, 6 656 B per sample and view; the fp32 MFMA "
                                       "(157 TFLOP/s) is the other bound: see executed_tflops"}
            del o32
        finally:
            eng.precision = "f16"
            eng.call_lanes = lanes_held
            for k in [k for k in eng._ws if "f32" in k]:
                del eng._ws[k]                                  # ~20 GB of chunk buffers
            eng._t32 = None
            torch.cuda.empty_cache()

    value = rays_per_step * world * args.steps / elapsed
    tables = True
    # executed FLOPs per ray: value/key projections folded; the 3 x 256 coarse channels of the first layer
    # are table taps (12 x 832 FMA per row) instead of a 768-deep contraction
    exec_per_ray = S * 6637056.0 + 4300000.0 - S * 4 * 2.0 * 832 * (768 - 12)
    cfg_name = ("configs[4]" if H == 512 else "configs[3]" if args.rig == "wide" else "configs[1]") \
        if (H, S) in ((256, 64), (512, 128)) else "custom"
    line = {
        "metric": f"rendered rays/sec ({'ACID' if args.rig == 'wide' else 'RealEstate10K'}-shaped {H}x{H} stereo pair, "
                  f"full-image render, {S} samples/ray)",
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16 (fp16-input/fp32-accumulate MFMA for the per-sample MLPs; f32 decoder; f64 geometry island)",
        "data": "synthetic",
        "config": {"workload": f"{cfg_name}: {H}x{H} stereo pair ({args.rig} rig), full-image render {R} rays x {S} "
                               f"samples, {B} pair(s) per GPU{' rendered pair by pair' if args.pair_by_pair else ''}, "
                               f"render path only (z/rel_pose/flow given, val=True)",
                   "chunk_rays": args.chunk_rays or model._engine._auto_chunk(S, dev), "lanes": args.lanes, "pairs_per_gpu": B,
                   "first_layer": "projected tables + K=80 MFMA + folded key_map layer on the register-resident slices (cpn_encode_key)"},
        "rays_per_s_calls_on_two_streams": None if overlapped is None else rays_per_step * world * args.steps / overlapped,
        # a new pair every step (tables, NHWC copies, camera upload and flow products rebuilt inside the timed region)
        "rays_per_s_fresh_pair": None if fresh is None else rays_per_step * world * args.steps / fresh,
        "ms_per_step_fresh_pair": None if fresh is None else 1e3 * fresh / args.steps,
        "rays_per_s_fresh_pair_announced": None if announced is None else rays_per_step * world * args.steps / announced,
        "ms_per_step_fresh_pair_announced": None if announced is None else 1e3 * announced / args.steps,
        # the reference's arithmetic (fp32 operands, layer by layer) on the same workload, and the image's distance from it
        "rays_per_s_f32": None if f32 is None else f32["rays_per_s_f32"],
        "rgb_max_abs_f16_vs_f32": None if f32 is None else f32["rgb_max_abs_f16_vs_f32"],
        "f32_mode": None if f32 is None else {k: f32[k] for k in ("form", "seconds_per_step", "executed_tflops", "roofline")},
        "path_tflops": value * f_ray(S) / 1e12,          # algorithmic FLOPs of the reference formulation
        "executed_tflops": value * exec_per_ray / 1e12,  # what the kernels execute after the restructurings
    }

    # ---- secondary figure: the whole image pipeline (get_z once per pair + the render pass), SURVEY.md §8(d)
    if H == 256 and not args.no_image:
        with torch.no_grad():
            for _ in range(5):                               # (the library picks its convolution kernels on first use)
                zz = model.get_z(inp)
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            for _ in range(10):
                zz = model.get_z(inp)
            torch.cuda.synchronize()
            getz_ms = (time.perf_counter() - g0) / 10 * 1e3
        line["get_z_ms"] = getz_ms
        # the same call replayed as a captured HIP graph (coponerf_amd/graphs.py): no host cost per launch, what the GPU needs
        from coponerf_amd.graphs import GraphedGetZ
        with torch.no_grad():
            gz = GraphedGetZ(model)
            for _ in range(2):
                zz = gz(inp)
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            for _ in range(10):
                zz = gz(inp)
            torch.cuda.synchronize()
            line["get_z_graph_ms"] = (time.perf_counter() - g0) / 10 * 1e3
            del gz
        line["image_rays_per_s"] = rays_per_step / (getz_ms * 1e-3 + elapsed / args.steps)
        del zz
        # the same pipeline as coponerf_amd/pipeline.render_images runs it by default: one stream, the host issuing get_z of
        # pair i+1 while the GPU renders pair i (pairs are independent, results identical to the serial order)
        if B == 1:
            from coponerf_amd.pipeline import render_images
            pairs = [inp] + [_to(syn.make_inputs(1, H, H, 0, seed=300 + rank * 16 + i, full_image=True), dev) for i in range(3)]
            with torch.no_grad():
                for _ in render_images(model, pairs[:2]):
                    pass
                torch.cuda.synchronize()
                p0 = time.perf_counter()
                nimg = 0
                for _ in render_images(model, pairs + pairs):
                    nimg += 1
                torch.cuda.synchronize()
                pdt = time.perf_counter() - p0
            line["image_rays_per_s_pipelined"] = nimg * R / pdt
            line["image_ms_pipelined"] = 1e3 * pdt / nimg
            # the same loop on a PARTITIONED chip: render pass on 192 CUs, get_z on the other 64 (CU-masked streams,
            # coponerf_amd/streams.py) — ordinary streams cannot overlap the two (the line above); and with each image
            # rendered the way the reference's callers do (18 forward() calls, test.py:176-212)
            try:
                for tag, kw in (("image", {}), ("image_ref_loop", {"nchunks": 18})):
                    with torch.no_grad():
                        for _ in render_images(model, pairs[:2], cu_split=(192, 64), **kw):
                            pass
                        torch.cuda.synchronize()
                        p0 = time.perf_counter()
                        nimg = 0
                        for _ in render_images(model, pairs + pairs, cu_split=(192, 64), **kw):
                            nimg += 1
                        torch.cuda.synchronize()
                        pdt = time.perf_counter() - p0
                    line[f"{tag}_rays_per_s_cu_partition_192_64"] = nimg * R / pdt
                    line[f"{tag}_ms_cu_partition_192_64"] = 1e3 * pdt / nimg
            except Exception as e:                     # a side figure must not take the headline line down
                line["cu_partition_error"] = f"{type(e).__name__}: {e}"[:300]
            # throughput form of the same loop: get_z batched over 4 consecutive pairs (one launch sequence for four)
            with torch.no_grad():
                for _ in render_images(model, pairs, getz_batch=4):
                    pass
                torch.cuda.synchronize()
                p0 = time.perf_counter()
                nimg = 0
                for _ in render_images(model, pairs + pairs, getz_batch=4):
                    nimg += 1
                torch.cuda.synchronize()
                pdt = time.perf_counter() - p0
            line["image_rays_per_s_getz_batch4"] = nimg * R / pdt
            line["image_ms_getz_batch4"] = 1e3 * pdt / nimg
            # get_z of each pair replayed as a captured HIP graph (coponerf_amd/graphs.py), one pair at a time
            with torch.no_grad():
                for _ in render_images(model, pairs[:2], graph=True):
                    pass
                torch.cuda.synchronize()
                p0 = time.perf_counter()
                nimg = 0
                for _ in render_images(model, pairs + pairs, graph=True):
                    nimg += 1
                torch.cuda.synchronize()
                pdt = time.perf_counter() - p0
            line["image_rays_per_s_getz_graph"] = nimg * R / pdt
            line["image_ms_getz_graph"] = 1e3 * pdt / nimg

    # ---- secondary figure: the loop the reference's callers run (test.py:164-212): get_z once, then 18 forward() calls
    #      on torch.chunk(uv, 18) with the callers' del / .cpu() / per-key concat, at batch 1 and batch 2 (test.py:130)
    if H == 256 and not args.no_ref_loop and B == 1:
        line["ref_loop"] = ref_loop_block(model, syn, dev, rank, H, inp, z, rel, flow, elapsed / args.steps)

    if rank == 0:
        line.update(roofline_block(prof, args, tables))
        if args.cpu_rays > 0 and world == 1:          # CPU baseline: rank 0 at N = 1 only
            line.update(cpu_baseline_block(args, syn, inp_cpu, z_cpu, rel_cpu, flow_cpu, out, Bc, H, S))

    # ---- secondary: configs[2] training step on the same ranks (exercises the RCCL gradient all-reduce at N > 1)
    if H == 256 and args.train_steps > 0:
        del out, model, z, flow
        torch.cuda.empty_cache()
        try:
            tr = measure_train(args, dev, rank, world, args.train_steps, 3)
        except Exception as e:                         # the headline line must survive a failure of the side figure
            tr = {"error": f"{type(e).__name__}: {e}"}
        line["train"] = tr

    if rank == 0:
        print(json.dumps(line))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


# --------------------------------------------------------------------------------------------------------------
def ref_loop_block(model, syn, dev, rank, H, inp1, z1, rel1, flow1, single_call_s):
    """coponerf_amd/evalloop.py (= /root/reference test.py:164-212) on the bench's pair, at batch 1 and at the batch 2
    test.py:130 uses: 18 forward(val=True) calls on torch.chunk(uv, 18) with the callers' del / .cpu() / per-key concat.
    `render_ms` = calls + join with latents given (wall clock until the GPU has drained), `release_ms` = dropping the
    joined result (the callers' 67 MB CPU pixel_val), `image_ms` = get_z + calls + join as one loop iteration of the
    caller runs, releases included."""
    from coponerf_amd.evalloop import render_in_chunks
    res = {"chunks": 18, "single_call_ms": 1e3 * single_call_s, "call_lanes": model._engine.call_lanes}
    for nb in (1, 2):
        if nb == 1:
            inp, lat = inp1, (z1, rel1, flow1)
        else:
            inp = _to(syn.make_inputs(nb, H, H, 0, seed=400 + rank, full_image=True), dev)
            zc, rc, fc = syn.make_latents(nb, H, H, seed=500 + rank)
            lat = (_to(zc, dev), rc.to(dev), _to(fc, dev))
        R = inp["query"]["uv"].shape[2]
        for _ in range(2):
            out = render_in_chunks(model, inp, 18, latents=lat)
            del out
        torch.cuda.synchronize()
        n, t_loop, t_free = 6, [], []
        for _ in range(n):
            t0 = time.perf_counter()
            out = render_in_chunks(model, inp, 18, latents=lat)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            del out
            t2 = time.perf_counter()
            t_loop.append(1e3 * (t1 - t0))
            t_free.append(1e3 * (t2 - t1))
        render_ms, release_ms = sum(t_loop) / n, sum(t_free) / n
        with torch.no_grad():
            for _ in range(2):
                model.get_z(inp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                model.get_z(inp)
            torch.cuda.synchronize()
            getz_ms = (time.perf_counter() - t0) / 3 * 1e3
            t0 = time.perf_counter()
            for _ in range(4):
                out = render_in_chunks(model, inp, 18)               # get_z inside, as the caller runs it
                del out
            torch.cuda.synchronize()
            image_ms = (time.perf_counter() - t0) / 4 * 1e3
        res[f"batch{nb}"] = {"render_ms_per_batch": render_ms, "render_ms_min_max": [min(t_loop), max(t_loop)],
                              "release_ms": release_ms, "render_rays_per_s": nb * R / (render_ms * 1e-3),
                              "get_z_ms": getz_ms, "image_ms_per_batch": image_ms,
                              "image_rays_per_s_ref_loop": nb * R / (image_ms * 1e-3),
                              "render_vs_single_call": render_ms / (nb * 1e3 * single_call_s)}
    return res


# --------------------------------------------------------------------------------------------------------------
def rocprof_view(sha, kernel, rows, alg_bytes):
    """The same roofline fraction from the committed rocprofv3 kernel statistics of this command (profiles/
    r*_render_kernel_stats.summary.csv + its .meta.json, written by tools/capture_profiles.sh), if they were taken on THIS
    kernel source and launch shape: the profiler's average duration runs a few per cent above the HIP-event one of the
    unprofiled run, and the two fractions are printed side by side so that profiles/ and this line agree to the digit."""
    import glob
    for meta in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_render_kernel_stats.meta.json")), reverse=True):
        try:
            with open(meta) as f:
                m = json.load(f)
            if m.get("kernel_source_sha16") != sha or m.get("rows_per_launch") != rows:
                continue
            csv_path = meta.replace(".meta.json", ".summary.csv")
            with open(csv_path) as f:
                for ln in f.read().splitlines()[1:]:
                    if kernel in ln:
                        cells = ln.rsplit(",", 4)
                        avg_ms = float(cells[3]) / 1e3
                        gbs = alg_bytes / (avg_ms * 1e-3) / 1e9
                        return {"rocprof": {"avg_launch_ms": avg_ms, "calls": int(cells[1]), "achieved": gbs,
                                            "frac": gbs / HBM_PEAK_GBS, "source": os.path.relpath(csv_path, ROOT)}}
        except (OSError, ValueError, IndexError):
            continue
    return {"rocprof": None}


def roofline_block(prof, args, tables):
    """Roofline of the dominant kernel from the HIP-event timings taken inside the timed region."""
    out = {}
    kern = {}
    for k, v in prof.items():
        tot = sum(a.elapsed_time(b) for a, b, _ in v)
        kern[k] = {"ms_per_step": tot / args.steps, "tflops": sum(f for _, _, f in v) / (tot * 1e-3) / 1e12}
    out["kernel_breakdown"] = kern
    fused = "encode_key" in prof                     # first layer + folded key layer in one kernel (csrc/encode_fused.hip)
    name = ("encode_key" if fused else "encode_hidden") if tables else "gemm_f16:query_encode_latent"
    evs = prof.get(name, [])
    if not evs:
        return out
    ms = [a.elapsed_time(b) for a, b, _ in evs]
    flops = evs[0][2]                                           # algorithmic: 2 * rows * 832 * 835 per launch
    avg_ms = sum(ms) / len(ms)
    achieved = flops / (avg_ms * 1e-3) / 1e12
    rows = int(round(flops / (2.0 * 832 * 835)))
    src = os.path.join(ROOT, "coponerf_amd", "csrc", ("encode_fused.hip" if fused else "encode.hip") if tables else "gemm_f16.hip")
    with open(src, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()[:16]
    # HBM bytes per launch come from separate rocprofv3 --pmc passes of this same command (they cannot be read
    # live); a committed record is used only if it was taken on THIS kernel source and launch shape
    traffic, tsrc = None, None
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True) if tables else \
        [os.path.join(ROOT, "profiles", "r01_v4_traffic.json")]
    for tpath in cands:
        if not os.path.exists(tpath):
            continue
        with open(tpath) as f:
            rec = json.load(f)
        same_src = rec.get("kernel_source_sha16") == sha if tables else True
        if same_src and rec.get("shape", {}).get("M") == rows:
            traffic, tsrc = rec["hbm_bytes"], os.path.relpath(tpath, ROOT)
            break
    if tables:
        # What bounds this kernel is its HBM stream: 832 fp16 written per row (the node tables and the full-resolution
        # map it reads are L2 / Infinity-Cache resident: FETCH_SIZE x 2 = 0.43 GB per launch).  The canonical work of the
        # layer it replaces (what the reference computes, 2*835*832 FLOP per row) is reported next to
        # it against the MFMA peak; the kernel itself executes 4 table taps + a K = 80 MFMA product per row.
        hh = args.height                                                                  # tables + level-3 map of a pair
        nimg_bytes = 2 * ((hh // 2 + 1) ** 2 + (hh // 2 + 9) ** 2) * 1664.0 + 2 * hh * hh * 128.0
        alg_bytes = rows * 832 * 2.0 + nimg_bytes
        key_flops = 0.0
        if fused:                                   # + kh written (256 B per sample = per 2 rows), the folded key matrix read once
            alg_bytes += rows / 2 * 256.0 + 128 * 1664 * 2.0
            key_flops = 2.0 * (rows / 2) * 128 * 1664
        gbs = alg_bytes / (avg_ms * 1e-3) / 1e9
        achieved = (flops + key_flops) / (avg_ms * 1e-3) / 1e12
        out["roofline"] = {
            "bound": "hbm", "kernel": ("encode_fused_kernel (cpn_encode_key: query_encode_latent 835->832 + ReLU with the gathers fused + the folded "
                                       "key_map 1664->128 layer on the slices of hid in registers)") if fused else
                      "encode_hidden_kernel (query_encode_latent 835->832 + ReLU with the gathers fused)",
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": tsrc, "avg_launch_ms": avg_ms, "launches": len(ms),
            "timing_source": "HIP events on the launch stream inside the timed region of THIS (unprofiled) run; the same "
                             "launches under rocprofv3 --kernel-trace run ~10 % slower (profiles/README.md)",
            "algorithmic_bytes_per_launch": alg_bytes, "kernel_source_sha16": sha,
            **rocprof_view(sha, "encode_fused_kernel" if fused else "encode_hidden_kernel", rows, alg_bytes),
            "canonical_mfma_view": {"bound": "mfma", "achieved": achieved, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": achieved / F16_MFMA_PEAK_TFLOPS, "flops_per_launch": flops + key_flops,
                                    "note": "FLOPs of the layer as the reference formulates it / launch time; the kernel "
                                            "executes 4 table taps + a K=80 MFMA product per row (DESIGN.md §4.1)"},
            "executed_tflops": (2.0 * rows * 832 * (80 + 4) + key_flops) / (avg_ms * 1e-3) / 1e12}
    else:
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_f16_kernel<13> (query_encode_latent 835->832)",
                           "achieved": achieved, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": achieved / F16_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": tsrc,
                           "avg_launch_ms": avg_ms, "launches": len(ms), "flops_per_launch": flops,
                           "kernel_source_sha16": sha}
    return out


def cpu_baseline_block(args, syn, inp_cpu, z_cpu, rel_cpu, flow_cpu, out, B, H, S):
    """The oracle on a bounded sample of the same workload on the host cores.  PyTorch CPU ops stop scaling (and
    regress) far below a 256-thread host, so the thread count is probed first and the best one is used and reported;
    the sample is sized for ~20 s of CPU work."""
    from oracle import render_ref as orc
    w = syn.make_render_weights()

    def cpu_run(n):
        sub = {"context": inp_cpu["context"],
               "query": {k: (v[:, :, :n].contiguous() if k in ("uv", "rgb") else v)
                         for k, v in inp_cpu["query"].items()}}
        with torch.no_grad():
            c0 = time.perf_counter()
            r = orc.forward(sub, z_cpu, rel_cpu, flow_cpu, True, w, npoints=S)
            return r, time.perf_counter() - c0

    ncpu = os.cpu_count() or 1
    best_t, best_rate, probe = 1, 0.0, {}
    for t in sorted({min(ncpu, c) for c in (8, 32, 96)}):
        torch.set_num_threads(t)
        cpu_run(32)                                              # warm-up at this thread count
        _, dt = cpu_run(128)
        probe[str(t)] = 128 / dt
        if 128 / dt > best_rate:
            best_t, best_rate = t, 128 / dt
    torch.set_num_threads(best_t)
    n = int(max(256, min(args.cpu_rays, best_rate * 20.0)))
    ref, cpu_s = cpu_run(n)
    res = {"cpu_baseline": {"value": B * n / cpu_s, "unit": "rays/s", "cores": best_t, "kind": "port",
                            "host_threads": ncpu, "thread_probe_rays_per_s": probe,
                            "sample": f"first {n} rays of each pair of the same {H}x{H}x{S} workload, "
                                      f"oracle/render_ref.py (PyTorch CPU ops), {best_t} of {ncpu} host "
                                      f"threads (best of a 3-point probe), {cpu_s:.1f} s"}}
    err = (out["rgb"][:, :, :n].cpu() - ref["rgb"]).abs()
    mse = float((err ** 2).mean())
    res["parity"] = {"rgb_max_abs_vs_oracle": float(err.max()),
                     "psnr_vs_oracle_db": float(10 * torch.log10(torch.tensor(4.0 / max(mse, 1e-20)))),
                     "pixel_val_bit_identical": bool(torch.equal(out["pixel_val"][:, :n], ref["pixel_val"]))}
    return res


# --------------------------------------------------------------------------------------------------------------
def _spawned(local_rank, argv, world, port):
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes)
    run(parse_args(argv))


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: become the launcher — one process per GPU over RCCL, like
        # /root/reference train.py:141-147 (mp.spawn(multigpu_train, nprocs=opt.gpus))
        if not torch.cuda.is_available() or torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only "
                             f"{torch.cuda.device_count() if torch.cuda.is_available() else 0} HIP device(s) visible")
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        import torch.multiprocessing as mp
        mp.spawn(_spawned, args=(sys.argv[1:], args.gpus, port), nprocs=args.gpus, join=True)
        return
    run(args)


if __name__ == "__main__":
    main()
