"""What a NEW stereo pair costs on top of a cached one (bench.py rays_per_s_fresh_pair vs value): GPU time of the per-pair
device work (NHWC copies, node features, table projection, flow products) and host time of the pose algebra + upload."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn      # noqa: E402
from coponerf_amd.aux_outputs import flow_products        # noqa: E402

dev = torch.device("cuda:0")
H = S = None
H, S = 256, 64
model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
model.load_state_dict(syn.make_render_weights(), strict=False)
model = model.to(dev).eval()
eng = model._engine
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else type(o)(mv(v) for v in o))
pairs = []
for j in range(6):
    ic = mv(syn.make_inputs(1, H, H, 0, seed=500 + j, full_image=True))
    zc, rc, fc = syn.make_latents(1, H, H, seed=600 + j)
    pairs.append((ic, mv(zc), rc.to(dev), mv(fc)))
w = eng._weights(model._render_params())


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


res = {"feature_maps_gpu_ms": [], "flow_products_gpu_ms": [], "camera_host_ms": [], "feature_maps_host_ms": [], "call_ms": []}
with torch.no_grad():
    for i, (ic, zc, rc, fc) in enumerate(pairs):
        torch.cuda.synchronize()
        h0 = time.perf_counter(); e0 = ev()
        eng._feature_maps(zc, w)
        e1 = ev(); h1 = time.perf_counter()
        flow_products(fc, H)
        e2 = ev()
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        ctx, qry = ic["context"], ic["query"]
        eng._camera(ctx["cam2world"], ctx["intrinsics"], qry["cam2world"], qry["intrinsics"], rc, True, H, dev)
        torch.cuda.synchronize()
        c1 = time.perf_counter()
        if i >= 2:
            res["feature_maps_gpu_ms"].append(e0.elapsed_time(e1)); res["flow_products_gpu_ms"].append(e1.elapsed_time(e2))
            res["camera_host_ms"].append((c1 - c0) * 1e3); res["feature_maps_host_ms"].append((h1 - h0) * 1e3)
    # whole calls: cached pair vs new pair
    for name, seq in (("cached", [pairs[0]] * 6), ("fresh", pairs)):
        for a in seq[:2]:
            model(a[0], z=a[1], rel_pose=a[2], val=True, flow=a[3])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a in seq[2:]:
            model(a[0], z=a[1], rel_pose=a[2], val=True, flow=a[3])
        torch.cuda.synchronize()
        res["call_ms_" + name] = (time.perf_counter() - t0) / 4 * 1e3
print({k: (round(sum(v) / len(v), 3) if isinstance(v, list) and v else v) for k, v in res.items()})
