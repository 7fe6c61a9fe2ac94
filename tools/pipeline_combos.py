"""Per-image time of coponerf_amd.pipeline.render_images under combinations of its options (CU partition, batched get_z, graph).
Usage: [NREP=4] python tools/pipeline_combos.py [indices, e.g. 4,0,4]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from coponerf_amd import CoPoNeRF, synthetic as syn
from coponerf_amd.pipeline import render_images
dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(syn.make_full_weights(shapes), strict=True)
model = model.to(dev).eval()
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
pairs = [mv(syn.make_inputs(1, 256, 256, 0, seed=300 + i, full_image=True)) for i in range(4)]
seq = pairs * int(os.environ.get("NREP", "4"))
def t(**kw):
    with torch.no_grad():
        for _ in render_images(model, seq[:4], **kw): pass
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 0
        for _ in render_images(model, seq, **kw): n += 1
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
ALL = ({}, {"overlap": "sums"}, {"overlap": True}, {"overlap": "sums", "graph": True}, {"graph": True}, {"getz_batch": 4}, {"getz_batch": 4, "cu_split": (224, 32)}, {"getz_batch": 4, "cu_split": (192, 64)}, {"cu_split": (192, 64)}, {"cu_split": (192, 64), "graph": True}, {"getz_batch": 4, "cu_split": (192, 64), "graph": True})
for i in ([int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else range(len(ALL))):
    print(ALL[i], round(t(**ALL[i]), 2), round(t(**ALL[i]), 2), flush=True)
