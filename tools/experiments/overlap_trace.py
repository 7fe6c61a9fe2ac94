"""Timeline of pipeline.render_images(overlap=...) from a rocprofv3 kernel trace: for every image the encoder's start / end,
the span of the get_z kernels that follow it, and the decoder's end (ms, relative to the first encoder of the excerpt).
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python tools/overlap_trace.py run <mode>
    python tools/overlap_trace.py read <kernel_trace.csv>"""
import csv
import os
import sys

if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from coponerf_amd import CoPoNeRF, synthetic as syn
    from coponerf_amd.pipeline import render_images
    dev = torch.device("cuda:0")
    model = CoPoNeRF.CoPoNeRF(n_view=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes), strict=True)
    model = model.to(dev).eval()
    mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
    pairs = [mv(syn.make_inputs(1, 256, 256, 0, seed=300 + i, full_image=True)) for i in range(4)]
    mode = {"serial": False, "sums": "sums", "all": True}[sys.argv[2]]
    with torch.no_grad():
        for _ in range(2):
            for _ in render_images(model, pairs * 2, overlap=mode):
                pass
            torch.cuda.synchronize()
else:
    rows = []
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    enc = [i for i, r in enumerate(rows) if "encode_key_kernel" in r[2]]
    enc = enc[-6:]
    t0 = rows[enc[0]][0]
    ms = lambda t: round((t - t0) / 1e6, 2)
    for a, b in zip(enc, enc[1:] + [len(rows)]):
        seg = rows[a:b]
        gz = [r for r in seg if any(k in r[2] for k in ("conv4d", "trunk_conv", "soft_argmax", "cva_", "gn_relu", "miopen", "Cijk", "pose_"))]
        lf = [r for r in seg if "lightfield_decode" in r[2]]
        att = [r for r in seg if "attend_hidden" in r[2]]
        if gz and lf:
            done = sum(1 for r in gz if r[1] <= lf[0][1])
            busy = sum(r[1] - r[0] for r in gz if r[1] <= lf[0][1]) / 1e6
            print(f"   get_z kernels finished before the decoder's end: {done} of {len(gz)}, their summed durations {busy:.2f} ms "
                  f"(all: {sum(r[1] - r[0] for r in gz) / 1e6:.2f} ms)")
        print("encoder", ms(rows[a][0]), "->", ms(rows[a][1]), "| get_z kernels", (ms(gz[0][0]), ms(gz[-1][1]), len(gz)) if gz else None,
              "| sums", [(ms(r[0]), ms(r[1])) for r in att], "| decoder end", ms(lf[0][1]) if lf else None)
