"""cpn_encode_key against cpn_encode_hidden + cpn_gemm_f16 on the same inputs: which rows / slices of hid and kh differ."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from coponerf_amd import CoPoNeRF, synthetic as syn
from coponerf_amd._hip import call
from tests.helpers import load_case, case_inputs, to_device

name = sys.argv[1] if len(sys.argv) > 1 else "wide_val"
cfg, _ = load_case(name)
dev = torch.device("cuda:0")
inp, z, rel, flow = case_inputs(cfg)
B, H, R, S, V = cfg["B"], cfg["H"], cfg["R"], cfg["S"], 2
model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
model.load_state_dict(syn.make_render_weights(), strict=False)
model = model.to(dev).eval()
eng = model._engine
inp, z, rel = to_device(inp, dev), to_device(z, dev), rel.to(dev)
w = eng._weights(model._render_params())
maps, tabs = eng._feature_maps(z, w)
ctx, qry = inp["context"], inp["query"]
g = eng._geometry(ctx["cam2world"], ctx["intrinsics"], qry["cam2world"], qry["intrinsics"], qry["uv"], rel, cfg["val"], S, H, H)
n = B * R
rows2 = n * V * S * 2
s = torch.cuda.current_stream().cuda_stream
hid_a = torch.full((rows2, 832), -1.0, dtype=torch.float16, device=dev)
hid_b = torch.full((rows2, 832), -1.0, dtype=torch.float16, device=dev)
kh = torch.full((rows2 // 2, 128), -1.0, dtype=torch.float16, device=dev)
args = (tabs[0].data_ptr(), maps[3].data_ptr(), H, H, g["pixel_val"].data_ptr(), g["sec_grid"].data_ptr(), g["pe6"].data_ptr(),
        w["enc.frag"].data_ptr(), w["query_encode_latent.b"].data_ptr())
call("cpn_encode_hidden", *args, B, V, R, S, 0, n, hid_a.data_ptr(), s)
call("cpn_encode_key", *args, w["enc.k80blk"].data_ptr(), int(os.environ.get("COPONERF_KEY_GROUP", "0")), w["key_fold.w16"].data_ptr(), w["key_fold.b"].data_ptr(), B, V, R, S, 0, n, hid_b.data_ptr(),
     kh.data_ptr(), s)
torch.cuda.synchronize()
kh_ref = torch.empty_like(kh)
call("cpn_gemm_f16", hid_a.data_ptr(), 1664, w["key_fold.w16"].data_ptr(), 1664, w["key_fold.b"].data_ptr(), kh_ref.data_ptr(), 128,
     rows2 // 2, 128, 1664, 1, 0, s)
torch.cuda.synchronize()
print(name, "rows2", rows2, "unwritten hid_a", int((hid_a == -1).sum()), "unwritten hid_b", int((hid_b == -1).sum()))
diff = (hid_a.float() - hid_b.float()).abs()
bad = diff > 0
print("hid mismatching elements", int(bad.sum()), "max", float(diff.max()))
if bad.any():
    rows = bad.any(dim=1).nonzero().flatten()
    print("rows with mismatch", rows.numel(), rows[:40].tolist())
    r0 = int(rows[0])
    cols = bad[r0].nonzero().flatten()
    print("row", r0, "(ray,v,s,j) =", (r0 // (V * S * 2), (r0 // (S * 2)) % V, (r0 // 2) % S, r0 % 2), "cols", cols[:20].tolist(), cols.numel())
    print("a", hid_a[r0, cols[:8]].tolist(), "b", hid_b[r0, cols[:8]].tolist())
    # pattern over (s, j, slice)
    import collections
    pat = collections.Counter()
    for rr in rows[:4000].tolist():
        for sl in bad[rr].view(13, 64).any(dim=1).nonzero().flatten().tolist():
            pat[((rr // 2) % S % 4, rr % 2, sl)] += 1
    print("pattern (s%4, j, slice) -> count:", sorted(pat.items())[:60])
dk = (kh.float() - kh_ref.float()).abs()
print("kh: unwritten", int((kh == -1).sum()), "max diff vs cpn_gemm_f16 on hid_a", float(dk.max()), "mismatch", int((dk > 0).sum()))
