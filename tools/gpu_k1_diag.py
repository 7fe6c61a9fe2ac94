"""Diagnostic: which op of K1 differs between HIP, oracle-on-CPU and the same torch code run on the GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coponerf_amd import synthetic as syn, _hip
from coponerf_amd.render import build_camera_block
from coponerf_amd._hip import call
from oracle import render_ref as orc

dev = torch.device("cuda:0")
B, H, R, S = 1, 64, 512, 32
inp = syn.make_inputs(B, H, H, R, seed=0)
z, rel, flow = syn.make_latents(B, H, H, seed=1)
ctx, qry = inp["context"], inp["query"]
cam, Tq = build_camera_block(ctx["cam2world"], ctx["intrinsics"], qry["cam2world"], qry["intrinsics"], rel, True, H)
N = 2
uv = qry["uv"].expand(-1, 2, -1, -1).reshape(N, R, 2)
Kq = qry["intrinsics"].expand(-1, 2, -1, -1).reshape(N, 4, 4)
Tq_f = Tq.reshape(N, 4, 4)

def mism(tag, a, b):
    a = a.cpu(); b = b.cpu()
    print(f"  {tag:28s} mismatched {float((a != b).float().mean()):.4%}  maxabs {float((a-b).abs().max()):.3e}")

# oracle on CPU and the same torch code on the GPU
d_c, m_c, o_c = orc.plucker_rays(Tq_f, uv, Kq)
d_g, m_g, o_g = orc.plucker_rays(Tq_f.to(dev), uv.to(dev), Kq.to(dev))
mism("torch cpu vs torch gpu: d", d_c, d_g)
mism("torch cpu vs torch gpu: m", m_c, m_g)
# HIP
coords9 = torch.empty(N, R, 9, device=dev); seg = torch.empty(N, R, 4, device=dev); ov = torch.empty(N, R, dtype=torch.uint8, device=dev)
camd = cam.to(dev); uvd = qry["uv"].reshape(B, R, 2).contiguous().to(dev)
call("cpn_project_rays", camd.data_ptr(), uvd.data_ptr(), 2 * R, B, 2, R, coords9.data_ptr(), seg.data_ptr(), ov.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
mism("hip vs torch cpu: d", coords9[..., 0:3], d_c)
mism("hip vs torch gpu: d", coords9[..., 0:3], d_g)
mism("hip vs torch cpu: m", coords9[..., 3:6], m_c)
mism("hip vs torch gpu: m", coords9[..., 3:6], m_g)
mism("hip vs torch cpu: o", coords9[..., 6:9], o_c[:, None].expand(-1, R, -1))
# sub-steps on CPU vs GPU torch
def steps(T, uv, K):
    fx, fy, cx, cy = (K[:, 0, 0, None], K[:, 1, 1, None], K[:, 0, 2, None], K[:, 1, 2, None])
    one = torch.ones_like(uv[..., 0])
    xl = (uv[..., 0] - cx) / fx * one
    yl = (uv[..., 1] - cy) / fy * one
    w = orc._affine_rows(T[:, None], (xl, yl, one, one))
    o = (T[:, None, 0, 3], T[:, None, 1, 3], T[:, None, 2, 3])
    v = (w[0] - o[0], w[1] - o[1], w[2] - o[2])
    n = orc._norm3(v)
    return {"xl": xl, "yl": yl, "w0": w[0], "w2": w[2], "v0": v[0], "sq": (v[0]*v[0]+v[1]*v[1])+v[2]*v[2], "n": n, "d0": v[0] / n}
sc = steps(Tq_f, uv, Kq); sg = steps(Tq_f.to(dev), uv.to(dev), Kq.to(dev))
for k in sc:
    mism("cpu vs gpu torch step " + k, sc[k], sg[k])
# isolated primitive checks
torch.manual_seed(0)
a = torch.rand(1 << 20) + 0.1; b = torch.rand(1 << 20) + 0.1
mism("primitive div", a / b, (a.to(dev) / b.to(dev)))
mism("primitive sqrt", torch.sqrt(a), torch.sqrt(a.to(dev)))
mism("primitive mul-add", a * b + a, a.to(dev) * b.to(dev) + a.to(dev))
mism("primitive div scalar 5", a / 5.0, a.to(dev) / 5.0)
mism("primitive div scalar 63", a / 63, a.to(dev) / 63)
mism("primitive tanh", torch.tanh(a), torch.tanh(a.to(dev)))
