import os, sys, torch
sys.path.insert(0, os.getcwd())
from coponerf_amd import CoPoNeRF, synthetic as syn, getz
from coponerf_amd.ufc_ops import HipOps
from tests.helpers import to_device
dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(syn.make_full_weights(shapes), strict=True)
model = model.to(dev).eval()
inp = to_device(syn.make_inputs(1, 256, 256, 64, seed=78), dev)
names = ["z0","z1","z2","z3","rel","f0","f1","f2","f3"]
with torch.no_grad():
    runs = []
    for _ in range(3):
        z, rel, flow = model.get_z(inp)
        runs.append([t.clone() for t in z] + [rel.clone()] + [f.clone() for f in flow])
    torch.cuda.synchronize()
    for r in runs[1:]:
        print({n: float((a-b).abs().max()) for n,a,b in zip(names, runs[0], r)})
    # stage by stage: encoder, conv4d op, correlation, linear attention
    x = getz.imagenet_normalise((inp["context"]["rgb"].flatten(0,1).permute(0,3,1,2)+1)/2.)
    e = [model.encoder(x) for _ in range(3)]
    print("encoder", [float((a-b).abs().max()) for a,b in zip(e[0], e[1])])
    ops = HipOps()
    c = torch.randn(1, 1, 16,16,16,16, device=dev)
    emb = model.feature_cost_aggregation.embedding[0]
    o = [emb(c, HipOps()) for _ in range(3)]
    print("encoder4d(embedding[0])", float((o[0]-o[1]).abs().max()), float((o[0]-o[2]).abs().max()))
    c8 = torch.randn(1, 8, 16,16,16,16, device=dev)
    lay = model.feature_cost_aggregation.layers[0][0]
    o = [lay.mlp_corr(c8, HipOps()) for _ in range(3)]
    print("mlp_corr", float((o[0]-o[1]).abs().max()), float((o[0]-o[2]).abs().max()))
    s = torch.randn(1, 256, 256, device=dev); t = torch.randn(1,256,256, device=dev)
    o = [ops.correlation_tokens(s, t, 16) for _ in range(3)]
    print("correlation", float((o[0]-o[1]).abs().max()))
    q = torch.randn(1,256,8,32, device=dev); k = torch.randn(1,256,8,32,device=dev); v = torch.randn(1,256,8,32,device=dev)
    o = [ops.linear_attention(q,k,v) for _ in range(3)]
    print("linattn", float((o[0]-o[1]).abs().max()))
    lin = torch.nn.Linear(2304, 256).to(dev); xx = torch.randn(1, 4096, 2304, device=dev)
    o = [lin(xx) for _ in range(3)]
    print("linear 4096x2304x256", float((o[0]-o[1]).abs().max()))
    lin = torch.nn.Linear(256, 1024).to(dev); xx = torch.randn(1, 256, 256, device=dev)
    o = [lin(xx) for _ in range(3)]
    print("linear 256x256x1024", float((o[0]-o[1]).abs().max()))
    o = [model.feature_cost_aggregation([t.clone() for t in e[0][:3]], 2, HipOps()) for _ in range(2)]
    print("ufc feats", [float((a-b).abs().max()) for a,b in zip(o[0][0], o[1][0])], "c", float((o[0][2]-o[1][2]).abs().max()))
