import torch, json
from coponerf_amd import CoPoNeRF, synthetic as syn, getz
from coponerf_amd.train_step import TrainStep
dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(syn.make_full_weights(shapes), strict=True)
model = model.to(dev).train()
inp = syn.make_inputs(4, 256, 256, 4096, seed=61)
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
inp = mv(inp)
step = TrainStep(model, lr=1e-4)
gt = inp["query"]["rgb"]
for it in range(6):
    getz.F16_BWD_TRACE = []
    step(inp, gt)
    torch.cuda.synchronize()
    tr = getz.F16_BWD_TRACE
    mx = max(float(t[1]) for t in tr if t[1] is not None), max(float(t[2]) for t in tr if t[2] is not None)
    print(it, len(tr), "max dx16 %.1f max dw16 %.1f" % mx)
for t in tr:
    print(t[0], None if t[1] is None else round(float(t[1]), 2), None if t[2] is None else round(float(t[2]), 2))
