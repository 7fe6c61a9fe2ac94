"""Where do the launches of one training step come from?  A TorchDispatchMode counts EVERY aten op that produces or
modifies a device tensor, grouped by the innermost coponerf_amd frame (forward) or by op name (backward, no Python
frame), and by op name overall.   python tools/launch_sites.py [top]"""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn      # noqa: E402

VIEW = {"view", "reshape", "_unsafe_view", "permute", "transpose", "t", "expand", "slice", "select", "unsqueeze", "squeeze",
        "detach", "alias", "as_strided", "unbind", "split", "split_with_sizes", "chunk", "narrow", "unflatten", "flatten",
        "_reshape_alias", "view_as", "size", "stride", "is_contiguous", "storage_offset", "numel", "dim", "sym_size",
        "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "lift_fresh", "_local_scalar_dense"}


class Rec(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by_frame = collections.Counter()
        self.by_op = collections.Counter()
        self.bytes_by = collections.Counter()             # (phase, frame, op, shape) -> bytes of the op's result (a time proxy)
        self.count_by = collections.Counter()
        self.phase = "fwd"

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in VIEW:
            return out
        t = out if torch.is_tensor(out) else (out[0] if isinstance(out, (tuple, list)) and out and torch.is_tensor(out[0]) else
                                              (args[0] if args and torch.is_tensor(args[0]) else None))
        if t is None or not t.is_cuda:
            return out
        frame = None
        for fs in reversed(traceback.extract_stack(limit=48)):
            if "coponerf_amd" in fs.filename:
                frame = f"{os.path.basename(fs.filename)}:{fs.lineno} {fs.line.strip()[:60]}"
                break
        if frame is None:
            frame = f"({self.phase}: autograd engine / library) {name}"
        self.by_frame[(self.phase, frame)] += 1
        key = (self.phase, frame, name, tuple(t.shape), t.is_contiguous())
        self.bytes_by[key] += t.numel() * t.element_size()
        self.count_by[key] += 1
        self.by_op[(self.phase, name)] += 1
        return out


dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(syn.make_full_weights(shapes), strict=True)
model = model.to(dev).train()
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
inp = mv(syn.make_inputs(4, 256, 256, 4096, seed=61))


def step(rec=None):
    model.zero_grad(set_to_none=True)
    loss = (model(inp, val=False)["rgb"] - inp["query"]["rgb"]).abs().mean()
    if rec is not None:
        rec.phase = "bwd"
    loss.backward()


step()
rec = Rec()
with rec:
    step(rec)
top = int(sys.argv[1]) if len(sys.argv) > 1 else 60
print("aten ops on device tensors (views excluded):", sum(rec.by_op.values()), "fwd", sum(v for (p, _), v in rec.by_op.items() if p == "fwd"),
      "bwd", sum(v for (p, _), v in rec.by_op.items() if p == "bwd"))
print("---- by op")
for (ph, name), n in rec.by_op.most_common(40):
    print(f"{n:5d} {ph} {name}")
print("---- by site")
for (ph, frame), n in rec.by_frame.most_common(top):
    print(f"{n:5d} {ph} {frame}")
print("---- by result bytes (phase, site, op, shape, contiguous result): elementwise / copy ops only")
ELT = {"copy_", "add", "add_", "mul", "mul_", "clone", "contiguous", "sub", "div", "fill_", "zero_", "cat", "stack", "sum", "neg",
       "where", "clamp_min", "clamp_min_", "relu", "threshold_backward", "gelu", "gelu_backward", "elu", "elu_backward", "zeros",
       "zeros_like", "ones_like", "repeat_interleave", "index_select", "native_layer_norm", "native_layer_norm_backward", "_to_copy"}
rows = [(b, k) for k, b in rec.bytes_by.items() if k[2] in ELT]
rows.sort(reverse=True)
for b, k in rows[:int(os.environ.get("SITES_TOP", "150"))]:
    print(f"{b / 1e6:9.1f} MB x{rec.count_by[k]:3d} {k[0]} {k[2]:14s} {str(k[3]):28s} {'c' if k[4] else 'nc'}  {k[1]}")
