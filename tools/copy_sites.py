"""Which Python lines of the get_z / render stack issue layout copies in one training step: a TorchDispatchMode records
every aten copy-like op (copy_, clone, _to_copy, cat, contiguous materialisations show up as clone / copy_) with its shape
and the innermost coponerf_amd frame; backward-side ops have no Python frame and are listed by shape only."""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn      # noqa: E402

MIN_ELEMS = int(os.environ.get("MIN_ELEMS", 1 << 16))
BY_COUNT = os.environ.get("BY_COUNT", "0") == "1"
WATCH = {"copy_", "clone", "_to_copy", "cat", "add", "add_", "mul", "sum", "fill_", "zero_", "zeros", "zeros_like", "stack"}


class Rec(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.Counter()
        self.bytes = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in WATCH:
            t = out if torch.is_tensor(out) else (args[0] if args and torch.is_tensor(args[0]) else None)
            if t is not None and t.is_cuda and t.numel() >= MIN_ELEMS:
                frame = "(autograd)"
                for fs in reversed(traceback.extract_stack(limit=40)):
                    if "coponerf_amd" in fs.filename:
                        frame = f"{os.path.basename(fs.filename)}:{fs.lineno} {fs.line.strip()[:70]}"
                        break
                key = (name, tuple(t.shape), frame)
                self.agg[key] += 1
                self.bytes[key] += t.numel() * t.element_size()
        return out


dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(syn.make_full_weights(shapes), strict=True)
model = model.to(dev).train()
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
GETZ = os.environ.get("GETZ", "0") == "1"          # profile one inference get_z at B = 1 instead of a training step
inp = mv(syn.make_inputs(1 if GETZ else 4, 256, 256, 4096, seed=61))
if GETZ:
    model.eval()


def step():
    if GETZ:
        with torch.no_grad():
            model.get_z(inp)
    else:
        model.zero_grad(set_to_none=True)
        (model(inp, val=False)["rgb"] - inp["query"]["rgb"]).abs().mean().backward()


step()
rec = Rec()
with rec:
    step()
tot = sum(rec.bytes.values())
print(f"{sum(rec.agg.values())} watched ops on tensors >= 64 K elements, {tot / 1e9:.2f} GB of outputs")
items = rec.agg.most_common if BY_COUNT else rec.bytes.most_common
for key, _ in items(int(sys.argv[1]) if len(sys.argv) > 1 else 45):
    b = rec.bytes[key]
    name, shape, frame = key
    print(f"{b / 1e6:9.1f} MB x{rec.agg[key]:4d} {name:10s} {str(shape):28s} {frame}")
