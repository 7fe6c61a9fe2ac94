"""Does get_z (side stream) overlap with a render pass (main stream) on this GPU?  Times: render alone, get_z alone,
both enqueued back to back from one host thread (render first / get_z first, side stream normal / high priority)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coponerf_amd import CoPoNeRF, synthetic as syn
dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(syn.make_full_weights(shapes), strict=True)
model = model.to(dev).eval()
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else type(o)(mv(v) for v in o) if isinstance(o, (list, tuple)) else o)
inp = mv(syn.make_inputs(1, 256, 256, 0, seed=1, full_image=True))
inp2 = mv(syn.make_inputs(1, 256, 256, 0, seed=2, full_image=True))
main = torch.cuda.current_stream()
with torch.no_grad():
    z, rel, flow = model.get_z(inp)
    def t(fn, n=4):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    render = lambda: model(inp, z=z, rel_pose=rel, val=True, flow=flow)
    print("render alone ms", t(render))
    print("get_z alone ms", t(lambda: model.get_z(inp2)))
    for prio in (0, -1):
        side = torch.cuda.Stream(priority=prio)
        def both_r_first():
            side.wait_stream(main)
            render()
            with torch.cuda.stream(side):
                model.get_z(inp2)
            main.wait_stream(side)
        def both_g_first():
            side.wait_stream(main)
            with torch.cuda.stream(side):
                model.get_z(inp2)
            render()
            main.wait_stream(side)
        print(f"priority {prio}: render first then get_z(side): {t(both_r_first):.2f} ms; get_z(side) first then render: {t(both_g_first):.2f} ms")
    # host-side cost of the enqueues
    torch.cuda.synchronize(); t0 = time.perf_counter(); render(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print("host time inside render() ms", (t1 - t0) * 1e3)
    t0 = time.perf_counter(); model.get_z(inp2); t1 = time.perf_counter(); torch.cuda.synchronize()
    print("host time inside get_z() ms", (t1 - t0) * 1e3)
