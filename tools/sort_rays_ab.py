"""Round 6: TrainStep with / without the per-step reordering of the query rays by image tile, alternated in ONE process
(boxes differ by more than the effect: 10 steps each way, 4 rounds)."""
import json
import time

import torch

from coponerf_amd import CoPoNeRF, synthetic as syn
from coponerf_amd.train_step import TrainStep


def main():
    dev = torch.device("cuda:0")
    model = CoPoNeRF.CoPoNeRF(n_view=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes), strict=True)
    model = model.to(dev).train()
    inp = syn.make_inputs(4, 256, 256, 4096, seed=61)
    mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
    inp = mv(inp)
    step = TrainStep(model, lr=1e-5)
    gt = inp["query"]["rgb"]
    res = {True: [], False: []}
    for _ in range(3):
        step(inp, gt)
    for rnd in range(4):
        for mode in (True, False):
            step.sort_rays = mode
            for _ in range(2):
                step(inp, gt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                step(inp, gt)
            torch.cuda.synchronize()
            res[mode].append((time.perf_counter() - t0) / 10 * 1e3)
    print(json.dumps({"sorted_by_tile_ms": [round(v, 2) for v in res[True]], "as_given_ms": [round(v, 2) for v in res[False]]}))


if __name__ == "__main__":
    main()
