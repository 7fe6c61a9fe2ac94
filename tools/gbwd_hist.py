"""How evenly the rows of a training step land on the accumulator tiles of cpn_gather_rows_bwd: per level, rows per (image, 8 x 4
pixel tile) from the step's sample coordinates — mean, maximum, and where the heavy tiles are."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coponerf_amd import synthetic as syn            # noqa: E402
from coponerf_amd.render import RenderEngine         # noqa: E402

B, R, S, H = 4, 4096, 64, 256
dev = torch.device("cuda:0")
inp = syn.make_inputs(B, H, H, R, seed=61)
eng = RenderEngine()
c, q = inp["context"], inp["query"]
g = eng._geometry(c["cam2world"].to(dev), c["intrinsics"].to(dev), q["cam2world"].to(dev), q["intrinsics"].to(dev),
                  q["uv"].to(dev), None, False, S, H, H)
TP, TPY = 8, 4
for lvl in range(4):
    shift = 4 - lvl - (lvl == 3)
    Wl = H >> shift
    tx, ty = (Wl + TP - 1) // TP, (Wl + TPY - 1) // TPY
    tot = torch.zeros(2 * B, ty, tx, device=dev)
    for j, grid in ((0, g["pixel_val"]), (1, g["sec_grid"])):
        # image that the rows read: own view for pixel_val, the other view for sec_grid
        gx, gy = grid[..., 0], grid[..., 1]                                  # (B*V, R, S)
        x = ((gx + 1) * Wl - 1) / 2
        y = ((gy + 1) * Wl - 1) / 2
        lo, hi = (0.0, Wl - 1.0) if j == 0 else (-2.0, Wl + 1.0)
        x, y = x.clamp(lo, hi), y.clamp(lo, hi)
        x0, y0 = x.floor().long(), y.floor().long()
        ok = (x0 >= -1) & (x0 < Wl) & (y0 >= -1) & (y0 < Wl)
        img = torch.arange(2 * B, device=dev).view(-1, 1, 1).expand_as(x0)
        if j == 1:
            img = img ^ 1
        t = (img * ty + (y0.clamp(0, Wl - 1) // TPY)) * tx + (x0.clamp(0, Wl - 1) // TP)
        tot.view(-1).index_add_(0, t[ok].reshape(-1), torch.ones_like(t[ok], dtype=torch.float32).reshape(-1))
    mean, mx = float(tot.mean()), float(tot.max())
    flat = tot.view(2 * B, -1)
    top = torch.topk(flat[0], min(6, flat.shape[1]))
    where = [(int(i) // tx, int(i) % tx, int(v)) for v, i in zip(top.values, top.indices)]
    print(f"level {lvl}: {Wl}x{Wl} px, {ty}x{tx} tiles/image: rows per tile mean {mean:9.0f} max {mx:9.0f} (x{mx / mean:5.1f}); "
          f"image 0 top (tile_y, tile_x, rows): {where}")
