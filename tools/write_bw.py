"""HBM write / read stream ceilings for the hid buffer of one 16 384-ray chunk (4 194 304 rows x 1664 B = 6.98 GB).
`python tools/write_bw.py --build` where hipcc is; `python tools/write_bw.py` on the GPU box."""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_build", "libwrite_bw.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(ROOT, "tools", "write_bw.hip"), "-o", SO])
    sys.exit(0)
import torch
lib = ctypes.CDLL(SO)
lib.wb_fill.argtypes = [ctypes.c_void_p, ctypes.c_longlong] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
lib.wb_read.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
rows = 16384 * 2 * 64 * 2
buf = torch.empty(rows * 1664, dtype=torch.uint8, device=dev)
sink = torch.zeros(1, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
def timed(fn, it=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
flush_buf = torch.zeros(1 << 30, dtype=torch.float32, device=dev)
def timed_flushed(fn, it=8):
    """each launch timed on its own after 8 GB of unrelated traffic (caches / TLBs cold for `buf`)"""
    tot = 0.0
    for _ in range(it):
        flush_buf.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / it
res = {}
gb = rows * 1664 / 1e9
for mode, name in ((0, "linear"), (1, "64B x 16 rows")):
    ms = timed_flushed(lambda: lib.wb_fill(buf.data_ptr(), rows, mode, 256, 512, 64, 2, 16384, s))
    res[f"write {name} grid=256x512 after a cache flush"] = {"ms": round(ms, 3), "TB/s": round(gb / ms, 3)}
ms = timed_flushed(lambda: lib.wb_read(buf.data_ptr(), rows * 104, 1024, sink.data_ptr(), s))
res["read linear grid=1024 after a cache flush"] = {"ms": round(ms, 3), "TB/s": round(gb / ms, 3)}
for mode, name in ((0, "linear"), (1, "64B x 16 rows"), (2, "128B x 8 rows"), (8, "nt linear"), (9, "nt 64B x 16 rows"),
                   (10, "nt 128B x 8 rows")):
    for grid, block in ((256, 512), (1024, 256), (16384, 256)):
        ms = timed(lambda: lib.wb_fill(buf.data_ptr(), rows, mode, grid, block, 64, 2, 16384, s))
        res[f"write {name} grid={grid}x{block}"] = {"ms": round(ms, 3), "TB/s": round(gb / ms, 3)}
for grid in (1024, 4096, 16384):
    ms = timed(lambda: lib.wb_read(buf.data_ptr(), rows * 104, grid, sink.data_ptr(), s))
    res[f"read linear grid={grid}"] = {"ms": round(ms, 3), "TB/s": round(gb / ms, 3)}
ms = timed(lambda: buf.zero_())
res["torch zero_"] = {"ms": round(ms, 3), "TB/s": round(gb / ms, 3)}
print(json.dumps(res, indent=1))

# ---- does a read fill the Infinity Cache (MALL, 256 MB)?  128 MB buffer: hot loop / after a flush / after flush + one read
small = torch.zeros(128 << 18, dtype=torch.float32, device=dev)      # 128 MB
n16s = small.numel() // 4
mall = {}
ms = timed(lambda: lib.wb_read(small.data_ptr(), n16s, 1024, sink.data_ptr(), s), it=20)
mall["128 MB read, hot loop"] = {"ms": round(ms, 4), "TB/s": round(0.134217728 / ms, 2)}
def one(after_flush_read):
    tot = 0.0
    for _ in range(8):
        flush_buf.add_(1)
        if after_flush_read:
            lib.wb_read(small.data_ptr(), n16s, 1024, sink.data_ptr(), s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.wb_read(small.data_ptr(), n16s, 1024, sink.data_ptr(), s); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / 8
ms = one(False); mall["128 MB read after an 8 GB flush"] = {"ms": round(ms, 4), "TB/s": round(0.134217728 / ms, 2)}
ms = one(True); mall["128 MB read after flush + one warming read"] = {"ms": round(ms, 4), "TB/s": round(0.134217728 / ms, 2)}
print(json.dumps(mall, indent=1))
