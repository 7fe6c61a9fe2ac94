"""One kernel form per invocation, a few launches, for tools/pmc_passes.sh (TCP_TOTAL_CACHE_ACCESSES, TA_BUSY, ... per dispatch):
the accesses that round 4 turned from MFMA-operand-layout reads of row-major memory (64 L1 tag look-ups per instruction) into
whole-line ones.   python tools/tag_probe.py decoder_before|decoder_after|rowdot_rowmajor|rowdot_frag|vfold_tiled|vfold_fewrows"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coponerf_amd import _hip                 # noqa: E402
from coponerf_amd._hip import call            # noqa: E402

what = sys.argv[1]
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
rnd = lambda *shape: torch.randn(*shape, generator=g)
if what.startswith("decoder"):
    M = 65536
    P, I = ctypes.c_void_p, ctypes.c_int
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "librayout_before.so")) if what == "decoder_before" else _hip.lib()
    lf = lib.cpn_lightfield_decode
    lf.argtypes = [P, P, P, P, I, I, I, P, P, P, P]
    pack = (rnd(_hip.LIGHTFIELD_PACK_FLOATS) * 0.05).to(dev)
    coords9, zl = rnd(2, M, 9).to(dev), rnd(M, 416).to(dev)
    over = torch.ones(2, M, dtype=torch.uint8, device=dev)
    rgb, valid = torch.empty(1, 1, M, 3, device=dev), torch.empty(1, M, 1, device=dev)
    for _ in range(5):
        assert lf(coords9.data_ptr(), zl.data_ptr(), pack.data_ptr(), over.data_ptr(), 1, 2, M, rgb.data_ptr(), valid.data_ptr(), 0, s) == 0
elif what.startswith("rowdot"):
    rows = 65536 * 128
    A, Q = (rnd(rows, 128) * 0.5).half().to(dev), (rnd(rows, 128) * 0.5).half().to(dev)
    W, b = (rnd(128, 128) * 0.05).half().to(dev), rnd(128).to(dev)
    lg = torch.empty(rows, device=dev)
    for _ in range(3):
        call("cpn_gemm_f16_rowdot", A.data_ptr(), 128, W.data_ptr(), 128, b.data_ptr(), Q.data_ptr(), 0 if what == "rowdot_frag" else 128,
             lg.data_ptr(), rows, 128, 128, s)
else:
    M, N, K = 3641, 416, 1664
    A, W, b = (rnd(M, K) * 0.5).half().to(dev), (rnd(N, K) * 0.05).half().to(dev), rnd(N).to(dev)
    Wp = torch.empty(N * K, dtype=torch.float16, device=dev)
    call("cpn_pack_gemm_frags", W.data_ptr(), K, N, K, Wp.data_ptr(), s)
    C = torch.empty(M, N, device=dev)
    for _ in range(10):
        if what == "vfold_fewrows":
            call("cpn_gemm_f16_fewrows", A.data_ptr(), K, Wp.data_ptr(), b.data_ptr(), C.data_ptr(), N, M, N, K, 0, s)
        else:
            call("cpn_gemm_f16", A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), C.data_ptr(), N, M, N, K, 0, 1, s)
torch.cuda.synchronize()
