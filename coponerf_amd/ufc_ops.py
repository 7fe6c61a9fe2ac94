"""HIP bindings of the UFC 4-D operators (the `ops` interface used by coponerf_amd.getz)."""
from __future__ import annotations

import torch

from ._hip import call


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class HipOps:
    """conv4d + GroupNorm + ReLU, cosine correlation and soft-argmax on gfx950 (csrc/ufc.hip)."""

    @staticmethod
    def _need_gpu(t):
        if t.device.type != "cuda":
            raise RuntimeError("coponerf_amd UFC operators run on a HIP device only (tensor on %s)" % t.device)

    def conv4d_gn_relu(self, x, wq, bq, ws, bs, k, s, p, gn_w, gn_b, eps):
        self._need_gpu(x)
        x = x.contiguous().float()
        B, Cin, Hq, Wq, Hs, Ws = x.shape
        Cout = wq.shape[0]
        o = lambda n: (n + 2 * p - k) // s + 1
        Hq2, Wq2, Hs2, Ws2 = o(Hq), o(Wq), o(Hs), o(Ws)
        y = torch.empty(B, Cout, Hq2, Wq2, Hs2, Ws2, device=x.device, dtype=torch.float32)
        stats = torch.zeros(B, 2, device=x.device, dtype=torch.float64)
        f = lambda t: t.detach().contiguous().float()
        wq_, bq_, ws_, bs_, gw, gb = f(wq), f(bq), f(ws), f(bs), f(gn_w), f(gn_b)
        call("cpn_conv4d_gn_relu", x.data_ptr(), wq_.data_ptr(), bq_.data_ptr(), ws_.data_ptr(), bs_.data_ptr(),
             gw.data_ptr(), gb.data_ptr(), float(eps), B, Cin, Cout, Hq, Wq, Hs, Ws, k, s, p, y.data_ptr(),
             stats.data_ptr(), _stream())
        return y

    def correlation_tokens(self, src, trg, fs):
        self._need_gpu(src)
        B, L, C = src.shape
        s_ = src.contiguous().float()
        t_ = trg.contiguous().float()
        out = torch.empty(B, 1, fs, fs, fs, fs, device=src.device, dtype=torch.float32)
        sn, tn = torch.empty_like(s_), torch.empty_like(t_)
        call("cpn_correlation", s_.data_ptr(), t_.data_ptr(), B, L, C, 1e-5, sn.data_ptr(), tn.data_ptr(),
             out.data_ptr(), _stream())
        return out

    def soft_argmax_pair(self, c):
        self._need_gpu(c)
        c = c.contiguous().float()
        B = c.shape[0]
        h = c.shape[-1]
        t_to_s = torch.empty(B, 2, h, h, device=c.device, dtype=torch.float32)
        s_to_t = torch.empty(B, 2, h, h, device=c.device, dtype=torch.float32)
        call("cpn_soft_argmax_pair", c.data_ptr(), B, h, 0.02, t_to_s.data_ptr(), s_to_t.data_ptr(), _stream())
        return t_to_s, s_to_t

    def resize_bilinear(self, x, size):
        """(N,C,h,w) -> (N,C,size,size), bilinear, align_corners=True."""
        self._need_gpu(x)
        x = x.contiguous().float()
        N, C, h, w = x.shape
        y = torch.empty(N, C, size, size, device=x.device, dtype=torch.float32)
        call("cpn_resize_bilinear_ac", x.data_ptr(), y.data_ptr(), N * C, h, w, size, size, _stream())
        return y
