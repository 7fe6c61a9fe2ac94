"""HIP bindings of the UFC 4-D operators (the `ops` interface used by coponerf_amd.getz).

Forward values ALWAYS come from the HIP kernels (csrc/ufc.hip).  When a caller trains (BASELINE config 3) the
operators are wrapped in `_HipForwardVjp`: the backward pass evaluates the vector-Jacobian product of the same
operator written with library ops (MIOpen conv2d / pooling / group_norm, hipBLASLt einsum) at the saved inputs.
That restatement is never used to produce a forward value.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _hip
from ._hip import call


def _stream() -> int:
    return _hip.stream_handle()


class _HipForwardVjp(Function):
    """forward: `hip_fn(*tensors)` (HIP kernels);  backward: VJP of `vjp_fn` (library ops) at the same inputs."""

    @staticmethod
    def forward(ctx, hip_fn, vjp_fn, *tensors):
        out = hip_fn(*tensors)
        ctx.vjp_fn = vjp_fn
        ctx.save_for_backward(*tensors)
        return out

    @staticmethod
    def backward(ctx, *gouts):
        need = ctx.needs_input_grad[2:]
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(n) for t, n in zip(ctx.saved_tensors, need)]
            outs = ctx.vjp_fn(*ins)
            outs = outs if isinstance(outs, tuple) else (outs,)
            wanted = [t for t, n in zip(ins, need) if n]
            pairs = [(o, g) for o, g in zip(outs, gouts) if g is not None]
            got = torch.autograd.grad([o for o, _ in pairs], wanted, [g for _, g in pairs], allow_unused=True)
        it = iter(got)
        return (None, None) + tuple(next(it) if n else None for n in need)


WGRAD_MIN_ROWS = 256        # below this many tokens the library's TN GEMM is as good


def wgrad_f32(dY2, X2, want_bias: bool):
    """dW (O, I) = dY2^T . X2 and db (O) = dY2.sum(0) of a Linear layer: cpn_wgrad_f32 (row slabs on the fp32 MFMA, one
    launch + a fixed-order reduction) where its layout rules hold, the stock products otherwise."""
    R, O = dY2.shape
    I = X2.shape[1]
    # (a handful of rows against a huge layer - the pose regressor's 512 x 134 144 first Linear at batch 4 - is an outer
    # product that is all output: the library's TN GEMM took 310 us for a 275 MB result)
    ok = (dY2.is_cuda and dY2.dtype == torch.float32 and X2.dtype == torch.float32
          and (R >= WGRAD_MIN_ROWS or O * I >= (1 << 24)) and O % 4 == 0 and I % 4 == 0)
    if ok:
        # cpn_wgrad_f32 wants unit column stride, 16-byte aligned rows and a leading dimension that covers a row (a row-expanded
        # gradient has stride(0) == 0, a column slice of a narrower parent stride(0) < O)
        if dY2.stride(1) != 1 or dY2.stride(0) % 4 or dY2.stride(0) < O or dY2.data_ptr() % 16:
            dY2 = dY2.contiguous()
        if X2.stride(1) != 1 or X2.stride(0) % 4 or X2.stride(0) < I or X2.data_ptr() % 16:
            X2 = X2.contiguous()
        dW = torch.empty(O, I, dtype=torch.float32, device=dY2.device)
        db = torch.empty(O, dtype=torch.float32, device=dY2.device) if want_bias else None
        scratch = torch.empty(_hip.lib().cpn_wgrad_f32_scratch_floats(R, O, I), dtype=torch.float32, device=dY2.device)
        call("cpn_wgrad_f32", dY2.data_ptr(), dY2.stride(0), X2.data_ptr(), X2.stride(0), R, O, I, dW.data_ptr(),
             0 if db is None else db.data_ptr(), scratch.data_ptr(), _stream())
        return dW, db
    return dY2.t() @ X2, (dY2.sum(0) if want_bias else None)


class LinearFn(Function):
    """y = x . W^T + b as the library computes it; backward: dx from the library, dW and db from `wgrad_f32`."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if (x.dim() == 2 and x.shape[0] <= 4 and x.shape[1] >= (1 << 15) and x.shape[1] % 4 == 0 and x.is_contiguous()
                and weight.is_contiguous() and x.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0):
            # a few rows against 275 MB of weights: the inference path's chunked GEMV (cpn_pose_gemv), 60 us where the
            # library's GEMM took 160
            nsplit, O = 8, weight.shape[0]
            h = torch.empty(x.shape[0], O, nsplit, dtype=torch.float32, device=x.device)
            call("cpn_pose_gemv", x.data_ptr(), weight.data_ptr(), x.shape[0], x.shape[1], O, nsplit, h.data_ptr(), _stream())
            y = h.sum(-1)
            return y if bias is None else y + bias
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        if not dy.is_contiguous():
            dy = dy.contiguous()          # ONE copy: the product below and wgrad_f32 each made their own (2 x 67 MB at the q/k split)
        dx = dy.matmul(w) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW, db = wgrad_f32(dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1]),
                               ctx.has_bias and ctx.needs_input_grad[2])
        return dx, dW, db


def linear(x, weight, bias=None):
    """torch.nn.functional.linear; under autograd on the GPU with the weight gradient on `cpn_wgrad_f32`."""
    if (x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
            and (x.numel() // x.shape[-1] >= WGRAD_MIN_ROWS or weight.numel() >= (1 << 24))
            and (weight.requires_grad or (bias is not None and bias.requires_grad))):
        return LinearFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)


class Linear(torch.nn.Linear):
    """nn.Linear (same parameters, same checkpoint keys) whose training backward uses `linear` above."""

    def forward(self, x):
        return linear(x, self.weight, self.bias)


class _CorrelationFn(Function):
    """correlation_tokens with a closed-form backward on the normalised tokens the forward kernel already wrote:
    C = sn tn^T, sn = s / (|s| + eps)  =>  dsn = dC tn, dtn = dC^T sn, ds = dsn / (|s| + eps) - sn (sn . dsn) / |s|.
    (The library VJP re-ran the L x L x C forward product first.)"""

    @staticmethod
    def forward(ctx, ops, src, trg, fs):
        out, sn, tn = ops._correlation(src, trg, fs)
        ctx.save_for_backward(src, trg, sn, tn)
        return out

    @staticmethod
    def backward(ctx, gout):
        src, trg, sn, tn = ctx.saved_tensors
        B, L, _ = src.shape
        dC = gout.reshape(B, L, L)

        def through_norm(x, xn, dxn):
            if x.is_cuda and x.shape[-1] <= 1024:
                x_, xn_, d_ = x.contiguous().float(), xn.contiguous(), dxn.contiguous()
                dx = torch.empty_like(d_)
                call("cpn_l2norm_rows_bwd", x_.data_ptr(), xn_.data_ptr(), d_.data_ptr(), x_.numel() // x_.shape[-1], x_.shape[-1],
                     1e-5, dx.data_ptr(), _stream())
                return dx
            r = x.norm(dim=-1, p=2, keepdim=True)
            return dxn / (r + 1e-5) - xn * ((xn * dxn).sum(-1, keepdim=True) / r.clamp_min(1e-30))

        ds = through_norm(src, sn, torch.bmm(dC, tn)) if ctx.needs_input_grad[1] else None
        dt = through_norm(trg, tn, torch.bmm(dC.transpose(1, 2), sn)) if ctx.needs_input_grad[2] else None
        return None, ds, dt, None


def _linear_attention_splits(L):
    return max(1, min(64, L // 64))          # 64-token slabs: the reduce pass is latency-bound per slab


class _LinearAttentionFn(Function):
    """linear_attention with cpn_linear_attention_bwd as its VJP (the library VJP re-ran the forward as ~15 ATen ops
    and differentiated those: 8 ms of the training step)."""

    @staticmethod
    def forward(ctx, ops, q, k, v, channel_major, eps):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        ctx.save_for_backward(q, k, v)
        ctx.cm, ctx.eps = channel_major, eps
        with torch.no_grad():
            return ops.linear_attention(q, k, v, channel_major, eps)

    @staticmethod
    def backward(ctx, gout):
        q, k, v = ctx.saved_tensors
        B, L, H, _ = q.shape
        Dv = v.shape[2] if ctx.cm else v.shape[3]
        nsplit = _linear_attention_splits(L)
        g = gout.contiguous().float()
        scr = torch.empty(_hip.lib().cpn_linear_attention_bwd_scratch(B, L, H, Dv, nsplit), dtype=torch.float32, device=q.device)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        call("cpn_linear_attention_bwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), g.data_ptr(), B, L, H, Dv, int(ctx.cm),
             float(ctx.eps), nsplit, scr.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), _stream())
        return None, dq, dk, dv, None, None


class _CrossAttentionFn(Function):
    """cross_attention with cpn_cross_attention_bwd as its VJP (three launches; the library VJP re-ran both softmaxes and
    einsums as ~25 ATen ops per call)."""

    @staticmethod
    def forward(ctx, ops, c, src_v, trg_v):
        c, src_v, trg_v = c.contiguous(), src_v.contiguous(), trg_v.contiguous()
        with torch.no_grad():
            sa, ta = ops.cross_attention(c, src_v, trg_v)
        ctx.save_for_backward(c, src_v, trg_v, sa, ta)
        return sa, ta

    @staticmethod
    def backward(ctx, g1, g2):
        c, sv, tv, sa, ta = ctx.saved_tensors
        B, H, S, T = c.shape
        g1 = torch.zeros_like(sa) if g1 is None else g1.contiguous().float()
        g2 = torch.zeros_like(ta) if g2 is None else g2.contiguous().float()
        scr = torch.empty(_hip.lib().cpn_cross_attention_bwd_scratch(B, H, S, T), dtype=torch.float32, device=c.device)
        dc, dsv, dtv = torch.empty_like(c), torch.empty_like(sv), torch.empty_like(tv)
        call("cpn_cross_attention_bwd", c.data_ptr(), sv.data_ptr(), tv.data_ptr(), sa.data_ptr(), ta.data_ptr(), g1.data_ptr(),
             g2.data_ptr(), B, H, S, T, sv.shape[-1], scr.data_ptr(), dc.data_ptr(), dsv.data_ptr(), dtv.data_ptr(), _stream())
        return None, dc, dsv, dtv


# strided Conv4d layers: HIP VJP (csrc/ufc_strided_bwd.hip); 0 = autograd through the library max-pool / conv2d graph
STRIDED_HIP_VJP = os.environ.get("CPN_STRIDED_CONV4D_VJP", "1") != "0"


def _wants_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t.requires_grad for t in tensors)


def _pool_pair(x, s, support: bool):
    """MaxPool4d of models/conv4d.py:7-30 over one pair of dims (kernel = stride = s, ceil_mode)."""
    if s == 1:
        return x
    B, C, Hq, Wq, Hs, Ws = x.shape
    if support:
        y = F.max_pool2d(x.reshape(B * C * Hq * Wq, 1, Hs, Ws), s, s, 0, ceil_mode=True)
        return y.reshape(B, C, Hq, Wq, y.shape[-2], y.shape[-1])
    y = F.max_pool2d(x.permute(0, 1, 4, 5, 2, 3).reshape(B * C * Hs * Ws, 1, Hq, Wq), s, s, 0, ceil_mode=True)
    return y.reshape(B, C, Hs, Ws, y.shape[-2], y.shape[-1]).permute(0, 1, 4, 5, 2, 3)


def _conv4d_lib(x, wq, bq, ws, bs, s, p):
    B, Cin = x.shape[:2]
    xq, xs = _pool_pair(x, s, True), _pool_pair(x, s, False)
    Hq, Wq, Hs2, Ws2 = xq.shape[2:]
    yq = F.conv2d(xq.permute(0, 4, 5, 1, 2, 3).reshape(B * Hs2 * Ws2, Cin, Hq, Wq), wq, bq, stride=s, padding=p)
    yq = yq.reshape(B, Hs2, Ws2, -1, yq.shape[-2], yq.shape[-1]).permute(0, 3, 4, 5, 1, 2)
    Hq2, Wq2, Hs, Ws = xs.shape[2:]
    ys = F.conv2d(xs.permute(0, 2, 3, 1, 4, 5).reshape(B * Hq2 * Wq2, Cin, Hs, Ws), ws, bs, stride=s, padding=p)
    ys = ys.reshape(B, Hq2, Wq2, -1, ys.shape[-2], ys.shape[-1]).permute(0, 3, 1, 2, 4, 5)
    return yq + ys


class _Conv4dGnReluFn(Function):
    """forward: cpn_conv4d + cpn_gn_relu, keeping the pre-normalisation volume.
    backward: GroupNorm(1 group)+ReLU on the HIP kernel cpn_gn_relu_bwd (two passes over the volume, sums from the
    forward); the data gradient of a stride-1 Conv4d is a Conv4d with flipped, transposed kernels -> the same HIP conv
    kernel; weight gradients through MIOpen's conv2d weight-gradient on the two separable branches.  Strided layers
    (max-pool routing) take the library VJP of the conv/pool graph."""

    @staticmethod
    def forward(ctx, ops, x, wq, bq, ws, bs, gn_w, gn_b, k, s, p, eps):
        y, out, stats = ops._conv4d_gn_relu_hip(x, wq, bq, ws, bs, k, s, p, gn_w, gn_b, eps, keep_pre=True)
        ctx.save_for_backward(x, wq, bq, ws, bs, gn_w, y, out, stats)
        ctx.cfg = (k, s, p, eps)
        ctx.ops = ops
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wq, bq, ws, bs, gn_w, y, out, stats = ctx.saved_tensors
        k, s, p, eps = ctx.cfg
        B, C = y.shape[:2]
        npos = y[0, 0].numel()
        dev = y.device
        dout = dout.contiguous().float()
        red = ctx.ops._zeros64(B * 2 + C * 2, dev)            # zeroed accumulators from the per-call pool
        dy = torch.empty_like(y)
        dgw = torch.empty(C, dtype=torch.float32, device=dev)
        dgb = torch.empty(C, dtype=torch.float32, device=dev)
        gw = gn_w.detach().contiguous().float()
        call("cpn_gn_relu_bwd", y.data_ptr(), out.data_ptr(), dout.data_ptr(), stats.data_ptr(), gw.data_ptr(), float(eps),
             B, C, npos, red.data_ptr(), dy.data_ptr(), dgw.data_ptr(), dgb.data_ptr(), _stream())
        need_x, need_w = ctx.needs_input_grad[1], any(ctx.needs_input_grad[2:6])
        gx = gwq = gbq = gws = gbs = None
        if k == 3 and s == 1 and p == 1:
            Bx, Cin, Hq, Wq, Hs, Ws = x.shape
            if need_x and Cin % 4 == 0:
                gx = torch.empty_like(x)
                fw = lambda w: w.detach().contiguous().float()            # read in place (transposed, flipped) by the kernel
                wq_c, ws_c = fw(wq), fw(ws)
                call("cpn_conv4d_dgrad", dy.data_ptr(), wq_c.data_ptr(), ws_c.data_ptr(), B, C, Cin, Hq, Wq, Hs, Ws,
                     gx.data_ptr(), _stream())
            elif need_x:
                flip = lambda w: w.detach().float().flip(-1, -2).transpose(0, 1).contiguous()
                zb = torch.zeros(Cin, dtype=torch.float32, device=dev)
                gx = torch.empty_like(x)
                scratch = torch.zeros(int(_hip.lib().cpn_gn_stats_doubles(B, Cin, Hq * Wq * Hs * Ws)), dtype=torch.float64,
                                      device=dev)
                wq_t, ws_t = flip(wq), flip(ws)
                call("cpn_conv4d", dy.data_ptr(), wq_t.data_ptr(), zb.data_ptr(), ws_t.data_ptr(), zb.data_ptr(), B, C, Cin,
                     Hq, Wq, Hs, Ws, 3, 1, 1, gx.data_ptr(), scratch.data_ptr(), 0, _stream())
            if need_w and Cin <= 32 and C <= 32 and Hs * Ws <= 256 and Hq * Wq <= 256 and min(Ws, Wq) >= 4:
                # both separable branches on the HIP weight-gradient kernel; the query branch sees the volumes with
                # the (query, support) index pairs swapped so that its 3x3 window runs over the last two dims too
                gwq = torch.empty(C, Cin, 3, 3, dtype=torch.float32, device=dev)
                gws = torch.empty(C, Cin, 3, 3, dtype=torch.float32, device=dev)
                gbs = torch.empty(C, dtype=torch.float32, device=dev)
                part = torch.empty(_hip.lib().cpn_conv_wgrad_scratch(Cin, C), dtype=torch.float32, device=dev)
                xf = x.detach().contiguous().float()
                call("cpn_conv_wgrad_planes", xf.data_ptr(), dy.data_ptr(), B, Cin, C, Hq * Wq, Hs, Ws, part.data_ptr(),
                     gws.data_ptr(), gbs.data_ptr(), _stream())
                xt = _swap_pairs_hip(xf)
                dt = _swap_pairs_hip(dy)
                call("cpn_conv_wgrad_planes", xt.data_ptr(), dt.data_ptr(), B, Cin, C, Hs * Ws, Hq, Wq, part.data_ptr(),
                     gwq.data_ptr(), 0, _stream())
                gbq = gbs
            elif need_w:
                cb = torch.ops.aten.convolution_backward
                xq = x.permute(0, 4, 5, 1, 2, 3).reshape(B * Hs * Ws, Cin, Hq, Wq)
                dq = dy.permute(0, 4, 5, 1, 2, 3).reshape(B * Hs * Ws, C, Hq, Wq)
                _, gwq, gbq = cb(dq, xq, wq.detach(), [C], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, True])
                xs = x.permute(0, 2, 3, 1, 4, 5).reshape(B * Hq * Wq, Cin, Hs, Ws)
                ds = dy.permute(0, 2, 3, 1, 4, 5).reshape(B * Hq * Wq, C, Hs, Ws)
                _, gws, _ = cb(ds, xs, ws.detach(), [C], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
                gbs = gbq
        elif (s > 1 and STRIDED_HIP_VJP and x.is_cuda and C == 8 and x.shape[1] in (1, 2, 8) and k <= 7 and (need_x or need_w)
              and x.numel() < 2 ** 31):
            # strided layers (max-pool routing): cpn_conv4d_strided_bwd instead of autograd through the library graph
            Bx, Cin, Hq, Wq, Hs, Ws = x.shape
            xf = x.detach().contiguous().float()
            fw = lambda w: w.detach().contiguous().float()
            scr = torch.empty(int(_hip.lib().cpn_conv4d_strided_bwd_scratch(Bx, Cin, C, Hq, Wq, Hs, Ws, k, s, p)),
                              dtype=torch.float32, device=dev)
            if need_x:
                gx = torch.empty_like(xf)
            if need_w:
                gwq = torch.empty(C, Cin, k, k, dtype=torch.float32, device=dev)
                gws = torch.empty_like(gwq)
                gbq = gbs = torch.empty(C, dtype=torch.float32, device=dev)
            ptr = lambda t: 0 if t is None else t.data_ptr()
            call("cpn_conv4d_strided_bwd", xf.data_ptr(), dy.data_ptr(), fw(wq).data_ptr(), fw(ws).data_ptr(), Bx, Cin, C, Hq, Wq,
                 Hs, Ws, k, s, p, scr.data_ptr(), ptr(gx), ptr(gwq), ptr(gws), ptr(gbq), _stream())
        else:
            need = ctx.needs_input_grad[1:6]
            with torch.enable_grad():
                ins = [t.detach().requires_grad_(n) for t, n in zip((x, wq, bq, ws, bs), need)]
                yy = _conv4d_lib(*ins, s, p)
            wanted = [t for t, n in zip(ins, need) if n]
            got = iter(torch.autograd.grad(yy, wanted, dy) if wanted else ())
            gx, gwq, gbq, gws, gbs = (next(got) if n else None for n in need)
        return None, gx, gwq, gbq, gws, gbs, dgw, dgb, None, None, None, None


class _DualSoftmaxFn(Function):
    """f = softmax(a, -1) * softmax(a, -2) on cpn_dual_softmax / cpn_dual_softmax_bwd."""

    @staticmethod
    def forward(ctx, a):
        a = a.contiguous().float()
        B, L, M = a.shape
        dev = a.device
        rstat = torch.empty(B, L, 2, dtype=torch.float32, device=dev)
        cstat = torch.empty(B, M, 2, dtype=torch.float32, device=dev)
        f = torch.empty_like(a)
        call("cpn_dual_softmax", a.data_ptr(), B, L, M, rstat.data_ptr(), cstat.data_ptr(), f.data_ptr(), _stream())
        ctx.save_for_backward(a, rstat, cstat, f)
        return f

    @staticmethod
    def backward(ctx, df):
        a, rstat, cstat, f = ctx.saved_tensors
        B, L, M = a.shape
        df = df.contiguous().float()
        srow = torch.empty(B, L, dtype=torch.float32, device=a.device)
        scol = torch.empty(B, M, dtype=torch.float32, device=a.device)
        da = torch.empty_like(a)
        call("cpn_dual_softmax_bwd", a.data_ptr(), rstat.data_ptr(), cstat.data_ptr(), f.data_ptr(), df.data_ptr(), B, L, M,
             srow.data_ptr(), scol.data_ptr(), da.data_ptr(), _stream())
        return da


class _SoftArgmaxPairFn(Function):
    """cpn_soft_argmax_pair / cpn_soft_argmax_pair_bwd."""

    @staticmethod
    def forward(ctx, ops, c):
        t_to_s, s_to_t = ops._soft_argmax_pair_hip(c)
        ctx.save_for_backward(c, t_to_s, s_to_t)
        return t_to_s, s_to_t

    @staticmethod
    def backward(ctx, g_ts, g_st):
        c, t_to_s, s_to_t = ctx.saved_tensors
        B, h = c.shape[0], c.shape[-1]
        zero = lambda g, like: torch.zeros_like(like) if g is None else g.contiguous().float()
        g_ts, g_st = zero(g_ts, t_to_s), zero(g_st, s_to_t)
        dc = torch.empty_like(c)
        call("cpn_soft_argmax_pair_bwd", c.data_ptr(), B, h, 0.02, t_to_s.data_ptr(), s_to_t.data_ptr(), g_ts.data_ptr(),
             g_st.data_ptr(), dc.data_ptr(), _stream())
        return None, dc


class DwConv3x3Fn(Function):
    """Depthwise 3x3 / stride 1 / pad 1 convolution of the UFC feed-forward blocks.  forward and data gradient are
    library depthwise convolutions; the weight / bias gradient (ten sums per channel) runs on cpn_dwconv3x3_wgrad."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return F.conv2d(x, w, b, 1, 1, 1, x.shape[1])

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, C, H, W = x.shape
        dy = dy.contiguous().float()
        dx = F.conv2d(dy, w.detach().flip(-1, -2), None, 1, 1, 1, C) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            xf = x.detach().contiguous().float()
            dw = torch.empty(C, 1, 3, 3, dtype=torch.float32, device=x.device)
            db = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.has_b else None
            call("cpn_dwconv3x3_wgrad", xf.data_ptr(), dy.data_ptr(), N, C, H, W, dw.data_ptr(),
                 0 if db is None else db.data_ptr(), _stream())
        return dx, dw, db


def _swap_pairs_hip(x):
    """(B,C,a,b,c,d) fp32 -> contiguous (B,C,c,d,a,b) on cpn_transpose_pairs (any layout in: made contiguous first)."""
    xc = x.contiguous().float()
    B, C, a, b, c, d = xc.shape
    y = torch.empty(B, C, c, d, a, b, dtype=torch.float32, device=xc.device)
    if B * C >= 65536:
        return xc.permute(0, 1, 4, 5, 2, 3).contiguous()
    call("cpn_transpose_pairs", xc.data_ptr(), B * C, a * b, c * d, y.data_ptr(), _stream())
    return y


class SwapPairsFn(Function):
    """x.permute(0,1,4,5,2,3).contiguous() of a 4-D correlation volume as one coalesced transpose kernel; the gradient is
    the same swap applied to the incoming gradient."""

    @staticmethod
    def forward(ctx, x):
        return _swap_pairs_hip(x)

    @staticmethod
    def backward(ctx, dy):
        return _swap_pairs_hip(dy)


def swap_pairs(x):
    """Contiguous (B,C,c,d,a,b) from (B,C,a,b,c,d): HIP transpose on the GPU, the library permute elsewhere."""
    if x.is_cuda and x.dim() == 6:
        return SwapPairsFn.apply(x) if torch.is_grad_enabled() and x.requires_grad else _swap_pairs_hip(x)
    return x.permute(0, 1, 4, 5, 2, 3).contiguous()


class DwConvTokensFn(Function):
    """The DWConv of the UFC feed-forward blocks on the token layout (B, L = size*size, C) itself: forward, data
    gradient and weight gradient on cpn_dwconv3x3_tokens / _wgrad — no (B,C,H,W) transposes on either side."""

    @staticmethod
    def forward(ctx, x, w, b, size: int):
        B, L, C = x.shape
        xc = x.detach().contiguous().float()
        wc = w.detach().reshape(C, 9).contiguous().float()
        bc = None if b is None else b.detach().contiguous().float()
        y = torch.empty_like(xc)
        call("cpn_dwconv3x3_tokens", xc.data_ptr(), wc.data_ptr(), 0 if bc is None else bc.data_ptr(), B, size, size, C, 0,
             y.data_ptr(), _stream())
        ctx.save_for_backward(xc, wc)
        ctx.size, ctx.has_b, ctx.wshape = size, b is not None, tuple(w.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        B, L, C = xc.shape
        dyc = dy.contiguous().float()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(xc)
            call("cpn_dwconv3x3_tokens", dyc.data_ptr(), wc.data_ptr(), 0, B, ctx.size, ctx.size, C, 1, dx.data_ptr(), _stream())
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw = torch.empty(C, 9, dtype=torch.float32, device=xc.device)
            db = torch.empty(C, dtype=torch.float32, device=xc.device) if ctx.has_b else None
            part = torch.empty(_hip.lib().cpn_dwconv3x3_tokens_wgrad_scratch(B, ctx.size, C), dtype=torch.float32,
                               device=xc.device)
            call("cpn_dwconv3x3_tokens_wgrad", xc.data_ptr(), dyc.data_ptr(), B, ctx.size, ctx.size, C, part.data_ptr(),
                 dw.data_ptr(), 0 if db is None else db.data_ptr(), _stream())
            dw = dw.view(ctx.wshape)
        return dx, dw, db, None


def _resize_adjoint_hip(g, h: int):
    """Adjoint of resize_bilinear(., n): g (N,C,n,n) -> (N,C,h,h) on cpn_resize_bilinear_ac_adjoint (a deterministic gather;
    the library's upsample backward scatters with atomics)."""
    g = g.contiguous().float()
    N, C, H, W = g.shape
    out = torch.empty(N, C, h, h, dtype=torch.float32, device=g.device)
    call("cpn_resize_bilinear_ac_adjoint", g.data_ptr(), out.data_ptr(), N * C, h, h, H, W, _stream())
    return out


class _ResizeFn(Function):
    """forward: cpn_resize_bilinear_ac.  backward: the adjoint of the (linear) interpolation, no forward re-run."""

    @staticmethod
    def forward(ctx, ops, x, size):
        ctx.in_shape, ctx.size = tuple(x.shape), size
        return ops.resize_bilinear(x, size)

    @staticmethod
    def backward(ctx, dout):
        if ctx.in_shape[-1] == ctx.in_shape[-2]:
            return None, _resize_adjoint_hip(dout, ctx.in_shape[-1]), None
        return None, torch.ops.aten.upsample_bilinear2d_backward(dout.contiguous(), [ctx.size, ctx.size],
                                                                 list(ctx.in_shape), True, None, None), None


class _CorrMean3Fn(Function):
    """forward: cpn_corr_mean3 (sum of the three interpolate4d's / 3 in one pass).  backward: the operator is linear —
    g / 3 for the finest level, and for each coarse level the adjoint of interpolate4d, contracted over the TARGET pair
    of dims first: that pass reads the 64^4 gradient through a contiguous (B, n*n, n, n) view, and everything behind it
    is h*h/n*n of the size (autograd through the composed ops ran the adjoint of the second forward pass first: two
    full-size permute copies and two full-size adds per step)."""

    @staticmethod
    def forward(ctx, ops, c0, c1, c2):
        ctx.hs = (c0.shape[-1], c1.shape[-1])
        return ops._corr_mean3_hip((c0, c1, c2))

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        B, n = g.shape[0], g.shape[-1]
        third = 1.0 / 3.0
        grads = []
        for need, h in zip(ctx.needs_input_grad[1:3], ctx.hs):
            if not need:
                grads.append(None)
                continue
            # (I, J | i, j): contract (i, j) -> (ht, wt), then (I, J) -> (hs, ws)
            t = _resize_adjoint_hip(g.view(B, n * n, n, n), h)                                     # (B, I*J, ht, wt)
            t = t.view(B, n, n, h * h).permute(0, 3, 1, 2)                                         # (B, ht*wt, I, J)
            t = _resize_adjoint_hip(t, h)                                                          # (B, ht*wt, hs, ws)
            grads.append((t.view(B, h, h, h, h).permute(0, 3, 4, 1, 2) * third).reshape(B, 1, h, h, h, h))
        g2 = g * third if ctx.needs_input_grad[3] else None
        return None, grads[0], grads[1], g2


class _ResizeAdjointFn(Function):
    """U^T: the adjoint of the bilinear (align_corners=True) upsampling size -> n, applied to x (N,C,n,n)."""

    @staticmethod
    def forward(ctx, ops, x, size):
        ctx.ops, ctx.n = ops, x.shape[-1]
        return _resize_adjoint_hip(x, size)

    @staticmethod
    def backward(ctx, g):
        return None, ctx.ops.resize_bilinear(g, ctx.n), None


def cost_volume_attention_torch(ops, q, k, v_corr, fs, residual=None, eps=1e-6):
    """HipOps.cost_volume_attention with stock differentiable ops (training), same association as the kernel."""
    import torch.nn.functional as F_
    B, H, Hs, Ws, Ht, Wt = v_corr.shape
    D = q.shape[-1]
    Q, K = F_.elu(q) + 1, F_.elu(k) + 1                                        # (B, L, H, D)
    Z = 1.0 / (torch.einsum("blhd,bhd->blh", Q, K.sum(dim=1)) + eps)
    as_map = lambda t: t.permute(0, 2, 3, 1).reshape(B, H * D, fs, fs)         # tokens (row-major over fs x fs) -> maps
    if fs == Hs:
        Qd, Kd = as_map(Q * Z[..., None]), as_map(K)
    else:
        Qd = ops.resize_bilinear(as_map(Q * Z[..., None]), Hs)                 # D (Z . Q')
        Kd = ops.resize_bilinear_adjoint(as_map(K), Hs)                        # U^T K'
    Qd, Kd = Qd.reshape(B, H, D, Hs * Ws), Kd.reshape(B, H, D, Hs * Ws)
    M = torch.matmul(Kd, v_corr.reshape(B, H, Hs * Ws, Ht * Wt))               # (B, H, D, Dv)
    msg = torch.matmul(Qd.transpose(-1, -2), M).reshape(B, H, Hs, Ws, Ht, Wt)
    return msg if residual is None else residual + msg


def cost_volume_attention_reference_order(ops, q, k, v_corr, fs, residual=None, eps=1e-6):
    """The reference's order (interpolate up, LinearAttention over fs*fs tokens, interpolate down) on the HIP operators: what
    rounds 1-2 ran; kept for A/B timing of the training step (CPN_CVA_REFERENCE_ORDER=1)."""
    B, H, Hs, Ws, Ht, Wt = v_corr.shape
    vc = ops.resize_bilinear(swap_pairs(v_corr).reshape(B, H * Ht * Wt, Hs, Ws), fs)
    msg = ops.linear_attention(q, k, vc.reshape(B, H, Ht * Wt, fs * fs), channel_major=True, eps=eps)
    msg = ops.resize_bilinear(msg.reshape(B, H * Ht * Wt, fs, fs), Hs)
    msg = msg.reshape(B, H, Ht, Wt, Hs, Ws).permute(0, 1, 4, 5, 2, 3)
    return msg if residual is None else residual + msg


_NORM_CONSTS = {}


def _conv_map_lib(rgb, w, b):
    """Library-op statement of cpn_conv_map7x7 (only its VJP is used): CoPoNeRF.py:182-187."""
    x = (rgb.permute(0, 3, 1, 2) + 1) / 2.
    # cached per device: torch.tensor(..., device=) is a pageable host -> device copy, ordered behind everything queued on
    # the stream — inside the backward pass it held the host until the render backward had drained (34 ms per step,
    # tools/host_step_profile.py --ops), and the rest of the backward was enqueued against an empty queue
    c = _NORM_CONSTS.get(x.device)
    if c is None:
        c = _NORM_CONSTS[x.device] = (torch.tensor((0.485, 0.456, 0.406), device=x.device).view(1, 3, 1, 1),
                                      torch.tensor((0.229, 0.224, 0.225), device=x.device).view(1, 3, 1, 1))
    mean, std = c
    return F.conv2d((x - mean) / std, w, b, stride=1, padding=3)


_LINSPACE = {}


class HipOps:
    """conv4d + GroupNorm + ReLU, cosine correlation, soft-argmax (csrc/ufc.hip) and the two attention forms of
    UFCLayer (csrc/ufc_attn.hip) on gfx950."""

    @staticmethod
    def _need_gpu(t):
        if t.device.type != "cuda":
            raise RuntimeError("coponerf_amd UFC operators run on a HIP device only (tensor on %s)" % t.device)

    def _stats(self, B, Cout, npos, device):
        """Zeroed float64 block for one layer's GroupNorm statistics (cpn_gn_stats_doubles: the sums, the arrival counters and
        the per-workgroup partials of the deterministic reduction).  A get_z call runs 63 Conv4d layers: their blocks are
        slices of one zeroed pool per HipOps instance (= per get_z call) instead of 63 fill launches."""
        n = int(_hip.lib().cpn_gn_stats_doubles(B, Cout, npos))
        pool = getattr(self, "_stats_pool", None)
        if pool is None or pool[0].device != device or pool[1] + n > pool[0].numel():
            pool = [torch.zeros(max(n, 1 << 20), device=device, dtype=torch.float64), 0]
            self._stats_pool = pool
        i = pool[1]
        pool[1] = i + n + (n & 1)
        return pool[0][i:i + n]

    def _zeros64(self, n, device):
        """n float64 zeros carved from a pooled, once-zeroed buffer (the backward of every Conv4d layer needs a few
        dozen zeroed accumulators: one fill per 64 KB instead of one per layer)."""
        pool = getattr(self, "_z64_pool", None)
        if pool is None or pool[0].device != device or pool[1] + n > pool[0].numel():
            pool = [torch.zeros(max(8192, n), device=device, dtype=torch.float64), 0]
            self._z64_pool = pool
        i = pool[1]
        pool[1] = i + ((n + 1) // 2) * 2                      # keep 16-byte alignment
        return pool[0][i:i + n]

    def conv4d_gn_relu(self, x, wq, bq, ws, bs, k, s, p, gn_w, gn_b, eps, residual=None):
        """residual: added to the output (x + Encoder4D(x) patterns); fused into the normalisation pass on the inference path."""
        self._need_gpu(x)
        if _wants_grad(x, wq, bq, ws, bs, gn_w, gn_b) or (residual is not None and _wants_grad(residual)):
            y = _Conv4dGnReluFn.apply(self, x.float(), wq, bq, ws, bs, gn_w, gn_b, k, s, p, eps)
            return y if residual is None else residual + y
        if residual is not None and not (residual.is_contiguous() and residual.dtype == torch.float32):
            return residual + self._conv4d_gn_relu_hip(x, wq, bq, ws, bs, k, s, p, gn_w, gn_b, eps)[0]
        return self._conv4d_gn_relu_hip(x, wq, bq, ws, bs, k, s, p, gn_w, gn_b, eps, residual=residual)[0]

    def _conv4d_gn_relu_hip(self, x, wq, bq, ws, bs, k, s, p, gn_w, gn_b, eps, keep_pre=False, residual=None):
        x = x.contiguous().float()
        B, Cin, Hq, Wq, Hs, Ws = x.shape
        Cout = wq.shape[0]
        o = lambda n: (n + 2 * p - k) // s + 1
        Hq2, Wq2, Hs2, Ws2 = o(Hq), o(Wq), o(Hs), o(Ws)
        y = torch.empty(B, Cout, Hq2, Wq2, Hs2, Ws2, device=x.device, dtype=torch.float32)
        stats = self._stats(B, Cout, Hq2 * Wq2 * Hs2 * Ws2, x.device)
        f = lambda t: t.detach().contiguous().float()
        wq_, bq_, ws_, bs_, gw, gb = f(wq), f(bq), f(ws), f(bs), f(gn_w), f(gn_b)
        from . import _hip
        nscr = _hip.lib().cpn_conv4d_scratch(B, Cin, Hq, Wq, Hs, Ws, s)      # pooled volumes of a strided layer
        scr = torch.empty(nscr, dtype=torch.float32, device=x.device) if nscr else None
        scr_p = scr.data_ptr() if scr is not None else 0
        if keep_pre:                                  # training: pre-normalisation volume kept for the backward pass
            call("cpn_conv4d", x.data_ptr(), wq_.data_ptr(), bq_.data_ptr(), ws_.data_ptr(), bs_.data_ptr(), B, Cin, Cout,
                 Hq, Wq, Hs, Ws, k, s, p, y.data_ptr(), stats.data_ptr(), scr_p, _stream())
            out = torch.empty_like(y)
            call("cpn_gn_relu", y.data_ptr(), stats.data_ptr(), gw.data_ptr(), gb.data_ptr(), 0, float(eps), B, Cout,
                 y[0, 0].numel(), out.data_ptr(), _stream())
            return y, out, stats
        if residual is not None and tuple(residual.shape) != tuple(y.shape):
            raise ValueError("conv4d_gn_relu: residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
        call("cpn_conv4d_gn_relu", x.data_ptr(), wq_.data_ptr(), bq_.data_ptr(), ws_.data_ptr(), bs_.data_ptr(),
             gw.data_ptr(), gb.data_ptr(), 0 if residual is None else residual.data_ptr(), float(eps), B, Cin, Cout, Hq, Wq, Hs, Ws, k, s, p, y.data_ptr(),
             stats.data_ptr(), scr_p, _stream())
        return y, stats

    def correlation_tokens(self, src, trg, fs):
        self._need_gpu(src)
        if _wants_grad(src, trg):
            return _CorrelationFn.apply(self, src.float(), trg.float(), fs)
        return self._correlation(src, trg, fs)[0]

    def _correlation(self, src, trg, fs):
        B, L, C = src.shape
        s_ = src.contiguous().float()
        t_ = trg.contiguous().float()
        out = torch.empty(B, 1, fs, fs, fs, fs, device=src.device, dtype=torch.float32)
        n2 = torch.empty(2, B, L, C, device=src.device, dtype=torch.float32)     # adjacent halves: cpn_correlation normalises
        sn, tn = n2[0], n2[1]                                                     # [src ; trg] in one launch when they are too
        call("cpn_correlation", s_.data_ptr(), t_.data_ptr(), B, L, C, 1e-5, sn.data_ptr(), tn.data_ptr(),
             out.data_ptr(), _stream())
        return out, sn, tn

    def conv_map(self, rgb, w, b, want_nhwc16=False):
        """conv_map of get_z (CoPoNeRF.py:69, 182-187) on cpn_conv_map7x7: rgb (N,H,W,3) in [-1,1] as the input dict
        holds it -> (N,64,H,W) fp32 [, (N,H,W,64) fp16 for the render path's gather]."""
        self._need_gpu(rgb)
        if _wants_grad(rgb, w, b):
            return _HipForwardVjp.apply(lambda r, ww, bb: self.conv_map(r, ww, bb)[0], _conv_map_lib,
                                        rgb.float(), w, b), None
        r_, w_, b_ = rgb.contiguous().float(), w.detach().contiguous().float(), b.detach().contiguous().float()
        N, H, W, _ = r_.shape
        out = torch.empty(N, 64, H, W, dtype=torch.float32, device=rgb.device)
        nhwc = torch.empty(N, H, W, 64, dtype=torch.float16, device=rgb.device) if want_nhwc16 else None
        call("cpn_conv_map7x7", r_.data_ptr(), w_.data_ptr(), b_.data_ptr(), N, H, W, out.data_ptr(),
             0 if nhwc is None else nhwc.data_ptr(), _stream())
        return out, nhwc

    def linear_attention(self, q, k, v, channel_major=False, eps=1e-6):
        """aggregation.LinearAttention (models/aggregation.py:84-117) on cpn_linear_attention.
        q, k (B,L,H,32); v / result (B,L,H,Dv) or, channel_major, (B,H,Dv,L)."""
        self._need_gpu(q)
        if _wants_grad(q, k, v):
            return _LinearAttentionFn.apply(self, q.float(), k.float(), v.float(), channel_major, eps)
        q_, k_, v_ = q.contiguous().float(), k.contiguous().float(), v.contiguous().float()
        B, L, H, D = q_.shape
        if D != 32:
            raise ValueError("cpn_linear_attention is built for head dimension 32 (got %d)" % D)
        Dv = v_.shape[2] if channel_major else v_.shape[3]
        nsplit = _linear_attention_splits(L)
        scr = torch.empty(_hip.lib().cpn_linear_attention_scratch(B, H, Dv, nsplit), dtype=torch.float32, device=q.device)
        out = torch.empty_like(v_)
        call("cpn_linear_attention", q_.data_ptr(), k_.data_ptr(), v_.data_ptr(), B, L, H, Dv, int(channel_major), float(eps),
             nsplit, scr.data_ptr(), out.data_ptr(), _stream())
        return out

    def cost_volume_attention(self, q, k, v_corr, fs, residual=None, eps=1e-6):
        """The cost-volume side of UFCLayer.forward_attention (models/aggregation.py:283-297, :301) WITHOUT the fs x fs
        tensors: residual + interp(LinearAttention(q, k, interp(v_corr, fs)), Hs) for v_corr / residual / result
        (B, H, Hs, Ws, Ht, Wt).  Up- and down-sampling are linear over the positions and the attention is linear in the
        values, so  msg = (D diag(Z) phi(q)) . ((U^T phi(k))^T v_low)  — P x 32 matrices per head instead of two
        2 048-channel resizes and a 256-wide attention over fs*fs tokens.  Inference: cpn_cost_volume_attention (two
        launches, fixed-order sums).  Training: the same association with stock differentiable ops."""
        self._need_gpu(q)
        B, H, Hs, Ws, Ht, Wt = v_corr.shape
        if _wants_grad(q, k, v_corr) or (residual is not None and _wants_grad(residual)) or Hs != Ws:
            if os.environ.get("CPN_CVA_REFERENCE_ORDER") == "1":             # A/B: the round-2 sequence on the HIP operators
                return cost_volume_attention_reference_order(self, q, k, v_corr, fs, residual, eps)
            return cost_volume_attention_torch(self, q, k, v_corr, fs, residual, eps)
        q_, k_, v_ = q.contiguous().float(), k.contiguous().float(), v_corr.contiguous().float()
        r_ = None if residual is None else residual.contiguous().float()
        P, Dv = Hs * Ws, Ht * Wt
        scr = torch.empty(_hip.lib().cpn_cost_volume_attention_scratch(B, fs * fs, H, P, Dv), dtype=torch.float32, device=q.device)
        out = torch.empty_like(v_)
        call("cpn_cost_volume_attention", q_.data_ptr(), k_.data_ptr(), v_.data_ptr(), 0 if r_ is None else r_.data_ptr(), B, fs,
             H, Hs, Dv, float(eps), scr.data_ptr(), out.data_ptr(), _stream())
        return out

    def resize_bilinear_adjoint(self, x, size):
        """Adjoint of resize_bilinear(., n) for x (N,C,n,n) -> (N,C,size,size): what its backward computes."""
        return _ResizeAdjointFn.apply(self, x, size)

    def cross_attention(self, c, src_v, trg_v):
        """UFCLayer.forward_cross's two softmax-weighted sums (models/aggregation.py:327-328) on cpn_cross_attention.
        c (B,H,S,T), src_v (B,S,H,32), trg_v (B,T,H,32) -> (B,S,H,32), (B,T,H,32)."""
        self._need_gpu(c)
        if _wants_grad(c, src_v, trg_v):
            return _CrossAttentionFn.apply(self, c.float(), src_v.float(), trg_v.float())
        c_, s_, t_ = c.contiguous().float(), src_v.contiguous().float(), trg_v.contiguous().float()
        B, H, S, T = c_.shape
        src_attn, trg_attn = torch.empty_like(s_), torch.empty_like(t_)
        call("cpn_cross_attention", c_.data_ptr(), s_.data_ptr(), t_.data_ptr(), B, H, S, T, s_.shape[-1],
             src_attn.data_ptr(), trg_attn.data_ptr(), _stream())
        return src_attn, trg_attn

    def soft_argmax_pair(self, c):
        self._need_gpu(c)
        if _wants_grad(c):
            return _SoftArgmaxPairFn.apply(self, c.contiguous().float())
        return self._soft_argmax_pair_hip(c)

    def _soft_argmax_pair_hip(self, c):
        c = c.contiguous().float()
        B = c.shape[0]
        h = c.shape[-1]
        t_to_s = torch.empty(B, 2, h, h, device=c.device, dtype=torch.float32)
        s_to_t = torch.empty(B, 2, h, h, device=c.device, dtype=torch.float32)
        call("cpn_soft_argmax_pair", c.data_ptr(), B, h, 0.02, t_to_s.data_ptr(), s_to_t.data_ptr(), _stream())
        return t_to_s, s_to_t

    def pose_positional(self, intrinsics, H: int, n: int):
        """getz.positional_encodings from the input dict's raw intrinsics (B, V, 4, 4), one launch (inference)."""
        K = intrinsics.detach().contiguous().float()
        self._need_gpu(K)
        key = (n, str(K.device))
        lin = _LINSPACE.get(key)
        if lin is None:
            lin = _LINSPACE[key] = torch.linspace(-1, 1, steps=n, device=K.device)
        B, V = K.shape[0], K.shape[1]
        out = torch.empty(B, n * n, 6, dtype=torch.float32, device=K.device)
        call("cpn_pose_positional", K.data_ptr(), B, V, float(H), lin.data_ptr(), n, out.data_ptr(), _stream())
        return out

    def pose_tail(self, pose_feat, pose_regressor, rotation_regressor, translation_regressor):
        """The pose regressor, the rotation / translation regressors, the 6-D rotation and the 4 x 4 assembly (inference):
        pose_feat (B, (16*16+6)*256*2) -> rel_pose (B, 4, 4).  The first Linear (275 MB of weights for a handful of rows) is
        cpn_pose_gemv's chunk partials at B <= 4 (the library's otherwise), everything behind it ONE launch."""
        import ctypes
        x = pose_feat.detach().contiguous().float()
        self._need_gpu(x)
        lin0 = pose_regressor[0]
        nsplit, bias0 = 0, 0
        if x.shape[0] <= 4 and x.shape[1] % 4 == 0 and lin0.weight.is_contiguous() and lin0.weight.shape[0] == 512:
            nsplit = 8
            h = torch.empty(x.shape[0], 512, nsplit, dtype=torch.float32, device=x.device)
            call("cpn_pose_gemv", x.data_ptr(), lin0.weight.data_ptr(), x.shape[0], x.shape[1], 512, nsplit, h.data_ptr(), _stream())
            bias0 = lin0.bias.data_ptr()
        else:
            h = lin0(x).contiguous()
        lins = [pose_regressor[2], pose_regressor[4]] + [m for m in rotation_regressor if isinstance(m, torch.nn.Linear)] + \
               [m for m in translation_regressor if isinstance(m, torch.nn.Linear)]
        shapes = [tuple(m.weight.shape) for m in lins]
        assert shapes == [(256, 512), (256, 256), (64, 128), (32, 64), (6, 32), (64, 128), (32, 64), (3, 32)], shapes
        ptrs = []
        for m in lins:
            for t in (m.weight, m.bias):
                assert t.is_contiguous() and t.dtype == torch.float32 and t.device == h.device
                ptrs.append(t.data_ptr())
        arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
        out = torch.empty(h.shape[0], 4, 4, dtype=torch.float32, device=h.device)
        call("cpn_pose_tail", h.data_ptr(), nsplit, bias0, arr, h.shape[0], out.data_ptr(), _stream())
        return out

    def dual_softmax(self, a):
        """(B,L,M) -> softmax(a,-1) * softmax(a,-2)."""
        self._need_gpu(a)
        return _DualSoftmaxFn.apply(a)

    def corr_mean3(self, corrs):
        """UFC.forward's final correlation (aggregation.py:549-553): sum(interpolate4d(c, n) for c in corrs) / 3 for the
        three levels' (B,1,h,h,h,h) volumes, on cpn_corr_mean3; under autograd _CorrMean3Fn supplies the adjoint."""
        if _wants_grad(*corrs):
            return _CorrMean3Fn.apply(self, *(c.float() for c in corrs))
        return self._corr_mean3_hip(corrs)

    def _corr_mean3_hip(self, corrs):
        c0, c1, c2 = (c.detach().contiguous().float() for c in corrs)
        self._need_gpu(c2)
        B, n = c2.shape[0], c2.shape[-1]
        out = torch.empty_like(c2)
        call("cpn_corr_mean3", c0.data_ptr(), c0.shape[-1], c1.data_ptr(), c1.shape[-1], c2.data_ptr(), n, B, out.data_ptr(),
             _stream())
        return out

    def resize_bilinear(self, x, size):
        """(N,C,h,w) -> (N,C,size,size), bilinear, align_corners=True."""
        self._need_gpu(x)
        if _wants_grad(x):
            return _ResizeFn.apply(self, x.float(), size)
        x = x.contiguous().float()
        N, C, h, w = x.shape
        y = torch.empty(N, C, size, size, device=x.device, dtype=torch.float32)
        call("cpn_resize_bilinear_ac", x.data_ptr(), y.data_ptr(), N * C, h, w, size, size, _stream())
        return y
