"""Autograd wrappers for the training path (BASELINE config 3; caller: /root/reference wrapper.py:107,138).

Every forward below calls the SAME HIP kernel the inference path uses (one forward path); only the backward is new:
  * plain GEMM gradients (dX = dY.W, dW = dY^T.X) are library GEMMs -> hipBLASLt via torch.matmul,
  * the two fused non-GEMM stages have their own kernels (csrc/backward.hip): the joint-softmax / weighted hidden
    sum and the bilinear gather (scatter-add into the feature maps),
  * ReLU masks, bias sums and the per-ray fp32 layers are a few elementwise / small-matmul torch ops.
No coordinate gradients exist on this path: sample coordinates derive from poses only and the 3-D points are
detached in the reference (models/CoPoNeRF.py:244, 380-381, 433), so geometry runs outside autograd.
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function

from . import _hip
from ._hip import call


def _stream() -> int:
    return _hip.stream_handle()


class GradScale:
    """Power-of-two scale carried by every fp16 activation gradient of ONE backward pass.

    The reference trains in fp32 (no GradScaler, /root/reference wrapper.py:107-151) and its loss is an L1 `.mean()`
    over B*R*3 ~ 5e4 elements, so dL/drgb ~ 2e-5 — already below fp16's smallest normal (6.1e-5) — and the gradient of
    the per-sample hidden activations (softmax weight ~1/128 times that) would flush to zero if stored as plain
    fp16.  Convention on this path: a gradient tensor of dtype fp16 holds `s * true gradient`, one of dtype fp32 holds
    the true gradient.  `s` is fixed by the first backward function that turns an fp32 gradient into an fp16 one
    (2^k such that its largest entry lands near `target`), lives on the device (no host sync), and is divided out
    wherever an fp16 gradient is reduced into an fp32 one (weight / bias / feature-map gradients).  An overflow shows
    up as a non-finite parameter gradient and makes the step's guard skip the update, like a lost AMP step."""

    def __init__(self, target: float = 256.0):
        self.target = float(target)
        self.s = None

    def ensure(self, d32: torch.Tensor) -> torch.Tensor:
        if self.s is None:
            amax = d32.detach().abs().amax().float().clamp_min(1e-30)
            self.s = torch.exp2(torch.floor(torch.log2(self.target / amax))).clamp(2.0 ** -24, 2.0 ** 60)
        return self.s

    def scaled16(self, d: torch.Tensor) -> torch.Tensor:
        """Incoming gradient -> the scaled representation (fp16 tensors already carry the scale)."""
        return d if d.dtype == torch.float16 else d * self.ensure(d)


# cpn_hid_grad_combine as the epilogue of the key path's data-gradient GEMM (cpn_gemm_f16_combine; round 6); 0 = two kernels
FUSE_COMBINE = os.environ.get("COPONERF_FUSE_COMBINE", "1") != "0"


class HidGradParts:
    """Gradient contributions to the hidden activations `hid` that are rank one per ray (w (x) dhbar, from the two
    attention-weighted hidden sums).  Their backward functions park them here instead of materialising a 7 GB tensor
    each; the backward of the layer that PRODUCED hid (autograd runs it after all of hid's consumers) combines them with
    the key path's gradient and the ReLU mask in one kernel (cpn_hid_grad_combine)."""

    def __init__(self):
        self.parts = []                     # (w (N,R,S) fp32, dhbar (rays,1664) fp32 scaled)
        self.combined = False               # the key path's data-gradient GEMM already formed the masked sum (FUSE_COMBINE)
        self.dqb = None                     # gradient of the shared qb (coords_embed) parked by the round that runs first
        self.dqb_done = False


# coords_embed feeds both attention rounds: the second backward call adds the first one's gradient in its kernel
SHARE_QB_GRAD = os.environ.get("COPONERF_SHARE_QB_GRAD", "1") != "0"
# grad_input of key_map_2 with the ReLU mask of its input as the GEMM's epilogue (cpn_gemm_f16_masked)
MASKED_DGRAD = os.environ.get("COPONERF_MASKED_DGRAD", "1") != "0"


class KeyForward:
    """Hand-over between EncodeFn and the GemmFn node of the folded key layer (kh = ReLU(W' . [hid_own ; hid_other] + c')): the
    first layer's kernel also forms kh (cpn_encode_key, as in inference: hid is not read back for it), and the GemmFn node only
    records what its backward needs.  W (128, 1664) / b (128): the folded, differentiable fp32 weights (render_train)."""

    def __init__(self, W, b):
        self.W, self.b = W, b
        self.W16 = None                      # (128, 1664) fp16 image of W, packed once for both nodes
        self.kh = None                       # (rows, 128) fp16, set by EncodeFn.forward, taken by GemmFn.forward


def _mm_f32(a16: torch.Tensor, b16: torch.Tensor) -> torch.Tensor:
    """(P,M) fp16 @ (M,Q) fp16 -> fp32 with fp32 accumulation AND fp32 output (hipBLASLt): M is millions of rows, an
    fp16 result would overflow / lose the tail of the sum."""
    return torch.mm(a16, b16, out_dtype=torch.float32)


def _colsum_f32(d: torch.Tensor) -> torch.Tensor:
    """d.sum(0) in fp32.  Over millions of 128-wide rows the library's column reduction runs at 1.8 TB/s (292 us for the key
    layer's 537 MB); the same rows viewed 64 at a time are 8 192 columns wide and it streams them at 5 TB/s (103 us)."""
    n, c = d.shape
    if d.is_contiguous() and n % 64 == 0 and n >= (1 << 16):
        return d.view(n // 64, 64 * c).sum(0, dtype=torch.float32).view(64, c).sum(0)
    return d.sum(0, dtype=torch.float32)


def _wgrad_tall(d16: torch.Tensor, x16: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """dW (N, K) fp32 = d16^T (N, M) . x16 (M, K) / scale on cpn_wgrad_tall_f16 (N % 208 == 0, K % 128 == 0; the library's
    split-K GEMM runs the 832 x 896 x 4.2 M case at 500 TFLOP/s, 12.5 ms of the training step)."""
    M, N = d16.shape
    K = x16.shape[1]
    part = torch.empty(_hip.lib().cpn_wgrad_tall_scratch(N, K), dtype=torch.float32, device=d16.device)
    dW = torch.empty(N, K, dtype=torch.float32, device=d16.device)
    call("cpn_wgrad_tall_f16", d16.data_ptr(), N, x16.data_ptr(), K, M, N, K, scale.reshape(1).data_ptr(), part.data_ptr(),
         dW.data_ptr(), _stream())
    return dW


def _data_grad(d16: torch.Tensor, W16: torch.Tensor, mask: torch.Tensor = None) -> torch.Tensor:
    """dA (M, lda) = d16 (M, N) . W16 (N, lda), fp16 in / fp32 accumulate / fp16 out.  It is the forward GEMM with the
    roles of N and K swapped, so it runs on cpn_gemm_f16 with the transposed weight image (hipBLASLt reaches 355 TFLOP/s
    on the 4.2 M x 832 x 896 case, the own kernel ~800); shapes outside the kernel's tile set go to the library."""
    M, N = d16.shape
    lda = W16.shape[1]
    if N % 32 or (lda % 208 and lda % 128) or not d16.is_contiguous():
        dA = torch.matmul(d16, W16)
        return dA if mask is None else torch.ops.aten.threshold_backward(dA, mask, 0)
    Wt = W16.t().contiguous()                                                      # (lda, N): K-contiguous rows
    dA = torch.empty(M, lda, dtype=torch.float16, device=d16.device)
    if mask is not None and M % 16 == 0 and mask.is_contiguous() and mask.shape == dA.shape:
        call("cpn_gemm_f16_masked", d16.data_ptr(), N, Wt.data_ptr(), N, mask.data_ptr(), lda, dA.data_ptr(), lda, M, lda, N,
             _stream())
        return dA
    zero = torch.zeros(lda, dtype=torch.float32, device=d16.device)
    call("cpn_gemm_f16", d16.data_ptr(), N, Wt.data_ptr(), N, zero.data_ptr(), dA.data_ptr(), lda, M, lda, N, 0, 0,
         _stream())
    return dA if mask is None else torch.ops.aten.threshold_backward(dA, mask, 0)


class GemmFn(Function):
    """C = act(A . W^T + b) through cpn_gemm_f16.  A (M, lda) fp16 (row stride lda >= K), W (N, K) fp32, b (N)."""

    @staticmethod
    def forward(ctx, A16, W, b, relu: bool, out_f32: bool, gs: GradScale, hid_parts=None, dims=None, in_parts=None, pre=None,
                in_relu: bool = False, premasked: bool = False):
        """in_relu: A16 is a ReLU output whose producer was told `premasked` - this node's backward returns the data gradient
        already masked by A16 > 0 (cpn_gemm_f16_masked), the producer skips its threshold pass."""
        M, lda = A16.shape
        N, K = W.shape
        Kp = ((K + 31) // 32) * 32
        assert Kp <= lda, (K, lda)
        if pre is not None and pre.kh is not None:
            # the product already exists (cpn_encode_key formed it with the first layer, same MFMA sequence): record only
            assert relu and not out_f32 and pre.kh.shape == (M, N) and pre.W16.shape == (N, lda)
            W16, C = pre.W16, pre.kh
            pre.kh = pre.W16 = None
        else:
            W16 = torch.zeros(N, lda, dtype=torch.float16, device=A16.device)
            Wc = W.detach().contiguous().float()
            call("cpn_pack_weight_f16", Wc.data_ptr(), N, K, W16.data_ptr(), lda, _stream())
            bc = b.detach().contiguous().float()
            C = torch.empty(M, N, dtype=torch.float32 if out_f32 else torch.float16, device=A16.device)
            call("cpn_gemm_f16", A16.data_ptr(), lda, W16.data_ptr(), lda, bc.data_ptr(), C.data_ptr(), N, M, N, Kp,
                 int(relu), int(out_f32), _stream())
        ctx.save_for_backward(A16, W16, C if relu else None)
        ctx.relu, ctx.K, ctx.gs = relu, K, gs
        ctx.hid_parts, ctx.dims, ctx.in_parts = hid_parts, dims, in_parts
        ctx.in_relu, ctx.premasked = in_relu and MASKED_DGRAD, premasked and MASKED_DGRAD
        return C

    @staticmethod
    def _combined_data_grad(ctx, d16, A16, W16):
        """A16 = hid (rows, 1664): dA and the gradients parked by the two attention sums, masked, in one kernel — this layer
        is the LAST consumer of hid that autograd runs (its gradient depends on both sums' backward passes)."""
        hp = ctx.in_parts
        B, V, R, S = ctx.dims
        K = d16.shape[1]
        if not (FUSE_COMBINE and hp is not None and hp.parts and A16.shape[1] == 1664 and S % 16 == 0 and K % 32 == 0
                and d16.is_contiguous() and A16.is_contiguous() and ctx.needs_input_grad[0]):
            return None
        parts = hp.parts
        (w1, dh1), (w2, dh2) = parts[0], (parts[1] if len(parts) > 1 else (None, None))
        Wt = W16.t().contiguous()                                              # (1664, K)
        out = torch.empty_like(A16)
        call("cpn_gemm_f16_combine", d16.data_ptr(), K, Wt.data_ptr(), K, A16.data_ptr(), w1.data_ptr(), dh1.data_ptr(),
             0 if w2 is None else w2.data_ptr(), 0 if dh2 is None else dh2.data_ptr(), B, V, R, S, 0, B * R, K, out.data_ptr(),
             _stream())
        hp.parts = []
        hp.combined = True
        return out

    @staticmethod
    def backward(ctx, dC):
        A16, W16, C = ctx.saved_tensors
        if ctx.hid_parts is not None and ctx.hid_parts.parts:
            # C = hid: key-path gradient (dC) + the parked rank-one contributions + ReLU mask in ONE pass
            B, V, R, S = ctx.dims
            d = ctx.gs.scaled16(dC.contiguous()).to(torch.float16)
            parts = ctx.hid_parts.parts
            (w1, dh1), (w2, dh2) = parts[0], (parts[1] if len(parts) > 1 else (None, None))
            d16 = torch.empty_like(C)
            call("cpn_hid_grad_combine", d.data_ptr(), C.data_ptr(), w1.data_ptr(), dh1.data_ptr(),
                 0 if w2 is None else w2.data_ptr(), 0 if dh2 is None else dh2.data_ptr(), B, V, R, S, 0, B * R,
                 d16.data_ptr(), _stream())
            ctx.hid_parts.parts = []
            ctx.hid_parts.combined = False
            d = d16
            inv = 1.0 / ctx.gs.s
        else:
            if ctx.hid_parts is not None:
                ctx.hid_parts.combined = False
            d = ctx.gs.scaled16(dC.contiguous())                                 # s * dC (GradScale convention)
            inv = 1.0 / ctx.gs.s
            if ctx.relu and not (ctx.premasked and d.dtype == torch.float16):
                d = torch.ops.aten.threshold_backward(d, C.to(d.dtype) if C.dtype != d.dtype else C, 0)   # d where C > 0
            d16 = d.to(torch.float16)
        dA = GemmFn._combined_data_grad(ctx, d16, A16, W16) if ctx.in_parts is not None else None
        if dA is None and ctx.in_relu and ctx.needs_input_grad[0]:
            dA = _data_grad(d16, W16, mask=A16)
        if dA is None:
            dA = _data_grad(d16, W16) if ctx.needs_input_grad[0] else None           # (M, lda) fp16, scaled; pad columns get 0
        if d16.shape[1] == 128 and A16.shape[1] == 128 and d16.is_contiguous() and ctx.needs_input_grad[1]:
            # 128 x 128 outputs over millions of rows: a streaming reduction, not a GEMM the library handles well
            dW = torch.zeros(128, 128, dtype=torch.float32, device=d16.device)
            db = torch.zeros(128, dtype=torch.float32, device=d16.device)
            call("cpn_wgrad_skinny_f16", d16.data_ptr(), A16.data_ptr(), 128, d16.shape[0], dW.data_ptr(), db.data_ptr(),
                 _stream())
            return (dA, dW[:, :ctx.K] * inv, (db * inv if ctx.needs_input_grad[2] else None)) + (None,) * 9
        n_out, k_in = d16.shape[1], A16.shape[1]
        both = d16.is_contiguous() and A16.is_contiguous()
        if ctx.needs_input_grad[1] and both and n_out % 208 == 0 and k_in % 128 == 0:
            dW = _wgrad_tall(d16, A16, ctx.gs.s)[:, :ctx.K]              # the 832 x 896 first layer in its gather form
        elif ctx.needs_input_grad[1] and both and k_in % 208 == 0 and n_out % 128 == 0:
            # 128 x 1664 (key map over the paired hidden rows): the same kernel on the transposed problem, dW^T = A^T . d
            dW = _wgrad_tall(A16, d16, ctx.gs.s).t()[:, :ctx.K]
        else:
            dW = _mm_f32(d16.t(), A16)[:, :ctx.K] * inv if ctx.needs_input_grad[1] else None
        db = _colsum_f32(d) * inv if ctx.needs_input_grad[2] else None
        return (dA, dW, db) + (None,) * 9


class LinearF32Fn(Function):
    """Y = act_out(act_in(X) . W^T + b + res) through cpn_linear_f32 (exact fp32).  K % 16 == 0, N <= 128."""

    @staticmethod
    def forward(ctx, X, W, b, res, relu_in: bool, relu_out: bool):
        M, K = X.shape
        N = W.shape[0]
        Xc, Wc = X.detach().contiguous().float(), W.detach().contiguous().float()
        bc = None if b is None else b.detach().contiguous().float()
        rc = None if res is None else res.detach().contiguous().float()
        Y = torch.empty(M, N, dtype=torch.float32, device=X.device)
        call("cpn_linear_f32", Xc.data_ptr(), K, Wc.data_ptr(), K, 0 if bc is None else bc.data_ptr(),
             0 if rc is None else rc.data_ptr(), N, Y.data_ptr(), N, M, N, K, int(relu_in), int(relu_out), _stream())
        ctx.save_for_backward(Xc, Wc, Y if relu_out else None)
        ctx.relu_in, ctx.relu_out = relu_in, relu_out
        ctx.has_b, ctx.has_res = b is not None, res is not None
        return Y

    @staticmethod
    def backward(ctx, dY):
        X, W, Y = ctx.saved_tensors
        d = dY * (Y > 0) if ctx.relu_out else dY
        Xa = torch.relu(X) if ctx.relu_in else X
        dX = d @ W
        if ctx.relu_in:
            dX = dX * (X > 0)
        from .ufc_ops import wgrad_f32
        dW, db = wgrad_f32(d, Xa, ctx.has_b)
        return dX, dW, db, (d if ctx.has_res else None), None, None


class LocalHiddenFn(Function):
    """out = fp16(relu(W . L(row) + b + add[ray])) through cpn_local_hidden."""

    @staticmethod
    def forward(ctx, loc8, coords9, W, b, add, dims, gs: GradScale):
        B, V, R, S = dims
        nrays = B * R
        Wc, bc = W.detach().contiguous().float(), b.detach().contiguous().float()
        ac = None if add is None else add.detach().contiguous().float()
        out = torch.empty(nrays * V * S, 128, dtype=torch.float16, device=loc8.device)
        call("cpn_local_hidden", loc8.data_ptr(), coords9.data_ptr(), Wc.data_ptr(), Wc.shape[1], bc.data_ptr(),
             0 if ac is None else ac.data_ptr(), B, V, R, S, 0, nrays, out.data_ptr(), _stream())
        ctx.save_for_backward(loc8, coords9, out)
        ctx.dims, ctx.has_add, ctx.gs = dims, add is not None, gs
        return out

    @staticmethod
    def backward(ctx, dout):
        loc8, coords9, out = ctx.saved_tensors
        B, V, R, S = ctx.dims
        ds = ctx.gs.scaled16(dout.contiguous()).to(torch.float16)
        dev = ds.device
        # ReLU mask, un-scaling, the 128 x 16 weight gradient, the bias gradient and the per-ray sum in one pass over ds
        dW = torch.zeros(128, 16, dtype=torch.float32, device=dev)
        db = torch.zeros(128, dtype=torch.float32, device=dev)
        dadd = torch.empty(B * R, 128, dtype=torch.float32, device=dev) if ctx.has_add else None
        scale = ctx.gs.s.reshape(1).float().contiguous()
        call("cpn_local_hidden_bwd", ds.data_ptr(), out.data_ptr(), loc8.data_ptr(), coords9.data_ptr(), scale.data_ptr(),
             B, V, R, S, dW.data_ptr(), db.data_ptr(), 0 if dadd is None else dadd.data_ptr(), _stream())
        return None, None, dW, db, dadd, None, None


class AttendHiddenFn(Function):
    """(hbar fp16 (rays,1664), w fp32 (N,R,S)) = cpn_attend_hidden(qa, qb, hid)."""

    @staticmethod
    def forward(ctx, qa, qb, hid2, dims, gs: GradScale, hid_parts=None, qb_last=None):
        """qb_last: None = qb is this node's own; False / True = qb (coords_embed) is shared by two nodes and this is the one
        whose backward runs first / last (round 2 / round 1: round 1's gradient depends on round 2's backward).  The first
        parks its dqb in hid_parts and returns none, the last adds it inside the kernel (an autograd accumulation of two
        0.5 GB tensors otherwise)."""
        B, V, R, S = dims
        nrays = B * R
        hbar = torch.empty(nrays, 1664, dtype=torch.float16, device=qa.device)
        w = torch.empty(B * V, R, S, dtype=torch.float32, device=qa.device)
        call("cpn_attend_hidden", qa.data_ptr(), qb.data_ptr(), 0, hid2.data_ptr(), B, V, R, S, 0, nrays, hbar.data_ptr(),
             w.data_ptr(), _stream())
        ctx.save_for_backward(qa, qb, hid2, w)
        ctx.dims, ctx.gs, ctx.hid_parts, ctx.qb_last = dims, gs, hid_parts, qb_last
        return hbar, w

    @staticmethod
    def backward(ctx, dhbar, dw):
        qa, qb, hid2, w = ctx.saved_tensors
        B, V, R, S = ctx.dims
        nrays = B * R
        gs = ctx.gs
        # the kernel is linear in (dhbar, dw): feed both in the scaled representation, get scaled fp16 gradients back.
        # dhbar (fp16) already carries the scale its producer fixed; if only the softmax weights received a gradient
        # (dhbar is the materialised zero tensor) the scale is fixed here from dw
        if gs.s is None:
            gs.ensure(dw)
        dh = dhbar.float().contiguous()
        dwc = None if dw is None else (dw.float() * gs.s).contiguous()
        dqa, dqb = torch.empty_like(qa), torch.empty_like(qb)
        park = ctx.hid_parts is not None
        dhid = None if park else torch.empty_like(hid2)
        hp = ctx.hid_parts
        share = SHARE_QB_GRAD and hp is not None and ctx.qb_last is not None
        acc = hp.dqb if (share and ctx.qb_last) else None
        call("cpn_attend_hidden_bwd", qa.data_ptr(), qb.data_ptr(), hid2.data_ptr(), w.data_ptr(), dh.data_ptr(),
             0 if dwc is None else dwc.data_ptr(), B, V, R, S, 0, nrays, dqa.data_ptr(), dqb.data_ptr(),
             0 if park else dhid.data_ptr(), 0 if acc is None else acc.data_ptr(), _stream())
        if park:                            # dhid = w (x) dhbar is formed by the producer of hid (cpn_hid_grad_combine)
            hp.parts.append((w, dh))
        if share:
            if ctx.qb_last:
                hp.dqb, hp.dqb_done = None, True
            elif not hp.dqb_done:           # (had the other round already run, nobody would pick the parked tensor up)
                hp.dqb = dqb
                dqb = None
        return dqa, dqb, dhid, None, None, None, None


class EncodeFn(Function):
    """hid fp16 (rows2, 832) = relu(query_encode_latent([gather | tanh(pt/5)])) through cpn_encode_key (node tables + K = 80
    MFMA + the folded key layer: the inference kernel) — the gathered 835-channel input is never materialised.
    The backward differentiates the layer in its table form (csrc/encode_bwd.hip, _backward_tables)."""

    @staticmethod
    def forward(ctx, z0, z1, z2, z3, W, b, pixel_val, sec_grid, pe6, dims, HW, gs: GradScale, hid_parts, key=None):
        B, V, R, S = dims
        H, Wd = HW
        s = _stream()
        dev = z0.device
        maps = []
        for t in (z0, z1, z2, z3):
            src = t.detach().float().contiguous()
            n, c, h, w_ = src.shape
            dst = torch.empty(n, h, w_, c, dtype=torch.float16, device=dev)
            call("cpn_nchw_to_nhwc_f16", src.data_ptr(), dst.data_ptr(), n, c, h, w_, s)
            maps.append(dst)
        Wc = W.detach().contiguous().float()
        bc = b.detach().contiguous().float()
        frag = torch.empty(13 * 3 * 4 * 64 * 8, dtype=torch.float16, device=dev)
        wtab = torch.empty(_hip.TAB_LD, 768, dtype=torch.float16, device=dev)
        call("cpn_pack_encode_weights", Wc.data_ptr(), Wc.shape[1], frag.data_ptr(), wtab.data_ptr(), s)
        nimg = z0.shape[0]
        nodes = nimg * int(_hip.lib().cpn_encode_table_nodes(H, Wd))
        feat = torch.empty(nodes, 768, dtype=torch.float16, device=dev)
        call("cpn_node_features", maps[0].data_ptr(), maps[1].data_ptr(), maps[2].data_ptr(), H, Wd, nimg, feat.data_ptr(), s)
        tab = torch.empty(nodes, _hip.TAB_LD, dtype=torch.float16, device=dev)
        zero = torch.zeros(_hip.TAB_LD, dtype=torch.float32, device=dev)
        call("cpn_gemm_f16", feat.data_ptr(), 768, wtab.data_ptr(), 768, zero.data_ptr(), tab.data_ptr(), _hip.TAB_LD, nodes,
             _hip.TAB_LD, 768, 0, 0, s)
        nrays = B * R
        hid = torch.empty(nrays * V * S * 2, 832, dtype=torch.float16, device=dev)
        if key is None:                      # a caller that only wants hid: a zero key layer rides along
            key = KeyForward(torch.zeros(128, 1664, device=dev), torch.zeros(128, device=dev))
        from .render import pack_key_ring
        Wk = key.W.detach().contiguous().float()
        key.W16 = torch.empty(128, 1664, dtype=torch.float16, device=dev)
        call("cpn_pack_weight_f16", Wk.data_ptr(), 128, 1664, key.W16.data_ptr(), 1664, s)
        ring = pack_key_ring(key.W16)
        kb = key.b.detach().contiguous().float()
        key.kh = torch.empty(nrays * V * S, 128, dtype=torch.float16, device=dev)
        call("cpn_encode_key", tab.data_ptr(), maps[3].data_ptr(), H, Wd, pixel_val.data_ptr(), sec_grid.data_ptr(),
             pe6.data_ptr(), frag.data_ptr(), bc.data_ptr(), ring.data_ptr(), kb.data_ptr(), B, V, R, S, 0, nrays,
             hid.data_ptr(), key.kh.data_ptr(), 0, s)
        W16 = torch.zeros(832, 896, dtype=torch.float16, device=dev)      # fp16 image of W, K padded (level-3 columns 768..831)
        call("cpn_pack_weight_f16", Wc.data_ptr(), 832, Wc.shape[1], W16.data_ptr(), 896, s)
        del tab, maps[0], maps[0], maps[0]                               # maps is now [level 3]
        ctx.save_for_backward(maps[0], pixel_val, sec_grid, pe6, W16, hid, feat, wtab)
        ctx.dims, ctx.HW, ctx.gs, ctx.hid_parts, ctx.K = dims, HW, gs, hid_parts, Wc.shape[1]
        ctx.shapes = [tuple(t.shape) for t in (z0, z1, z2, z3)]
        return hid

    @staticmethod
    def _backward_tables(ctx, d16):
        """The layer differentiated in its table form (csrc/encode_bwd.hip): one 832-wide scatter into the node tables,
        two small GEMMs over the 0.28 M nodes, the adjoint of the node sampling, and the K = 80 tail (full-resolution
        level, point encoding, bias) on 128-wide operands — instead of a re-gather, a 4.2 M x 832 x 896 weight-gradient
        GEMM, a 4.2 M x 832 x 832 data-gradient GEMM and an 832-column scatter into four maps."""
        m3, pixel_val, sec_grid, pe6, W16, hid, feat, wtab = ctx.saved_tensors
        B, V, R, S = ctx.dims
        H, Wd = ctx.HW
        s = _stream()
        dev = d16.device
        lib = _hip.lib()
        nimg = B * V
        nodes = nimg * int(lib.cpn_encode_table_nodes(H, Wd))
        gs = ctx.gs.s
        # ---- table gradient (fp32, carries the pass's scale gs)
        dT = torch.zeros(nodes, _hip.TAB_LD, dtype=torch.float32, device=dev)
        boxes = torch.empty(int(lib.cpn_scatter_tables_scratch(H, Wd, B, V, R, S)), dtype=torch.int32, device=dev)
        call("cpn_scatter_rows_tables", d16.data_ptr(), d16.shape[1], H, Wd, pixel_val.data_ptr(), sec_grid.data_ptr(), B, V,
             R, S, 0, B * R, dT.data_ptr(), boxes.data_ptr(), s)
        # a node sums up to thousands of rows: its own power-of-two scale for the fp16 GEMM operands (device side, no sync)
        dT16 = torch.empty(nodes, _hip.TAB_LD, dtype=torch.float16, device=dev)
        sc = torch.zeros(3, dtype=torch.float32, device=dev)                  # [amax bits scratch, chosen scale, 1 / scale]
        call("cpn_scale_to_f16", dT.data_ptr(), dT.numel(), 4096.0, sc.data_ptr(), dT16.data_ptr(), sc[1:].data_ptr(), s)
        s2 = sc[1]
        del dT
        both = (gs * s2).reshape(1)
        dWtab = _wgrad_tall(dT16, feat, both) if ctx.needs_input_grad[4] else None               # (832, 768)
        g = [None] * 4
        if any(ctx.needs_input_grad[:4]):
            dfeat = _mm_f32(dT16, wtab)                                                          # (nodes, 768), scale gs*s2
            shapes = ctx.shapes
            dm = [torch.empty(n, h, w_, c, dtype=torch.float32, device=dev) for (n, c, h, w_) in shapes[:3]]
            call("cpn_node_features_bwd", dfeat.data_ptr(), H, Wd, nimg, dm[0].data_ptr(), dm[1].data_ptr(), dm[2].data_ptr(), s)
            del dfeat
            inv2 = 1.0 / both
            g[:3] = [m.mul_(inv2).permute(0, 3, 1, 2).contiguous() for m in dm]
            # ---- full-resolution level: data gradient of its 64 columns, scattered into the map
            Wt = torch.zeros(832, 128, dtype=torch.float16, device=dev)
            Wt[:, :64] = W16[:, 768:832]
            dA = _data_grad(d16, Wt)                                                             # (rows, 128) fp16, scale gs
            n3, c3, h3, w3 = shapes[3]
            dm3 = torch.zeros(n3, h3, w3, c3, dtype=torch.float32, device=dev)
            boxes3 = torch.empty(B * V * lib.cpn_gather_bwd_chunks(R, S) * 16, dtype=torch.int32, device=dev)
            call("cpn_gather_rows_bwd_level3", dA.data_ptr(), dA.shape[1], 0, H, Wd, pixel_val.data_ptr(), sec_grid.data_ptr(),
                 B, V, R, S, 0, B * R, dm3.data_ptr(), boxes3.data_ptr(), s)
            del dA
            # handed on as the NHWC buffer it is (an NCHW view with channels-last strides): its only consumer is conv_map's
            # backward, a library convolution that takes that layout; the .contiguous() here was a 134 MB transpose
            g[3] = dm3.mul_(1.0 / gs).permute(0, 3, 1, 2)
            if os.environ.get("COPONERF_NCHW_LEVEL3_GRAD") == "1":
                g[3] = g[3].contiguous()
        dW = db = None
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[5]:
            xt = torch.empty(d16.shape[0], 128, dtype=torch.float16, device=dev)
            call("cpn_gather_tail", m3.data_ptr(), H, Wd, pixel_val.data_ptr(), sec_grid.data_ptr(), pe6.data_ptr(), B, V, R, S,
                 0, B * R, xt.data_ptr(), s)
            dWt = _wgrad_tall(d16, xt, gs)                                                       # (832, 128)
            if ctx.needs_input_grad[4]:
                dW = torch.cat((dWtab, dWt[:, :ctx.K - 768]), dim=1)
            if ctx.needs_input_grad[5]:
                db = dWt[:, ctx.K - 768].contiguous()
        return g[0], g[1], g[2], g[3], dW, db, None, None, None, None, None, None, None, None

    @staticmethod
    def backward(ctx, dC):
        hid = ctx.saved_tensors[5]
        B, V, R, S = ctx.dims
        H, Wd = ctx.HW
        s = _stream()
        d = ctx.gs.scaled16(dC.contiguous()).to(torch.float16)
        if ctx.hid_parts is not None and ctx.hid_parts.combined and not ctx.hid_parts.parts:
            # the key path's data-gradient GEMM already added the parked parts and applied the mask (cpn_gemm_f16_combine)
            ctx.hid_parts.combined = False
            d16 = d.view(hid.shape)
        else:                                                # (parts parked after a fused combine are still added here)
            parts = ctx.hid_parts.parts if ctx.hid_parts is not None else []
            (w1, dh1) = parts[0] if len(parts) > 0 else (None, None)
            (w2, dh2) = parts[1] if len(parts) > 1 else (None, None)
            d16 = torch.empty_like(hid)                      # relu mask (.) (key-path gradient + parked rank-one parts)
            call("cpn_hid_grad_combine", d.data_ptr(), hid.data_ptr(), 0 if w1 is None else w1.data_ptr(),
                 0 if dh1 is None else dh1.data_ptr(), 0 if w2 is None else w2.data_ptr(), 0 if dh2 is None else dh2.data_ptr(),
                 B, V, R, S, 0, B * R, d16.data_ptr(), s)
            if ctx.hid_parts is not None:
                ctx.hid_parts.parts = []
                ctx.hid_parts.combined = False
        del d
        return EncodeFn._backward_tables(ctx, d16)
