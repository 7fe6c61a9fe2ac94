"""ORACLE — CPU restatement of the 4-D operators of UFC.   *** TEST INFRASTRUCTURE ***

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  It restates, with
stock PyTorch CPU ops, the operators that coponerf_amd/getz.py dispatches to HIP kernels on the GPU:

  conv4d_gn_relu      conv4d.Conv4d + MaxPool4d + GroupNorm + ReLU   (/root/reference models/conv4d.py:7-30,57-163)
  correlation_tokens  aggregation.correlation / correlation_token     (models/aggregation.py:70-80)
  soft_argmax_pair    aggregation.soft_argmax x2 + softmax_with_temperature (models/aggregation.py:119-144,555-560)

Pinned by tests/golden/ufc_ops.npz and tests/golden/getz.npz, generated from the imported upstream model
(tests/golden/make_golden_getz.py).  `TorchOps` plugs into coponerf_amd.getz.get_z so that the (stock-op) glue of
the product is itself checked against the upstream get_z on CPU.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _pool_pair(x, s, dims):
    """max-pool (kernel = stride = s, ceil_mode) over the pair of dims `dims` of a 6-D tensor (conv4d.py:7-30)."""
    if s == 1:
        return x
    B, C = x.shape[:2]
    if dims == "support":                                      # pool over the last two dims
        y = F.max_pool2d(x.reshape(B * C * x.shape[2] * x.shape[3], 1, x.shape[4], x.shape[5]), s, s, 0, ceil_mode=True)
        return y.reshape(B, C, x.shape[2], x.shape[3], y.shape[-2], y.shape[-1])
    xp = x.permute(0, 1, 4, 5, 2, 3)
    y = F.max_pool2d(xp.reshape(B * C * x.shape[4] * x.shape[5], 1, x.shape[2], x.shape[3]), s, s, 0, ceil_mode=True)
    return y.reshape(B, C, x.shape[4], x.shape[5], y.shape[-2], y.shape[-1]).permute(0, 1, 4, 5, 2, 3)


def conv4d(x, wq, bq, ws, bs, k, s, p):
    """x (B,Cin,Hq,Wq,Hs,Ws) -> (B,Cout,Hq',Wq',Hs',Ws')  (conv4d.py:108-135)."""
    B, Cin = x.shape[:2]
    xq = _pool_pair(x, s, "support")                           # query branch sees support-pooled input
    xs = _pool_pair(x, s, "query")
    Hq, Wq, Hs2, Ws2 = xq.shape[2:]
    yq = F.conv2d(xq.permute(0, 4, 5, 1, 2, 3).reshape(B * Hs2 * Ws2, Cin, Hq, Wq), wq, bq, stride=s, padding=p)
    yq = yq.reshape(B, Hs2, Ws2, -1, yq.shape[-2], yq.shape[-1]).permute(0, 3, 4, 5, 1, 2)
    Hq2, Wq2, Hs, Ws = xs.shape[2:]
    ys = F.conv2d(xs.permute(0, 2, 3, 1, 4, 5).reshape(B * Hq2 * Wq2, Cin, Hs, Ws), ws, bs, stride=s, padding=p)
    ys = ys.reshape(B, Hq2, Wq2, -1, ys.shape[-2], ys.shape[-1]).permute(0, 3, 1, 2, 4, 5)
    return yq + ys


def l2_normalise_tokens(x, eps=1e-5):
    return x / (x.norm(dim=-1, p=2, keepdim=True) + eps)


class TorchOps:
    """The `ops` interface of coponerf_amd.getz with CPU restatements."""

    @staticmethod
    def conv4d_gn_relu(x, wq, bq, ws, bs, k, s, p, gn_w, gn_b, eps, residual=None):
        y = conv4d(x, wq, bq, ws, bs, k, s, p)
        out = F.relu(F.group_norm(y, 1, gn_w, gn_b, eps))
        return out if residual is None else residual + out          # the caller's `x + Encoder4D(...)`

    @staticmethod
    def dual_softmax(a):
        """models/backbone.py:296-330: the product of the row-wise and the column-wise softmax."""
        return a.softmax(dim=-1) * a.softmax(dim=-2)

    @staticmethod
    def resize_bilinear(x, size):
        return F.interpolate(x, size=(size, size), mode="bilinear", align_corners=True)

    @staticmethod
    def correlation_tokens(src, trg, fs):
        B = src.shape[0]
        c = torch.einsum("bsc,btc->bst", l2_normalise_tokens(src), l2_normalise_tokens(trg))
        return c.reshape(B, 1, fs, fs, fs, fs)

    @staticmethod
    def conv_map(rgb, w, b, want_nhwc16=False):
        """models/CoPoNeRF.py:182-187: (rgb+1)/2, ImageNet normalisation, 7x7 convolution.  rgb (N,H,W,3)."""
        x = (rgb.permute(0, 3, 1, 2) + 1) / 2.
        mean = torch.tensor((0.485, 0.456, 0.406)).view(1, 3, 1, 1)
        std = torch.tensor((0.229, 0.224, 0.225)).view(1, 3, 1, 1)
        return F.conv2d((x - mean) / std, w, b, stride=1, padding=3), None

    @staticmethod
    def linear_attention(q, k, v, channel_major=False, eps=1e-6):
        """aggregation.LinearAttention.forward (models/aggregation.py:84-117), phi = ELU + 1.
        q, k (B,L,H,D); v / result (B,L,H,Dv), or (B,H,Dv,L) when channel_major."""
        if channel_major:
            v = v.permute(0, 3, 1, 2)
        Q, K = F.elu(q) + 1, F.elu(k) + 1
        L = v.shape[1]
        KV = torch.einsum("nshd,nshv->nhdv", K, v / L)
        Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
        out = torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * L
        return out.permute(0, 2, 3, 1).contiguous() if channel_major else out.contiguous()

    @staticmethod
    def cost_volume_attention(q, k, v_corr, fs, residual=None, eps=1e-6):
        """The cost-volume side of UFCLayer.forward_attention AS THE REFERENCE ORDERS IT (models/aggregation.py:283-297,
        :301): value_corr interpolated to fs x fs, LinearAttention over the fs*fs tokens, the message interpolated back."""
        B, H, Hs, Ws, Ht, Wt = v_corr.shape
        vc = v_corr.permute(0, 1, 4, 5, 2, 3).reshape(B, H * Ht * Wt, Hs, Ws)
        vc = F.interpolate(vc, size=(fs, fs), mode="bilinear", align_corners=True)
        vc = vc.reshape(B, H, Ht * Wt, fs * fs).permute(0, 3, 1, 2)                              # (B, L, H, Dv)
        msg = TorchOps.linear_attention(q, k, vc, eps=eps)                                       # (B, L, H, Dv)
        msg = msg.permute(0, 2, 3, 1).reshape(B, H * Ht * Wt, fs, fs)
        msg = F.interpolate(msg, size=(Hs, Ws), mode="bilinear", align_corners=True)
        msg = msg.reshape(B, H, Ht, Wt, Hs, Ws).permute(0, 1, 4, 5, 2, 3)
        return msg if residual is None else residual + msg

    @staticmethod
    def cross_attention(c, src_v, trg_v):
        """UFCLayer.forward_cross, models/aggregation.py:327-328: c (B,H,S,T), src_v (B,S,H,C), trg_v (B,T,H,C)."""
        src_attn = torch.einsum("bhst,bthc->bshc", c.softmax(-1), trg_v)
        trg_attn = torch.einsum("bhst,bshc->bthc", c.softmax(-2), src_v)
        return src_attn.contiguous(), trg_attn.contiguous()

    @staticmethod
    def _soft_argmax(corr, beta=0.02):
        b, _, h, w = corr.shape
        m, _ = corr.max(dim=1, keepdim=True)
        e = torch.exp((corr - m) / beta)
        pr = (e / e.sum(dim=1, keepdim=True)).view(-1, h, w, h, w)
        xn = torch.linspace(-1, 1, w).view(1, w, 1, 1)
        yn = torch.linspace(-1, 1, h).view(1, h, 1, 1)
        gx = (pr.sum(dim=1) * xn).sum(dim=1, keepdim=True)
        gy = (pr.sum(dim=2) * yn).sum(dim=1, keepdim=True)
        return gx, gy

    @classmethod
    def soft_argmax_pair(cls, c):
        gx, gy = cls._soft_argmax(c.permute(0, 1, 4, 5, 2, 3).flatten(1, 3))
        t_to_s = torch.cat((gx, gy), dim=1)
        gx, gy = cls._soft_argmax(c.flatten(1, 3))
        return t_to_s, torch.cat((gx, gy), dim=1)
