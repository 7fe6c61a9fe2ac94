"""Training path on the MI355X: gradients of the HIP render path against autograd through the CPU oracle."""
import pytest
import torch

from coponerf_amd import synthetic as syn
from tests.helpers import to_device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def _loss(out, coef, cw):
    return (out["rgb"] * coef).sum() + (out["at_wt"] * cw).sum()


def test_render_gradients_match_oracle_autograd(dev):
    from coponerf_amd import CoPoNeRF
    from oracle import render_ref as orc
    B, H, R, S = 2, 64, 80, 32
    weights = syn.make_render_weights(seed=17)
    inp = syn.make_inputs(B, H, H, R, seed=51)
    z, rel, flow = syn.make_latents(B, H, H, seed=52)
    coef = syn.normal((B, 1, R, 3), seed=53)
    cw = syn.normal((2 * B, R, S), seed=54) * 0.3
    # ---- oracle gradients (float32 autograd on the CPU)
    w_ref = {k: v.clone().requires_grad_(True) for k, v in weights.items()}
    z_ref = [t.clone().requires_grad_(True) for t in z]
    out_ref = orc.forward(inp, z_ref, rel, flow, False, w_ref, npoints=S)
    _loss(out_ref, coef, cw).backward()
    # ---- HIP path
    model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
    model.load_state_dict(weights, strict=False)
    model = model.to(dev)
    model.train()
    z_hip = [t.to(dev).requires_grad_(True) for t in z]
    out = model(to_device(inp, dev), z=z_hip, rel_pose=rel.to(dev), val=False, flow=to_device(flow, dev))
    assert out["rgb"].requires_grad and out["at_wt"].requires_grad
    assert (out["rgb"].detach().cpu() - out_ref["rgb"].detach()).abs().max() <= 1e-3
    _loss(out, coef.to(dev), cw.to(dev)).backward()
    params = dict(model.named_parameters())

    report = []

    def close(name, got, want):
        # fp16 activations flip a few ReLU masks relative to the fp32 oracle, which moves single entries of the
        # small per-ray layers by ~1/rays; judge the whole tensor (relative L2) and bound the worst entry loosely.
        got = got.cpu()
        scale = float(want.abs().max())
        err = float((got - want).abs().max())
        rel = float((got - want).norm() / (want.norm() + 1e-12))
        report.append((name, rel, err, scale))
        return rel <= 3e-2 and err <= 0.1 * scale + 1e-6

    for name, wr in w_ref.items():
        assert params[name].grad is not None, name
    ok = [close(name, params[name].grad, wr.grad) for name, wr in w_ref.items()]
    ok += [close(f"z[{i}]", zh.grad, zr.grad) for i, (zh, zr) in enumerate(zip(z_hip, z_ref))]
    table = "\n".join(f"{n:48s} relL2 {r:.3e} maxerr {e:.3e} scale {s_:.3e}" for n, r, e, s_ in report)
    print(table)
    assert all(ok), table
    # layers the render path never touches stay without gradient (skip-None contract of wrapper.py:26)
    assert params["corr_embed.weight"].grad is None
    # ... and against the upstream reference's own gradients on the same case (tests/golden/grads.npz)
    from tests.test_oracle_grads import check_against_fixture
    grads = {name: params[name].grad for name in w_ref}
    grads.update({f"z{i}": t.grad for i, t in enumerate(z_hip)})
    check_against_fixture(grads, rel_l2=5e-2, rel_max=0.1)          # fixture holds a 1-in-61 sample per tensor


def test_render_training_converges_like_the_fp32_oracle(dev):
    """the HIP backward carries fp16 activation gradients (one static scale per pass) and agrees with upstream's
    gradients only to ~1e-2 per tensor - what does that do to TRAINING?  Twelve Adam steps on the render weights (latents given,
    a target image rendered by a perturbed copy of the weights): the HIP path on the device beside float32 autograd through the
    CPU oracle from the same start.  The loss curves must coincide (every step within 3 % of the loss; measured: 4 digits at first, 1 % after twelve steps),
    and the weights must have moved to the same place (distance between the two end points small against the distance moved)."""
    from coponerf_amd import CoPoNeRF
    from oracle import render_ref as orc
    B, H, R, S, steps, lr = 1, 64, 192, 32, 12, 2e-4
    start = syn.make_render_weights(seed=23)
    teacher = {k: v + 0.3 * syn.normal(tuple(v.shape), seed=900 + i) * v.abs().mean() for i, (k, v) in enumerate(start.items())}
    inp = syn.make_inputs(B, H, H, R, seed=71)
    z, rel, flow = syn.make_latents(B, H, H, seed=72)
    with torch.no_grad():
        target = orc.forward(inp, z, rel, flow, False, teacher, npoints=S)["rgb"]
    trained = [k for k in start if not k.startswith(("corr_embed", "latent_avg"))]
    # ---- float32 autograd through the oracle (CPU)
    w_ref = {k: v.clone().requires_grad_(k in trained) for k, v in start.items()}
    opt_ref = torch.optim.Adam([w_ref[k] for k in trained], lr=lr)
    loss_ref = []
    for _ in range(steps):
        opt_ref.zero_grad()
        l = (orc.forward(inp, z, rel, flow, False, w_ref, npoints=S)["rgb"] - target).abs().mean()
        l.backward()
        opt_ref.step()
        loss_ref.append(float(l))
    # ---- the HIP training path (device)
    model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
    model.load_state_dict(start, strict=False)
    model = model.to(dev).train()
    P = dict(model.named_parameters())
    opt = torch.optim.Adam([P[k] for k in trained], lr=lr)
    d_inp, d_z, d_flow, d_tgt = to_device(inp, dev), to_device(z, dev), to_device(flow, dev), target.to(dev)
    loss_hip = []
    for _ in range(steps):
        opt.zero_grad()
        l = (model(d_inp, z=d_z, rel_pose=rel.to(dev), val=False, flow=d_flow)["rgb"] - d_tgt).abs().mean()
        l.backward()
        opt.step()
        loss_hip.append(float(l))
    print("loss, fp32 oracle:", [round(x, 5) for x in loss_ref])
    print("loss, HIP path   :", [round(x, 5) for x in loss_hip])
    # (at this learning rate Adam's first steps throw the loss from 0.04 to 0.35 and it comes back over the next ten: a
    # trajectory that amplifies differences - the comparison is only meaningful if the run moves the loss a lot)
    assert max(loss_ref) > 2 * min(loss_ref)
    for k, (a, b) in enumerate(zip(loss_hip, loss_ref)):
        assert abs(a - b) <= 3e-2 * b, (k, a, b)
    moved = sum(float((w_ref[k].detach() - start[k]).pow(2).sum()) for k in trained) ** 0.5
    apart = sum(float((P[k].detach().cpu() - w_ref[k].detach()).pow(2).sum()) for k in trained) ** 0.5
    print(f"weights moved {moved:.4f}, HIP end point {apart:.4f} from the oracle's ({apart / moved:.3f} of the way)")
    assert apart <= 0.15 * moved, (apart, moved)


def test_full_training_step_end_to_end(dev):
    """get_z + render + backward + SGD step at 256x256: gradients reach encoder, UFC and render weights, loss moves."""
    from coponerf_amd import CoPoNeRF
    model = CoPoNeRF.CoPoNeRF(n_view=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes), strict=True)
    model = model.to(dev).train()
    inp = to_device(syn.make_inputs(1, 256, 256, 128, seed=61), dev)
    target = inp["query"]["rgb"]
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        out = model(inp, val=False)
        loss = (out["rgb"] - target).abs().mean()
        loss.backward()
        losses.append(float(loss.detach()))
        opt.step()
    P = dict(model.named_parameters())
    for name in ("query_encode_latent.weight", "phi.lin_out.weight", "encoder.model.conv1.weight", "conv_map.weight",
                 "feature_cost_aggregation.layers.0.0.q_proj.weight",
                 "feature_cost_aggregation.embedding.2.conv4d.0.0.query_conv.weight"):
        assert P[name].grad is not None and torch.isfinite(P[name].grad).all() and float(P[name].grad.abs().max()) > 0, name
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0]


def test_render_gradients_ragged_shapes(dev):
    """B = 3, 37 rays, 24 samples (not a power of two): gradients against autograd through the oracle."""
    from coponerf_amd import CoPoNeRF
    from oracle import render_ref as orc
    B, H, R, S = 3, 48, 37, 24
    weights = syn.make_render_weights(seed=19)
    inp = syn.make_inputs(B, H, H, R, seed=95)
    z, rel, flow = syn.make_latents(B, H, H, seed=96)
    coef = syn.normal((B, 1, R, 3), seed=97)
    w_ref = {k: v.clone().requires_grad_(True) for k, v in weights.items()}
    z_ref = [t.clone().requires_grad_(True) for t in z]
    (orc.forward(inp, z_ref, rel, flow, False, w_ref, npoints=S)["rgb"] * coef).sum().backward()
    model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
    model.load_state_dict(weights, strict=False)
    model = model.to(dev).train()
    z_hip = [t.to(dev).requires_grad_(True) for t in z]
    out = model(to_device(inp, dev), z=z_hip, rel_pose=rel.to(dev), val=False, flow=to_device(flow, dev))
    (out["rgb"] * coef.to(dev)).sum().backward()
    params = dict(model.named_parameters())
    bad = []
    for name, ref in list(w_ref.items()) + [(f"z{i}", t) for i, t in enumerate(z_ref)]:
        got = (z_hip[int(name[1:])] if name[0] == "z" and name[1:].isdigit() else params[name]).grad
        if ref.grad is None or float(ref.grad.norm()) == 0.0:
            continue
        rel_err = float((got.cpu() - ref.grad).norm() / ref.grad.norm())
        # with ~100 rays one ReLU of the per-ray decoder flipping between the fp16 and the fp32 forward moves the
        # phi gradients by a few percent; everything per-sample averages over 10^4 rows
        if rel_err > (0.12 if name.startswith("phi.") else 4e-2):
            bad.append((name, rel_err))
    assert not bad, bad


def test_ufc_operator_gradients_match_oracle(dev):
    """HipOps forward (HIP) + library-op VJP against autograd through the CPU oracle operators."""
    from coponerf_amd.ufc_ops import HipOps
    from oracle.ufc_ref import TorchOps
    hip, ref = HipOps(), TorchOps()

    def both(fn_name, tensors, extra_hip, extra_ref=None):
        extra_ref = extra_hip if extra_ref is None else extra_ref
        a = [t.clone().requires_grad_(True) for t in tensors]
        b = [t.clone().to(dev).requires_grad_(True) for t in tensors]
        oa = extra_ref(getattr(ref, fn_name), a)
        ob = extra_hip(getattr(hip, fn_name), b)
        oa = oa if isinstance(oa, tuple) else (oa,)
        ob = ob if isinstance(ob, tuple) else (ob,)
        la = sum((o * syn.normal(tuple(o.shape), seed=90 + i)).sum() for i, o in enumerate(oa))
        lb = sum((o * syn.normal(tuple(o.shape), seed=90 + i).to(dev)).sum() for i, o in enumerate(ob))
        la.backward(), lb.backward()
        for i, (x, y) in enumerate(zip(a, b)):
            rel = float((y.grad.cpu() - x.grad).norm() / (x.grad.norm() + 1e-12))
            assert rel <= 2e-3, (fn_name, i, rel)

    x = syn.normal((1, 4, 8, 8, 8, 8), seed=81)
    for k, s, p, cout in ((3, 1, 1, 8), (3, 2, 1, 6)):
        wq, ws = syn.normal((cout, 4, k, k), seed=82) * 0.2, syn.normal((cout, 4, k, k), seed=83) * 0.2
        bq, bs = syn.normal((cout,), seed=84) * 0.1, syn.normal((cout,), seed=85) * 0.1
        gw, gb = 1 + 0.1 * syn.normal((cout,), seed=86), 0.1 * syn.normal((cout,), seed=87)
        both("conv4d_gn_relu", [x, wq, bq, ws, bs, gw, gb],
             lambda f, t: f(t[0], t[1], t[2], t[3], t[4], k, s, p, t[5], t[6], 1e-5))
    src, trg = syn.normal((2, 64, 32), seed=88), syn.normal((2, 64, 32), seed=89)
    both("correlation_tokens", [src, trg], lambda f, t: f(t[0], t[1], 8))
    both("soft_argmax_pair", [syn.normal((2, 1, 8, 8, 8, 8), seed=80) * 0.05], lambda f, t: f(t[0]))
    # linear attention, both value layouts, ragged L / Dv and more than one token slab (L >= 128 -> nsplit 2)
    q, k = syn.normal((2, 150, 3, 32), seed=60), syn.normal((2, 150, 3, 32), seed=61)
    both("linear_attention", [q, k, syn.normal((2, 150, 3, 40), seed=62)], lambda f, t: f(t[0], t[1], t[2]))
    both("linear_attention", [q, k, syn.normal((2, 3, 72, 150), seed=63)], lambda f, t: f(t[0], t[1], t[2], True))
    both("linear_attention", [q[:, :33], k[:, :33], syn.normal((2, 33, 3, 32), seed=64)], lambda f, t: f(t[0], t[1], t[2]))
    both("cross_attention", [syn.normal((2, 3, 40, 56), seed=65) * 2.0, syn.normal((2, 40, 3, 32), seed=66),
                             syn.normal((2, 56, 3, 32), seed=67)], lambda f, t: f(t[0], t[1], t[2]))
    # the shapes of the training step: 256 x 256 cost volumes, 8 heads; 1024 tokens with 256-wide channel-major values (16 slabs)
    both("cross_attention", [syn.normal((1, 8, 256, 256), seed=55) * 3.0, syn.normal((1, 256, 8, 32), seed=56),
                             syn.normal((1, 256, 8, 32), seed=57)], lambda f, t: f(t[0], t[1], t[2]))
    both("linear_attention", [syn.normal((1, 1024, 8, 32), seed=58), syn.normal((1, 1024, 8, 32), seed=59),
                              syn.normal((1, 8, 256, 1024), seed=54)], lambda f, t: f(t[0], t[1], t[2], True))
    both("resize_bilinear", [syn.normal((2, 3, 8, 8), seed=79)], lambda f, t: f(t[0], 16))
    both("dual_softmax", [syn.normal((2, 70, 130), seed=78) * 2.0], lambda f, t: f(t[0]))


def test_conv_weight_gradient_kernels(dev):
    """cpn_conv_wgrad_planes / cpn_dwconv3x3_wgrad against autograd of the stock convolutions (fp32, GPU)."""
    import torch.nn.functional as F
    from coponerf_amd import _hip
    from coponerf_amd._hip import call
    st = torch.cuda.current_stream().cuda_stream
    for (B, Cin, Cout, G, H, W) in ((2, 32, 32, 9, 16, 16), (1, 4, 8, 5, 8, 8), (2, 8, 32, 3, 6, 10), (1, 1, 8, 4, 16, 16),
                                  (2, 8, 8, 4, 16, 16), (1, 32, 8, 3, 16, 16), (2, 7, 20, 3, 16, 16)):
        x = syn.normal((B, Cin, G, H, W), seed=70).to(dev)
        dy = syn.normal((B, Cout, G, H, W), seed=71).to(dev)
        w = torch.zeros(Cout, Cin, 3, 3, device=dev, requires_grad=True)
        b = torch.zeros(Cout, device=dev, requires_grad=True)
        x2 = x.permute(0, 2, 1, 3, 4).reshape(B * G, Cin, H, W)
        d2 = dy.permute(0, 2, 1, 3, 4).reshape(B * G, Cout, H, W)
        (F.conv2d(x2, w, b, 1, 1) * d2).sum().backward()
        part = torch.empty(_hip.lib().cpn_conv_wgrad_scratch(Cin, Cout), device=dev)
        dw, db = torch.empty(Cout, Cin, 3, 3, device=dev), torch.empty(Cout, device=dev)
        call("cpn_conv_wgrad_planes", x.data_ptr(), dy.data_ptr(), B, Cin, Cout, G, H, W, part.data_ptr(), dw.data_ptr(),
             db.data_ptr(), st)
        assert (dw - w.grad).abs().max() <= 2e-4 * (1 + w.grad.abs().max()), (B, Cin, Cout, G, H, W)
        assert (db - b.grad).abs().max() <= 2e-4 * (1 + b.grad.abs().max())
    from coponerf_amd.ufc_ops import DwConv3x3Fn
    x = syn.normal((3, 40, 16, 16), seed=72).to(dev)
    wa = (syn.normal((40, 1, 3, 3), seed=73) * 0.3).to(dev)
    ba = (syn.normal((40,), seed=74) * 0.1).to(dev)
    coef = syn.normal((3, 40, 16, 16), seed=75).to(dev)
    grads = []
    for fn in (lambda a, w_, b_: F.conv2d(a, w_, b_, 1, 1, 1, 40), DwConv3x3Fn.apply):
        xs, ws, bs = (t.clone().requires_grad_(True) for t in (x, wa, ba))
        (fn(xs, ws, bs) * coef).sum().backward()
        grads.append((xs.grad, ws.grad, bs.grad))
    for g_ref, g_hip in zip(*grads):
        assert (g_ref - g_hip).abs().max() <= 2e-4 * (1 + g_ref.abs().max())
    # the token-layout form (B, H*W, C) used by the feed-forward blocks: value and all three gradients against the
    # reference's transpose -> grouped Conv2d -> transpose (models/aggregation.py:18-28); non-square check via H != W is
    # not needed (size x size maps only), but an odd size and C = 44 (not a multiple of 8) are
    from coponerf_amd.ufc_ops import DwConvTokensFn
    for (Bq, size, C) in ((2, 7, 44), (3, 16, 256)):
        xt = syn.normal((Bq, size * size, C), seed=76 + C).to(dev)
        wt = (syn.normal((C, 1, 3, 3), seed=77 + C) * 0.3).to(dev)
        bt = (syn.normal((C,), seed=78 + C) * 0.1).to(dev)
        ct = syn.normal((Bq, size * size, C), seed=79 + C).to(dev)

        def ref(a, w_, b_):
            m = a.transpose(1, 2).reshape(Bq, C, size, size)
            return F.conv2d(m, w_, b_, 1, 1, 1, C).flatten(2).transpose(1, 2)
        outs, grads = [], []
        for fn in (ref, lambda a, w_, b_: DwConvTokensFn.apply(a, w_, b_, size)):
            xs, ws, bs = (t.clone().requires_grad_(True) for t in (xt, wt, bt))
            y = fn(xs, ws, bs)
            (y * ct).sum().backward()
            outs.append(y.detach())
            grads.append((xs.grad, ws.grad, bs.grad))
        assert (outs[0] - outs[1]).abs().max() <= 1e-5 * (1 + outs[0].abs().max())
        for g_ref, g_hip in zip(*grads):
            assert g_ref.shape == g_hip.shape
            assert (g_ref - g_hip).abs().max() <= 2e-4 * (1 + g_ref.abs().max())


def test_mean_loss_gradients_survive_fp16(dev):
    """the reference's loss is an L1 `.mean()` over B*R*3 ~ 5e4 values, so dL/drgb ~ 2e-5 and the
    per-sample activation gradients (~1e-8) underflow plain fp16.  With the per-pass power-of-two scale of
    train_fns.GradScale every gradient must match fp32 autograd through the oracle at THAT magnitude (the loss below
    is the small case's L1 sum divided by 4*4096*3, i.e. the configs[2] normalisation)."""
    from coponerf_amd import CoPoNeRF
    from oracle import render_ref as orc
    B, H, R, S = 2, 64, 80, 32
    denom = 4 * 4096 * 3
    weights = syn.make_render_weights(seed=17)
    inp = syn.make_inputs(B, H, H, R, seed=51)
    z, rel, flow = syn.make_latents(B, H, H, seed=52)
    gt = inp["query"]["rgb"]
    w_ref = {k: v.clone().requires_grad_(True) for k, v in weights.items()}
    z_ref = [t.clone().requires_grad_(True) for t in z]
    ((orc.forward(inp, z_ref, rel, flow, False, w_ref, npoints=S)["rgb"] - gt).abs().sum() / denom).backward()
    model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
    model.load_state_dict(weights, strict=False)
    model = model.to(dev).train()
    z_hip = [t.to(dev).requires_grad_(True) for t in z]
    out = model(to_device(inp, dev), z=z_hip, rel_pose=rel.to(dev), val=False, flow=to_device(flow, dev))
    ((out["rgb"] - gt.to(dev)).abs().sum() / denom).backward()
    params = dict(model.named_parameters())
    bad, seen_small = [], False
    for name, ref in list(w_ref.items()) + [(f"z{i}", t) for i, t in enumerate(z_ref)]:
        got = (z_hip[int(name[1:])] if name[0] == "z" and name[1:].isdigit() else params[name]).grad
        if ref.grad is None or float(ref.grad.norm()) == 0.0:
            continue
        assert got is not None and torch.isfinite(got).all(), name
        seen_small |= float(ref.grad.abs().max()) < 6.1e-5               # below fp16's smallest normal
        rel_err = float((got.cpu() - ref.grad).norm() / ref.grad.norm())
        if rel_err > (0.12 if name.startswith("phi.") else 4e-2):
            bad.append((name, rel_err, float(ref.grad.abs().max())))
    assert seen_small, "the case does not exercise sub-fp16-normal gradients"
    assert not bad, bad


def test_config3_size_training_step(dev):
    """BASELINE configs[2] at its real per-GPU size (batch 4 pairs x 4096 rays x 64 samples, 256x256): one full step
    (get_z + render + L1 mean loss + backward) is finite, and because stereo pairs are independent work units the
    batched render gradient equals the sum of the four per-pair gradients."""
    from coponerf_amd import CoPoNeRF
    B, H, R, S = 4, 256, 4096, 64
    weights = syn.make_render_weights(seed=17)
    model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
    model.load_state_dict(weights, strict=False)
    model = model.to(dev).train()
    inp = to_device(syn.make_inputs(B, H, H, R, seed=61), dev)
    z, rel, flow = syn.make_latents(B, H, H, seed=62)
    z, rel, flow = to_device(z, dev), rel.to(dev), to_device(flow, dev)
    gt = inp["query"]["rgb"]
    names = ("query_encode_latent.weight", "key_map.weight", "latent_value.weight", "phi.lin_out.weight",
             "query_embed.weight", "query_repeat_embed_2.bias")
    P = dict(model.named_parameters())

    def grads(sel):
        model.zero_grad(set_to_none=True)
        sub = {"context": {k: v[sel] for k, v in inp["context"].items()}, "query": {k: v[sel] for k, v in inp["query"].items()}}
        zz = [t.view(B, 2, *t.shape[1:])[sel].reshape(-1, *t.shape[1:]) for t in z]
        out = model(sub, z=zz, rel_pose=rel[sel], val=False, flow=[f[sel] for f in flow])
        loss = (out["rgb"] - gt[sel]).abs().sum() / (B * R * 3)
        loss.backward()
        return float(loss.detach()), {n: P[n].grad.detach().clone() for n in names}

    loss_all, g_all = grads(slice(0, B))
    assert torch.isfinite(torch.tensor(loss_all))
    parts = [grads(slice(b, b + 1)) for b in range(B)]
    assert abs(sum(p[0] for p in parts) - loss_all) <= 1e-5 * max(1.0, abs(loss_all))
    for n in names:
        want = sum(p[1][n] for p in parts)
        assert torch.isfinite(g_all[n]).all() and float(g_all[n].abs().max()) > 0, n
        rel_err = float((g_all[n] - want).norm() / (want.norm() + 1e-20))
        # the batched and the per-pair passes pick their fp16 gradient scale separately: equal up to fp16 rounding
        assert rel_err <= 2e-2, (n, rel_err)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 63, 64, 4099, 70000])
def test_skinny_weight_gradient_kernels(M):
    """cpn_wgrad_skinny_f16 (dW = dY^T . X over M rows for the 128-wide layers) against float64."""
    from coponerf_amd._hip import call
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M)
    dY = (torch.randn(M, 128, generator=g) * 0.5).half()
    X = torch.randn(M, 160, generator=g).half()                     # row stride 160 > 128: only the first 128 columns count
    s = torch.cuda.current_stream().cuda_stream
    dYd, Xd = dY.to(dev), X.to(dev)
    dW = torch.zeros(128, 128, device=dev)
    db = torch.zeros(128, device=dev)
    call("cpn_wgrad_skinny_f16", dYd.data_ptr(), Xd.data_ptr(), 160, M, dW.data_ptr(), db.data_ptr(), s)
    want = dY.double().t() @ X[:, :128].double()
    wantb = dY.double().sum(0)
    tol = 2e-4 * max(1.0, float(want.abs().max()))
    assert float((dW.cpu().double() - want).abs().max()) <= tol, float((dW.cpu().double() - want).abs().max())
    assert float((db.cpu().double() - wantb).abs().max()) <= 2e-4 * max(1.0, float(wantb.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 65, 1024, 16 * 64 * 3 + 37, 150001])
def test_tall_weight_gradient_kernel(M):
    """cpn_wgrad_tall_f16 (dW = dY^T . X / scale over M rows, the 832 x 896 first-layer gradient) against float64."""
    from coponerf_amd import _hip
    from coponerf_amd._hip import call
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M)
    N, K, ldy, ldx = 416, 256, 424, 264                            # two n tiles, two k tiles; padded row strides
    dY = (torch.randn(M, ldy, generator=g) * 0.5).half()
    X = torch.randn(M, ldx, generator=g).half()
    s = torch.cuda.current_stream().cuda_stream
    dYd, Xd = dY.to(dev), X.to(dev)
    scale = torch.tensor([4.0], device=dev)
    part = torch.empty(_hip.lib().cpn_wgrad_tall_scratch(N, K), device=dev)
    dW = torch.empty(N, K, device=dev)
    call("cpn_wgrad_tall_f16", dYd.data_ptr(), ldy, Xd.data_ptr(), ldx, M, N, K, scale.data_ptr(), part.data_ptr(),
         dW.data_ptr(), s)
    want = dY[:, :N].double().t() @ X[:, :K].double() / 4.0
    err = float((dW.cpu().double() - want).abs().max())
    assert err <= 2e-4 * max(1.0, float(want.abs().max())), err
    dW2 = torch.empty_like(dW)
    call("cpn_wgrad_tall_f16", dYd.data_ptr(), ldy, Xd.data_ptr(), ldx, M, N, K, scale.data_ptr(), part.data_ptr(),
         dW2.data_ptr(), s)
    assert torch.equal(dW, dW2)                                    # fixed summation order


@pytest.mark.gpu
def test_backward_entry_points_reject_bad_shapes():
    """The training-side entry points fail loudly (RuntimeError carrying cpn_last_error's reason) on shapes outside their
    tile sets instead of computing something else."""
    from coponerf_amd import _hip
    from coponerf_amd._hip import call
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    a = torch.zeros(64, 512, dtype=torch.float16, device=dev)
    f = torch.zeros(1 << 16, dtype=torch.float32, device=dev)
    with pytest.raises(RuntimeError, match="208"):                      # N = 100: not a multiple of the 208-row tile
        call("cpn_wgrad_tall_f16", a.data_ptr(), 512, a.data_ptr(), 512, 64, 100, 128, 0, f.data_ptr(), f.data_ptr(), st)
    with pytest.raises(RuntimeError, match="null"):
        call("cpn_wgrad_tall_f16", 0, 512, a.data_ptr(), 512, 64, 208, 128, 0, f.data_ptr(), f.data_ptr(), st)
    with pytest.raises(RuntimeError, match="C == 32"):                  # cross attention: 16 channels per head
        call("cpn_cross_attention_bwd", f.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(),
             f.data_ptr(), 1, 1, 8, 8, 16, f.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(), st)
    with pytest.raises(RuntimeError, match="bad shape"):
        call("cpn_linear_attention_bwd", f.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(), 1, 0, 1, 8, 0, 1e-6, 1,
             f.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(), st)
    assert _hip.lib().cpn_wgrad_tall_scratch(100, 128) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,s,p,cin,ties", [(32, 3, 2, 1, 1, False), (64, 5, 4, 2, 1, False), (10, 5, 4, 2, 2, False),
                                               (13, 3, 2, 1, 1, True), (16, 3, 2, 1, 1, True)])
def test_strided_conv4d_vjp_equals_library_graph(n, k, s, p, cin, ties):
    """cpn_conv4d_strided_bwd (csrc/ufc_strided_bwd.hip) against autograd through the library graph of the same layer
    (two max_pool2d + two conv2d, models/conv4d.py:57-135): input, weight and bias gradients; ragged (ceil-mode) windows;
    and — with a ReLU'd, quantised input — windows full of equal maxima, where the routing rule decides (first maximum in
    scan order)."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from coponerf_amd import getz, ufc_ops
    dev = torch.device("cuda:0")
    ops = ufc_ops.HipOps()
    torch.manual_seed(3)
    enc = getz.Encoder4D((cin, 8), k, s, p).to(dev)
    B = 2
    x = torch.randn(B, cin, n, n, n, n, device=dev)
    if ties:
        x = (torch.relu(x) * 2).round() / 2                  # many exact zeros and repeated values inside every window
    x.requires_grad_(True)
    params = list(enc.parameters())
    got = {}
    for flag in (True, False):
        old, ufc_ops.STRIDED_HIP_VJP = ufc_ops.STRIDED_HIP_VJP, flag
        try:
            y = enc(x, ops)
            g = torch.cos(torch.arange(y.numel(), device=dev, dtype=torch.float32)).view_as(y)
            got[flag] = torch.autograd.grad((y * g).sum(), [x] + params)
        finally:
            ufc_ops.STRIDED_HIP_VJP = old
    for name, a, b in zip(["x"] + [f"param{i}" for i in range(len(params))], got[True], got[False]):
        scale = float(b.abs().max()) + 1e-12
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 2e-5 * scale, (name, float((a - b).abs().max()), scale)
    # the routing itself: exactly the same elements receive a gradient
    assert torch.equal(got[True][0] != 0, got[False][0] != 0)


@pytest.mark.gpu
def test_corr_mean3_gradients_equal_composed_interpolations():
    """_CorrMean3Fn's backward (adjoint of interpolate4d contracted over the target dims first) against autograd through the
    composed resize / add / divide ops it replaces (UFC.forward, aggregation.py:549-553)."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from coponerf_amd import getz, ufc_ops
    dev = torch.device("cuda:0")
    ops = ufc_ops.HipOps()
    g0 = torch.Generator().manual_seed(4)
    corrs = [torch.randn(2, 1, h, h, h, h, generator=g0).to(dev).requires_grad_(True) for h in (16, 32, 64)]
    w = torch.cos(torch.arange(2 * 64 ** 4, device=dev, dtype=torch.float32)).view(2, 1, 64, 64, 64, 64)
    fused = ops.corr_mean3(corrs)
    got = torch.autograd.grad((fused * w).sum(), corrs)
    up = [getz._interp4d(x, 64, ops) for x in corrs]
    composed = ((up[0] + up[1]) + up[2]) / 3
    want = torch.autograd.grad((composed * w).sum(), corrs)
    assert float((fused - composed).abs().max()) <= 2.4e-7 * float(composed.abs().max())
    for a, b in zip(got, want):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), float((a - b).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("h,n", [(16, 64), (32, 64), (16, 32), (16, 16), (1, 8), (5, 13), (64, 16)])
def test_resize_adjoint_kernel_equals_library_backward(h, n):
    """cpn_resize_bilinear_ac_adjoint (deterministic gather) against aten's upsample_bilinear2d_backward (atomic scatter),
    the VJP of F.interpolate(mode='bilinear', align_corners=True) from (h,h) to (n,n) — up- and down-sampling."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from coponerf_amd import ufc_ops
    dev = torch.device("cuda:0")
    g = torch.randn(3, 7, n, n, device=dev)
    got = ufc_ops._resize_adjoint_hip(g, h)
    want = torch.ops.aten.upsample_bilinear2d_backward(g, [n, n], [3, 7, h, h], True, None, None)
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 5e-6 * max(1.0, float(want.abs().max()))      # a few ulps over <= 100 terms
    assert torch.equal(got, ufc_ops._resize_adjoint_hip(g, h))              # run-to-run identical


@pytest.mark.gpu
@pytest.mark.parametrize("R,O,I,ldy,ldx", [(256, 4, 4, 4, 4), (257, 64, 64, 64, 64), (1000, 128, 144, 128, 144),
                                           (4099, 68, 260, 72, 264), (8192, 256, 1024, 256, 1024),
                                           (32768, 1024, 256, 1024, 256), (16384, 128, 416, 128, 416)])
def test_linear_weight_gradient_kernel(R, O, I, ldy, ldx):
    """cpn_wgrad_f32 (dW = dY^T . X, db = column sums, row slabs summed in a fixed order) against float64 products: exact
    fp32 products and fp32 accumulation, so the error is that of an fp32 sum of R terms; run twice -> bit-identical;
    strided operands (a view with a wider row) give the same bits as their contiguous copies."""
    from coponerf_amd.ufc_ops import wgrad_f32
    dev = torch.device("cuda:0")
    dYw = syn.normal((R, ldy), seed=R + O).to(dev)
    Xw = syn.normal((R, ldx), seed=R + I + 1).to(dev)
    dY, X = dYw[:, :O], Xw[:, :I]
    dW, db = wgrad_f32(dY, X, True)
    ref_w = (dY.double().t() @ X.double())
    ref_b = dY.double().sum(0)
    scale = (R ** 0.5)
    assert float((dW.double() - ref_w).abs().max()) <= 2e-5 * scale
    assert float((db.double() - ref_b).abs().max()) <= 2e-5 * scale
    dW2, db2 = wgrad_f32(dY, X, True)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)
    dW3, db3 = wgrad_f32(dY.contiguous(), X.contiguous(), True)
    assert torch.equal(dW, dW3) and torch.equal(db, db3)
    dW4, none = wgrad_f32(dY, X, False)
    assert none is None and torch.equal(dW, dW4)


@pytest.mark.gpu
def test_linear_layer_backward_equals_library_backward():
    """ufc_ops.Linear (nn.Linear whose training backward runs dW / db on cpn_wgrad_f32): same forward bits as nn.Linear,
    gradients equal to the library's to fp32 summation order; small token counts and no-grad calls take the stock path."""
    from coponerf_amd.ufc_ops import Linear, LinearFn
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    ref = torch.nn.Linear(256, 1024).to(dev)
    mine = Linear(256, 1024).to(dev)
    mine.load_state_dict(ref.state_dict())
    x = syn.normal((2, 4096, 256), seed=9).to(dev)
    g = syn.normal((2, 4096, 1024), seed=10).to(dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = ref(xa), mine(xb)
    assert type(yb.grad_fn).__name__ == "LinearFnBackward"
    assert torch.equal(ya, yb)
    ya.backward(g)
    yb.backward(g)
    assert torch.equal(xa.grad, xb.grad)
    for pa, pb in zip(ref.parameters(), mine.parameters()):
        assert float((pa.grad - pb.grad).abs().max()) <= 2e-4 * float(pa.grad.abs().max())
    small = mine(x[:, :100].reshape(-1, 256).requires_grad_(True))
    assert type(small.grad_fn).__name__ != "LinearFnBackward"
    with torch.no_grad():
        assert mine(x).grad_fn is None


@pytest.mark.gpu
def test_one_launch_adam_equals_library_adam():
    """optim.OneLaunchAdam (cpn_adam_step: every tensor in one launch, moments in flat buffers) against torch.optim.Adam over
    four steps: odd sizes (vector body + scalar tail, one block and many), a tensor that has no gradient in step 2 (skipped,
    keeps its own bias-correction count), a gradient scale applied inside the update (= scaling the gradients first)."""
    from coponerf_amd.optim import OneLaunchAdam
    dev = torch.device("cuda:0")
    shapes = [(3,), (5, 1), (2049,), (4100,), (64, 33), (256, 1024), (1,)]
    mine = [torch.nn.Parameter(syn.normal(s, seed=40 + i).to(dev)) for i, s in enumerate(shapes)]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    a, b = OneLaunchAdam(mine, lr=2e-4), torch.optim.Adam(ref, lr=2e-4)
    for step in range(4):
        scale = torch.tensor(0.37 if step % 2 else 1.0, device=dev)
        for i, (p, q) in enumerate(zip(mine, ref)):
            if step == 2 and i == 3:
                p.grad = q.grad = None
                continue
            g = syn.normal(shapes[i], seed=100 * step + i).to(dev) * (10.0 ** (i - 3))
            p.grad, q.grad = g.clone(), g * scale
        a.step(gscale=scale)
        b.step()
        for i, (p, q) in enumerate(zip(mine, ref)):
            assert float((p - q).abs().max()) <= 2e-6 * float(q.abs().max()) + 1e-9, (step, i)
    for i, q in enumerate(ref):
        m, v = a.moments(i)
        st = b.state[q]
        assert float((m - st["exp_avg"]).abs().max()) <= 1e-6 * float(st["exp_avg"].abs().max()) + 1e-12
        assert float((v - st["exp_avg_sq"]).abs().max()) <= 1e-6 * float(st["exp_avg_sq"].abs().max()) + 1e-12
    assert a.steps.tolist() == [4, 4, 4, 3, 4, 4, 4]                     # kept on the device, per tensor
    # a gated step is a no-op on the device: parameters, moments and counts stay
    before = [p.detach().clone() for p in mine]
    for i, p in enumerate(mine):
        p.grad = syn.normal(shapes[i], seed=900 + i).to(dev)
    a.step(gate=torch.zeros((), device=dev))
    assert all(torch.equal(p, q) for p, q in zip(mine, before)) and a.steps.tolist() == [4, 4, 4, 3, 4, 4, 4]
    a.step(gate=torch.ones((), device=dev))
    assert not torch.equal(mine[5], before[5]) and a.steps.tolist() == [5, 5, 5, 4, 5, 5, 5]
    a.zero_grad()
    assert all(p.grad is None for p in mine)


@pytest.mark.gpu
def test_one_launch_adam_is_a_torch_optimizer():
    """the reference checkpoints `optimizer.state_dict()` and drives the learning rate with ExponentialLR(0.95) through
    its MultiLR wrapper (/root/reference train.py:102-108, wrapper.py:98, 134-136).  OneLaunchAdam: a torch.optim.Optimizer with
    param_groups a scheduler writes; state_dict() in torch.optim.Adam's layout, loadable by either optimizer; a resumed
    optimizer continues exactly; parameter versions are bumped by a step (the inference caches are keyed on them)."""
    from coponerf_amd.optim import OneLaunchAdam
    dev = torch.device("cuda:0")
    shapes = [(7,), (130, 3), (2049,), (64, 33)]
    new = lambda: [torch.nn.Parameter(syn.normal(s, seed=40 + i).to(dev)) for i, s in enumerate(shapes)]
    mine, ref = new(), new()
    a = OneLaunchAdam([{"params": mine[:2]}, {"params": mine[2:]}], lr=1e-3)
    b = torch.optim.Adam([{"params": ref[:2]}, {"params": ref[2:]}], lr=1e-3)
    assert isinstance(a, torch.optim.Optimizer) and len(a.param_groups) == 2
    sa, sb = torch.optim.lr_scheduler.ExponentialLR(a, 0.95), torch.optim.lr_scheduler.ExponentialLR(b, 0.95)

    def grads(ps, qs, step):
        for i, (p, q) in enumerate(zip(ps, qs)):
            if step == 1 and i == 1:
                p.grad = q.grad = None
                continue
            g = syn.normal(shapes[i], seed=100 * step + i).to(dev)
            p.grad, q.grad = g.clone(), g.clone()
    for step in range(3):
        grads(mine, ref, step)
        v0 = [p._version for p in mine]
        a.step()
        b.step()
        assert all(p._version > v for p, v, in zip(mine, v0) if p.grad is not None), "a step must bump the parameters' versions"
        sa.step()
        sb.step()
        assert a.param_groups[0]["lr"] == pytest.approx(b.param_groups[0]["lr"]) and a.lr == pytest.approx(1e-3 * 0.95 ** (step + 1))
    for p, q in zip(mine, ref):
        assert float((p - q).abs().max()) <= 2e-6 * float(q.abs().max()) + 1e-9
    sd_a, sd_b = a.state_dict(), b.state_dict()
    assert sd_a["param_groups"][1]["params"] == sd_b["param_groups"][1]["params"] == [2, 3]
    assert set(sd_a["state"]) == set(sd_b["state"]) == {0, 1, 2, 3}
    for k in sd_b["state"]:
        assert float(sd_a["state"][k]["step"]) == float(sd_b["state"][k]["step"]), k         # index 1 missed one update
        for name in ("exp_avg", "exp_avg_sq"):
            x, y = sd_a["state"][k][name], sd_b["state"][k][name]
            assert x.shape == y.shape and float((x - y).abs().max()) <= 1e-6 * float(y.abs().max()) + 1e-12
    # resume: a fresh OneLaunchAdam from ITS checkpoint and one from the library's continue like the library's
    cont = []
    for sd in (sd_a, sd_b):
        ps = [torch.nn.Parameter(p.detach().clone()) for p in mine]
        o = OneLaunchAdam([{"params": ps[:2]}, {"params": ps[2:]}], lr=123.0)
        o.load_state_dict(sd)
        assert o.lr == pytest.approx(a.lr) and o.steps.tolist() == [3, 2, 3, 3]
        cont.append((ps, o))
    lib = torch.optim.Adam([{"params": ref[:2]}, {"params": ref[2:]}], lr=1.0)
    lib.load_state_dict(sd_a)                                                 # and the library's Adam reads ours
    for step in (3, 4):
        for ps, o in cont:
            grads(ps, ref, step)
            o.step()
        lib.step()
        for ps, _ in cont:
            for p, q in zip(ps, ref):
                assert float((p - q).abs().max()) <= 2e-6 * float(q.abs().max()) + 1e-9
    a.param_groups[1]["lr"] = 0.5
    with pytest.raises(NotImplementedError):
        grads(mine, ref, 5)
        a.step()


@pytest.mark.gpu
def test_eval_after_training_sees_the_updated_weights():
    """the loop validates between training steps (/root/reference wrapper.py:134, 178-188).  Every
    derived-weight cache of the inference path is keyed on the parameters' versions, and the one-launch optimizer writes
    them through raw pointers: eval, train k steps, eval again must render the second image with the NEW weights - i.e.
    equal to a fresh model that loads the trained state_dict."""
    from coponerf_amd import CoPoNeRF
    from coponerf_amd.train_step import TrainStep
    dev = torch.device("cuda:0")
    model = CoPoNeRF.CoPoNeRF(n_view=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes), strict=True)
    model = model.to(dev)
    inp = to_device(syn.make_inputs(1, 256, 256, 128, seed=61), dev)

    def evaluate(m):
        m.eval()
        with torch.no_grad():
            z, rel, flow = m.get_z(inp)
            out = m(inp, z=z, rel_pose=rel, val=True, flow=flow)
        return out["rgb"].clone(), rel.clone()
    rgb0, rel0 = evaluate(model)
    model.train()
    step = TrainStep(model, lr=2e-3)
    for _ in range(3):
        r = step(inp, inp["query"]["rgb"])
    assert bool(r["stepped"])
    rgb1, rel1 = evaluate(model)
    fresh = CoPoNeRF.CoPoNeRF(n_view=2)
    fresh.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
    rgb2, rel2 = evaluate(fresh.to(dev))
    assert float((rgb1 - rgb0).abs().max()) > 1e-4, "three steps at lr 2e-3 must move the image"
    assert float((rgb1 - rgb2).abs().max()) <= 2e-6, float((rgb1 - rgb2).abs().max())
    assert float((rel1 - rel2).abs().max()) <= 1e-6


@pytest.mark.gpu
def test_train_step_guards_on_the_device():
    """TrainStep on one device (no exchange): the finite-gradient guard and the clip coefficient stay on the device, the
    update kernel is gated by the flag.  A step whose gradients hold a NaN changes nothing (parameters, moments, counts)
    and reports stepped == False when asked; the next clean step updates; no call of the step waits for the GPU."""
    from coponerf_amd import CoPoNeRF
    from coponerf_amd.train_step import TrainStep, _LazyFlag
    dev = torch.device("cuda:0")
    model = CoPoNeRF.CoPoNeRF(n_view=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes), strict=True)
    model = model.to(dev).train()
    inp = to_device(syn.make_inputs(1, 256, 256, 128, seed=61), dev)
    step = TrainStep(model)
    params = [p for p in model.parameters()]
    probe = model.query_encode_latent.weight
    w0 = [p.detach().clone() for p in params]
    hook = probe.register_hook(lambda g: g * float("nan"))
    r = step(inp, inp["query"]["rgb"])
    hook.remove()
    assert isinstance(r["stepped"], _LazyFlag) and bool(r["stepped"]) is False
    assert all(torch.equal(p.detach(), q) for p, q in zip(params, w0))
    assert int(step.opt.steps.max()) == 0 and float(step.opt.exp_avg.abs().max()) == 0.0
    assert all(p.grad is None for p in params)
    scale_before = model._engine.grad_scale_target
    r = step(inp, inp["query"]["rgb"])                      # the skipped step reaches the scale adaptation here
    assert bool(r["stepped"]) is True and step.skipped_total == 1
    assert model._engine.grad_scale_target <= scale_before
    idx = [i for i, p in enumerate(params) if p is probe][0]
    assert not torch.equal(probe.detach(), w0[idx])
    got = step.opt.steps
    assert int(got.max()) == 1 and int(got.min()) == 0      # the parameters without a gradient keep their zero
    r = step(inp, inp["query"]["rgb"])
    assert bool(r["stepped"]) and float(r["loss"]) == float(r["loss"])


@pytest.mark.gpu
@pytest.mark.parametrize("B,R,S,two", [(1, 5, 16, True), (2, 37, 64, True), (1, 300, 32, False), (3, 64, 128, True)])
def test_combine_gemm_equals_gemm_then_combine(B, R, S, two):
    """cpn_gemm_f16_combine (the key path's data-gradient GEMM with cpn_hid_grad_combine as its epilogue, round 6) gives the
    bits of cpn_gemm_f16 followed by cpn_hid_grad_combine: same MFMA sequence, same fp16 rounding of the product, same fp32
    sums — ragged tile tails (rows not a multiple of 256), one or two parked parts."""
    from coponerf_amd._hip import call
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    V, K = 2, 128
    rows = B * R * V * S
    g = torch.Generator().manual_seed(rows + S)
    dkh = (torch.randn(rows, K, generator=g) * 0.5).half().to(dev)
    Wt = (torch.randn(1664, K, generator=g) * 0.1).half().to(dev)
    hid = torch.relu(torch.randn(rows, 1664, generator=g)).half().to(dev)
    w1 = torch.rand(B * V, R, S, generator=g).to(dev)
    w2 = torch.rand(B * V, R, S, generator=g).to(dev)
    dh1 = torch.randn(B * R, 1664, generator=g).to(dev)
    dh2 = torch.randn(B * R, 1664, generator=g).to(dev)
    zero = torch.zeros(1664, device=dev)
    d = torch.empty(rows, 1664, dtype=torch.float16, device=dev)
    call("cpn_gemm_f16", dkh.data_ptr(), K, Wt.data_ptr(), K, zero.data_ptr(), d.data_ptr(), 1664, rows, 1664, K, 0, 0, st)
    want = torch.empty_like(d)
    p2 = (w2.data_ptr(), dh2.data_ptr()) if two else (0, 0)
    call("cpn_hid_grad_combine", d.data_ptr(), hid.data_ptr(), w1.data_ptr(), dh1.data_ptr(), p2[0], p2[1], B, V, R, S, 0, B * R,
         want.data_ptr(), st)
    got = torch.full_like(d, float("nan"))
    call("cpn_gemm_f16_combine", dkh.data_ptr(), K, Wt.data_ptr(), K, hid.data_ptr(), w1.data_ptr(), dh1.data_ptr(), p2[0], p2[1],
         B, V, R, S, 0, B * R, K, got.data_ptr(), st)
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    # and against float64 on a sample of rows
    idx = torch.arange(0, rows, max(1, rows // 64), device=dev)
    t = idx // (V * S)
    v = (idx // S) % V
    s_ = idx % S
    b, r = t // R, t % R
    wa = w1[b * V + v, r, s_][:, None].double()
    wb = w2[b * V + v, r, s_][:, None].double() if two else 0.0
    ref = dkh[idx].double() @ Wt.double().t() + wa * dh1[t].double() + (wb * dh2[t].double() if two else 0.0)
    ref = torch.where(hid[idx] > 0, ref, torch.zeros_like(ref))
    assert float((got[idx].double() - ref).abs().max()) <= 2e-2 * max(1.0, float(ref.abs().max()))


def test_fused_combine_training_matches_two_kernel_form(dev, monkeypatch):
    """render_train with COPONERF_FUSE_COMBINE on / off: the same gradients (the fused epilogue is bit-identical)."""
    from coponerf_amd import CoPoNeRF, train_fns
    B, H, R, S = 2, 64, 40, 32
    weights = syn.make_render_weights(seed=31)
    inp = to_device(syn.make_inputs(B, H, H, R, seed=65), dev)
    z, rel, flow = syn.make_latents(B, H, H, seed=66)
    coef = syn.normal((B, 1, R, 3), seed=67).to(dev)
    res = []
    for fuse in (True, False):
        monkeypatch.setattr(train_fns, "FUSE_COMBINE", fuse)
        model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
        model.load_state_dict(weights, strict=False)
        model = model.to(dev).train()
        zz = [t.to(dev).requires_grad_(True) for t in z]
        out = model(inp, z=zz, rel_pose=rel.to(dev), val=False, flow=to_device(flow, dev))
        (out["rgb"] * coef).sum().backward()
        grads = {"z%d" % i: t.grad for i, t in enumerate(zz)}
        grads.update({k: p.grad for k, p in model.named_parameters() if p.grad is not None})
        res.append(grads)
    assert set(res[0]) == set(res[1]) and "query_encode_latent.weight" in res[0]
    for k in res[0]:                                         # (fp32 atomics in the level-3 scatter: not bit-reproducible)
        rel_err = float((res[0][k] - res[1][k]).norm() / (res[1][k].norm() + 1e-30))
        assert rel_err <= 1e-5, (k, rel_err)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(16, 128, 128), (4000, 128, 128), (272, 1664, 128), (1040, 416, 64)])
def test_masked_data_gradient_gemm(M, N, K):
    """cpn_gemm_f16_masked = cpn_gemm_f16 followed by the ReLU mask of the layer's input, bit for bit."""
    from coponerf_amd._hip import call
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).half().to(dev)
    Wt = (torch.randn(N, K, generator=g) * 0.2).half().to(dev)
    mask = torch.relu(torch.randn(M, N, generator=g)).half().to(dev)
    zero = torch.zeros(N, device=dev)
    want = torch.empty(M, N, dtype=torch.float16, device=dev)
    call("cpn_gemm_f16", A.data_ptr(), K, Wt.data_ptr(), K, zero.data_ptr(), want.data_ptr(), N, M, N, K, 0, 0, st)
    want = torch.where(mask > 0, want, torch.zeros_like(want))
    got = torch.full_like(want, float("nan"))
    call("cpn_gemm_f16_masked", A.data_ptr(), K, Wt.data_ptr(), K, mask.data_ptr(), N, got.data_ptr(), N, M, N, K, st)
    assert torch.equal(got, want)
    with pytest.raises(RuntimeError, match="M %"):
        call("cpn_gemm_f16_masked", A.data_ptr(), K, Wt.data_ptr(), K, mask.data_ptr(), N, got.data_ptr(), N, 17, N, K, st)


def test_backward_fusions_leave_the_gradients_alone(dev, monkeypatch):
    """render_train with the round-6 backward fusions (shared coords_embed gradient summed inside cpn_attend_hidden_bwd, the
    ReLU mask as the epilogue of key_map_2's data-gradient GEMM) against autograd's accumulate / threshold passes."""
    from coponerf_amd import CoPoNeRF, train_fns
    B, H, R, S = 2, 64, 40, 32
    weights = syn.make_render_weights(seed=35)
    inp = to_device(syn.make_inputs(B, H, H, R, seed=85), dev)
    z, rel, flow = syn.make_latents(B, H, H, seed=86)
    coef = syn.normal((B, 1, R, 3), seed=87).to(dev)
    cw = syn.normal((B * 2, R, S), seed=88).to(dev)
    res = []
    for on in (True, False):
        monkeypatch.setattr(train_fns, "SHARE_QB_GRAD", on)
        monkeypatch.setattr(train_fns, "MASKED_DGRAD", on)
        model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
        model.load_state_dict(weights, strict=False)
        model = model.to(dev).train()
        zz = [t.to(dev).requires_grad_(True) for t in z]
        out = model(inp, z=zz, rel_pose=rel.to(dev), val=False, flow=to_device(flow, dev))
        _loss(out, coef, cw).backward()
        grads = {"z%d" % i: t.grad for i, t in enumerate(zz)}
        grads.update({k: p.grad for k, p in model.named_parameters() if p.grad is not None})
        res.append(grads)
    assert set(res[0]) == set(res[1]) and "query_embed.weight" in res[0] and "key_map.weight" in res[0]
    for k in res[0]:                                         # the shared gradient is summed in fp32 before ONE fp16 rounding
        rel_err = float((res[0][k] - res[1][k]).norm() / (res[1][k].norm() + 1e-30))
        assert rel_err <= 2e-3, (k, rel_err)


def test_local_hidden_backward_kernel(dev):
    """cpn_local_hidden_bwd (ReLU mask, un-scaling, the 128 x 16 weight gradient, the bias gradient and the per-ray sums in
    one pass, on the fp32 MFMA) against the same sums formed by float64 tensor ops; shapes with rays of 48 and 128 rows (a
    partial and two full 64-row chunks)."""
    from coponerf_amd._hip import call
    st = torch.cuda.current_stream().cuda_stream
    for (B, V, R, S, scale) in ((2, 2, 5, 24, 8.0), (1, 2, 3, 64, 0.5), (3, 2, 9, 64, 1.0)):
        nrays, rpr = B * R, V * S
        rows = nrays * rpr
        ds = (syn.normal((rows, 128), seed=81) * 0.5).to(dev).half()
        out = syn.normal((rows, 128), seed=82).to(dev).half()
        loc8 = syn.normal((B * V * R * S, 8), seed=83).to(dev)
        coords9 = syn.normal((B * V * R, 9), seed=84).to(dev)
        sc_t = torch.tensor([scale], device=dev)
        dW = torch.zeros(128, 16, device=dev)
        db = torch.zeros(128, device=dev)
        dadd = torch.empty(nrays, 128, device=dev)
        call("cpn_local_hidden_bwd", ds.data_ptr(), out.data_ptr(), loc8.data_ptr(), coords9.data_ptr(), sc_t.data_ptr(), B, V, R, S,
             dW.data_ptr(), db.data_ptr(), dadd.data_ptr(), st)
        # row (ray, v, s) of the launch -> L from loc8[((b V + v) R + r) S + s] and coords9[(b V + v) R + r]
        ray = torch.arange(nrays, device=dev).repeat_interleave(rpr)
        m = torch.arange(rpr, device=dev).repeat(nrays)
        v, s = m // S, m % S
        b, r = ray // R, ray % R
        nr = (b * V + v) * R + r
        l8, c9 = loc8[nr * S + s].double(), coords9[nr].double()
        z = torch.zeros(rows, dtype=torch.float64, device=dev)
        L = torch.stack([l8[:, 0], l8[:, 1], l8[:, 2], z, z, z, c9[:, 0], c9[:, 1], c9[:, 2], l8[:, 3], l8[:, 4], l8[:, 5], l8[:, 6],
                         c9[:, 6], c9[:, 7], c9[:, 8]], dim=1)
        d = torch.where(out.double() > 0, ds.double(), torch.zeros((), dtype=torch.float64, device=dev)) / scale
        want_dW, want_db = d.t() @ L, d.sum(0)
        want_dadd = d.view(nrays, rpr, 128).sum(1)
        tol = lambda w: 2e-5 * (1 + w.abs().max())
        assert (dW.double() - want_dW).abs().max() <= tol(want_dW), (B, V, R, S)
        assert (db.double() - want_db).abs().max() <= tol(want_db), (B, V, R, S)
        assert (dadd.double() - want_dadd).abs().max() <= tol(want_dadd), (B, V, R, S)
