"""bf16-MFMA kernels (BASELINE configs[4]).  New functionality with no reference semantics beyond "fp32 math, rounded"
(SURVEY.md section 2.1): parity unpinned by the reference.  What IS pinned: each kernel must equal the fp32-accumulated
result on operands rounded to bf16 (nearest even) -- checked against fp64 on the rounded operands to 1e-5 normwise --
and must stay within the stated distance (2e-2 normwise, measured 2-4e-3) of the fp64 result on the unrounded operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from drqv2_amd import ops as o
    return o


def r16(t):
    return t.to(torch.bfloat16).to(torch.float64)


def gerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("hin,nb", [(41, 3), (39, 2), (37, 5), (41, 200)])
def test_conv_bf16_fwd_dgrad_wgrad(ops, hin, nb):
    import torch.nn.functional as Fn
    g = torch.Generator(device="cuda").manual_seed(hin + nb)
    rn = lambda *sh: torch.randn(*sh, device="cuda", generator=g)
    hout = hin - 2
    x, w, b = rn(nb, 32, hin, hin), rn(32, 32, 3, 3) * 0.1, rn(32) * 0.1
    # forward
    y = ops.conv3x3_fwd(x, w, b, 1, bf16=True)
    ref_r = torch.relu(Fn.conv2d(r16(x), r16(w), b.double()))
    ref = torch.relu(Fn.conv2d(x.double(), w.double(), b.double()))
    assert gerr(y, ref_r) <= 1e-5, gerr(y, ref_r)
    assert gerr(y, ref) <= 2e-2
    # dgrad (layer output size hout; dy stored zero-padded by 2) with ReLU mask
    dy = rn(nb, 32, hout, hout)
    dy_pad = torch.zeros(nb, 32, hout + 4, hout + 4, device="cuda")
    dy_pad[:, :, 2:-2, 2:-2] = dy
    mask = rn(nb, 32, hin, hin)
    dx = ops.conv3x3_dgrad(dy_pad, w, mask, bf16=True)
    gx_r = Fn.conv_transpose2d(r16(dy), r16(w)) * (mask.double() > 0)
    gx = Fn.conv_transpose2d(dy.double(), w.double()) * (mask.double() > 0)
    assert gerr(dx, gx_r) <= 1e-5, gerr(dx, gx_r)
    assert gerr(dx, gx) <= 2e-2
    # wgrad + bias gradient (the bias gradient sums the UNROUNDED dy)
    dw, db = ops.conv3x3_wgrad(x, dy_pad[:, :, 2:-2, 2:-2], 1, bf16=True)
    wd = w.double().requires_grad_(True)
    (gw_r,) = torch.autograd.grad(Fn.conv2d(r16(x), wd), wd, r16(dy))
    (gw,) = torch.autograd.grad(Fn.conv2d(x.double(), wd), wd, dy.double())
    assert gerr(dw, gw_r) <= 1e-5, gerr(dw, gw_r)
    assert gerr(dw, gw) <= 2e-2
    assert gerr(db, dy.double().sum((0, 2, 3))) <= 3e-6


@pytest.mark.parametrize("hin,nb", [(41, 3), (39, 2), (37, 5), (41, 130)])
def test_conv_bf16_channel_contiguous_layout_is_the_same_arithmetic(ops, hin, nb):
    """The layout the bf16 update keeps between the encoder layers (bf16 [frame][y][x][32 channels]): a value stored in
    it is the rounding the fp32-storage kernels apply when they stage the value, so forward (either side in either
    layout), masked input gradient and weight gradient must equal the fp32-storage kernels BIT FOR BIT."""
    g = torch.Generator(device="cuda").manual_seed(7 * hin + nb)
    rn = lambda *sh: torch.randn(*sh, device="cuda", generator=g)
    hout = hin - 2
    x, w, b = rn(nb, 32, hin, hin), rn(32, 32, 3, 3) * 0.1, rn(32) * 0.1
    xn = ops.to_nhwc_bf16(x)
    assert torch.equal(ops.from_nhwc_bf16(xn), x.to(torch.bfloat16).float())          # the helpers invert each other
    y = ops.conv3x3_fwd(x, w, b, 1, bf16=True)                                         # fp32 in, fp32 out
    y_r = y.to(torch.bfloat16).float()
    for x_in in (x, xn):
        yo = ops.conv3x3_fwd_bf16_nhwc(x_in, w, b, relu=True, y_nhwc=False)
        assert torch.equal(yo, y)
        yn = ops.conv3x3_fwd_bf16_nhwc(x_in, w, b, relu=True, y_nhwc=True)
        assert torch.equal(ops.from_nhwc_bf16(yn), y_r)
    # without ReLU too (the sign of a rounded value is the value's)
    y0 = ops.conv3x3_fwd(x, w, b, 1, relu=False, bf16=True)
    assert torch.equal(ops.from_nhwc_bf16(ops.conv3x3_fwd_bf16_nhwc(xn, w, b, relu=False)), y0.to(torch.bfloat16).float())
    # masked input gradient: the mask in the layout
    dy = rn(nb, 32, hout, hout)
    dy_pad = torch.zeros(nb, 32, hout + 4, hout + 4, device="cuda")
    dy_pad[:, :, 2:-2, 2:-2] = dy
    mask = torch.relu(rn(nb, 32, hin, hin))                                            # an activation: zeros and positives
    dx = ops.conv3x3_dgrad(dy_pad, w, mask, bf16=True)
    assert torch.equal(ops.conv3x3_dgrad_bf16_nhwc(dy_pad, w, ops.to_nhwc_bf16(mask)), dx)
    # ... the incoming gradient in the layout too (bf16, zero border of 2), and the result written into the interior of
    # a padded bf16 buffer in the layout (what the next input gradient and the weight gradient read)
    dyn = ops.to_nhwc_bf16(dy_pad)
    maskn = ops.to_nhwc_bf16(mask)
    assert torch.equal(ops.conv3x3_dgrad_bf16_nhwc(dyn, w, maskn), dx)
    for d_in in (dy_pad, dyn):
        buf = ops.conv3x3_dgrad_bf16_nhwc(d_in, w, maskn, dx_nhwc=True)
        full = ops.from_nhwc_bf16(buf)
        assert torch.equal(full[:, :, 2:-2, 2:-2], dx.to(torch.bfloat16).float())
        full[:, :, 2:-2, 2:-2] = 0
        assert not bool(full.any())                                                    # the border stays zero
    # weight gradient: the layer input in the layout
    dw, db = ops.conv3x3_wgrad(x, dy_pad[:, :, 2:-2, 2:-2], 1, bf16=True)
    dwn, dbn = ops.conv3x3_wgrad_bf16_nhwc(xn, dy_pad[:, :, 2:-2, 2:-2])
    assert torch.equal(dwn, dw) and torch.equal(dbn, db)
    # ... and the gradient as the padded bf16 buffer: the same products; the bias gradient sums the rounded values
    dwp, dbp = ops.conv3x3_wgrad_bf16_nhwc(xn, dyn)
    assert torch.equal(dwp, dw)
    assert gerr(dbp, r16(dy).sum((0, 2, 3))) <= 3e-6


@pytest.mark.parametrize("n", [2, 70])
def test_fused_aug_conv1_bf16(ops, n):
    """The bf16-MFMA form of the fused aug + conv1 launch: the stored encoder input is the fp32 one bit for bit; the
    layer output equals the fp32-accumulated convolution of that input and the weights rounded to bf16."""
    import torch.nn.functional as Fn
    g = torch.Generator().manual_seed(n)
    obs = torch.randint(0, 256, (n, 9, 84, 84), generator=g, dtype=torch.uint8).cuda()
    obs1 = torch.randint(0, 256, (n, 9, 84, 84), generator=g, dtype=torch.uint8).cuda()
    sh = torch.randint(0, 9, (n, 2), generator=g).float().cuda()
    sh1 = torch.randint(0, 9, (n, 2), generator=g).float().cuda()
    w = (torch.randn(32, 9, 3, 3, generator=g) * 0.2).cuda()
    b = (torch.randn(32, generator=g) * 0.1).cuda()
    y, xaug = ops.conv1_aug_fwd(obs, sh, obs1, sh1, w, b, n_store=2 * n, bf16=True)
    y32, xaug32 = ops.conv1_aug_fwd(obs, sh, obs1, sh1, w, b, n_store=2 * n)
    assert torch.equal(xaug, xaug32)
    ref_r = torch.relu(Fn.conv2d(r16(xaug32), r16(w), b.double(), stride=2))
    assert gerr(y, ref_r) <= 1e-5, gerr(y, ref_r)
    assert gerr(y, y32.double()) <= 2e-2
    # the same launch writing its output in the bf16 [frame][y][x][32] layout: y rounded, nothing else
    yn, xaug_n = ops.conv1_aug_fwd(obs, sh, obs1, sh1, w, b, n_store=2 * n, bf16=True, y_nhwc=True)
    assert torch.equal(xaug_n, xaug)
    assert torch.equal(ops.from_nhwc_bf16(yn), y.to(torch.bfloat16).float())


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("M,N,K", [(256, 1024, 1024), (70, 100, 56), (33, 21, 130), (512, 64, 2048)])
def test_gemm_bf16_three_layouts(ops, M, N, K):
    """forward (x W^T + b, relu), dgrad (dy W with mask), wgrad (dy^T x with bias gradient): the shapes of the update
    incl. ragged ones (K = 56 / 62, N = A) and a split-K case."""
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    # forward: A = x [M][K] (k-contiguous), B = w [N][K] (k-contiguous)
    (y,), _ = ops.gemm_batched([x], True, [w], True, M, N, K, K, K, biases=[b], relu=True, bf16=True)
    ref_r = torch.relu(r16(x) @ r16(w).T + b.double())
    ref = torch.relu(x.double() @ w.double().T + b.double())
    assert gerr(y, ref_r) <= 1e-5, gerr(y, ref_r)
    assert gerr(y, ref) <= 2e-2
    # dgrad: dx[M][K] = dy[M][N] w[N][K], masked: A = dy (k = N contiguous), B(k,n) = w[k*ldb + n]
    dy, mask = rnd(M, N, seed=4), rnd(M, K, seed=5)
    (dx,), _ = ops.gemm_batched([dy], True, [w], False, M, K, N, N, K, auxs=[mask], bf16=True)
    gx_r = (r16(dy) @ r16(w)) * (mask.double() > 0)
    assert gerr(dx, gx_r) <= 1e-5, gerr(dx, gx_r)
    assert gerr(dx, (dy.double() @ w.double()) * (mask.double() > 0)) <= 2e-2
    # wgrad: dw[N][K] = dy^T x, db = column sums of dy: A(m=n_out, k=batch) = dy[k*lda + m], B(k, n) = x[k*ldb + n]
    (dw,), (db,) = ops.gemm_batched([dy], False, [x], False, N, K, M, N, K, rowsum=True, bf16=True)
    assert gerr(dw, r16(dy).T @ r16(x)) <= 1e-5
    assert gerr(dw, dy.double().T @ x.double()) <= 2e-2
    assert gerr(db, dy.double().sum(0)) <= 3e-6


def test_gemm_bf16_first_layer_wgrad_with_split_k(ops):
    """Weight gradient of a first layer (few output tiles, long batch reduction): split-K plus a separate column-sum
    pass for the bias gradient."""
    Brows, N, K = 2048, 1024, 121
    dy, x = rnd(Brows, N, seed=50), rnd(Brows, K, seed=51)
    (dw,), (db,) = ops.gemm_batched([dy], False, [x], False, N, K, Brows, N, K, rowsum=True, bf16=True)
    assert gerr(dw, r16(dy).T @ r16(x)) <= 1e-5
    assert gerr(db, dy.double().sum(0)) <= 3e-6


@pytest.mark.parametrize("F", [50, 100])
def test_gemm_bf16_trunk_shapes(ops, F):
    """Linear(39200 -> F): forward with split-K (several problems at once; F = 100 takes the 128-column tile), weight
    gradient, and the input gradient scattered into the zero-padded conv-gradient layout with the ReLU mask."""
    B, R, hw = 64, 39200, 35
    feat = [rnd(B, R, seed=10 + i, scale=0.05) for i in range(2)]
    w = [rnd(F, R, seed=20 + i, scale=R ** -0.5) for i in range(2)]
    bias = [rnd(F, seed=30 + i) for i in range(2)]
    ys, _ = ops.gemm_batched(feat, True, w, True, B, F, R, R, R, biases=bias, bf16=True)
    for i in range(2):
        assert gerr(ys[i], r16(feat[i]) @ r16(w[i]).T + bias[i].double()) <= 1e-5
        assert gerr(ys[i], feat[i].double() @ w[i].double().T + bias[i].double()) <= 2e-2
    dz = rnd(B, F, seed=40)
    (dw,), (db,) = ops.gemm_batched([dz], False, [feat[0]], False, F, R, B, F, R, rowsum=True, bf16=True)
    assert gerr(dw, r16(dz).T @ r16(feat[0])) <= 1e-5 and gerr(db, dz.double().sum(0)) <= 3e-6
    pad = torch.zeros(B, 32, hw + 4, hw + 4, device="cuda")
    ops.gemm_batched([dz], True, [w[0]], False, B, R, F, F, R, auxs=[feat[0]], scatter_hw=hw, Cs=[pad], bf16=True)
    want = (r16(dz) @ r16(w[0])) * (feat[0].double() > 0)
    assert gerr(pad[:, :, 2:-2, 2:-2].reshape(B, R), want) <= 1e-5
    border = pad.clone()
    border[:, :, 2:-2, 2:-2] = 0
    assert not bool(border.any())


def _bf16_update_checks(name, ag, cfg, m, m32, m64, o64):
    """metrics 1e-2 relative; features 2e-2 normwise; every weight tensor (>= 1024 elements) cosine >= 0.97 and |g| within
    10 % (actor: 0.95, 25 %); each network's whole gradient cosine >= 0.98 (actor 0.95)."""
    worst = 0.0
    for k in m64:
        e = abs(m[k] - m64[k]) / max(1.0, abs(m64[k]))
        worst = max(worst, e)
        assert e <= 1e-2, (k, m[k], m32[k], m64[k])
    B = cfg["B"]
    feat = ag._engine.ws_view("FEAT", B, (2 * B, 39200))
    ferr = gerr(feat[:B], o64.last["feat"])
    assert ferr <= 2e-2, ferr
    cmin, nmax, netcos = 1.0, 0.0, {}
    cosf = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))
    for nm, mod, key in (("enc", ag.encoder, "g_enc"), ("critic", ag.critic, "g_critic"), ("actor", ag.actor, "g_actor")):
        allg, allr = [], []
        for (pn, p), g64 in zip(mod.named_parameters(), o64.last[key].values()):
            a, b = p.grad.detach().double().cpu().reshape(-1), g64.reshape(-1)
            cos = cosf(a, b)
            nr = abs(float(a.norm() / b.norm().clamp_min(1e-300)) - 1.0)
            if a.numel() >= 1024:          # biases / scalars are small sums of cancelling terms: whole-network check only
                cmin, nmax = min(cmin, cos), max(nmax, nr)
                # (actor: its gradient is taken through the critic after the critic's sign-like first Adam step, which
                # amplifies every upstream difference -- 1e-3 already in fp32, SURVEY finding 3)
                assert cos >= (0.95 if nm == "actor" else 0.97) and nr <= (0.25 if nm == "actor" else 0.1), (nm, pn, cos, nr)
            allg.append(a); allr.append(b)
        netcos[nm] = cosf(torch.cat(allg), torch.cat(allr))
        # the actor's gradient is taken through the critic AFTER its Adam step at t=1 (a sign-like step that
        # amplifies every upstream difference: SURVEY finding 3; 1e-3 already in fp32)
        assert netcos[nm] >= (0.95 if nm == "actor" else 0.98), (nm, netcos[nm])
    print(f"{name} bf16: worst metric rel err {worst:.2e}, features {ferr:.2e}, min tensor cosine {cmin:.5f}, "
          f"max norm deviation {nmax:.2e}, whole-network cosines {netcos}")


@pytest.mark.parametrize("name", ["cheetah_b64", "humanoid_b32"])
def test_bf16_update_against_fp64_oracle(name):
    """Whole update with set_compute_dtype("bf16") against the fp64 oracle of the reference's fp32 arithmetic, same
    batch, shifts and noise.  There is no reference behaviour to match here (parity unpinned); the bounds are this
    implementation's measured distances with a margin (see _bf16_update_checks).
    Element-wise agreement is not the criterion: bf16 rounding of the activations flips a fraction of the ReLU
    decisions near zero, and the bias gradients are small sums of large cancelling terms."""
    from tests.test_hip_step import WIDE, make_agent, make_oracle, run_hip
    cfg = WIDE[name]
    ag = make_agent(cfg).set_compute_dtype("bf16")
    ref = make_agent(cfg)                                   # the fp32 HIP path on the same inputs
    o64 = make_oracle(cfg, torch.float64)
    m, batch, (sh_o, sh_n, n_c, n_a) = run_hip(ag, cfg, 0)
    m32, _, _ = run_hip(ref, cfg, 0)
    m64 = o64.update(batch, cfg["step0"], sh_o, sh_n, n_c, n_a, keep=True)
    _bf16_update_checks(name, ag, cfg, m, m32, m64, o64)
    # the parameters moved and stayed finite; a second bf16 update runs from the updated state
    m2, _, _ = run_hip(ag, cfg, 1)
    assert all(v == v and abs(v) < 1e6 for v in m2.values())
    assert bool(torch.isfinite(ag._engine.params).all())


@pytest.mark.parametrize("name", ["cheetah_b64", "humanoid_b32", "cheetah_b256"])
def test_bf16_update_activation_storage_does_not_change_the_update(name):
    """The bf16 update keeps the outputs of conv1..conv3 as bf16 [frame][y][x][32]; DRQ_STEP_BF16_FP32_ACTS keeps them
    fp32 NCHW (round 2's storage, rounded when staged).  Same operands in the same sums: two updates from the same
    state must leave the same parameters, Adam moments and metrics bit for bit (the gradients between the layers fp32
    in both runs: that switch has its own test below)."""
    from tests.test_hip_step import WIDE, make_agent, run_hip
    cfg = WIDE[name]
    outs = []
    for flags in (8, 12):
        ag = make_agent(cfg).set_compute_dtype("bf16")
        ag._engine.step_flags = flags
        ms = [run_hip(ag, cfg, u)[0] for u in range(2)]
        torch.cuda.synchronize()
        eng = ag._engine
        outs.append((ms, eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.grads.clone()))
    (m0, *a0), (m1, *a1) = outs
    assert m0 == m1
    for x, y in zip(a0, a1):
        assert torch.equal(x, y)


@pytest.mark.parametrize("name", ["cheetah_b64", "humanoid_b32"])
def test_bf16_update_gradient_storage_changes_two_bias_gradients_only(name):
    """DRQ_STEP_BF16_FP32_GRADS keeps the gradients handed between the encoder's input-gradient launches fp32 instead
    of bf16 [frame][y][x][32].  The operands of every product are the same, so the first update's metrics and every
    gradient are identical bit for bit -- except the bias gradients of conv2 and conv3, which sum those gradients' values
    (rounded in one form, unrounded in the other; sums of cancelling terms: within 1e-2 of their norm, measured 1.6e-3)."""
    from tests.test_hip_step import WIDE, make_agent, run_hip
    cfg = WIDE[name]
    outs = []
    for flags in (0, 8):
        ag = make_agent(cfg).set_compute_dtype("bf16")
        ag._engine.step_flags = flags
        m = run_hip(ag, cfg, 0)[0]
        torch.cuda.synchronize()
        outs.append((m, {n: p.grad.detach().clone() for n, p in ag.encoder.named_parameters()},
                     ag._engine.grads.clone(), ag._engine.layout["seg"]["enc"]))
    (m0, e0, g0, seg), (m1, e1, g1, _) = outs
    assert m0 == m1
    assert torch.equal(g0[seg[1]:], g1[seg[1]:])                  # critic, actor: everything behind the encoder segment
    names = list(e0)
    bias_23 = [n for n in names if n.endswith("bias")][1:3]       # conv2, conv3 (convnet.2, convnet.4)
    for n in names:
        if n in bias_23:
            assert gerr(e0[n], e1[n]) <= 1e-2, (n, gerr(e0[n], e1[n]))
        else:
            assert torch.equal(e0[n], e1[n]), n


def test_config5_shape_humanoid_batch_2048_fp32_and_bf16():
    """BASELINE configs[4] at ITS shape: humanoid_run (A=21, feature_dim=100), batch_size=2048, one GPU.
    (a) the fp32 update at that shape against the pinned oracle, held to the bounds of test_wide_batch_update_matches_oracle
        (metrics 1e-5, features 2e-6, gradients no worse than 2x the fp32 oracle's own error against fp64; ReLU decisions
        that differ from the fp64 oracle's counted and inside the fp32 error bound) -- so the shape is oracle-covered;
    (b) the bf16 update on the same batch, shifts and noise against the same fp64 oracle run WITHOUT injected decisions,
        bounds of _bf16_update_checks (new functionality: parity unpinned by the reference)."""
    from tests.test_hip_step import (make_agent, make_oracle, run_hip, check_encoder_inputs_bitwise, check_relu_decisions,
                                     hip_masks, check_critic_decisions, nerr)
    cfg = dict(C=9, A=21, F=100, H=1024, B=2048, lr=8e-5, sched="linear(1.0,0.1,2000000)", wseed=11, bseed=110,
               updates=1, step0=1000, smooth=True)
    B = cfg["B"]
    # ---- (a) fp32
    ag = make_agent(cfg)
    o32, o64 = make_oracle(cfg, torch.float32), make_oracle(cfg, torch.float64)
    m, batch, (sh_o, sh_n, n_c, n_a) = run_hip(ag, cfg, 0)
    xin = check_encoder_inputs_bitwise(ag, cfg, batch, sh_o, sh_n)
    flips, units = check_relu_decisions(ag, cfg, o64, xin[:B])
    acts, crit_masks = hip_masks(ag, cfg)
    critic_before = {k: v.clone() for k, v in o64.critic.items()}
    kw = dict(enc_in_override=(xin[:B], xin[B:]), keep=True, relu_masks=acts, critic_relu_masks=crit_masks)
    m32 = o32.update(batch, cfg["step0"], sh_o, sh_n, n_c, n_a, **kw)
    m64 = o64.update(batch, cfg["step0"], sh_o, sh_n, n_c, n_a, **kw)
    cflips, cunits = check_critic_decisions(o64, critic_before, crit_masks)
    print(f"humanoid_b2048 fp32: {flips}/{units} encoder and {cflips}/{cunits} critic ReLU decisions differ from fp64")
    for k in m64:
        assert m[k] == pytest.approx(m64[k], rel=1e-5, abs=1e-5), (k, m[k], m32[k], m64[k])
    feat = ag._engine.ws_view("FEAT", B, (2 * B, 39200))
    assert nerr(feat[:B], o64.last["feat"]) <= 2e-6
    for nm, mod, key in (("enc", ag.encoder, "g_enc"), ("critic", ag.critic, "g_critic"), ("actor", ag.actor, "g_actor")):
        for (pn, p), g64, g32 in zip(mod.named_parameters(), o64.last[key].values(), o32.last[key].values()):
            e_hip, e_o32 = nerr(p.grad, g64), nerr(g32, g64)
            assert e_hip <= max(2.0 * e_o32, 2e-5 if nm != "actor" else 2e-3), (nm, pn, e_hip, e_o32)
    del ag, o32, xin, acts, crit_masks
    torch.cuda.empty_cache()
    # ---- (b) bf16, against the fp64 oracle's own (un-injected) update
    o64b = make_oracle(cfg, torch.float64)
    m64b = o64b.update(batch, cfg["step0"], sh_o, sh_n, n_c, n_a, keep=True)
    agb = make_agent(cfg).set_compute_dtype("bf16")
    mb, _, _ = run_hip(agb, cfg, 0)
    _bf16_update_checks("humanoid_b2048", agb, cfg, mb, m, m64b, o64b)
    m2, _, _ = run_hip(agb, cfg, 1)
    assert all(v == v and abs(v) < 1e6 for v in m2.values())
    assert bool(torch.isfinite(agb._engine.params).all())
