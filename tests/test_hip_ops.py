"""Op-level parity of the HIP kernels (through the C ABI) against fp64 references and the oracle.
Tolerances follow SURVEY.md App. B: every kernel's error vs an fp64 evaluation of the same op, with
identical injected inputs, must be at the fp32 rounding floor (normwise <= 1e-5, most <= 2e-6)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from drqv2_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    from drqv2_amd import ops as o, _lib
    _lib.load()
    assert torch.cuda.is_available()
    return o


def nerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rnd(*shape, seed=0, scale=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))


# ---------------------------------------------------------------------------------------------
def test_aug_vs_oracle_and_reference(ops):
    from oracle import drq_oracle as O
    d = np.load(os.path.join(G, "aug.npz"))
    base = torch.from_numpy(d["base_grid"])
    for nm, obs in (("smooth", synth.make_batch(4, 1, 9, seed=3, smooth=True)[0]),
                    ("noise", synth.make_batch(4, 1, 9, seed=4, smooth=False)[0])):
        sh = torch.from_numpy(d[f"{nm}_shifts"])
        out = ops.random_shifts_aug(obs.cuda(), sh.float().cuda(), 4, base.cuda()).cpu()
        ora = O.random_shifts_aug(obs.float(), sh, 4, base)
        d_ora = (out - ora).abs().max().item()
        print(f"aug[{nm}] max |hip - oracle| = {d_ora:.3e} (0..255 scale), exact={torch.equal(out, ora)}")
        assert torch.equal(out, ora)                             # same scalar formula, same rounding order: bit-exact
        ref = torch.from_numpy(d[f"{nm}_sub"])                     # the reference's own output
        assert (out[:, ::4, ::5, ::3] - ref).abs().max().item() <= 1e-3
        crop = O.aug_integer_crop(obs.float(), sh)
        assert (out - crop).abs().max().item() <= 4e-3
        # float-input entry (RandomShiftsAug.forward is handed obs.float())
        out_f = ops.random_shifts_aug(obs.float().cuda(), sh.float().cuda(), 4, base.cuda()).cpu()
        assert torch.equal(out_f, out)
        # fused /255 - 0.5
        out_n = ops.random_shifts_aug(obs.cuda(), sh.float().cuda(), 4, base.cuda(), fuse_norm=True).cpu()
        assert torch.equal(out_n, out / 255.0 - 0.5)


def test_aug_fused_normalisation_equals_true_division(ops):
    """obs / 255.0 - 0.5 (drqv2.py:54) fused into the augmentation: the kernel's division-free sequence must give
    the correctly rounded quotient on every value a training batch produces (uint8 and float entry, both kernels)."""
    obs = synth.make_batch(64, 1, 9, seed=21, smooth=False)[0]
    sh = torch.from_numpy(np.random.RandomState(5).randint(0, 9, (64, 2))).float()
    out = ops.random_shifts_aug(obs.cuda(), sh.cuda(), 4).cpu()
    out_n = ops.random_shifts_aug(obs.cuda(), sh.cuda(), 4, fuse_norm=True).cpu()
    assert torch.equal(out_n, out / 255.0 - 0.5)
    assert torch.equal(ops.random_shifts_aug(obs.float().cuda(), sh.cuda(), 4).cpu(), out)


def test_aug_shift_indices_bit_exact_all_81(ops):
    """every (sx,sy) in [0,8]^2: the output is the integer crop up to fp32 dust -> rounding recovers
    the exact uint8 crop, i.e. the shift the kernel applied is the shift that was drawn."""
    from oracle import drq_oracle as O
    obs = synth.make_batch(81, 1, 9, seed=9, smooth=False)[0][:, :2].contiguous()
    sh = torch.tensor([[x, y] for x in range(9) for y in range(9)], dtype=torch.int32)
    out = ops.random_shifts_aug(obs.cuda(), sh.float().cuda(), 4).cpu()
    crop = O.aug_integer_crop(obs.float(), sh)
    assert torch.equal(out.round(), crop)


@pytest.mark.parametrize("cin,hin,stride,nb", [(9, 84, 2, 5), (32, 41, 1, 3), (32, 39, 1, 4), (32, 37, 1, 2)])
def test_conv_fwd(ops, cin, hin, stride, nb):
    x = rnd(nb, cin, hin, hin, seed=1)
    w = rnd(32, cin, 3, 3, seed=2, scale=0.2)
    b = rnd(32, seed=3, scale=0.1)
    y = ops.conv3x3_fwd(x.cuda(), w.cuda(), b.cuda(), stride, relu=True)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), stride=stride))
    assert nerr(y, ref) <= 2e-6
    y2 = ops.conv3x3_fwd(x.cuda(), w.cuda(), b.cuda(), stride, relu=False)
    assert nerr(y2, F.conv2d(x.double(), w.double(), b.double(), stride=stride)) <= 2e-6


@pytest.mark.parametrize("hout,nb", [(35, 3), (37, 2), (39, 5)])
def test_conv_dgrad(ops, hout, nb):
    hin = hout + 2
    dy = rnd(nb, 32, hout, hout, seed=4)
    w = rnd(32, 32, 3, 3, seed=5, scale=0.2)
    act = rnd(nb, 32, hin, hin, seed=6).clamp_min(0)          # post-ReLU input of the layer
    dy_pad = F.pad(dy, (2, 2, 2, 2)).contiguous()
    dx = ops.conv3x3_dgrad(dy_pad.cuda(), w.cuda(), act.cuda())
    ref = F.conv_transpose2d(dy.double(), w.double()) * (act > 0).double()
    assert nerr(dx, ref) <= 2e-6
    dx2 = ops.conv3x3_dgrad(dy_pad.cuda(), w.cuda(), None)
    assert nerr(dx2, F.conv_transpose2d(dy.double(), w.double())) <= 2e-6


@pytest.mark.parametrize("hin,nb", [(41, 3), (39, 4), (37, 2), (41, 1), (37, 33)])
def test_conv_fwd_winograd(ops, hin, nb):
    """The update's form of the 32->32 layers (Winograd F(2x2,3x3), drq_conv3x3_fwd_wino): same bound against fp64 as
    the direct kernel, ragged last tile row / column (the output sizes are odd), ragged last unit of 16 tiles."""
    x = rnd(nb, 32, hin, hin, seed=1)
    w = rnd(32, 32, 3, 3, seed=2, scale=0.2)
    b = rnd(32, seed=3, scale=0.1)
    y = ops.conv3x3_fwd(x.cuda(), w.cuda(), b.cuda(), 1, relu=True, wino=True)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double()))
    assert nerr(y, ref) <= 2e-6
    y2 = ops.conv3x3_fwd(x.cuda(), w.cuda(), b.cuda(), 1, relu=False, wino=True)
    assert nerr(y2, F.conv2d(x.double(), w.double(), b.double())) <= 2e-6
    # against the direct kernel: rounding-level difference only
    assert nerr(y2, ops.conv3x3_fwd(x.cuda(), w.cuda(), b.cuda(), 1, relu=False).double().cpu()) <= 2e-6


@pytest.mark.parametrize("hout,nb", [(35, 3), (37, 2), (39, 5), (35, 19)])
def test_conv_dgrad_winograd(ops, hout, nb):
    hin = hout + 2
    dy = rnd(nb, 32, hout, hout, seed=4)
    w = rnd(32, 32, 3, 3, seed=5, scale=0.2)
    act = rnd(nb, 32, hin, hin, seed=6).clamp_min(0)          # post-ReLU input of the layer
    dy_pad = F.pad(dy, (2, 2, 2, 2)).contiguous()
    dx = ops.conv3x3_dgrad(dy_pad.cuda(), w.cuda(), act.cuda(), wino=True)
    ref = F.conv_transpose2d(dy.double(), w.double()) * (act > 0).double()
    assert nerr(dx, ref) <= 2e-6
    dx2 = ops.conv3x3_dgrad(dy_pad.cuda(), w.cuda(), None, wino=True)
    assert nerr(dx2, F.conv_transpose2d(dy.double(), w.double())) <= 2e-6
    # strided (padded) destination, as the update writes it: the border must stay untouched
    lib, hp = ops._lib.load(), hin + 4
    dst = torch.full((nb, 32, hp, hp), 7.0, device="cuda")
    dyc, wc, ac = dy_pad.cuda(), w.cuda(), act.cuda()
    ops.check(lib.drq_conv3x3_dgrad_wino(dyc.data_ptr(), wc.data_ptr(), ac.data_ptr(), dst.data_ptr(), nb, hout,
                                         32 * hp * hp, hp * hp, hp, 2 * hp + 2, None), "wino")
    torch.cuda.synchronize()
    assert torch.equal(dst[:, :, 2:-2, 2:-2].cpu(), dx.cpu())
    border = dst.clone()
    border[:, :, 2:-2, 2:-2] = 7.0
    assert bool((border == 7.0).all())


@pytest.mark.parametrize("hin,nb", [(41, 5), (39, 2), (37, 7), (41, 1), (37, 64), (39, 256)])
def test_conv_wgrad_winograd(ops, hin, nb):
    """The update's weight / bias gradient of the 32->32 layers (Winograd form, drq_conv3x3_wgrad_wino): same bound
    against fp64 as the direct kernel; dY is the interior view of the zero-padded buffer, as the update stores it
    (the kernel relies on that padding for the half-empty last tile row / column and for the ragged last step)."""
    hout = hin - 2
    x = rnd(nb, 32, hin, hin, seed=7).clamp_min(0)
    dy = rnd(nb, 32, hout, hout, seed=8)
    dy_pad = F.pad(dy, (2, 2, 2, 2)).contiguous().cuda()
    dw, db = ops.conv3x3_wgrad(x.cuda(), dy_pad[:, :, 2:-2, 2:-2], 1, wino=True)
    xd = x.double()
    wd = torch.zeros(32, 32, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xd, wd) * dy.double()).sum().backward()
    assert nerr(dw, wd.grad) <= 3e-6
    assert nerr(db, dy.double().sum((0, 2, 3))) <= 3e-6
    dw_d, db_d = ops.conv3x3_wgrad(x.cuda(), dy_pad[:, :, 2:-2, 2:-2], 1)
    assert nerr(dw, dw_d.double().cpu()) <= 3e-6 and nerr(db, db_d.double().cpu()) <= 3e-6


def test_conv_wgrad_winograd_refuses_a_contiguous_dy(ops):
    """The Winograd weight-gradient kernel reads one row / column past every dy plane as zeros: dy must be the interior
    of a zero-padded buffer (include/drqv2_hip.h).  A contiguous dy would give wrong sums silently: DRQ_EARG instead."""
    from drqv2_amd._lib import DrqError
    x = torch.relu(rnd(2, 32, 39, 39, seed=1)).cuda()
    dy = rnd(2, 32, 37, 37, seed=2).cuda()
    with pytest.raises(DrqError):
        ops.conv3x3_wgrad(x, dy, 1, wino=True)
    pad = torch.zeros(2, 32, 41, 41).cuda()
    pad[:, :, 2:-2, 2:-2] = dy
    dw, db = ops.conv3x3_wgrad(x, pad[:, :, 2:-2, 2:-2], 1, wino=True)
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), (32, 32, 3, 3), dy.double().cpu())
    assert nerr(dw, ref) <= 3e-6


@pytest.mark.parametrize("cin,hin,stride,nb", [(9, 84, 2, 3), (32, 41, 1, 5), (32, 39, 1, 2), (32, 37, 1, 7)])
def test_conv_wgrad(ops, cin, hin, stride, nb):
    hout = (hin - 3) // stride + 1
    x = rnd(nb, cin, hin, hin, seed=7)
    dy = rnd(nb, 32, hout, hout, seed=8)
    dy_pad = F.pad(dy, (2, 2, 2, 2)).contiguous().cuda()
    dw, db = ops.conv3x3_wgrad(x.cuda(), dy_pad[:, :, 2:-2, 2:-2], stride)      # strided view, as in the step
    xd = x.double().requires_grad_(False)
    wd = torch.zeros(32, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xd, wd, stride=stride) * dy.double()).sum().backward()
    assert nerr(dw, wd.grad) <= 2e-6
    assert nerr(db, dy.double().sum((0, 2, 3))) <= 2e-6


@pytest.mark.parametrize("M,N,K", [(256, 1024, 1024), (8, 50, 39200), (37, 56, 1024), (256, 6, 1024), (5, 1, 1024),
                                   (64, 1024, 56), (33, 100, 50)])
def test_linear_fwd(ops, M, N, K):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    y = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda(), relu=True)
    assert nerr(y, torch.relu(x.double() @ w.double().t() + b.double())) <= 3e-6
    for tile, sk in ((1, 1), (2, 1), (1, 3), (2, 2)):
        y = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda(), tile=tile, splitk=sk)
        assert nerr(y, x.double() @ w.double().t() + b.double()) <= 3e-6, (tile, sk)


@pytest.mark.parametrize("M,N,K", [(256, 1024, 1024), (7, 56, 1024), (256, 39200, 50), (19, 1024, 1), (32, 50, 1024)])
def test_linear_dgrad(ops, M, N, K):
    dy, w = rnd(M, K, seed=1), rnd(K, N, seed=2, scale=K ** -0.5)
    mask = rnd(M, N, seed=3)
    dx = ops.linear_dgrad(dy.cuda(), w.cuda(), mask.cuda())
    assert nerr(dx, (dy.double() @ w.double()) * (mask > 0).double()) <= 3e-6


@pytest.mark.parametrize("Brows,N,K", [(256, 1024, 1024), (9, 50, 39200), (256, 1, 1024), (31, 1024, 56), (12, 6, 1024),
                                       (2048, 100, 64), (1030, 1024, 32),    # >= 1,024 rows: the column sum's 16-column shape
                                       (512, 50, 39200), (128, 100, 39200)])  # the trunk kernel's other batch sizes
def test_linear_wgrad(ops, Brows, N, K):
    dy, x = rnd(Brows, N, seed=1), rnd(Brows, K, seed=2)
    dw, db = ops.linear_wgrad(dy.cuda(), x.cuda())
    assert nerr(dw, dy.double().t() @ x.double()) <= 3e-6
    assert nerr(db, dy.double().sum(0)) <= 3e-6


@pytest.mark.parametrize("rows,Fd", [(256, 50), (7, 100), (3, 20), (5, 256)])
def test_ln_tanh(ops, rows, Fd):
    z = rnd(rows, Fd, seed=1, scale=2.0)
    g = 1 + 0.1 * rnd(Fd, seed=2)
    b = 0.1 * rnd(Fd, seed=3)
    dh = rnd(rows, Fd, seed=4)
    h, xhat, rstd = ops.ln_tanh_fwd(z.cuda(), g.cuda(), b.cuda())
    zd = z.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.tanh(F.layer_norm(zd, (Fd,), gd, bd, 1e-5))
    assert nerr(h, ref) <= 2e-6
    (ref * dh.double()).sum().backward()
    dz, dg, dbeta = ops.ln_tanh_bwd(dh.cuda(), h, xhat, rstd, g.cuda())
    assert nerr(dz, zd.grad) <= 1e-5
    assert nerr(dg, gd.grad) <= 1e-5
    assert nerr(dbeta, bd.grad) <= 1e-5


def test_adam_and_polyak_bitwise(ops):
    """vs torch.optim.Adam / utils.soft_update_params outputs captured from the reference stack"""
    d = np.load(os.path.join(G, "elementwise.npz"))
    p = torch.from_numpy(d["adam_p0"].copy()).cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for t in range(3):
        ops.adam_flat(p, torch.from_numpy(d["adam_g"][t]).cuda(), m, v, 1e-4, t + 1)
        assert torch.equal(m.cpu(), torch.from_numpy(d["adam_m"][t]))
        assert torch.equal(v.cpu(), torch.from_numpy(d["adam_v"][t]))
        assert torch.equal(p.cpu(), torch.from_numpy(d["adam_p"][t]))
    tg = torch.from_numpy(d["ema_tgt0"].copy()).cuda().view(-1)
    ops.ema_flat(torch.from_numpy(d["ema_net"]).cuda().view(-1), tg, 0.01)
    assert torch.equal(tg.cpu().view(65, 64), torch.from_numpy(d["ema_tgt1"]))


def test_adam_fused_polyak_and_gscale(ops):
    from oracle import drq_oracle as O
    n = 10007
    p0, g = rnd(n, seed=1), rnd(n, seed=2, scale=1e-3)
    t0 = rnd(n, seed=3)
    p, m, v, tg = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda(), t0.clone().cuda()
    ops.adam_flat(p, (g * 4).cuda(), m, v, 8e-5, 1, gscale=0.25, tgt=tg, tau=0.01)
    pr, mr, vr, tr = p0.clone(), torch.zeros(n), torch.zeros(n), t0.clone()
    O.adam_step(pr, g, mr, vr, 1, 8e-5)
    O.polyak(pr, tr, 0.01)
    assert torch.equal(p.cpu(), pr) and torch.equal(tg.cpu(), tr)


def test_trunc_normal_sample_bitwise(ops):
    d = np.load(os.path.join(G, "elementwise.npz"))
    mu = torch.from_numpy(d["tn_mu"])
    pre = torch.atanh(mu.double()).float()
    mu_k, a = ops.trunc_normal_sample(pre.cuda(), torch.from_numpy(d["tn_noise"]).cuda(), 0.37, 0.3)
    assert (mu_k.cpu() - mu).abs().max() <= 2e-7
    # with the kernel's own mu the clamp chain is exact
    from oracle import drq_oracle as O
    a_ref = O.trunc_normal_sample(mu_k.cpu(), torch.from_numpy(d["tn_noise"]), 0.37, 0.3)
    assert torch.equal(a.cpu(), a_ref)
    assert a.abs().max().item() <= 1 - 1e-6 + 1e-9


def test_unsupported_shapes_are_refused(ops):
    from drqv2_amd import _lib
    x = torch.zeros(2, 16, 20, 20, device="cuda")
    w = torch.zeros(32, 16, 3, 3, device="cuda")
    with pytest.raises(_lib.DrqError):
        ops.conv3x3_fwd(x, w, torch.zeros(32, device="cuda"), 1)
    with pytest.raises(_lib.DrqError):
        ops.conv3x3_fwd(x.cpu(), w, torch.zeros(32), 1)


@pytest.mark.parametrize("Brows,N,K,n", [(256, 1024, 1024, 2), (9, 50, 39200, 1), (31, 1024, 56, 2), (12, 6, 1024, 1),
                                         (40, 70, 33, 3), (256, 50, 39200, 1), (300, 100, 8192, 1), (5, 21, 4096, 1),
                                         (128, 50, 39200, 1), (256, 64, 4096, 1), (256, 33, 8192, 1),
                                         (128, 1, 4128, 1), (256, 100, 39200, 1), (128, 128, 4096, 1)])
def test_gemm_batched_wgrad_with_fused_bias_grad(ops, Brows, N, K, n):
    """dW_i = dy_i^T x_i and db_i = column sums of dy_i from ONE launch, independent pointers per problem"""
    dys = [rnd(Brows, N, seed=10 + i) for i in range(n)]
    xs = [rnd(Brows, K, seed=20 + i) for i in range(n)]
    Cs, rs = ops.gemm_batched([t.cuda() for t in dys], False, [t.cuda() for t in xs], False, N, K, Brows, N, K,
                              rowsum=True)
    for dy, x, c, r in zip(dys, xs, Cs, rs):
        assert nerr(c, dy.double().t() @ x.double()) <= 3e-6
        assert nerr(r, dy.double().sum(0)) <= 3e-6


@pytest.mark.parametrize("n", [1, 3, 32, 200])
def test_fused_aug_conv1_is_bit_identical_to_the_two_kernel_path(ops, n):
    """drq_conv1_aug_fwd (what the update runs) against drq_aug_fwd(fuse_norm) + drq_conv3x3_fwd: same encoder
    input (also equal to the oracle's aug bit for bit, test_aug_vs_oracle_and_reference) and same layer output, both
    views, every one of the 81 shifts present; frames >= n_store are not written."""
    g = torch.Generator().manual_seed(n)
    obs = torch.randint(0, 256, (n, 9, 84, 84), generator=g, dtype=torch.uint8).cuda()
    obs1 = torch.randint(0, 256, (n, 9, 84, 84), generator=g, dtype=torch.uint8).cuda()
    sh = torch.randint(0, 9, (n, 2), generator=g).float()
    sh1 = torch.randint(0, 9, (n, 2), generator=g).float()
    if n >= 81:
        sh[:81] = torch.tensor([[a, b] for a in range(9) for b in range(9)], dtype=torch.float32)
    sh, sh1 = sh.cuda(), sh1.cuda()
    w, b = (rnd(32, 9, 3, 3, seed=5) * 0.2).cuda(), (rnd(32, seed=6) * 0.1).cuda()
    y, xaug = ops.conv1_aug_fwd(obs, sh, obs1, sh1, w, b, n_store=2 * n)
    x0 = ops.random_shifts_aug(obs, sh, 4, fuse_norm=True)
    x1 = ops.random_shifts_aug(obs1, sh1, 4, fuse_norm=True)
    assert torch.equal(xaug[:n], x0) and torch.equal(xaug[n:], x1)
    assert torch.equal(y[:n], ops.conv3x3_fwd(x0, w, b, 2)) and torch.equal(y[n:], ops.conv3x3_fwd(x1, w, b, 2))
    y2, xaug2 = ops.conv1_aug_fwd(obs, sh, obs1, sh1, w, b)            # the update's form: obs view stored only
    assert torch.equal(y2, y) and torch.equal(xaug2[:n], x0) and not bool(xaug2[n:].any())


@pytest.mark.parametrize("cin,hin,stride,nb", [(9, 84, 2, 256), (32, 41, 1, 512), (32, 39, 1, 256), (32, 37, 1, 96)])
def test_conv_kernels_at_training_batch_sizes(ops, cin, hin, stride, nb):
    """Forward, dgrad and wgrad at the batch sizes of the training step, where a wave walks several pixel tiles
    (delayed stores, prefetch across tiles) and a wgrad wave several output rows (rolling row tiles) -- paths the
    small cases above never reach.  Reference: torch's own convolution on the GPU in fp64."""
    import torch.nn.functional as Fn
    g = torch.Generator(device="cuda").manual_seed(11)
    rn = lambda *sh: torch.randn(*sh, device="cuda", generator=g)
    gerr = lambda a, b: float((a.double() - b).norm() / b.norm())
    hout = (hin - 3) // stride + 1
    x, w, b = rn(nb, cin, hin, hin), rn(32, cin, 3, 3) * 0.1, rn(32) * 0.1
    y = ops.conv3x3_fwd(x, w, b, stride)
    assert gerr(y, torch.relu(Fn.conv2d(x.double(), w.double(), b.double(), stride=stride))) <= 2e-6
    dy = rn(nb, 32, hout, hout)
    dy_pad = torch.zeros(nb, 32, hout + 4, hout + 4, device="cuda")
    dy_pad[:, :, 2:-2, 2:-2] = dy
    dw, db = ops.conv3x3_wgrad(x, dy_pad[:, :, 2:-2, 2:-2], stride)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    gx, gw, gb = torch.autograd.grad(Fn.conv2d(xd, wd, bd, stride=stride), (xd, wd, bd), dy.double())
    assert gerr(dw, gw) <= 3e-6 and gerr(db, gb) <= 3e-6
    if stride == 1:
        mask = rn(nb, 32, hin, hin)
        dx = ops.conv3x3_dgrad(dy_pad, w, mask)
        assert gerr(dx, gx * (mask.double() > 0)) <= 2e-6


@pytest.mark.parametrize("M,K,hw", [(256, 50, 35), (8, 50, 35), (70, 100, 12), (33, 21, 16)])
def test_skinny_trunk_dgrad_matches_generic_path_and_fp64(ops, M, K, hw):
    """dX = (dz W) * (mask > 0) for the trunk shape (short K, N = 32*hw*hw): the dedicated kernel (automatic
    dispatch) against the generic tiled GEMM (explicit tile) and fp64, dense and padded-scatter output."""
    N = 32 * hw * hw
    dz, w = rnd(M, K, seed=1), rnd(K, N, seed=2, scale=K ** -0.5)
    mask = rnd(M, N, seed=3)
    ref = (dz.double() @ w.double()) * (mask.double() > 0)
    args = ([dz.cuda()], True, [w.cuda()], False, M, N, K, K, N)
    (c_fast,), _ = ops.gemm_batched(*args, auxs=[mask.cuda()])
    (c_gen,), _ = ops.gemm_batched(*args, auxs=[mask.cuda()], tile=2, splitk=1)
    assert nerr(c_fast, ref) <= 3e-6 and nerr(c_gen, ref) <= 3e-6
    (c_nomask,), _ = ops.gemm_batched(*args)
    assert nerr(c_nomask, dz.double() @ w.double()) <= 3e-6
    hp = hw + 4
    pads = [torch.zeros(M, 32, hp, hp, device="cuda") for _ in range(2)]
    ops.gemm_batched(*args, auxs=[mask.cuda()], scatter_hw=hw, Cs=[pads[0]])
    ops.gemm_batched(*args, auxs=[mask.cuda()], scatter_hw=hw, Cs=[pads[1]], tile=2, splitk=1)
    assert torch.equal(pads[0][:, :, 2:-2, 2:-2].reshape(M, N), c_fast)          # same values, scattered
    assert nerr(pads[1][:, :, 2:-2, 2:-2].reshape(M, N), ref) <= 3e-6
    border = pads[0].clone()
    border[:, :, 2:-2, 2:-2] = 0
    assert float(border.abs().sum()) == 0.0                                     # the zero border is never written


def test_gemm_batched_forward_four_problems(ops):
    M, N, K = 64, 50, 2048
    xs = [rnd(M, K, seed=i) for i in range(2)]
    wts = [rnd(N, K, seed=5 + i, scale=K ** -0.5) for i in range(4)]
    bs = [rnd(N, seed=9 + i) for i in range(4)]
    A = [xs[0].cuda(), xs[0].cuda(), xs[1].cuda(), xs[1].cuda()]
    Cs, _ = ops.gemm_batched(A, True, [w.cuda() for w in wts], True, M, N, K, K, K, biases=[b.cuda() for b in bs],
                             relu=True)
    for i, c in enumerate(Cs):
        ref = torch.relu(xs[i // 2].double() @ wts[i].double().t() + bs[i].double())
        assert nerr(c, ref) <= 3e-6


@pytest.mark.parametrize("M,N,K,n", [(256, 50, 39200, 4), (256, 50, 39200, 1), (64, 50, 4096, 2), (32, 64, 4128, 1),
                                     (96, 7, 8192, 3), (512, 50, 39200, 1), (256, 100, 39200, 4), (32, 100, 39200, 1),
                                     (64, 65, 4096, 2), (32, 128, 4096, 1)])
def test_trunk_forward_partials(ops, M, N, K, n):
    """the trunk kernel (z = feat W^T, k-contiguous operands, N <= 64, long K): row tiles of 32 and 64, slices of
    unequal length, weight rows past N read as zero, one problem and several; the sum of its records is the GEMM."""
    xs = [rnd(M, K, seed=i) for i in range(n)]
    wts = [rnd(N, K, seed=5 + i, scale=K ** -0.5) for i in range(n)]
    got, sk = ops.gemm_batched_partial([x.cuda() for x in xs], [w.cuda() for w in wts], M, N, K, K, K)
    assert sk > 1
    for i in range(n):
        assert nerr(got[i], xs[i].double() @ wts[i].double().t()) <= 3e-6


@pytest.mark.parametrize("B,H,n", [(256, 1024, 4), (7, 64, 2), (33, 100, 1), (2048, 1024, 2), (1030, 100, 1)])
def test_qout_fwd_bwd(ops, B, H, n):     # >= 1,024 rows: the backward kernel's 16-column shape
    hs = [rnd(B, H, seed=i).clamp_min(0) for i in range(n)]
    ws_ = [rnd(H, seed=10 + i, scale=H ** -0.5) for i in range(n)]
    bs = [rnd(1, seed=20 + i) for i in range(n)]
    dqs = [rnd(B, seed=30 + i) for i in range(n)]
    qs = ops.qout_fwd([t.cuda() for t in hs], [t.cuda() for t in ws_], [t.cuda() for t in bs])
    for h, w, b, q in zip(hs, ws_, bs, qs):
        assert nerr(q, h.double() @ w.double() + b.double()) <= 2e-6
    dhs, dws, dbs = ops.qout_bwd([t.cuda() for t in dqs], [t.cuda() for t in hs], [t.cuda() for t in ws_])
    for dq, h, w, dh, dw, db in zip(dqs, hs, ws_, dhs, dws, dbs):
        assert nerr(dh, torch.outer(dq.double(), w.double()) * (h > 0).double()) <= 2e-6
        assert nerr(dw, dq.double() @ h.double()) <= 3e-6
        assert nerr(db, dq.double().sum().view(1)) <= 3e-6
    dhs2, dws2, _ = ops.qout_bwd([t.cuda() for t in dqs], [t.cuda() for t in hs], [t.cuda() for t in ws_], False)
    assert dws2 is None and all(torch.equal(a, b) for a, b in zip(dhs, dhs2))
