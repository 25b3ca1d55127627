"""Row-local stages fused with the first MLP layer (csrc/rowblock.hip) through the C ABI: LayerNorm+tanh [+ action
columns] -> first layers, and policy output layer + sample -> the target critic's first layers.  The LayerNorm part
must equal drq_ln_tanh_fwd bit for bit; products are held to fp64 at the fp32 rounding floor."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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


def ln64(z, g, b):
    z = z.double()
    m = z.mean(1, keepdim=True)
    v = ((z - m) ** 2).mean(1, keepdim=True)
    return torch.tanh((z - m) / torch.sqrt(v + 1e-5) * g.double() + b.double())


@pytest.mark.parametrize("rows,F,A,H", [(256, 50, 6, 1024), (32, 100, 21, 1024), (19, 50, 1, 256), (64, 20, 12, 320),
                                        (5, 64, 3, 64)])
def test_ln_l1_matches_the_separate_kernels(ops, rows, F, A, H):
    z = [rnd(rows, F, seed=s, scale=3.0) for s in (1, 2, 3)]
    g = [1 + 0.1 * rnd(F, seed=10 + s) for s in range(3)]
    bt = [0.1 * rnd(F, seed=20 + s) for s in range(3)]
    act = rnd(rows, A, seed=4)
    wq = [rnd(H, F + A, seed=30 + h, scale=(F + A) ** -0.5) for h in range(2)]
    bq = [rnd(H, seed=40 + h) for h in range(2)]
    wp, bp = rnd(H, F, seed=50, scale=F ** -0.5), rnd(H, seed=51)
    c = lambda t: t.cuda()
    jobs = [dict(z=c(z[0]), gamma=c(g[0]), beta=c(bt[0]), rows=rows, tail=c(act), heads=[(c(wq[0]), c(bq[0])), (c(wq[1]), c(bq[1]))]),
            dict(z=c(z[1]), gamma=c(g[1]), beta=c(bt[1]), rows=rows, heads=[(c(wp), c(bp))]),
            dict(z=c(z[2]), gamma=c(g[2]), beta=c(bt[2]), rows=rows, save=False)]
    res = ops.ln_l1_fwd(jobs, F, H)
    for j in range(3):
        h, xh, rs = ops.ln_tanh_fwd(c(z[j]), c(g[j]), c(bt[j]))
        assert torch.equal(res[j]["out"][:, :F], h), j
        if j < 2:
            assert torch.equal(res[j]["xhat"], xh) and torch.equal(res[j]["rstd"], rs)
        assert nerr(h, ln64(z[j], g[j], bt[j])) <= 1e-6
    assert torch.equal(res[0]["out"][:, F:], c(act))
    x0 = torch.cat([res[0]["out"][:, :F].double().cpu(), act.double()], 1)
    for hd in range(2):
        assert nerr(res[0]["ys"][hd], torch.relu(x0 @ wq[hd].double().t() + bq[hd].double())) <= 3e-6
    x1 = res[1]["out"].double().cpu()
    assert nerr(res[1]["ys"][0], torch.relu(x1 @ wp.double().t() + bp.double())) <= 3e-6


@pytest.mark.parametrize("rows,F,splitk", [(256, 50, 16), (64, 100, 64), (7, 50, 3), (32, 50, 37)])
def test_ln_l1_from_split_k_records(ops, rows, F, splitk):
    H = 256
    parts = rnd(2, splitk, rows, F, seed=5)
    bias = [rnd(F, seed=6), rnd(F, seed=7)]
    g, bt = 1 + 0.1 * rnd(F, seed=8), 0.1 * rnd(F, seed=9)
    w, b = rnd(H, F, seed=11, scale=F ** -0.5), rnd(H, seed=12)
    c = lambda t: t.cuda()
    pc = c(parts)
    jobs = [dict(part=pc[0], bias=c(bias[0]), gamma=c(g), beta=c(bt), rows=rows, heads=[(c(w), c(b))]),
            dict(part=pc[1], bias=c(bias[1]), gamma=c(g), beta=c(bt), rows=rows)]
    res = ops.ln_l1_fwd(jobs, F, H, splitk=splitk, slab=rows * F)
    for j in range(2):
        zz = parts[j].double().sum(0) + bias[j].double()
        ref = ln64(zz, g, bt)
        assert nerr(res[j]["out"], ref) <= 2e-6
    assert nerr(res[0]["ys"][0], torch.relu(res[0]["out"].double().cpu() @ w.double().t() + b.double())) <= 3e-6


@pytest.mark.parametrize("B,F,A,H", [(256, 50, 6, 1024), (32, 100, 21, 1024), (19, 50, 1, 256), (8, 50, 12, 512)])
def test_policy_out_sample_and_target_first_layers(ops, B, F, A, H):
    p2 = torch.relu(rnd(2 * B, H, seed=1))
    w3, b3 = rnd(A, H, seed=2, scale=H ** -0.5), rnd(A, seed=3)
    nz_hi, nz_lo = rnd(B, A, seed=4), rnd(B, A, seed=5)
    h_t = torch.tanh(rnd(B, F, seed=6))
    wq = [rnd(H, F + A, seed=30 + h, scale=(F + A) ** -0.5) for h in range(2)]
    bq = [rnd(H, seed=40 + h) for h in range(2)]
    c = lambda t: t.cuda()
    ha_hi = torch.zeros(B, F + A).cuda()
    ha_hi[:, :F] = c(h_t)
    ha_lo = torch.zeros(B, F + A).cuda()
    std, clip = 0.7, 0.3
    r = ops.policy_out_l1_fwd(c(p2), c(w3), c(b3), B, F, std, clip, c(nz_hi), ha_hi, c(nz_lo), ha_lo,
                              heads=[(c(wq[0]), c(bq[0])), (c(wq[1]), c(bq[1]))])
    ref = p2.double() @ w3.double().t() + b3.double()
    assert nerr(r["p3"], ref) <= 3e-6
    # the sample is the reference's arithmetic on the kernel's own pre-activation: bit-exact against the sampling op
    mu_hi, a_hi = ops.trunc_normal_sample(r["p3"][B:].contiguous(), c(nz_hi), std, clip)
    mu_lo, a_lo = ops.trunc_normal_sample(r["p3"][:B].contiguous(), c(nz_lo), std, clip)
    assert torch.equal(r["mu_hi"], mu_hi) and torch.equal(ha_hi[:, F:], a_hi)
    assert torch.equal(r["mu_lo"], mu_lo) and torch.equal(ha_lo[:, F:], a_lo)
    assert torch.equal(ha_hi[:, :F], c(h_t))
    x = ha_hi.double().cpu()
    for hd in range(2):
        assert nerr(r["ys"][hd], torch.relu(x @ wq[hd].double().t() + bq[hd].double())) <= 3e-6
    # without first layers and without the lo job
    ha2 = torch.zeros(B, F + A).cuda()
    r2 = ops.policy_out_l1_fwd(c(p2), c(w3), c(b3), B, F, std, clip, c(nz_hi), ha2)
    assert torch.equal(r2["p3"], r["p3"]) and torch.equal(ha2[:, F:], ha_hi[:, F:])
