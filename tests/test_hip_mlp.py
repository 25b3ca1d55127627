"""The hidden-layer GEMMs on the LDS-DMA ring kernel (csrc/gemm3.hip) through the C ABI against fp64, at the shapes
the update issues (batch 64..512 rows, hidden 1024) plus small / ragged-in-tiles ones: forward with bias + ReLU and
the fused partial dots of the Q heads' output layer, the input gradient with its ReLU mask, and the one-launch
weight + input gradient pair.  Bounds: the fp32 rounding floor of SURVEY.md App. B (normwise <= 3e-6)."""
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


@pytest.mark.parametrize("n,M,N,K", [(4, 256, 1024, 1024), (2, 256, 1024, 1024), (1, 512, 1024, 1024), (1, 64, 64, 64),
                                     (3, 128, 192, 96), (2, 64, 1024, 1024), (1, 1024, 256, 320)])
def test_mlp_fwd_bias_relu_and_partial_dots(ops, n, M, N, K):
    xs = [rnd(M, K, seed=10 + i) for i in range(n)]
    ws = [rnd(N, K, seed=20 + i, scale=K ** -0.5) for i in range(n)]
    bs = [rnd(N, seed=30 + i) for i in range(n)]
    qw = [rnd(N, seed=40 + i, scale=N ** -0.5) for i in range(n)]
    cu = lambda ts: [t.cuda() for t in ts]
    ys, _ = ops.mlp_fwd(cu(xs), cu(ws), cu(bs), relu=True)
    ys2, qps = ops.mlp_fwd(cu(xs), cu(ws), cu(bs), relu=True, qws=cu(qw))
    for i in range(n):
        ref = torch.relu(xs[i].double() @ ws[i].double().t() + bs[i].double())
        assert nerr(ys[i], ref) <= 3e-6, i
        assert torch.equal(ys[i], ys2[i]) or nerr(ys2[i], ref) <= 3e-6
        q = qps[i].double().sum(1).cpu()
        qref = ref @ qw[i].double()
        assert float((q - qref).abs().max() / qref.abs().max()) <= 3e-6, i
        # every partial is the dot over its own column tile
        nq = qps[i].shape[1]
        tile = N // nq
        part_ref = (ref * qw[i].double()).view(M, nq, tile).sum(2)
        assert float((qps[i].double().cpu() - part_ref).abs().max() / part_ref.abs().max()) <= 3e-6
    # no bias, no ReLU
    ys, _ = ops.mlp_fwd(cu(xs), cu(ws), None, relu=False)
    assert nerr(ys[0], xs[0].double() @ ws[0].double().t()) <= 3e-6


@pytest.mark.parametrize("n,M,N,K", [(2, 256, 1024, 1024), (1, 256, 1024, 1024), (4, 128, 1024, 1024), (1, 64, 64, 64),
                                     (2, 192, 128, 160), (2, 512, 1024, 1024)])
def test_mlp_dgrad_masked(ops, n, M, N, K):
    dys = [rnd(M, K, seed=1 + i) for i in range(n)]
    ws = [rnd(K, N, seed=5 + i, scale=K ** -0.5) for i in range(n)]
    mk = [rnd(M, N, seed=9 + i) for i in range(n)]
    cu = lambda ts: [t.cuda() for t in ts]
    dxs = ops.mlp_dgrad(cu(dys), cu(ws), cu(mk))
    for i in range(n):
        assert nerr(dxs[i], (dys[i].double() @ ws[i].double()) * (mk[i] > 0).double()) <= 3e-6
    dxs = ops.mlp_dgrad(cu(dys), cu(ws), None)
    assert nerr(dxs[0], dys[0].double() @ ws[0].double()) <= 3e-6


@pytest.mark.parametrize("n,B,Nout,Kin", [(2, 256, 1024, 1024), (1, 256, 1024, 1024), (2, 512, 1024, 1024),
                                          (1, 64, 64, 64), (2, 128, 192, 128), (1, 64, 1024, 1024)])
def test_mlp_wgrad_dgrad_pair(ops, n, B, Nout, Kin):
    dys = [rnd(B, Nout, seed=1 + i) for i in range(n)]
    xs = [rnd(B, Kin, seed=5 + i) for i in range(n)]
    ws = [rnd(Nout, Kin, seed=9 + i, scale=Nout ** -0.5) for i in range(n)]
    cu = lambda ts: [t.cuda() for t in ts]
    xc = cu(xs)
    dws, dbs, dxs = ops.mlp_wgrad_dgrad(cu(dys), xc, cu(ws), xc)      # mask = the layer's (post-ReLU) input
    for i in range(n):
        assert nerr(dws[i], dys[i].double().t() @ xs[i].double()) <= 3e-6
        assert nerr(dbs[i], dys[i].double().sum(0)) <= 3e-6
        assert nerr(dxs[i], (dys[i].double() @ ws[i].double()) * (xs[i] > 0).double()) <= 3e-6


def test_mlp_refuses_ineligible_shapes(ops):
    from drqv2_amd._lib import DrqError
    x, w = torch.zeros(32, 64).cuda(), torch.zeros(64, 64).cuda()
    with pytest.raises(DrqError):
        ops.mlp_fwd([x], [w], None)                  # M not a multiple of 64
    x, w = torch.zeros(64, 48).cuda(), torch.zeros(64, 48).cuda()
    with pytest.raises(DrqError):
        ops.mlp_fwd([x], [w], None)                  # K not a multiple of 32
