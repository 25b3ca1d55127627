"""The Winograd F(2x2,3x3) identities the encoder kernels rely on (drqv2_amd/csrc/conv_wino.hip,
conv_wino_wgrad.hip), checked on the CPU in fp64 against torch's own convolution, and the fp32 error level of the
forward form against the direct form's (the numbers quoted in DESIGN.md section 3).  No GPU, no library call: this
pins the MATH (matrices, tiling of odd output sizes, zero-padded dY, the sign-flipped basis row of the weight
gradient); the kernels themselves are held to fp64 by tests/test_hip_ops.py on the GPU."""
import numpy as np
import torch
import torch.nn.functional as F

Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def tiles(x, th):
    """x [N,C,H,W] -> 4x4 patches of the 2x2-output tiling: [N,C,th,th,4,4] (zero beyond the frame)."""
    N, C, H, W = x.shape
    xp = np.zeros((N, C, 2 * th + 2, 2 * th + 2), dtype=x.dtype)
    xp[:, :, :H, :W] = x
    d = np.empty((N, C, th, th, 4, 4), dtype=x.dtype)
    for i in range(4):
        for j in range(4):
            d[..., i, j] = xp[:, :, i:i + 2 * th:2, j:j + 2 * th:2]
    return d


def wino_fwd(x, w, dt):
    N, C, H, _ = x.shape
    ho = H - 2
    th = (ho + 1) // 2
    U = np.einsum('ij,kcjl,ml->kcim', G.astype(dt), w.astype(dt), G.astype(dt)).astype(dt)
    d = tiles(x.astype(dt), th)
    V = np.einsum('ij,nctsjl,ml->nctsim', Bt.astype(dt), d, Bt.astype(dt)).astype(dt)
    M = np.zeros((N, w.shape[0], th, th, 4, 4), dtype=dt)
    for c in range(C):                       # sequential accumulation over the input channels, like the MFMA chain
        M += (U[None, :, c, None, None] * V[:, None, c]).astype(dt)
    Y = np.einsum('ij,nktsjl,ml->nktsim', At.astype(dt), M, At.astype(dt)).astype(dt)
    return Y.transpose(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], 2 * th, 2 * th)[:, :, :ho, :ho]


def test_forward_identity_and_fp32_error_level():
    torch.manual_seed(0)
    for H in (41, 39, 37):
        w = torch.empty(32, 32, 3, 3)
        torch.nn.init.orthogonal_(w, torch.nn.init.calculate_gain('relu'))
        x = torch.relu(torch.randn(3, 32, H, H)) * 0.5
        ref = F.conv2d(x.double(), w.double()).numpy()
        y64 = wino_fwd(x.numpy(), w.numpy(), np.float64)
        assert np.abs(y64 - ref).max() <= 1e-12 * np.abs(ref).max()
        y32 = wino_fwd(x.numpy(), w.numpy(), np.float32)
        d32 = F.conv2d(x, w).numpy()
        e_w = np.linalg.norm(y32 - ref) / np.linalg.norm(ref)
        e_d = np.linalg.norm(d32 - ref) / np.linalg.norm(ref)
        assert e_w <= 1e-6 and e_w <= 4 * e_d + 1e-7, (e_w, e_d)     # same level as the direct form (measured 1.9e-7 / 1.6e-7)


def test_weight_gradient_identity_with_padded_dy_and_flipped_basis_row():
    """dg = (SG)^T [ sum_tiles (A' dY A'^T) .* (B^T d B) ] (SG) with A' = A with the sign of its last row flipped and
    S = diag(1,1,1,-1): what conv3x3_wgrad_wino_kernel computes, including the half-empty last tile row / column
    (odd output sizes) read from dY's zero padding."""
    torch.manual_seed(1)
    A = At.T.copy()                                            # [[1,0],[1,1],[1,-1],[0,-1]]
    Ap = A.copy()
    Ap[3] = -Ap[3]
    SG = G.copy()
    SG[3] = -SG[3]
    for H in (41, 39, 37):
        ho = H - 2
        th = (ho + 1) // 2
        x = torch.relu(torch.randn(2, 32, H, H)).double()
        dy = torch.randn(2, 32, ho, ho).double()
        wd = torch.zeros(32, 32, 3, 3, dtype=torch.float64, requires_grad=True)
        (F.conv2d(x, wd) * dy).sum().backward()
        d = tiles(x.numpy(), th)                               # [N,C,th,th,4,4]
        dyp = np.zeros((2, 32, 2 * th, 2 * th))
        dyp[:, :, :ho, :ho] = dy.numpy()                       # the zero padding beyond the odd output size
        dyt = dyp.reshape(2, 32, th, 2, th, 2).transpose(0, 1, 2, 4, 3, 5)           # [N,K,th,th,2,2]
        V = np.einsum('ij,nctsjl,ml->nctsim', Bt, d, Bt)
        dM = np.einsum('ij,nktsjl,ml->nktsim', Ap, dyt, Ap)
        dU = np.einsum('nktsim,nctsim->kcim', dM, V)
        dg = np.einsum('ia,kcij,jb->kcab', SG, dU, SG)
        ref = wd.grad.numpy()
        assert np.abs(dg - ref).max() <= 1e-11 * np.abs(ref).max()


def test_input_gradient_is_the_forward_form_on_padded_dy_with_flipped_transposed_weights():
    torch.manual_seed(2)
    w = torch.randn(32, 32, 3, 3).double() * 0.1
    dy = torch.randn(2, 32, 35, 35).double()
    ref = F.conv_transpose2d(dy, w).numpy()                    # [2,32,37,37]
    wt = w.numpy().transpose(1, 0, 2, 3)[:, :, ::-1, ::-1]     # wmode 1 of conv_wino.hip
    dyp = F.pad(dy, (2, 2, 2, 2)).numpy()
    got = wino_fwd(dyp, np.ascontiguousarray(wt), np.float64)
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
