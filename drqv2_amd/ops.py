"""Thin tensor-level wrappers over the C ABI (include/drqv2_hip.h): shape checks on the host, raw
device pointers into the library, work enqueued on torch's current stream.  GPU tensors only."""
import torch

from . import _lib
from ._lib import check, ptr

ENC_H = (84, 41, 39, 37, 35)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need(t, dtype=torch.float32, name="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise _lib.DrqError(f"{name}: a GPU tensor is required (the HIP path has no CPU fallback)")
    if t.device.index != torch.cuda.current_device():
        # these wrappers launch on the current device's current stream (the update path guards the device itself)
        raise _lib.DrqError(f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                            "wrap the call in torch.cuda.device(...)")
    if t.dtype != dtype or not t.is_contiguous():
        raise _lib.DrqError(f"{name}: contiguous {dtype} required, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


def aug_base_grid(h, pad, device):
    """The table torch.linspace gives the reference on the same device (drqv2.py:24-29)."""
    S = h + 2 * pad
    eps = 1.0 / S
    return torch.linspace(-1.0 + eps, 1.0 - eps, S, device=device, dtype=torch.float32)[:h].contiguous()


def random_shifts_aug(x, shift, pad=4, base=None, fuse_norm=False):
    """x: u8 or f32 [n,c,h,h]; shift: f32 [n,1,1,2] or [n,2] (the torch.randint draw)."""
    lib = _lib.load()
    n, c, h, w = x.shape
    assert h == w
    shift = _need(shift.reshape(n, 2), name="shift")
    base = aug_base_grid(h, pad, x.device) if base is None else _need(base, name="base")
    out = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    if x.dtype == torch.uint8:
        _need(x, torch.uint8, "obs")
        check(lib.drq_aug_fwd(ptr(x), ptr(shift), ptr(base), ptr(out), n, c, h, pad, int(fuse_norm), _stream()),
              "drq_aug_fwd")
    else:
        _need(x, name="obs")
        if fuse_norm:
            raise _lib.DrqError("fuse_norm needs uint8 input")
        check(lib.drq_aug_fwd_f32(ptr(x), ptr(shift), ptr(base), ptr(out), n, c, h, pad, _stream()),
              "drq_aug_fwd_f32")
    return out


def conv1_aug_fwd(obs, shift, obs1, shift1, w, b, n_store=None, base=None, bf16=False, y_nhwc=False):
    """Fused RandomShiftsAug + /255-0.5 + conv1 + ReLU on both views (drq_conv1_aug_fwd).
    Returns (y [2n,32,41,41], xaug [2n,9,84,84] with frames [0,n_store) written, the rest zero); y_nhwc (with bf16): y
    is bf16 [2n,41,41,32] (drq_conv1_aug_fwd_bf16_nhwc)."""
    lib = _lib.load()
    n = obs.shape[0]
    _need(obs, torch.uint8, "obs")
    _need(obs1, torch.uint8, "obs1")
    if tuple(obs.shape[1:]) != (9, 84, 84) or obs1.shape != obs.shape:
        raise _lib.DrqError("conv1_aug_fwd: frames must be [n,9,84,84]")
    shift, shift1 = _need(shift.reshape(n, 2), name="shift"), _need(shift1.reshape(n, 2), name="shift1")
    base = aug_base_grid(84, 4, obs.device) if base is None else _need(base, name="base")
    n_store = n if n_store is None else n_store
    y = (torch.empty((2 * n, 41, 41, 32), device=obs.device, dtype=torch.bfloat16) if y_nhwc else
         torch.empty((2 * n, 32, 41, 41), device=obs.device, dtype=torch.float32))
    xaug = torch.zeros((2 * n, 9, 84, 84), device=obs.device, dtype=torch.float32)
    if y_nhwc and not bf16:
        raise _lib.DrqError("conv1_aug_fwd: the channel-contiguous bf16 output belongs to the bf16 form")
    fn = lib.drq_conv1_aug_fwd_bf16_nhwc if y_nhwc else lib.drq_conv1_aug_fwd_bf16 if bf16 else lib.drq_conv1_aug_fwd
    check(fn(ptr(obs), ptr(shift), ptr(obs1), ptr(shift1), ptr(base), ptr(_need(w, name="w")), ptr(_need(b, name="b")),
             ptr(xaug), ptr(y), n, n_store, _stream()), "drq_conv1_aug_fwd")
    return y, xaug


def conv1_aug_fwd_indexed(frames, idx, shift, frames1, idx1, shift1, w, b, n_store=None, base=None):
    """drq_conv1_aug_fwd_indexed: the two views are rows idx / idx1 (int64) of frame stores [slots, 9*84*84] uint8."""
    lib = _lib.load()
    n = idx.numel()
    shift, shift1 = _need(shift.reshape(n, 2), name="shift"), _need(shift1.reshape(n, 2), name="shift1")
    base = aug_base_grid(84, 4, frames.device) if base is None else _need(base, name="base")
    n_store = n if n_store is None else n_store
    y = torch.empty((2 * n, 32, 41, 41), device=frames.device, dtype=torch.float32)
    xaug = torch.zeros((2 * n, 9, 84, 84), device=frames.device, dtype=torch.float32)
    check(lib.drq_conv1_aug_fwd_indexed(ptr(frames), ptr(idx), ptr(shift), ptr(frames1), ptr(idx1), ptr(shift1), ptr(base),
                                        ptr(_need(w, name="w")), ptr(_need(b, name="b")), ptr(xaug), ptr(y), n, n_store,
                                        _stream()), "drq_conv1_aug_fwd_indexed")
    return y, xaug


def u8_normalize(x):
    lib = _lib.load()
    _need(x, torch.uint8, "obs")
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(lib.drq_u8_normalize(ptr(x), ptr(y), x.numel(), _stream()), "drq_u8_normalize")
    return y


def conv3x3_fwd(x, w, b, stride, relu=True, bf16=False, wino=False):
    lib = _lib.load()
    _need(x, name="x"), _need(w, name="w"), _need(b, name="b")
    nb, cin, hin, _ = x.shape
    hout = (hin - 3) // stride + 1
    y = torch.empty((nb, 32, hout, hout), device=x.device, dtype=torch.float32)
    if wino:
        check(lib.drq_conv3x3_fwd_wino(ptr(x), ptr(w), ptr(b), ptr(y), nb, hin, int(relu), 32 * hout * hout, hout * hout,
                                       hout, 0, _stream()), "drq_conv3x3_fwd_wino")
        return y
    if bf16:
        check(lib.drq_conv3x3_fwd_bf16(ptr(x), ptr(w), ptr(b), ptr(y), nb, hin, int(relu), 32 * hout * hout, hout * hout,
                                       hout, 0, _stream()), "drq_conv3x3_fwd_bf16")
        return y
    check(lib.drq_conv3x3_fwd(ptr(x), ptr(w), ptr(b), ptr(y), nb, cin, hin, stride, int(relu), 32 * hout * hout,
                              hout * hout, hout, 0, _stream()), "drq_conv3x3_fwd")
    return y


def conv3x3_dgrad(dy_pad, w, mask, bf16=False, wino=False):
    """dy_pad [nb,32,hout+4,hout+4] (zero border of 2) -> dx [nb,32,hout+2,hout+2] * (mask>0)."""
    lib = _lib.load()
    _need(dy_pad, name="dy_pad"), _need(w, name="w")
    nb, _, hp, _ = dy_pad.shape
    hout = hp - 4
    hin = hout + 2
    if mask is not None:
        _need(mask, name="mask")
        assert tuple(mask.shape) == (nb, 32, hin, hin)
    dx = torch.empty((nb, 32, hin, hin), device=dy_pad.device, dtype=torch.float32)
    if wino:
        check(lib.drq_conv3x3_dgrad_wino(ptr(dy_pad), ptr(w), ptr(mask), ptr(dx), nb, hout, 32 * hin * hin, hin * hin, hin,
                                         0, _stream()), "drq_conv3x3_dgrad_wino")
        return dx
    if bf16:
        check(lib.drq_conv3x3_dgrad_bf16(ptr(dy_pad), ptr(w), ptr(mask), ptr(dx), nb, hout, 32 * hin * hin, hin * hin, hin,
                                         0, _stream()), "drq_conv3x3_dgrad_bf16")
        return dx
    check(lib.drq_conv3x3_dgrad(ptr(dy_pad), ptr(w), ptr(mask), ptr(dx), nb, hout, 32 * hin * hin, hin * hin, hin, 0,
                                _stream()), "drq_conv3x3_dgrad")
    return dx


def conv3x3_wgrad(x, dy, stride, bf16=False, wino=False):
    """x [nb,cin,hin,hin], dy [nb,32,hout,hout] (any strides with unit x-stride) -> dw, db."""
    lib = _lib.load()
    _need(x, name="x")
    if not (dy.is_cuda and dy.dtype == torch.float32 and dy.stride(3) == 1):
        raise _lib.DrqError("dy: GPU fp32 with unit innermost stride required")
    nb, cin, hin, _ = x.shape
    dw = torch.empty((32, cin, 3, 3), device=x.device, dtype=torch.float32)
    db = torch.empty((32,), device=x.device, dtype=torch.float32)
    nbytes = lib.drq_conv3x3_wgrad_ws_bytes()
    ws = torch.empty((nbytes // 4,), device=x.device, dtype=torch.float32)
    if wino:        # dy must be the interior view of a buffer zero-padded by 2 (the kernel reads the padding as zeros)
        check(lib.drq_conv3x3_wgrad_wino(ptr(x), dy.data_ptr(), ptr(dw), ptr(db), nb, hin, dy.stride(0), dy.stride(1),
                                         dy.stride(2), 0, ptr(ws), nbytes, _stream()), "drq_conv3x3_wgrad_wino")
        return dw, db
    if bf16:
        check(lib.drq_conv3x3_wgrad_bf16(ptr(x), dy.data_ptr(), ptr(dw), ptr(db), nb, hin, dy.stride(0), dy.stride(1),
                                         dy.stride(2), 0, ptr(ws), nbytes, _stream()), "drq_conv3x3_wgrad_bf16")
        return dw, db
    check(lib.drq_conv3x3_wgrad(ptr(x), dy.data_ptr(), ptr(dw), ptr(db), nb, cin, hin, stride, dy.stride(0),
                                dy.stride(1), dy.stride(2), 0, ptr(ws), nbytes, _stream()), "drq_conv3x3_wgrad")
    return dw, db


def to_nhwc_bf16(x):
    """fp32 [n,32,h,w] -> the bf16 channel-contiguous layout of the bf16 update ([n,h,w,32], include/drqv2_hip.h)."""
    return x.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous()


def from_nhwc_bf16(y):
    """the inverse, as fp32 [n,32,h,w] (the bf16 values, exactly)."""
    return y.to(torch.float32).permute(0, 3, 1, 2).contiguous()


def conv3x3_fwd_bf16_nhwc(x, w, b, relu=True, y_nhwc=True):
    """x: fp32 [nb,32,hin,hin] or bf16 [nb,hin,hin,32] (layout above) -> y in that layout (bf16) or fp32 NCHW."""
    lib = _lib.load()
    x_nhwc = x.dtype == torch.bfloat16
    nb = x.shape[0]
    hin = x.shape[1] if x_nhwc else x.shape[2]
    hout = hin - 2
    assert x.is_cuda and x.is_contiguous() and (x.shape[3] == 32 if x_nhwc else x.shape[1] == 32)
    y = (torch.empty((nb, hout, hout, 32), device=x.device, dtype=torch.bfloat16) if y_nhwc else
         torch.empty((nb, 32, hout, hout), device=x.device, dtype=torch.float32))
    check(lib.drq_conv3x3_fwd_bf16_nhwc(ptr(x), ptr(w), ptr(b), ptr(y), nb, hin, int(relu), int(x_nhwc), int(y_nhwc),
                                        _stream()), "drq_conv3x3_fwd_bf16_nhwc")
    return y


def conv3x3_dgrad_bf16_nhwc(dy_pad, w, mask_nhwc, dx_nhwc=False):
    """dy_pad: fp32 [nb,32,hout+4,hout+4] or bf16 [nb,hout+4,hout+4,32] (zero border of 2); the mask bf16 [nb,hin,hin,32].
    -> dx fp32 [nb,32,hin,hin], or (dx_nhwc) a zeroed bf16 [nb,hin+4,hin+4,32] buffer with dx in its interior."""
    lib = _lib.load()
    _need(w, name="w")
    dy_nhwc = dy_pad.dtype == torch.bfloat16
    nb = dy_pad.shape[0]
    hp = dy_pad.shape[1] if dy_nhwc else dy_pad.shape[2]
    hout = hp - 4
    hin = hout + 2
    assert dy_pad.is_cuda and dy_pad.is_contiguous()
    assert mask_nhwc.dtype == torch.bfloat16 and tuple(mask_nhwc.shape) == (nb, hin, hin, 32) and mask_nhwc.is_contiguous()
    dx = (torch.zeros((nb, hin + 4, hin + 4, 32), device=dy_pad.device, dtype=torch.bfloat16) if dx_nhwc else
          torch.empty((nb, 32, hin, hin), device=dy_pad.device, dtype=torch.float32))
    check(lib.drq_conv3x3_dgrad_bf16_nhwc(ptr(dy_pad), ptr(w), ptr(mask_nhwc), ptr(dx), nb, hout, int(dy_nhwc), int(dx_nhwc),
                                          32 * hin * hin, hin * hin, hin, 0, _stream()), "drq_conv3x3_dgrad_bf16_nhwc")
    return dx


def conv3x3_wgrad_bf16_nhwc(x_nhwc, dy):
    """x bf16 [nb,hin,hin,32]; dy: an fp32 strided view [nb,32,hout,hout] or the padded bf16 [nb,hout+4,hout+4,32] buffer."""
    lib = _lib.load()
    nb, hin, _, _ = x_nhwc.shape
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.is_contiguous() and x_nhwc.shape[3] == 32
    dw = torch.empty((32, 32, 3, 3), device=dy.device, dtype=torch.float32)
    db = torch.empty((32,), device=dy.device, dtype=torch.float32)
    nbytes = lib.drq_conv3x3_wgrad_ws_bytes()
    ws = torch.empty((nbytes // 4,), device=dy.device, dtype=torch.float32)
    if dy.dtype == torch.bfloat16:
        assert tuple(dy.shape) == (nb, hin + 2, hin + 2, 32) and dy.is_contiguous()
        check(lib.drq_conv3x3_wgrad_bf16_nhwc(ptr(x_nhwc), ptr(dy), ptr(dw), ptr(db), nb, hin, 1, 0, 0, 0, 0, ptr(ws), nbytes,
                                              _stream()), "drq_conv3x3_wgrad_bf16_nhwc")
    else:
        check(lib.drq_conv3x3_wgrad_bf16_nhwc(ptr(x_nhwc), dy.data_ptr(), ptr(dw), ptr(db), nb, hin, 0, dy.stride(0),
                                              dy.stride(1), dy.stride(2), 0, ptr(ws), nbytes, _stream()),
              "drq_conv3x3_wgrad_bf16_nhwc")
    return dw, db


def gemm(A, a_kc, B, b_kc, M, N, K, lda=None, ldb=None, bias=None, relu=False, aux=None, nbatch=1, a_bs=0,
         b_bs=0, c_bs=None, bias_bs=0, aux_bs=0, tile=0, splitk=0, out=None, ldc=None):
    lib = _lib.load()
    dev = A.device
    lda = lda if lda is not None else (K if a_kc else M)
    ldb = ldb if ldb is not None else (K if b_kc else N)
    ldc = ldc if ldc is not None else N
    c_bs = c_bs if c_bs is not None else M * N
    if out is None:
        out = torch.empty((nbatch, M, N) if nbatch > 1 else (M, N), device=dev, dtype=torch.float32)
    ws = torch.empty((16 * 1024 * 1024,), device=dev, dtype=torch.float32)
    check(lib.drq_gemm_f32(ptr(A), lda, int(a_kc), ptr(B), ldb, int(b_kc), ptr(out), ldc, M, N, K, nbatch, a_bs, b_bs,
                           c_bs, ptr(bias), bias_bs, int(relu), ptr(aux), (aux.shape[-1] if aux is not None else 0),
                           aux_bs, 0, tile, splitk, ptr(ws), ws.numel() * 4, _stream()), "drq_gemm_f32")
    return out


def linear_fwd(x, w, b, relu=False, **kw):
    M, K = x.shape
    N = w.shape[0]
    return gemm(x, True, w, True, M, N, K, bias=b, relu=relu, **kw)


def linear_dgrad(dy, w, mask=None, **kw):
    M, K = dy.shape           # K = out features
    N = w.shape[1]
    return gemm(dy, True, w, False, M, N, K, aux=mask, **kw)


def linear_wgrad(dy, x, **kw):
    Brows, N = dy.shape
    K = x.shape[1]
    dw = gemm(dy, False, x, False, N, K, Brows, lda=N, ldb=K, **kw)
    lib = _lib.load()
    db = torch.empty((N,), device=dy.device, dtype=torch.float32)
    check(lib.drq_colsum(ptr(dy), N, 0, ptr(db), 0, Brows, N, 1, _stream()), "drq_colsum")
    return dw, db


def ln_tanh_fwd(z, gamma, beta, save=True):
    lib = _lib.load()
    rows, F = z.shape
    out = torch.empty_like(z)
    xhat = torch.empty_like(z) if save else None
    rstd = torch.empty((rows,), device=z.device, dtype=torch.float32) if save else None
    check(lib.drq_ln_tanh_fwd(ptr(z), F, ptr(gamma), ptr(beta), ptr(out), F, ptr(xhat), ptr(rstd), rows, F, _stream()),
          "drq_ln_tanh_fwd")
    return out, xhat, rstd


def ln_tanh_bwd(dh, h, xhat, rstd, gamma):
    lib = _lib.load()
    rows, F = dh.shape
    dz, dln = torch.empty_like(dh), torch.empty_like(dh)
    dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
    check(lib.drq_ln_tanh_bwd(ptr(dh), F, None, 0, ptr(h), F, ptr(xhat), ptr(rstd), ptr(gamma), ptr(dz), ptr(dln),
                              ptr(dg), ptr(db), rows, F, _stream()), "drq_ln_tanh_bwd")
    return dz, dg, db


def trunc_normal_sample(pre_tanh, noise, std, clip):
    lib = _lib.load()
    B, A = pre_tanh.shape
    mu, a = torch.empty_like(pre_tanh), torch.empty_like(pre_tanh)
    check(lib.drq_trunc_normal_sample(ptr(pre_tanh), ptr(noise), float(std), float(clip if clip is not None else 0.0),
                                      int(clip is not None), ptr(mu), ptr(a), A, B, A, _stream()),
          "drq_trunc_normal_sample")
    return mu, a


def adam_flat(p, g, m, v, lr, step, gscale=1.0, tgt=None, tau=0.0):
    lib = _lib.load()
    check(lib.drq_adam_flat(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), int(step), float(gscale), ptr(tgt),
                            float(tau), _stream()), "drq_adam_flat")


def ema_flat(p, t, tau):
    lib = _lib.load()
    check(lib.drq_ema_flat(ptr(p), ptr(t), p.numel(), float(tau), _stream()), "drq_ema_flat")


def tanh(x):
    lib = _lib.load()
    y = torch.empty_like(x)
    check(lib.drq_tanh(ptr(x), ptr(y), x.numel(), _stream()), "drq_tanh")
    return y


def _ptr_array(ts):
    """host array of device pointers (None entries allowed) for the batched entry points."""
    import ctypes
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def gemm_batched(As, a_kc, Bs, b_kc, M, N, K, lda, ldb, biases=None, relu=False, auxs=None, rowsum=False, tile=0,
                 splitk=0, scatter_hw=0, Cs=None, bf16=False):
    """n independent problems of one shape in a single launch.  Returns (list of C, list of rowsum or None).
    scatter_hw > 0: C_i is a caller-provided zero-padded [M][32][hw+4][hw+4] gradient buffer (pass Cs)."""
    lib = _lib.load()
    n = len(As)
    dev = As[0].device
    if Cs is None:
        Cs = [torch.empty((M, N), device=dev, dtype=torch.float32) for _ in range(n)]
    rs = [torch.empty((M,), device=dev, dtype=torch.float32) for _ in range(n)] if rowsum else None
    ws = torch.empty((16 * 1024 * 1024,), device=dev, dtype=torch.float32)
    if bf16:
        check(lib.drq_gemm_batched_bf16(n, _ptr_array(As), lda, int(a_kc), _ptr_array(Bs), ldb, int(b_kc), _ptr_array(Cs),
                                        N, M, N, K, _ptr_array(biases) if biases else None, int(relu),
                                        _ptr_array(auxs) if auxs else None, (auxs[0].shape[-1] if auxs else 0),
                                        _ptr_array(rs) if rs else None, scatter_hw, splitk, ptr(ws), ws.numel() * 4,
                                        _stream()), "drq_gemm_batched_bf16")
        return Cs, rs
    check(lib.drq_gemm_batched_f32(n, _ptr_array(As), lda, int(a_kc), _ptr_array(Bs), ldb, int(b_kc), _ptr_array(Cs),
                                   N, M, N, K, _ptr_array(biases) if biases else None, int(relu),
                                   _ptr_array(auxs) if auxs else None, (auxs[0].shape[-1] if auxs else 0),
                                   _ptr_array(rs) if rs else None, scatter_hw, tile, splitk, ptr(ws), ws.numel() * 4,
                                   _stream()), "drq_gemm_batched_f32")
    return Cs, rs


def gemm_batched_partial(As, Bs, M, N, K, lda, ldb):
    """Forward form (both operands k-contiguous) through the internal entry the update uses for the trunk:
    returns (sum of the split-K partial records per problem [n][M][N], split count).  Test / tool helper."""
    import ctypes
    lib = _lib.load()
    n = len(As)
    dev = As[0].device
    Cs = [torch.zeros((M, N), device=dev, dtype=torch.float32) for _ in range(n)]
    ws = torch.zeros((16 * 1024 * 1024,), device=dev, dtype=torch.float32)
    sk = ctypes.c_int(0)
    check(lib.drq_gemm_batched_partial(n, _ptr_array(As), lda, 1, _ptr_array(Bs), ldb, 1, _ptr_array(Cs), N, M, N, K, None, ptr(ws),
             ws.numel() * 4, ctypes.byref(sk), _stream()), "drq_gemm_batched_partial")
    k = sk.value
    if k <= 1:
        return torch.stack(Cs), k
    return ws[: n * k * M * N].view(n, k, M, N).sum(dim=1), k


def qout_fwd(hs, ws_, bs):
    lib = _lib.load()
    B, H = hs[0].shape
    qs = [torch.empty((B,), device=hs[0].device, dtype=torch.float32) for _ in hs]
    check(lib.drq_qout_fwd(len(hs), _ptr_array(hs), _ptr_array(ws_), _ptr_array(bs), _ptr_array(qs), B, H, _stream()),
          "drq_qout_fwd")
    return qs


def qout_bwd(dqs, hs, ws_, want_wgrad=True):
    lib = _lib.load()
    B, H = hs[0].shape
    dev = hs[0].device
    dhs = [torch.empty((B, H), device=dev, dtype=torch.float32) for _ in hs]
    dws = [torch.empty((H,), device=dev, dtype=torch.float32) for _ in hs] if want_wgrad else None
    dbs = [torch.empty((1,), device=dev, dtype=torch.float32) for _ in hs] if want_wgrad else None
    check(lib.drq_qout_bwd(len(hs), _ptr_array(dqs), _ptr_array(hs), _ptr_array(ws_), _ptr_array(dhs),
                           _ptr_array(dws) if dws else None, _ptr_array(dbs) if dbs else None, B, H, _stream()),
          "drq_qout_bwd")
    return dhs, dws, dbs


def mlp_fwd(xs, ws_, bs, relu=True, qws=None):
    """hidden-layer forward on the LDS-DMA ring kernel; qws: weight rows of a following Linear(N,1) -> partial dots.
    Returns (list of y, list of qpart [M][nq] or None)."""
    import ctypes
    lib = _lib.load()
    M, K = xs[0].shape
    N = ws_[0].shape[0]
    dev = xs[0].device
    ys = [torch.empty((M, N), device=dev, dtype=torch.float32) for _ in xs]
    qps = [torch.zeros((M, N // 32), device=dev, dtype=torch.float32) for _ in xs] if qws else None
    nq = ctypes.c_int(0)
    check(lib.drq_mlp_fwd(len(xs), _ptr_array(xs), xs[0].stride(0), _ptr_array(ws_), ws_[0].stride(0), _ptr_array(ys), N,
                          M, N, K, _ptr_array(bs) if bs else None, int(relu), _ptr_array(qws) if qws else None,
                          _ptr_array(qps) if qps else None, ctypes.byref(nq), _stream()), "drq_mlp_fwd")
    if qps:
        qps = [q.view(-1)[: M * nq.value].view(M, nq.value) for q in qps]
    return ys, qps


def mlp_dgrad(dys, ws_, masks=None):
    lib = _lib.load()
    M, K = dys[0].shape
    N = ws_[0].shape[1]
    dxs = [torch.empty((M, N), device=dys[0].device, dtype=torch.float32) for _ in dys]
    check(lib.drq_mlp_dgrad(len(dys), _ptr_array(dys), dys[0].stride(0), _ptr_array(ws_), ws_[0].stride(0), _ptr_array(dxs),
                            N, M, N, K, _ptr_array(masks) if masks else None, (masks[0].stride(0) if masks else 0),
                            _stream()), "drq_mlp_dgrad")
    return dxs


def mlp_wgrad_dgrad(dys, xs, ws_, masks=None):
    """both gradients of one hidden layer in one launch: returns (dws, dbs, dxs)."""
    lib = _lib.load()
    Brows, Nout = dys[0].shape
    Kin = xs[0].shape[1]
    dev = dys[0].device
    dws = [torch.empty((Nout, Kin), device=dev, dtype=torch.float32) for _ in dys]
    dbs = [torch.empty((Nout,), device=dev, dtype=torch.float32) for _ in dys]
    dxs = [torch.empty((Brows, Kin), device=dev, dtype=torch.float32) for _ in dys]
    check(lib.drq_mlp_wgrad_dgrad(len(dys), _ptr_array(dys), dys[0].stride(0), _ptr_array(xs), xs[0].stride(0),
                                  _ptr_array(dws), _ptr_array(dbs), _ptr_array(ws_), ws_[0].stride(0), _ptr_array(dxs),
                                  Kin, _ptr_array(masks) if masks else None, (masks[0].stride(0) if masks else 0),
                                  Brows, Nout, Kin, _stream()), "drq_mlp_wgrad_dgrad")
    return dws, dbs, dxs


def _int_array(vals):
    import ctypes
    return (ctypes.c_int * len(vals))(*vals)


def ln_l1_fwd(jobs, F, H, splitk=0, slab=0):
    """jobs: list of dicts with keys z or part (+bias), gamma, beta, rows, optional tail [rows][A], save (bool),
    heads: list of (w [H][K], b [H]).  Returns per job dict(out [rows][F+tail_n], xhat, rstd, ys=[...])."""
    lib = _lib.load()
    dev = jobs[0]["gamma"].device
    res, part, z, bias, gamma, beta, out, ldo, xhat, rstd, tail, tld, tn, rows, nh, w, b, y = ([] for _ in range(18))
    for j in jobs:
        r = j["rows"]
        t = j.get("tail")
        n_t = t.shape[1] if t is not None else 0
        o = torch.zeros((r, F + n_t), device=dev, dtype=torch.float32)
        xh = torch.empty((r, F), device=dev, dtype=torch.float32) if j.get("save", True) else None
        rs = torch.empty((r,), device=dev, dtype=torch.float32) if j.get("save", True) else None
        heads = j.get("heads", [])
        ys = [torch.empty((r, H), device=dev, dtype=torch.float32) for _ in heads]
        res.append(dict(out=o, xhat=xh, rstd=rs, ys=ys))
        part.append(j.get("part")); z.append(j.get("z")); bias.append(j.get("bias"))
        gamma.append(j["gamma"]); beta.append(j["beta"]); out.append(o); ldo.append(F + n_t)
        xhat.append(xh); rstd.append(rs); tail.append(t); tld.append(t.stride(0) if t is not None else 0); tn.append(n_t)
        rows.append(r); nh.append(len(heads))
        for h in range(2):
            w.append(heads[h][0] if h < len(heads) else None)
            b.append(heads[h][1] if h < len(heads) else None)
            y.append(ys[h] if h < len(heads) else None)
    check(lib.drq_ln_l1_fwd(len(jobs), _ptr_array(part) if splitk else None, _ptr_array(z) if not splitk else None,
                            _ptr_array(bias), _ptr_array(gamma), _ptr_array(beta), _ptr_array(out), _int_array(ldo),
                            _ptr_array(xhat), _ptr_array(rstd), _ptr_array(tail), _int_array(tld), _int_array(tn),
                            _int_array(rows), _int_array(nh), _ptr_array(w), _ptr_array(b), _ptr_array(y), F, H, splitk,
                            slab, _stream()), "drq_ln_l1_fwd")
    return res


def policy_out_l1_fwd(p2, w3, b3, srow0, F, std, clip, noise_hi, ha_hi, noise_lo=None, ha_lo=None, heads=None):
    """policy output layer + sample (+ first layers on the hi rows).  ha_hi [rows-srow0][F+A] holds h in its first F
    columns; returns dict(p3, mu_hi, mu_lo, ys)."""
    lib = _lib.load()
    rows, H = p2.shape
    A = w3.shape[0]
    dev = p2.device
    p3 = torch.empty((rows, A), device=dev, dtype=torch.float32)
    mu_hi = torch.empty((rows - srow0, A), device=dev, dtype=torch.float32)
    mu_lo = torch.empty((srow0, A), device=dev, dtype=torch.float32) if noise_lo is not None else None
    ys = [torch.empty((rows - srow0, H), device=dev, dtype=torch.float32) for _ in (heads or [])]
    check(lib.drq_policy_out_l1_fwd(ptr(p2), ptr(w3), ptr(b3), ptr(p3), rows, srow0, H, A, F, float(std),
                                    float(clip if clip is not None else 0.0), int(clip is not None), ptr(noise_hi),
                                    ptr(mu_hi), ptr(ha_hi), ha_hi.stride(0), ptr(noise_lo), ptr(mu_lo), ptr(ha_lo),
                                    (ha_lo.stride(0) if ha_lo is not None else 0), len(heads or []),
                                    _ptr_array([h[0] for h in heads]) if heads else None,
                                    _ptr_array([h[1] for h in heads]) if heads else None,
                                    _ptr_array(ys) if heads else None, _stream()), "drq_policy_out_l1_fwd")
    return dict(p3=p3, mu_hi=mu_hi, mu_lo=mu_lo, ys=ys)
