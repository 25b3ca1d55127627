"""Trunk forward (z = feat W^T, K = 39200, N = 50) timing: dedicated kernel vs the LDS-tiled one (dev tool).
DRQ_NO_TRUNK_KERNEL=1 selects the LDS-tiled kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops
M, N, K = 256, 50, 39200
for n in (4, 1):
    xs = [torch.randn(M, K, device="cuda") for _ in range(2)]
    ws = [torch.randn(N, K, device="cuda") * K ** -0.5 for _ in range(4)]
    A = [xs[0], xs[0], xs[1], xs[1]][:n]
    for _ in range(3):
        got, sk = ops.gemm_batched_partial(A, ws[:n], M, N, K, K, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    import ctypes
    from drqv2_amd import _lib
    lib = _lib.load(dev=True)
    Cs = [torch.zeros((M, N), device="cuda") for _ in range(n)]
    wsb = torch.zeros((16 * 1024 * 1024,), device="cuda")
    skc = ctypes.c_int(0)
    pa, pb, pc = ops._ptr_array(A), ops._ptr_array(ws[:n]), ops._ptr_array(Cs)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        e0.record()
        lib.drq_gemm_batched_partial(n, pa, K, 1, pb, K, 1, pc, N, M, N, K, None, wsb.data_ptr(), wsb.numel() * 4,
                                     ctypes.byref(skc), st)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000)
    ts.sort()
    ref = A[0].double() @ ws[0].double().t()
    err = float((got[0].double() - ref).abs().max() / ref.abs().max())
    print(f"problems={n}: median {ts[len(ts) // 2]:.1f} us  min {ts[0]:.1f} us  split {skc.value}  err {err:.2e}")

# trunk weight gradient dW = dz^T feat (M = 50, N = 39200, K = batch)
for Bk in (256, 128):
    dz = torch.randn(Bk, N, device="cuda")
    feat = torch.randn(Bk, K, device="cuda")
    for _ in range(3):
        (dw,), (db,) = ops.gemm_batched([dz], False, [feat], False, N, K, Bk, N, K, rowsum=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0.record()
        ops.gemm_batched([dz], False, [feat], False, N, K, Bk, N, K, rowsum=True, Cs=[dw])
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000)
    ts.sort()
    ref = dz.double().t() @ feat.double()
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    print(f"wgrad batch={Bk}: median {ts[len(ts) // 2]:.1f} us (includes a 64 MB workspace allocation per call)  err {err:.2e}")
