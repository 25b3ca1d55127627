"""bf16-MFMA conv kernels against the fp32 ones, in isolation (conv2's shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drqv2_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *sh: torch.randn(*sh, device="cuda", generator=g)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for hin in (41, 39, 37):
    hout = hin - 2
    w, b = rn(32, 32, 3, 3) * 0.1, rn(32) * 0.1
    x2 = rn(2 * B, 32, hin, hin)
    x = x2[:B].contiguous()
    dy_pad = torch.zeros(B, 32, hout + 4, hout + 4, device="cuda")
    dy_pad[:, :, 2:-2, 2:-2] = rn(B, 32, hout, hout)
    mask = rn(B, 32, hin, hin)
    for bf in (False, True):
        tf = timeit(lambda: ops.conv3x3_fwd(x2, w, b, 1, bf16=bf))
        td = timeit(lambda: ops.conv3x3_dgrad(dy_pad, w, mask, bf16=bf))
        tw = timeit(lambda: ops.conv3x3_wgrad(x, dy_pad[:, :, 2:-2, 2:-2], 1, bf16=bf))
        print(f"hin={hin} B={B} {'bf16' if bf else 'fp32'}: fwd(2B) {tf:7.1f} us  dgrad {td:7.1f} us  wgrad(+reduce, +ws alloc) {tw:7.1f} us", flush=True)
