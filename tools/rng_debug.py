"""Why does the fused RNG launch differ from torch's calls?  Prints per-draw equality and the generator offsets."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import _lib
from drqv2_amd._lib import ptr
from torch.distributions.utils import _standard_normal
lib = _lib.load()
torch.cuda.init()
dev = torch.device("cuda", 0)
gen = torch.cuda.default_generators[0]
for n, A in ((4, 3), (256, 6)):
    torch.manual_seed(99)
    st = gen.get_state()
    print(f"n={n} A={A}: seed {gen.initial_seed()} offset before {gen.get_offset()}")
    ref = []
    for k in range(4):
        ref.append(torch.randint(0, 9, size=(n, 1, 1, 2), device=dev, dtype=torch.float32) if k < 2
                   else _standard_normal((n, A), dtype=torch.float32, device=dev))
        print(f"  after torch draw {k}: offset {gen.get_offset()}")
    gen.set_state(st)
    seed, off = gen.initial_seed(), gen.get_offset()
    bufs = [torch.empty(n, 1, 1, 2, device=dev), torch.empty(n, 1, 1, 2, device=dev), torch.empty(n, A, device=dev),
            torch.empty(n, A, device=dev)]
    for trial_off in (off, off // 4, off * 4):
        rc = lib.drq_rng_draws(seed, trial_off, 2 * n, n * A, 9, *(ptr(b) for b in bufs), None)
        torch.cuda.synchronize()
        eq = [bool(torch.equal(a.view(-1), b.view(-1))) for a, b in zip(ref, bufs)]
        md = [float((a.view(-1) - b.view(-1)).abs().max()) for a, b in zip(ref, bufs)]
        nd = [int((a.view(-1) != b.view(-1)).sum()) for a, b in zip(ref, bufs)]
        print(f"  offset {trial_off}: rc {rc} equal {eq} max|diff| {md} ndiff {nd}")
    print("  ref shift[:8]", ref[0].view(-1)[:8].tolist(), "got", bufs[0].view(-1)[:8].tolist())
    print("  ref noise[:4]", ref[2].view(-1)[:4].tolist(), "got", bufs[2].view(-1)[:4].tolist())
    # hypotheses: second value of the quadruple, other elements
