"""Runs every tools/bin/rngprobe_<k>.so (tools/rng_flags_probe.hip built with different hipcc flags) against
torch.empty(n).normal_() from the same generator state and prints how many elements differ."""
import ctypes, glob, os, sys
import torch
torch.cuda.init()
gen = torch.cuda.default_generators[0]
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin")
flags = dict(l.strip().split(": ", 1) for l in open(os.path.join(here, "rngprobe_flags.txt")))
n = 1 << 16
torch.manual_seed(7)
st = gen.get_state()
ref = torch.empty(n, device="cuda").normal_()
gen.set_state(st)
seed, off = gen.initial_seed(), gen.get_offset()
for k in sorted(flags, key=int):
    lib = ctypes.CDLL(os.path.join(here, f"rngprobe_{k}.so"))
    lib.probe.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
    out = torch.zeros(n, device="cuda")
    rc = lib.probe(seed, off, n, out.data_ptr())
    nd = int((out != ref).sum())
    print(f"variant {k} [{flags[k]}]: rc {rc} differing {nd} of {n}, max |diff| {float((out - ref).abs().max()):.3e}", flush=True)
