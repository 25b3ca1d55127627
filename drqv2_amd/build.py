"""Build libdrqv2_hip.so (gfx950 only) in-tree with hipcc.  `python -m drqv2_amd.build [--force] [--dev]`.

--dev builds libdrqv2_hip_dev.so with -DDRQ_DEV: the same kernels plus the timing ablations, time-stamp hooks and
environment knobs that tools/ uses.  The product library has none of them (no getenv, no drq_dev_* exports)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdrqv2_hip.so")
LIB_DEV = os.path.join(HERE, "libdrqv2_hip_dev.so")
SOURCES = ["conv.hip", "conv_wino.hip", "conv_wino_wgrad.hip", "conv1aug.hip", "conv_bf16.hip", "gemm.hip", "gemm2.hip", "gemm3.hip", "rowblock.hip", "skinny.hip", "elementwise.hip", "rng.hip", "step.hip"]
# per-file additions.  conv_wino.hip: hipcc's SLP vectoriser packs the transform adds into v_pk_add_f32 plus the
# v_mov shuffles that feed them -- more VALU issue slots beside the MFMAs, not fewer (136 moves per unit)
# rng.hip: hipRAND's Box-Muller (logf, sincosf, the scaling multiply-adds) must round like the copy inside torch's
# own normal_() kernel: -ffp-contract=on does (0 of 65,536 values differ; hipcc's default fast contraction: 15 %
# differ in the last bit; contraction off: 0.7 %) -- tools/rng_flags_probe.py, and the engine's self test at run time
FILE_FLAGS = {"conv_wino.hip": ["-fno-slp-vectorize"], "conv_wino_wgrad.hip": ["-fno-slp-vectorize"], "rng.hip": ["-ffp-contract=on"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fvisibility=hidden"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "drqv2_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True, dev=False):
    lib = LIB_DEV if dev else LIB
    if not force and not needs_build(lib):
        return lib
    objdir = os.path.join(HERE, "build", "dev" if dev else "prod")
    os.makedirs(objdir, exist_ok=True)
    flags = FLAGS + (["-DDRQ_DEV"] if dev else [])
    procs = []
    for src in _sources():
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [_hipcc()] + flags + FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((obj, subprocess.Popen(cmd)))
    objs = []
    for obj, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed for {obj}")
        objs.append(obj)
    # -z defs: an internal entry point declared with the wrong linkage fails here, not at dlopen on the GPU box
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,defs", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, dev="--dev" in sys.argv))
