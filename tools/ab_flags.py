#!/usr/bin/env python3
"""Same-process, interleaved A/B of the update's schedule switches (DrqStep.flags): 0 = production, 1 = no row fusion,
2 = no gemm3, 3 = neither (the round-2 schedule).  Usage: tools/ab_flags.py [task] [batch] [rounds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    task = sys.argv[1] if len(sys.argv) > 1 else "cheetah_run"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    flags = [int(f) for f in os.environ.get("AB_FLAGS", "0,1,2,3").split(",")]
    r = bench.Runner(task, B, B, torch.device("cuda", 0), 0, 1, False)
    res = {f: [] for f in flags}
    r.run(30, 10)
    for _ in range(rounds):
        for f in flags:
            r.agent._engine.step_flags = f
            out = r.run(60, 5)
            res[f].append(out["ms_per_step"])
    for f in flags:
        v = sorted(res[f])
        print(f"flags={f}: median {v[len(v)//2]*1e3:8.1f} us  min {v[0]*1e3:8.1f} us   ({task} B={B})", flush=True)


if __name__ == "__main__":
    main()
