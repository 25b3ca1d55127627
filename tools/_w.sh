bash tools/gpu_trace.sh t32c --task cartpole_swingup --batch 32 > gpurun_out/t32c.log 2>&1; tail -62 gpurun_out/t32c.log | head -58
bash tools/gpu_trace.sh t32h --task humanoid_run --batch 32 > gpurun_out/t32h.log 2>&1; tail -62 gpurun_out/t32h.log | head -58
