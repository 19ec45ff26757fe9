#!/bin/bash
# round 3, session f: merged localities -- parity with real peers, then the one-GPU probe, two localities against merged
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merged" > $OUT/r3f_pytest.log 2>&1 || { tail -30 $OUT/r3f_pytest.log; exit 1; }
tail -3 $OUT/r3f_pytest.log
for m in two merged two merged; do
  HALO_GPU_HOST_TIMING=1 timeout -k 10 300 python tools/dd_single_gpu_probe.py 96k xyz rccl $m 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tee -a $OUT/r3f_ddprobe.txt || exit 1
done
