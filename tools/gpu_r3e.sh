#!/bin/bash
# round 3, session e: the domain step on one GPU (self-links), both transports; the --dd rehearsal with its first-step parity check;
# the 1M-atom test incl. the C++ step over the peer-copy transport
OUT=gpurun_out; mkdir -p $OUT
for t in rccl peer; do
  HALO_GPU_HOST_TIMING=1 timeout -k 10 300 python tools/dd_single_gpu_probe.py 96k xyz $t 2>&1 | grep -v amdgpu.ids | tee -a $OUT/r3e_ddprobe.txt || exit 1
done
timeout -k 10 300 python bench.py --dd --atoms 96k --steps 100 --warmup 10 2>&1 | grep -v amdgpu.ids | tee $OUT/r3e_dd_bench.txt || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "1m" > $OUT/r3e_pytest.log 2>&1; tail -5 $OUT/r3e_pytest.log
