#!/bin/bash
# round 3, session g: the energy flavours' trailing workgroups -- parity, then energy / dH/dl step times by tail level
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nblib.py tests/test_gpu_update.py -x -q -m gpu > $OUT/r3g_pytest.log 2>&1 || { tail -30 $OUT/r3g_pytest.log; exit 1; }
tail -3 $OUT/r3g_pytest.log
for rep in 1 2; do for lvl in 0 1 2; do
  echo "== NBNXM_HIP_ENERGY_TAIL=$lvl"; NBNXM_HIP_ENERGY_TAIL=$lvl FLAVOURS=ewald/cut,rf/cut timeout -k 10 300 python tools/flavour_probe.py 300 2>&1 | grep -v amdgpu.ids | tee -a $OUT/r3g_energy_tail.txt
done; done
