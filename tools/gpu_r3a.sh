#!/bin/bash
# round 3, session a: contiguous i-force adds -- parity subset, A/B against the strided form, flavours that changed registers
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nblib.py -x -q -m gpu > $OUT/r3a_pytest.log 2>&1 || { tail -30 $OUT/r3a_pytest.log; exit 1; }
tail -3 $OUT/r3a_pytest.log
L=gromacs-fep-gpu_amd/lib/libnbnxm_hip.so
bash tools/gpu_ab.sh $L variants/istrided.so | tee $OUT/r3a_ab.txt || exit 1
for v in $L variants/istrided.so; do
  echo "== $v"; NBNXM_HIP_LIB=$v FLAVOURS=ewald/cut,ewald/fswitch,ewald/pswitch timeout -k 10 300 python tools/flavour_probe.py 300 2>&1 | tee -a $OUT/r3a_flavours.txt
done
