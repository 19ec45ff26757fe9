#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...]: an alternative build of libnbnxm_hip.so into variants/NAME.so
# (run it with NBNXM_HIP_LIB=variants/NAME.so); variants/ is not tracked.
set -e
cd "$(dirname "$0")/../gromacs-fep-gpu_amd"
NAME=$1; shift
B=../variants/build_$NAME
mkdir -p $B
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-honor-nans -fno-slp-vectorize ${PRELOAD--mllvm -amdgpu-kernarg-preload-count=4} -I../include -Icsrc $@"
pids=()
for f in csrc/*.hip; do
  o=$B/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc $FLAGS -c $f -o $o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$NAME.so $B/*.o -ldl
echo built variants/$NAME.so
