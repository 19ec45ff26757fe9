#!/bin/bash
# tools/regs.sh FILE.hip [extra flags]: VGPRs, scratch bytes and waves/SIMD of every kernel of one translation unit (CPU only)
D=$(mktemp -d); cd $D
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fno-honor-nans -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=4 -I/root/repo/include -I/root/repo/gromacs-fep-gpu_amd/csrc "${@:2}" --save-temps -c /root/repo/gromacs-fep-gpu_amd/csrc/$1 -o x.o 2>&1 | grep -v warning | grep -i "error" | head -5
F=$(ls *gfx950*.s 2>/dev/null | head -1); [ -z "$F" ] && { echo "compile failed"; exit 1; }
awk '/^_Z.*:/{name=$1} /^; NumVgprs:/{v=$3} /^; ScratchSize:/{s=$3} /^; Occupancy:/{print substr(name,1,60), "vgpr",v,"scratch",s,"occ",$3}' $F | grep -v "^ " | sort -u
cp $F /tmp/last_regs.s
rm -rf $D
