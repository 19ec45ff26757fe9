#!/bin/bash
# CPU only: the host producers (libnbnxm_host.so) and the oracle (liboracle.so) rebuilt with AddressSanitizer + UBSan in a scratch copy, and
# the CPU test files that load them run with the sanitizer runtimes preloaded; prints the number of sanitizer reports (0 expected).
set -e
SRC=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
cp -r $SRC/gromacs-fep-gpu_amd $SRC/include $SRC/oracle $SRC/tests $SRC/pytest.ini $SRC/__graft_entry__.py $T/
cd $T
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
make -s -C gromacs-fep-gpu_amd host CXXFLAGS="-O1 -std=c++17 -fPIC -fopenmp -Wall $SAN" -B
make -s -C oracle CFLAGS="-O1 -fPIC -std=c11 -Wall -Wno-unused-parameter -ffp-contract=off -fopenmp $SAN" -B
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest tests/test_domdec_gloo.py tests/test_oracle_golden.py tests/test_oracle_nblib.py tests/test_pairlist_cpu.py tests/test_launch_plan.py \
  tests/test_launch_shape.py tests/test_abi_symbols.py -q -s -m "not gpu" -p no:cacheprovider > out.log 2>&1 || true
tail -1 out.log
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' out.log || true)"
cd /; rm -rf $T
