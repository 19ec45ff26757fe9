#!/usr/bin/env python3
"""How the 64 lanes of the listed 8 x 8 pair blocks are used (diagnostics build: tools/build_variant.sh stats -DNBNXM_BLOCK_STATS, then
NBNXM_HIP_LIB=variants/stats.so): executed blocks, blocks with an empty 32-lane half (what per-half masks could skip if halves of
different j-clusters could be packed into one instruction), active lanes per executed block.  usage: block_stats_probe.py [24k|96k]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fep_testlib as tl
pkg = tl.pkg
nm = {"24k": (20, 20, 20), "96k": (40, 40, 20)}[sys.argv[1] if len(sys.argv) > 1 else "96k"]
case = tl.make_case(nm=nm, num_perturbed_molecules=16, elec="ewald", seed=2026, n_lambda=11, max_cjpacked_per_sci=16)
nb = tl.setup_gpu(case, fused=True, use_dynamic_pruning=True)
sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
lib = pkg.hip_lib()
n = 16384
buf = (C.c_ulonglong * (4 * n))()


def counters():
    torch.cuda.synchronize()
    lib.nbnxm_gpu_debug_timeline(C.c_void_p(nb._h), buf, n)
    return np.frombuffer(buf, dtype=np.uint64)[4 * n - 64:4 * n - 57].astype(np.int64).copy()


for _ in range(3):      # the first launch prunes the list
    nb.clear_outputs(False); nb.launch_kernel(sw)
c0 = counters()
nb.clear_outputs(False); nb.launch_kernel(sw)
c = counters() - c0
listed, executed, lo_empty, hi_empty, lanes, any_quarter, quarters = [int(v) for v in c]
print("listed pair blocks %d (cluster pairs of the pruned list); executed (some lane within the cut-off) %d = %.3f" % (listed, executed, executed / listed))
print("active lanes per executed block: %.1f of 64" % (lanes / executed))
print("executed blocks with an empty 32-lane half: lower %d, upper %d = %.3f of the executed blocks" % (lo_empty, hi_empty, (lo_empty + hi_empty) / executed))
print("  -> perfectly packed (two such halves of different j-clusters into one instruction) these would cost half: %.3f of today's executed blocks"
      % (1.0 - 0.5 * (lo_empty + hi_empty) / executed))
print("executed blocks with at least one empty 16-lane quarter: %.3f; empty quarters per executed block %.2f of 4" % (any_quarter / executed, quarters / executed))
nb.free()
