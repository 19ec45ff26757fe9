"""Diagnostics: range sizes of the work partition on a small box with / without perturbed molecules."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
wl = importlib.import_module("gromacs_fep_gpu_amd.workload")
lib = pkg.hip_lib()
lib.nbnxm_gpu_debug_get_work_ranges.restype = C.c_void_p
for pm in (0, 16):
    c = wl.make_case(nm=(10, 10, 10), num_perturbed_molecules=pm, elec="ewald", seed=2026, n_lambda=11, max_cjpacked_per_sci=16)
    nb = wl.setup_gpu(c, fused=True, use_dynamic_pruning=True)
    sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
    for _ in range(3):
        nb.clear_outputs(False)
        nb.launch_kernel(sw)
    nr = C.c_int(0)
    ptr = lib.nbnxm_gpu_debug_get_work_ranges(nb.h, 0, 1, C.byref(nr))
    n = nr.value
    ranges = np.zeros(n + 1, np.int32)
    lib.nbnxm_gpu_debug_download(nb.h, C.c_void_p(ptr), ranges.ctypes.data_as(C.c_void_p), C.c_size_t(ranges.nbytes))
    sizes = np.diff(ranges)
    cj = pkg.download_cjpacked(nb, len(c.plist_fused.cjPacked))
    bits = np.array([bin(int(m)).count("1") for m in cj["imei"][:, 0]["imask"]])
    per_range = np.array([bits[ranges[i]:ranges[i + 1]].sum() for i in range(n)])
    print("perturbed molecules %d: groups %d, ranges %d, groups per range min/mean/max %d / %.1f / %d, cluster pairs per range min/mean/max %d / %.0f / %d"
          % (pm, len(bits), n, sizes.min(), sizes.mean(), sizes.max(), per_range.min(), per_range.mean(), per_range.max()), flush=True)
    nb.free()
