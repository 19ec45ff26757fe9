#!/usr/bin/env python3
"""Experiment behind the measured-share partition (diagnostics build: make HIPFLAGS+=-DNBNXM_WAVE_TIMELINE): read the per-wave
timeline of a launch, give the ranges of a slow SIMD a smaller share of the work, launch again."""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fep_testlib as tl
pkg = tl.pkg
alpha = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
case = tl.make_case(nm=(40, 40, 20), num_perturbed_molecules=16, elec="ewald", seed=2026, n_lambda=11, max_cjpacked_per_sci=16)
nb = tl.setup_gpu(case, fused=True, use_dynamic_pruning=True)
sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
lib = pkg.hip_lib()
n = 5120
buf = (C.c_ulonglong * (4 * n))()


def run(k=30):
    for _ in range(5):
        nb.clear_outputs(False); nb.launch_kernel(sw)
    torch.cuda.synchronize()
    nb.set_timing(True); nb.reset_timings()
    for _ in range(k):
        nb.clear_outputs(False); nb.launch_kernel(sw)
    torch.cuda.synchronize()
    f = np.zeros((case.grid.num_atoms, 3), np.float32)
    nb.launch_cpyback(f, sw); nb.wait_finish_task(sw, case.have_soft_core)
    tm = nb.get_timings(); nb.set_timing(False)
    lib.nbnxm_gpu_debug_timeline(C.c_void_p(nb._h), buf, n)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).copy()
    end = (a[:, 2] - a[:, 0].min()).astype(np.float64) / 100.0
    hw = a[:, 3] & np.uint64(0xFFFFFFFF); xcc = (a[:, 3] >> np.uint64(32)) & np.uint64(0xF)
    simd = (hw >> np.uint64(4)) & np.uint64(3); cu = (hw >> np.uint64(8)) & np.uint64(0xF); sh = (hw >> np.uint64(12)) & np.uint64(1)
    se = (hw >> np.uint64(13)) & np.uint64(7)
    return 1e3 * tm.nb_k_ms / max(1, tm.nb_k_count), end, (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd


share = np.ones(n)
for it in range(5):
    us, end, key = run()
    uk, inv = np.unique(key, return_inverse=True)
    simd_end = np.zeros(len(uk)); np.maximum.at(simd_end, inv, end)
    print("iteration %d: kernel %.1f us; per-SIMD finish mean %.1f std %.2f max %.1f; wave ends p10 %.1f p50 %.1f p90 %.1f"
          % (it, us, simd_end.mean(), simd_end.std(), simd_end.max(), *np.percentile(end, [10, 50, 90])))
    worst = np.argsort(simd_end)[-6:]
    for k in worst:
        ws = np.nonzero(inv == k)[0]
        print("   slow SIMD %5d: end %.1f, waves %s ends %s shares %s" % (uk[k], simd_end[k], ws.tolist(), np.round(end[ws], 1).tolist(), np.round(share[ws], 2).tolist()))
    # within a SIMD: every wave should end with the last one; across SIMDs: every SIMD with the mean
    target = simd_end.mean()
    share *= (target / end) ** alpha if "--per-wave" in sys.argv else (target / simd_end[inv]) ** alpha
    share = np.clip(share / share.mean(), 0.7, 1.4)
    sh = np.ascontiguousarray(share, np.float32)
    lib.nbnxm_gpu_debug_set_work_shares(C.c_void_p(nb._h), 0, 1, sh.ctypes.data_as(C.c_void_p), n)
