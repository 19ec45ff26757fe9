#!/usr/bin/env python3
"""Per-wave timeline of the cluster-pair kernel (diagnostics build: make HIPFLAGS+=-DNBNXM_WAVE_TIMELINE).
Prints start / end distributions per wave and per SIMD in microseconds (100 MHz wall clock)."""
import ctypes, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fep_testlib as tl
pkg = tl.pkg
mode = sys.argv[1] if len(sys.argv) > 1 else "split"
nm = {"3k": (10, 10, 10), "24k": (20, 20, 20), "96k": (40, 40, 20)}[sys.argv[2] if len(sys.argv) > 2 else "96k"]
case = tl.make_case(nm=nm, num_perturbed_molecules=16, elec="ewald", seed=2026, n_lambda=11, max_cjpacked_per_sci=16)
nb = tl.setup_gpu(case, fused=(mode == "fused"))
sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
for _ in range(5):
    nb.clear_outputs(False); nb.launch_kernel(sw)
torch.cuda.synchronize()
lib = pkg.hip_lib()
n = 5120 if len(sys.argv) < 4 else int(sys.argv[3])
buf = (ctypes.c_ulonglong * (4 * n))()
lib.nbnxm_gpu_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
lib.nbnxm_gpu_debug_timeline(ctypes.c_void_p(nb._h), buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4)
t0 = a[:, 0].min()
start = (a[:, 0] - t0).astype(np.float64) / 100.0
end = (a[:, 2] - t0).astype(np.float64) / 100.0
main = (a[:, 1] - t0).astype(np.float64) / 100.0
hw = a[:, 3] & np.uint64(0xFFFFFFFF)
xcc = (a[:, 3] >> np.uint64(32)) & np.uint64(0xF)
simd = (hw >> np.uint64(4)) & np.uint64(3); cu = (hw >> np.uint64(8)) & np.uint64(0xF); sh = (hw >> np.uint64(12)) & np.uint64(1); se = (hw >> np.uint64(13)) & np.uint64(7)
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
print("waves %d  start us: min %.1f p50 %.1f p99 %.1f max %.1f" % (n, start.min(), np.median(start), np.percentile(start, 99), start.max()))
print("end us:   min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (end.min(), np.percentile(end, 10), np.median(end), np.percentile(end, 90), np.percentile(end, 99), end.max()))
print("first group data at us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(main, [10, 50, 90, 100])))
print("duration us: p10 %.1f p50 %.1f p90 %.1f max %.1f; first group data at p50 %.1f" % (tuple(np.percentile(end - start, [10, 50, 90, 100])) + (np.median(main),)))
uk, cnt = np.unique(key, return_counts=True)
print("distinct SIMDs %d, waves per SIMD: min %d max %d, histogram %s" % (len(uk), cnt.min(), cnt.max(), np.bincount(cnt).tolist()))
simd_end = np.array([end[key == k].max() for k in uk])
print("per-SIMD last end us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(simd_end, [10, 50, 90, 100])))
print("xcc counts", np.bincount(xcc.astype(np.int64)).tolist())
# the trailing workgroups' waves (records behind the ranges' ones; word 1 = 1 rolling prune, 2 perturbed pairs, 3 buffer clear)
nt = 16384
buf2 = (ctypes.c_ulonglong * (4 * nt))()
lib.nbnxm_gpu_debug_timeline(ctypes.c_void_p(nb._h), buf2, nt)
b = np.frombuffer(buf2, dtype=np.uint64).reshape(nt, 4)[n:]
b = b[(b[:, 0] != 0) & (b[:, 1] <= 3)]
for kind, name in ((1, "rolling prune"), (2, "perturbed pairs"), (3, "buffer clear")):
    t = b[b[:, 1] == kind]
    if len(t) == 0:
        continue
    ts = (t[:, 0] - t0).astype(np.float64) / 100.0
    te = (t[:, 2] - t0).astype(np.float64) / 100.0
    print("tail waves, %-15s %5d: start us min %.1f p50 %.1f max %.1f | end us p50 %.1f p90 %.1f max %.1f | duration p50 %.1f max %.1f"
          % (name, len(t), ts.min(), np.median(ts), ts.max(), np.median(te), np.percentile(te, 90), te.max(), np.median(te - ts), (te - ts).max()))
    print("   waves ending after the last range wave (%.1f us): %d" % (end.max(), int((te > end.max()).sum())))
np.save(os.path.join(ROOT, "gpurun_out", "timeline_%s.npy" % mode), a)
