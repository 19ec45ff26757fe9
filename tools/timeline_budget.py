#!/usr/bin/env python3
"""Where a launch of the cluster kernel goes that is not pair work (diagnostics build -DNBNXM_WAVE_TIMELINE, variants/timeline.so).
usage: timeline_budget.py [24k|96k|1m] [ewald|rf]
Per wave the kernel records {start, first group's data arrived, end, HW_ID} in 100 MHz ticks.  Printed: a budget whose rows add up to the
launch as the device's clock sees it (first wave start -> last wave end), next to the event-clock time of the same kernel per step."""
import ctypes, sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fep_testlib as tl
pkg = tl.pkg
size = sys.argv[1] if len(sys.argv) > 1 else "96k"
elec = sys.argv[2] if len(sys.argv) > 2 else "ewald"
nm = {"3k": (10, 10, 10), "12k": (20, 20, 10), "24k": (20, 20, 20), "48k": (40, 20, 20), "96k": (40, 40, 20), "1m": (88, 88, 44)}[size]
case = tl.make_case(nm=nm, num_perturbed_molecules=16 if size != "24k" else 3, elec=elec, seed=2026, n_lambda=11, max_cjpacked_per_sci=16)
nb = tl.setup_gpu(case, fused=True, use_dynamic_pruning=True)   # the bench's list: dynamically pruned
step = os.environ.get("TIMELINE_STEP", "force")     # force | energy | dhdl: which flavour of the step is timed (energy flavours: 4 waves per SIMD)
sw = pkg.step_workload(energy=(step != "force"), virial=(step != "force"), dhdl=(step == "dhdl"))
for _ in range(300):
    nb.clear_outputs(step != "force"); nb.launch_kernel(sw)
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    nb.clear_outputs(step != "force"); nb.launch_kernel(sw)
torch.cuda.synchronize()
step_us = (time.perf_counter() - t0) / K * 1e6
lib = pkg.hip_lib()
nt = 16384
buf = (ctypes.c_ulonglong * (4 * nt))()
lib.nbnxm_gpu_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
lib.nbnxm_gpu_debug_timeline(ctypes.c_void_p(nb._h), buf, nt)
allrec = np.frombuffer(buf, dtype=np.uint64).reshape(nt, 4)
main = (allrec[:, 0] != 0) & (allrec[:, 1] > 3)   # range waves: word 1 is a time stamp; trailing waves: 1, 2 or 3
n = int(np.nonzero(main)[0].max()) + 1
a = allrec[:n]
a = a[a[:, 0] != 0]
t0 = a[:, 0].min()
start = (a[:, 0] - t0).astype(np.float64) / 100.0
first = (a[:, 1] - t0).astype(np.float64) / 100.0
end = (a[:, 2] - t0).astype(np.float64) / 100.0
hw = a[:, 3] & np.uint64(0xFFFFFFFF)
xcc = (a[:, 3] >> np.uint64(32)) & np.uint64(0xF)
simd = (hw >> np.uint64(4)) & np.uint64(3); cu = (hw >> np.uint64(8)) & np.uint64(0xF); sh = (hw >> np.uint64(12)) & np.uint64(1); se = (hw >> np.uint64(13)) & np.uint64(7)
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
uk = np.unique(key)
simd_end = np.array([end[key == k].max() for k in uk])
simd_first = np.array([first[key == k].min() for k in uk])
tail = allrec[n:]
tail = tail[(tail[:, 0] != 0) & (tail[:, 1] <= 3) & (tail[:, 0] >= t0)]   # (records older than this launch: left by a launch with more trailing waves)
tail_end = ((tail[:, 2] - t0).astype(np.float64) / 100.0).max() if len(tail) else 0.0
last = max(end.max(), tail_end)
print("box %s %s: %d range waves (%d with work), %d SIMDs, waves per SIMD %.2f" % (size, elec, n, len(a), len(uk), len(a) / len(uk)))
print("step by the host clock (instrumented build, %d steps): %.2f us" % (K, step_us))
print("budget (us, device clock, first wave start = 0):")
rows = [
    ("wave start skew: median wave start", np.median(start)),
    ("wave start -> first group's data in LDS, median", np.median(first - start)),
    ("  ... 90th percentile / max", None),
    ("pair work: median SIMD's first data -> mean over SIMDs of the last range wave's end", simd_end.mean() - np.median(simd_first)),
    ("drain: mean SIMD end -> slowest SIMD's end", simd_end.max() - simd_end.mean()),
    ("trailing workgroups beyond the last range wave", max(0.0, tail_end - end.max())),
]
acc = 0.0
for name, v in rows:
    if v is None:
        print("    %-86s %6.1f / %.1f" % (name, np.percentile(first - start, 90), (first - start).max()))
        continue
    acc += v
    print("    %-86s %6.2f   (running sum %6.2f)" % (name, v, acc))
print("    %-86s %6.2f" % ("first wave start -> last wave end (device clock)", last))
print("    %-86s %6.2f" % ("host step time minus that: dispatch, end of kernel, launch gap", step_us - last))
print("detail: wave start p50 %.2f p99 %.2f max %.2f | first data p10 %.2f p50 %.2f p90 %.2f max %.2f" % (
    np.median(start), np.percentile(start, 99), start.max(), *np.percentile(first, [10, 50, 90, 100])))
print("        wave end min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f | per-SIMD last end p10 %.1f p50 %.1f p90 %.1f max %.1f mean %.1f" % (
    end.min(), *np.percentile(end, [10, 50, 90, 100]), *np.percentile(simd_end, [10, 50, 90, 100]), simd_end.mean()))
print("        wave duration p10 %.1f p50 %.1f p90 %.1f max %.1f; sum of wave durations / (SIMDs x last end) = %.3f" % (
    *np.percentile(end - start, [10, 50, 90, 100]), (end - start).sum() / (len(uk) * last * (len(a) / len(uk)))))
# the same by XCD: mean over an XCD's SIMDs of their last range wave's end, and whether workgroup b of the launch runs on XCD b mod 8
kx = np.array([int(xcc[key == k][0]) for k in uk])
print("        per-SIMD last end by XCD: " + " ".join("%.2f" % simd_end[kx == x].mean() for x in range(8)) + " | mean %.2f" % simd_end.mean())
wi = np.nonzero((allrec[:n, 0] != 0))[0]
print("        XCD of the first 24 workgroups (4 waves each): " + " ".join(str(int(xcc[wi == 4 * b][0])) if (wi == 4 * b).any() else "-" for b in range(24)))
if len(tail):
    ts = (tail[:, 0] - t0).astype(np.float64) / 100.0
    te = (tail[:, 2] - t0).astype(np.float64) / 100.0
    print("        trailing waves %d: start min %.1f p50 %.1f | end p50 %.1f max %.1f | duration p50 %.1f p90 %.1f max %.1f" % (
        len(tail), ts.min(), np.median(ts), np.median(te), te.max(), np.median(te - ts), np.percentile(te - ts, 90), (te - ts).max()))
    for kind, name in ((1, "rolling prune"), (2, "perturbed pairs"), (3, "buffer clear")):
        m = tail[:, 1] == kind
        if m.any():
            print("           %-16s %5d waves: start p10 %.1f p50 %.1f p90 %.1f | end p50 %.1f p90 %.1f max %.1f | duration p50 %.1f max %.1f" % (
                name, int(m.sum()), *np.percentile(ts[m], [10, 50, 90]), *np.percentile(te[m], [50, 90, 100]), np.median((te - ts)[m]), (te - ts)[m].max()))
np.save(os.path.join(ROOT, "gpurun_out", "timeline_budget_%s_%s.npy" % (size, elec)), allrec)
# the steps of the prologue (second half of the buffer; builds that record them)
buf2 = (ctypes.c_ulonglong * (4 * 49152))()
if os.environ.get("TIMELINE_PROLOGUE", "0") != "1":
    sys.exit(0)
try:
    lib.nbnxm_gpu_debug_timeline(ctypes.c_void_p(nb._h), buf2, 49152)
    allex = np.frombuffer(buf2, dtype=np.uint64).reshape(49152, 4)
    ex = allex[16384:16384 + n]
    ok = ex[:, 0] != 0
    if ok.sum() > 0:
        ex = ex[ok]; st = allrec[:n][ok]
        top = (ex[:, 0] - t0).astype(np.int64) / 100.0
        desc = (ex[:, 1] - t0).astype(np.int64) / 100.0
        issued = (ex[:, 2] - t0).astype(np.int64) / 100.0
        lo32 = lambda v: ((v.astype(np.int64) - (ex[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)) & 0xFFFFFFFF) / 100.0 + top
        arrived = lo32(ex[:, 3] >> np.uint64(32)); barrier = lo32(ex[:, 3] & np.uint64(0xFFFFFFFF))
        fd = (st[:, 1] - t0).astype(np.int64) / 100.0
        f = lambda v: "p10 %.2f p50 %.2f p90 %.2f" % tuple(np.percentile(v, [10, 50, 90]))
        print("prologue steps (us from the first wave's start): first instruction %s | start record %s | batch issued %s | batch arrived %s | behind the barrier %s | first group's data %s"
              % (f(top), f(desc), f(issued), f(arrived), f(barrier), f(fd)))
        print("   per wave, medians of the differences: first instruction -> record %.2f, -> issued %.2f, -> arrived %.2f, -> barrier %.2f, -> first data %.2f"
              % (np.median(desc - top), np.median(issued - desc), np.median(arrived - issued), np.median(barrier - arrived), np.median(fd - barrier)))
    tr = allex[32768:32768 + n].astype(np.float64)
    cnt = tr[:, 0].sum()
    if cnt > 0:
        dur = (end - start)
        print("piece transitions: %.2f per wave; per transition: loop end -> request (i-force reduction, atomics, next entry) %.2f us, request -> arrived "
              "(behind the atomics) %.2f us, collect %.2f us; share of the waves' time: %.1f %% / %.1f %% / %.1f %%"
              % (cnt / n, tr[:, 1].sum() / cnt / 100, tr[:, 2].sum() / cnt / 100, tr[:, 3].sum() / cnt / 100,
                 tr[:, 1].sum() / dur.sum(), tr[:, 2].sum() / dur.sum(), tr[:, 3].sum() / dur.sum()))
except Exception as e:   # older builds: no second half
    print("no prologue stamps:", e)
