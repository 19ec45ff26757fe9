#!/usr/bin/env python3
"""Fits the cost model of the work partition (csrc/nbnxm_work_partition.h: c_weightPair/Slot/Group/Entry) to measured
per-SIMD busy times.  Needs the diagnostics build (make HIPFLAGS+=-DNBNXM_WAVE_TIMELINE).

For every wave: features of its range (cluster pairs, non-empty j-slots, non-empty groups, i-entry starts); the waves
are grouped by the SIMD they ran on (HW_ID); least squares of the SIMD's last finish time against the summed features."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fep_testlib as tl
pkg = tl.pkg

BOX = os.environ.get("CAL_BOX", "96k")       # CAL_BOX=24k CAL_ELEC=rf: BASELINE configs[1]
NM = {"12k": (20, 20, 10), "24k": (20, 20, 20), "48k": (40, 20, 20), "96k": (40, 40, 20)}[BOX]
case = tl.make_case(nm=NM, num_perturbed_molecules=3 if BOX == "24k" else 16, elec=os.environ.get("CAL_ELEC", "ewald"), seed=2026, n_lambda=11,
                    max_cjpacked_per_sci=16)
FUSED = "--split" not in sys.argv   # split mode: the atom-pair FEP kernels run beside the cluster kernel and disturb the SIMD times
nb = tl.setup_gpu(case, fused=FUSED, use_dynamic_pruning=True)
sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
for _ in range(8):
    nb.clear_outputs(False); nb.launch_kernel(sw)
torch.cuda.synchronize()
lib = pkg.hip_lib()
lib.nbnxm_gpu_debug_get_work_ranges.restype = C.c_void_p
nr = C.c_int(0)
ptr = lib.nbnxm_gpu_debug_get_work_ranges(C.c_void_p(nb._h), 0, 1, C.byref(nr))
n = nr.value
ranges = np.zeros(n + 1, np.int32)
lib.nbnxm_gpu_debug_download(C.c_void_p(nb._h), C.c_void_p(ptr), ranges.ctypes.data_as(C.c_void_p), C.c_size_t(ranges.nbytes))
pl = case.plist_fused if FUSED else case.plist
cj = pkg.download_cjpacked(nb, len(pl.cjPacked))
imask = cj["imei"][:, 0]["imask"].astype(np.uint32)
popc = np.array([bin(int(m)).count("1") for m in imask])
slots = sum(((imask >> (8 * k)) & 0xFF) != 0 for k in range(4)).astype(np.int64)
groups = (imask != 0).astype(np.int64)
excl_ind = cj["imei"][:, 0]["excl_ind"]
xpairs = np.where(excl_ind != 0, popc, 0).astype(np.int64)      # cluster pairs of the groups that carry an exclusion mask
starts = np.zeros(len(imask), np.int64)
# the kernel walks the caller's entries joined again where they are pieces of one (super-cluster, shift) j-list (nbnxm_gpu_init_pairlist)
sci_sorted = np.sort(pl.sci[pl.sci["cjPackedEnd"] > pl.sci["cjPackedBegin"]], order="cjPackedBegin")
joined = np.ones(len(sci_sorted), bool)
joined[1:] = ~((sci_sorted["sci"][1:] == sci_sorted["sci"][:-1]) & (sci_sorted["shift"][1:] == sci_sorted["shift"][:-1])
               & (sci_sorted["cjPackedBegin"][1:] == sci_sorted["cjPackedEnd"][:-1]))
starts[sci_sorted["cjPackedBegin"][joined]] = 1
empty = (imask == 0).astype(np.int64)
cs = lambda a: np.concatenate([[0], np.cumsum(a)])
feat = np.stack([cs(popc)[ranges[1:]] - cs(popc)[ranges[:-1]], cs(slots)[ranges[1:]] - cs(slots)[ranges[:-1]],
                 cs(groups)[ranges[1:]] - cs(groups)[ranges[:-1]], cs(starts)[ranges[1:]] - cs(starts)[ranges[:-1]] + 1,
                 cs(xpairs)[ranges[1:]] - cs(xpairs)[ranges[:-1]]], axis=1).astype(np.float64)
if "--executed" in sys.argv:
    # sixth feature: the cluster pairs of the range that have an atom pair within the cut-off, i.e. whose pair block executes its inner
    # part (26 of the block's 33 vector instructions) -- the partition weighs every listed cluster pair alike
    xq = np.asarray(case.grid.xq, np.float64)[:, :3].reshape(-1, 8, 3)
    shift_vec = np.asarray(case.grid.shift_vec, np.float64).reshape(-1, 3)
    sci_of = np.zeros(len(imask), np.int64); shift_of = np.zeros(len(imask), np.int64)
    for e in pl.sci:
        sci_of[e["cjPackedBegin"]:e["cjPackedEnd"]] = e["sci"]; shift_of[e["cjPackedBegin"]:e["cjPackedEnd"]] = e["shift"]
    cjs = cj["cj"].astype(np.int64)                       # (groups, 4)
    executed = np.zeros(len(imask), np.int64)
    rc2 = float(case.rc) ** 2 if hasattr(case, "rc") else 1.0
    for g0 in range(0, len(imask), 4096):
        g1 = min(g0 + 4096, len(imask))
        xi = xq[sci_of[g0:g1, None] * 8 + np.arange(8)[None, :]] + shift_vec[shift_of[g0:g1]][:, None, None, :]      # (G, 8 ci, 8 atoms, 3)
        xj = xq[np.clip(cjs[g0:g1], 0, len(xq) - 1)]                                                                   # (G, 4 cj, 8 atoms, 3)
        d2 = ((xi[:, None, :, :, None, :] - xj[:, :, None, None, :, :]) ** 2).sum(-1)                                  # (G, 4, 8, 8, 8)
        within = (d2 < rc2).any(axis=(-1, -2))                                                                         # (G, 4, 8)
        bits = ((imask[g0:g1, None, None] >> (np.arange(4)[None, :, None] * 8 + np.arange(8)[None, None, :])) & 1).astype(bool)
        executed[g0:g1] = (within & bits).sum(axis=(1, 2))
    print("cluster pairs with an atom pair within the cut-off: %.3f of the listed ones" % (executed.sum() / popc.sum()))
    feat = np.concatenate([feat, (cs(executed)[ranges[1:]] - cs(executed)[ranges[:-1]])[:, None].astype(np.float64)], axis=1)
buf = (C.c_ulonglong * (4 * n))()
lib.nbnxm_gpu_debug_timeline(C.c_void_p(nb._h), buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4)
t0 = a[:, 0].min()
end = (a[:, 2] - t0).astype(np.float64) / 100.0
hw = a[:, 3] & np.uint64(0xFFFFFFFF); xcc = (a[:, 3] >> np.uint64(32)) & np.uint64(0xF)
simd = (hw >> np.uint64(4)) & np.uint64(3); cu = (hw >> np.uint64(8)) & np.uint64(0xF); sh = (hw >> np.uint64(12)) & np.uint64(1); se = (hw >> np.uint64(13)) & np.uint64(7)
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
uk = np.unique(key)
X = np.array([feat[key == k].sum(axis=0) for k in uk])
y = np.array([end[key == k].max() for k in uk])
A = np.concatenate([X, np.ones((len(uk), 1))], axis=1)
coef, res, *_ = np.linalg.lstsq(A, y, rcond=None)
pred = A @ coef
print("SIMDs %d; per-SIMD finish: mean %.1f std %.2f us; features per SIMD mean %s" % (len(uk), y.mean(), y.std(), X.mean(axis=0).round(1)))
if X.shape[1] == 6:
    A5 = np.concatenate([X[:, :5], np.ones((len(uk), 1))], axis=1)
    c5 = np.linalg.lstsq(A5, y, rcond=None)[0]
    print("WITH the executed-block feature: residual std %.2f us (without it %.2f us); us per listed cluster pair %.5f, per executed one %.5f; "
          "executed per SIMD std/mean %.4f, listed %.4f" % ((y - pred).std(), (y - A5 @ c5).std(), coef[0], coef[5], X[:, 5].std() / X[:, 5].mean(), X[:, 0].std() / X[:, 0].mean()))
    # per wave: duration against the wave's own features (waves share their SIMD, so this is only indicative)
    Aw = np.concatenate([feat, np.ones((n, 1))], axis=1)
    durw = end - (a[:, 0] - t0).astype(np.float64) / 100.0
    cw = np.linalg.lstsq(Aw, durw, rcond=None)[0]
    cw5 = np.linalg.lstsq(Aw[:, [0, 1, 2, 3, 4, 6]], durw, rcond=None)[0]
    print("per wave: duration std %.2f us, residual with / without the executed feature %.2f / %.2f us" % (durw.std(), (durw - Aw @ cw).std(), (durw - Aw[:, [0, 1, 2, 3, 4, 6]] @ cw5).std()))
    coef = c5; X = X[:, :5]; A = A5; pred = A5 @ c5; feat = feat[:, :5]
print("fit: us per cluster pair %.5f, per slot %.5f, per group %.5f, per piece %.5f, per cluster pair with exclusion mask %.5f, const %.2f; "
      "residual std %.2f us" % (*coef, (y - pred).std()))
w = coef[:5] / coef[0] * 8
print("weights relative to 8 per cluster pair: slot %.1f group %.1f piece %.1f excl-pair %.1f (current 4 / 16 / 128 / 0)" % (w[1], w[2], w[3], w[4]))
A4 = np.concatenate([X[:, :4], np.ones((len(uk), 1))], axis=1)
c4 = np.linalg.lstsq(A4, y, rcond=None)[0]
print("without the exclusion feature: residual std %.2f us" % (y - A4 @ c4).std())
print("per-SIMD features: std/mean %s" % (X.std(axis=0) / X.mean(axis=0)).round(3))
start_t = (a[:, 0] - t0).astype(np.float64) / 100.0
dur = end - start_t
nempty = cs(empty)[ranges[1:]] - cs(empty)[ranges[:-1]]
np.savez(os.path.join(ROOT, "gpurun_out", "calibrate_%s_%s.npz" % (BOX, os.environ.get("CAL_TAG", "x"))), feat=feat, ranges=ranges, start=start_t, end=end, key=key,
         nempty=nempty, sci=np.asarray(sci_sorted), joined=joined, imask=imask)
print("slowest waves (work item: end us, duration us | cluster pairs, slots, non-empty groups, pieces, empty groups | waves' mean):")
for w_ in np.argsort(end)[-12:][::-1]:
    print("  %5d: %.1f %.1f | %d %d %d %d %d" % (w_, end[w_], dur[w_], feat[w_, 0], feat[w_, 1], feat[w_, 2], feat[w_, 3], nempty[w_]))
print("  mean : %.1f %.1f | %.0f %.0f %.0f %.1f %.1f" % (end.mean(), dur.mean(), feat[:, 0].mean(), feat[:, 1].mean(), feat[:, 2].mean(), feat[:, 3].mean(), nempty.mean()))
order = np.argsort(end)[-200:]
print("the 200 last waves: mean cluster pairs %.0f slots %.0f groups %.1f pieces %.2f empty %.2f; SIMD-mates' mean end %.1f" % (
    feat[order, 0].mean(), feat[order, 1].mean(), feat[order, 2].mean(), feat[order, 3].mean(), nempty[order].mean(),
    np.mean([end[(key == key[w_])].mean() for w_ in order])))
cur = feat[:, :4] @ np.array([8, 4, 16, 128.0])
print("current weight per wave: mean %.0f std %.1f (%.2f %%)" % (cur.mean(), cur.std(), 100 * cur.std() / cur.mean()))
# a second launch of the same work: is a slow SIMD slow again?
nb.clear_outputs(False); nb.launch_kernel(sw)
torch.cuda.synchronize()
lib.nbnxm_gpu_debug_timeline(C.c_void_p(nb._h), buf, n)
b = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4)
end2 = (b[:, 2] - b[:, 0].min()).astype(np.float64) / 100.0
same_place = np.array_equal(a[:, 3], b[:, 3])
y2 = np.array([end2[key == k].max() for k in uk])
print("second launch: waves on the same SIMDs: %s; correlation of the per-SIMD finish times %.3f; of the residuals %.3f"
      % (same_place, np.corrcoef(y, y2)[0, 1], np.corrcoef(y - pred, y2 - A @ np.linalg.lstsq(A, y2, rcond=None)[0])[0, 1]))
cuk = uk // 4
print("residual: std of CU means %.2f us, of XCC means %.2f us" % (np.std([np.mean((y - pred)[cuk == c]) for c in np.unique(cuk)]),
      np.std([np.mean((y - pred)[uk // 1024 == x]) for x in range(8)])))
