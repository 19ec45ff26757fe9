"""Force-only step on a dual pair list as a production run has it (nstlist ~ 100): outer list at rlistOuter, first + rolling prune to
rlistInner; against the static list of the bench (rlist 1.1).  usage: python tools/dual_list_probe.py"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
wl = importlib.import_module("gromacs_fep_gpu_amd.workload")
import torch  # noqa: E402

steps = 300
for outer, inner in ((1.1, 1.1), (1.2, 1.05), (1.3, 1.05), (1.3, 1.02)):
    c = wl.make_case(nm=(40, 40, 20), num_perturbed_molecules=16, elec="ewald", seed=2026, n_lambda=11, max_cjpacked_per_sci=16, rlist=outer)
    c.rlist_inner = inner
    nb = wl.setup_gpu(c, fused=True, use_dynamic_pruning=True)
    nb.set_timing(False)
    sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
    for _ in range(40):
        nb.clear_outputs(False)
        nb.launch_kernel(sw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        nb.clear_outputs(False)
        nb.launch_kernel(sw)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    cj = pkg.download_cjpacked(nb, len(c.plist_fused.cjPacked))
    masks = np.ascontiguousarray(cj["imei"][:, 0]["imask"])
    pairs = int(np.unpackbits(masks.view(np.uint8)).sum())
    print("rlistOuter %.2f rlistInner %.2f: groups %d (empty after prune %d), cluster pairs %d, force step %.4f ms" % (
        outer, inner, len(masks), int((masks == 0).sum()), pairs, ms), flush=True)
    nb.free()
