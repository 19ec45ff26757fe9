"""Host time to queue one force-only step (clear_outputs + launch_kernel through the C ABI) against the GPU time of the step, by box size.
If the two are equal the step is launch-bound.  usage: python tools/host_enqueue_probe.py"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
wl = importlib.import_module("gromacs_fep_gpu_amd.workload")
import torch  # noqa: E402

for name, nm, npert in (("3k", (10, 10, 10), 16), ("3k no perturbed atoms", (10, 10, 10), 0), ("24k", (20, 20, 20), 16), ("96k", (40, 40, 20), 16)):
    c = wl.make_case(nm=nm, num_perturbed_molecules=npert, elec="ewald", seed=2026, n_lambda=11, max_cjpacked_per_sci=16)
    nb = wl.setup_gpu(c, fused=True, use_dynamic_pruning=True)
    nb.set_timing(False)
    sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
    for _ in range(50):
        nb.clear_outputs(False)
        nb.launch_kernel(sw)
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        nb.clear_outputs(False)
        nb.launch_kernel(sw)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%-24s host enqueue %.2f us per step, step %.2f us" % (name, 1e6 * t_enq / n, 1e6 * t_all / n), flush=True)
    nb.free()
