import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package
pkg = load_package()
wl = importlib.import_module("gromacs_fep_gpu_amd.workload")
import torch
for name, nm in (("3k", (10, 10, 10)), ("96k", (40, 40, 20))):
    c = wl.make_case(nm=nm, num_perturbed_molecules=16, elec="ewald", seed=2026, n_lambda=11, max_cjpacked_per_sci=16)
    nb = wl.setup_gpu(c, fused=True, use_dynamic_pruning=True)
    nb.set_timing(False)
    sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
    for _ in range(20):
        nb.clear_outputs(False); nb.launch_kernel(sw)
    torch.cuda.synchronize()
    t_c = t_l = 0.0
    n = 400
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); nb.clear_outputs(False); t1 = time.perf_counter(); nb.launch_kernel(sw); t2 = time.perf_counter()
        t_c += t1 - t0; t_l += t2 - t1
    torch.cuda.synchronize()
    print("%s: clear_outputs %.2f us, launch_kernel %.2f us per call from Python (empty queue)" % (name, 1e6 * t_c / n, 1e6 * t_l / n), flush=True)
    nb.free()
