#!/usr/bin/env python3
"""Which kernel pays for the shift forces of a virial step: N force-only steps, then N virial-only steps; run under
rocprofv3 --kernel-trace and compare the two halves of the trace (tools: see the end of the output)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fep_testlib as tl
pkg = tl.pkg
case = tl.make_case(nm=(40, 40, 20), num_perturbed_molecules=16, elec="ewald", seed=2026, n_lambda=11, max_cjpacked_per_sci=16)
nb = tl.setup_gpu(case, fused=True, use_dynamic_pruning=True)
for virial in (False, True):
    sw = pkg.step_workload(energy=False, virial=virial, dhdl=False)
    for _ in range(110):
        nb.clear_outputs(virial); nb.launch_kernel(sw)
    torch.cuda.synchronize()
nb.free()
