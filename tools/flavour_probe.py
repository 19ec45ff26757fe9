"""Force-only and energy step time of the 96k box for the kernel flavours a production run can end up in (fused mode):
plain cut-off LJ, twin cut-offs (what PME tuning leaves: rcoulomb > rvdw), force / potential switch, LJ-PME, combination rules,
reaction field, tabulated Ewald.  usage: python tools/flavour_probe.py [steps]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
wl = importlib.import_module("gromacs_fep_gpu_amd.workload")
import torch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
FLAVOURS = [("ewald", "cut", None), ("ewald", "cut", 0.9), ("ewald", "fswitch", None), ("ewald", "pswitch", None), ("ewald", "ewald_geom", None),
            ("ewald", "comb_geom", None), ("ewald", "comb_lb", None), ("rf", "cut", None), ("ewald_tab", "cut", None), ("ewald_tab", "cut", 0.9)]
only = os.environ.get("FLAVOURS")
conditioned = False
for elec, vdw, rvdw in FLAVOURS:
    name = "%s/%s%s" % (elec, vdw, "" if rvdw is None else "/twin")
    if only and name not in only.split(","):
        continue
    # (a tabulated pick runs the analytical kernels by default, csrc/nbnxm_gpu.hip: kernelElecType; these rows time the tabulated kernels)
    elec = "ewald_tab_kept" if elec == "ewald_tab" else elec   # the tabulated kernels themselves
    c = wl.make_case(nm=(40, 40, 20), num_perturbed_molecules=16, elec=elec, vdw=vdw, rvdw=rvdw, seed=2026, n_lambda=11, max_cjpacked_per_sci=16)
    nb = wl.setup_gpu(c, fused=True, use_dynamic_pruning=True)
    nb.set_timing(False)
    res = {}
    if not conditioned:
        # the device's clock needs a few hundred ms of work to settle (bench.py does the same ahead of its warm-up)
        sw0 = pkg.step_workload(energy=False, virial=False, dhdl=False)
        for _ in range(2000):
            nb.clear_outputs(False)
            nb.launch_kernel(sw0)
        torch.cuda.synchronize()
        conditioned = True
    for label, virial in (("force", False), ("energy", True)):
        sw = pkg.step_workload(energy=virial, virial=virial, dhdl=False)
        for _ in range(30):
            nb.clear_outputs(virial)
            nb.launch_kernel(sw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            nb.clear_outputs(virial)
            nb.launch_kernel(sw)
        torch.cuda.synchronize()
        res[label] = 1e3 * (time.perf_counter() - t0) / steps
    print("%-22s force step %.4f ms   energy step %.4f ms" % (name, res["force"], res["energy"]), flush=True)
    nb.free()
