"""All ranks of a decomposition on ONE GPU, each on its own host thread, through the C++ step (halo_gpu_domain_force_step) over the
in-process peer-copy transport: what the decomposed box costs the device in all (the ranks' kernels share the GPU, so the time per step
divided by the number of ranks is a rank's device-busy time for its real domain size — the figure the multi-GPU projection of DESIGN.md
section 6 needs), next to the same box as a single domain.  usage: dd_virtual_ranks_probe.py [96k|768k|1m] [AxBxC] [two|merged]"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
import torch  # noqa: E402

wl = importlib.import_module("gromacs_fep_gpu_amd.workload")
domdec = importlib.import_module("gromacs_fep_gpu_amd.domdec")
size = sys.argv[1] if len(sys.argv) > 1 else "1m"
nm = {"96k": (40, 40, 20), "768k": (80, 80, 40), "1m": (88, 88, 44)}[size]
ncells = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "2x2x2").split("x"))
merged = not (len(sys.argv) > 3 and sys.argv[3] == "two")
case = wl.make_case(nm=nm, num_perturbed_molecules=16, elec="ewald", seed=2026, n_lambda=11, build_lists=False)
dd = domdec.DomainDecomposition(case.sys["x"], case.sys["box"], case.sys["molId"], ncells, case.rlist)
uid = domdec.new_halo_id(pkg, domdec.TRANSPORT_PEER_COPY)
steps, pairs = [], 0
for r in range(dd.num_ranks):
    plan = dd.plan(r)
    system = domdec.RankSystem(pkg, plan, case.sys["box"], case.sys["qA"], case.sys["qB"], case.sys["typeA"], case.sys["typeB"], case.ntype,
                               case.sys["molId"], case.rlist, perturbed=case.perturbed)
    nb = domdec.make_rank_gpu(pkg, wl, case, system, merged=merged)
    nb.set_timing(False)
    halo = domdec.RcclHalo(pkg, None, r, dd.num_ranks, nb.stream(pkg.LOCAL if merged else pkg.NONLOCAL), unique_id=uid)
    steps.append(domdec.DomainStep(pkg, nb, system, halo))
sw = pkg.step_workload()


def loop(st, n):
    def run():
        for _ in range(n):
            st.step(sw)
    return run


domdec.run_ranks_in_threads([loop(st, 20) for st in steps])
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
domdec.run_ranks_in_threads([loop(st, n) for st in steps])
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(json.dumps({"atoms": case.natoms, "grid": "%dx%dx%d" % ncells, "ranks_on_one_gpu": dd.num_ranks, "merged_localities": merged,
                  "home_atoms_per_rank": int(np.mean([st.num_home for st in steps])), "halo_atoms_per_rank": int(np.mean([st.num_all - st.num_home for st in steps])),
                  "ms_per_step_all_ranks": 1e3 * t_all / n, "ms_per_step_per_rank_device_share": 1e3 * t_all / n / dd.num_ranks,
                  "ms_host_threads_done_queueing": 1e3 * t_enq / n}))
for st in steps:
    st.halo.free()
    st.nb.free()
