"""One-GPU rehearsal of bench.py's domain-decomposition leg: ONE rank that is its own neighbour along x (self-links), the real RCCL
transport (or, third argument "peer", the in-process peer-copy transport), the real two-stream step, timed like the leg.  Numbers are
for orientation only (a self-exchange is a device copy).  usage: dd_single_gpu_probe.py [24k|96k|768k] [x|xyz] [rccl|peer|push] [two|merged]"""
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
nm = {"24k": (20, 20, 20), "96k": (40, 40, 20), "768k": (80, 80, 40)}[sys.argv[1] if len(sys.argv) > 1 else "96k"]
links = (True, True, True) if (len(sys.argv) > 2 and sys.argv[2] == "xyz") else (True, False, False)
case = wl.make_case(nm=nm, num_perturbed_molecules=16, elec="ewald", seed=2026, n_lambda=11, build_lists=False)
t0 = time.time()
dd = domdec.DomainDecomposition(case.sys["x"], case.sys["box"], case.sys["molId"], (1, 1, 1), case.rlist, self_links=links)
plan = dd.plan(0)
t_plan = time.time() - t0
t0 = time.time()
system = domdec.RankSystem(pkg, plan, case.sys["box"], case.sys["qA"], case.sys["qB"], case.sys["typeA"], case.sys["typeB"], case.ntype,
                           case.sys["molId"], case.rlist, perturbed=case.perturbed)
t_lists = time.time() - t0
merged = len(sys.argv) > 4 and sys.argv[4] == "merged"
nb = domdec.make_rank_gpu(pkg, wl, case, system, merged=merged)
nb.set_timing(False)
transport = sys.argv[3] if len(sys.argv) > 3 else "rccl"
uid = (domdec.new_halo_id(pkg, domdec.TRANSPORT_PEER_COPY) if transport == "peer"
       else domdec.new_halo_id(pkg, domdec.TRANSPORT_PEER_PUSH) if transport == "push" else None)   # push: the one-sided transport (merged only)
halo = domdec.RcclHalo(pkg, None, 0, 1, nb.stream(pkg.LOCAL if merged else pkg.NONLOCAL), unique_id=uid)
st = domdec.DomainStep(pkg, nb, system, halo)
sw = pkg.step_workload()
for _ in range(20):
    st.step(sw)
torch.cuda.synchronize()
t1 = time.perf_counter()
n = 200
for _ in range(n):
    st.step(sw)
ms_enqueue = 1e3 * (time.perf_counter() - t1) / n      # host time to queue a step (if it equals ms_per_step, the step is launch-bound)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t1) / n
# the host's own cost of queueing a step: with an empty queue every time (no back-pressure from a device that is behind)
t_q = 0.0
for _ in range(100):
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    st.step(sw)
    t_q += time.perf_counter() - t2
torch.cuda.synchronize()
ms_enqueue_idle = 1e3 * t_q / 100
print(json.dumps({"ms_host_enqueue_per_step_empty_queue": ms_enqueue_idle, "atoms": case.natoms, "transport": transport, "merged_localities": merged, "self_links": links, "home": plan.num_home, "halo": plan.num_halo, "ms_per_step": ms, "ms_host_enqueue_per_step": ms_enqueue,
                  "host_plan_s": t_plan, "host_rank_lists_s": t_lists, "halo_bytes": halo.bytes_per_step()}))
halo.free()     # (HALO_GPU_HOST_TIMING=1: prints the host time spent queueing each part of the step)
nb.free()
