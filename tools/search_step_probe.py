"""GPU-side cost of a search step (what the non-bonded module does every nstlist steps once the host has a new list), 96k box, fused mode:
atom data + list upload, first (outer + inner) prune, work partition, perturbed-cluster-pair list; then the first force step.
usage: python tools/search_step_probe.py"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
wl = importlib.import_module("gromacs_fep_gpu_amd.workload")
import torch  # noqa: E402

c = wl.make_case(nm=(40, 40, 20), num_perturbed_molecules=16, elec="ewald", seed=2026, n_lambda=11, max_cjpacked_per_sci=16)
SPLIT = "split" in sys.argv      # the reference's shape: carved cluster list + atom-pair list through gpu_init_feppairlist
g, pl = c.grid, (c.plist if SPLIT else c.plist_fused)
nb = wl.setup_gpu(c, fused=not SPLIT, use_dynamic_pruning=True)
nb.set_timing(False)
sw = pkg.step_workload(energy=False, virial=False, dhdl=False)


def sync():
    torch.cuda.synchronize()


def timed(fn, n=1):
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return 1e3 * (time.perf_counter() - t0) / n


# the caller's lists in page-locked memory, as the reference keeps them (argument "pageable": plain numpy arrays, staged by the library)
if "pageable" not in sys.argv:
    pl_sci, pl_cj, pl_excl = pkg.pinned_copy(pl.sci), pkg.pinned_copy(pl.cjPacked), pkg.pinned_copy(pl.excl)
else:
    pl_sci, pl_cj, pl_excl = pl.sci, pl.cjPacked, pl.excl
for rep in range(4):
    t_atom = timed(lambda: nb.init_atomdata(g.num_atoms, g.type, qA=g.qA, qB=g.qB, typeA=g.typeA, typeB=g.typeB))
    t_list = timed(lambda: nb.init_pairlist(pl_sci, pl_cj, pl_excl))
    t_bits = timed((lambda: nb.init_feppairlist(c.plist.fep, g.atomIndices)) if SPLIT else (lambda: nb.init_fep_cluster_bits(g.fepBits)))
    t_xq = timed(lambda: nb.copy_xq_to_gpu(g.xq))

    def first_step():
        nb.clear_outputs(False)
        nb.launch_kernel(sw)
    t_first = timed(first_step)
    t_next = timed(first_step, 100)
    print("search step %d: atom data %.3f  list upload %.3f  fep bits / atom-pair list %.3f  xq %.3f  first step (prune + partition + kernel) %.3f  | later steps %.4f ms"
          % (rep, t_atom, t_list, t_bits, t_xq, t_first, t_next), flush=True)
# host time of each call without waiting for the device (what the caller's thread spends), then the wait for everything
if "host" in sys.argv:
    p_xq = pkg.pinned_copy(g.xq)
    for rep in range(4):
        sync()
        t = [time.perf_counter()]
        nb.init_atomdata(g.num_atoms, g.type, qA=g.qA, qB=g.qB, typeA=g.typeA, typeB=g.typeB); t.append(time.perf_counter())
        nb.init_pairlist(pl_sci, pl_cj, pl_excl); t.append(time.perf_counter())
        nb.init_fep_cluster_bits(g.fepBits); t.append(time.perf_counter())
        nb.copy_xq_to_gpu(p_xq); t.append(time.perf_counter())
        nb.clear_outputs(False); nb.launch_kernel(sw); t.append(time.perf_counter())
        sync(); t.append(time.perf_counter())
        d = [1e3 * (b - a) for a, b in zip(t[:-1], t[1:])]
        print("host ms: atom data %.3f  list %.3f  bits %.3f  xq %.3f  first launch %.3f  | wait for the device %.3f  | total %.3f"
              % (d[0], d[1], d[2], d[3], d[4], d[5], 1e3 * (t[-1] - t[0])), flush=True)
nb.free()
