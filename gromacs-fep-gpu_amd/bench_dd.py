"""bench.py --dd: one box decomposed over the ranks (BASELINE configs[4]); kept out of bench.py so that the contract file stays short."""
import importlib
import json
import time

import numpy as np


def run(args, rank, world, dist, torch, nm, npert, metric, dt_fs, rehearsal, rccl_ranks):
    """A step = halo x (pack, RCCL point-to-point, unpack), x -> xq, fused cluster kernel on the rank's share of the list,
    force reduction, halo f.  value = pair interactions of the WHOLE list per second (strong scaling)."""
    pkg = importlib.import_module("gromacs_fep_gpu_amd")
    wl = importlib.import_module("gromacs_fep_gpu_amd.workload")
    domdec = importlib.import_module("gromacs_fep_gpu_amd.domdec")
    replica = importlib.import_module("gromacs_fep_gpu_amd.replica")
    t0 = time.time()
    case = wl.make_case(nm=nm, num_perturbed_molecules=npert, elec="ewald", seed=2026, n_lambda=11)
    dd = domdec.SlabDecomposition(case.grid, case.plist_fused, world)
    plan = dd.plan(rank)
    t_build = time.time() - t0
    nb = wl.setup_gpu(case, fused=True, use_dynamic_pruning=not args.no_prune,
                      list_override=(plan.sci, plan.cjPacked, case.plist_fused.excl))
    halo = domdec.HaloExchange(plan, "cuda")
    st = domdec.DomainStep(nb, case.grid, plan, halo)
    g = case.grid
    real = g.atomIndices >= 0
    x = np.zeros((case.natoms, 3), np.float32)
    x[g.atomIndices[real]] = g.xq.reshape(-1, 4)[real, :3]
    st.d_x.copy_(torch.from_numpy(x))
    comm = domdec.TorchDistComm(dist)
    sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
    for _ in range(1 + args.warmup):
        st.step(comm, sw)
    torch.cuda.synchronize()
    cj_dev = pkg.download_cjpacked(nb, len(plan.cjPacked))
    imask = np.ascontiguousarray(cj_dev["imei"][:, 0]["imask"])
    red = "cpu" if rehearsal else "cuda"
    my_pairs = torch.tensor([float(np.unpackbits(imask.view(np.uint8)).sum())], device=red, dtype=torch.float64)
    dist.all_reduce(my_pairs)
    dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        st.step(comm, sw)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = replica.max_over_ranks(time.perf_counter() - t_start, dist, device=red)
    if rank == 0:
        pairs = 64.0 * float(my_pairs.item())
        print(json.dumps({
            "metric": metric, "value": pairs * args.steps / elapsed, "unit": "pair-interactions/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "rccl_ranks": rccl_ranks,
            "data": "synthetic (seeded SPC/E-like water box + 48-atom decoupled ligand)",
            "config": {"workload": "configs[4]-style: %d-atom box decomposed into %d slabs, halo exchange over RCCL" % (case.natoms, world),
                       "mode": "fused", "atoms": int(case.natoms), "cluster_pairs": int(my_pairs.item()),
                       "halo_bytes_per_step_rank0": halo.bytes_per_step(), "parallelism": "dd%d" % world},
            "ns_per_day_kernel_bound": 86400.0 / (elapsed / args.steps) * dt_fs * 1e-6,
            "roofline": None, "host_list_build_s": t_build}), flush=True)
    nb.free()
    dist.destroy_process_group()
