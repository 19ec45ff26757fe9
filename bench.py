#!/usr/bin/env python3
"""Benchmark of the FEP non-bonded hot path on MI355X (driver contract: one JSON line on rank 0).

Workload (BASELINE.json configs[2]): 96,000-atom SPC/E-like water box (the "100k box",
40x40x20 molecules, 12.43 x 12.43 x 6.21 nm, 100 atoms/nm^3) with a 48-atom perturbed ligand
(A = coupled, B = decoupled), lambda_coul = lambda_vdw = 0.5, soft-core alpha 0.5 / power 1 /
sigma 0.3, Ewald real-space electrostatics (analytical correction, beta from rtol 1e-5 at 1.0 nm),
LJ cut-off with potential shift, rc = rvdw = 1.0 nm, rlist = 1.1 nm, dynamically pruned list.

One "step" = one pass of the hot path over the resident inputs: clear outputs + gpu_launch_kernel
(fused cluster-pair kernel with in-line perturbed pairs; forces only, as in a normal MD step).
Inputs (xq, parameters, lists) are in HBM before the timed region.  Synthetic data, seeded.

N > 1: `python bench.py --gpus N` starts the N ranks itself (torch.distributed.run, one process per GPU, RCCL);
under a launcher that has already set RANK / WORLD_SIZE it is one of the ranks.  Default (BASELINE configs[3]):
every rank an independent lambda replica of the same box (window = rank mod 11), no data-path collective ->
weak scaling.  --dd (configs[4]): one box decomposed over the ranks on a 3-D domain grid with a halo exchange
(gromacs-fep-gpu_amd/domdec.py) -> strong scaling.

The product legs (everything but `cpu_baseline`) use only gromacs-fep-gpu_amd/ (libnbnxm_hip.so, libnbnxm_host.so);
the CPU baseline leg alone loads the oracle.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3   # vector fp32
DT_FS = 2.0                # MD time step for the kernel-bound ns/day figure
METRIC = "ns/day + pair-interactions/s, 100k-atom FEP box @ λ=0.5, 1/2/4/8 MI355X"
# molecules per box edge of the synthetic water boxes (3 atoms each): configs[1], configs[2], 8 x configs[2], configs[4]'s 1.02 M atoms
BOXES = {"3k": (10, 10, 10), "12k": (20, 20, 10), "24k": (20, 20, 20), "48k": (40, 20, 20), "96k": (40, 40, 20), "192k": (40, 40, 40), "768k": (80, 80, 40), "1m": (88, 88, 44)}
COUNTERS_FILE = os.path.join("profiles", "r04", "counters_fused_force_kernel.json")   # written by tools/summarize_counters.py
# the two other roofs of the dominant kernel ("which roof"):
PEAK_CLOCK_GHZ = 2.4                 # MI355X_MICROARCH.md; under this kernel's load the chip holds ~2.0 GHz
VALU_CYCLES_PER_WAVE64_INSTRUCTION = 2   # SIMD-32 issues a wave64 VALU instruction over 2 cycles (quarter-rate ones take 8)
NUM_SIMDS = 1024                     # 256 CUs x 4
# memory-side float atomics: one 64-byte request per line an instruction touches, 13.5 ns per request and CU whatever the shape
# (tools/ubench/atomic_shapes.hip, profiles/r03/ubench_atomic_shapes.txt; the guide's 1.3 TB/s of 256-byte instructions is the same rate)
ATOMIC_REQUESTS_PER_S = 256 / 13.5e-9


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--mode", choices=["fused", "split"], default="fused")
    ap.add_argument("--atoms", choices=list(BOXES), default="96k")
    ap.add_argument("--elec", choices=["ewald", "rf"], default="ewald", help="rf: BASELINE configs[1] (with --atoms 24k)")
    ap.add_argument("--max-cjpacked-per-sci", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--condition-steps", type=int, default=2000,
                    help="untimed steps ahead of the --warmup steps that bring the device out of its idle power state (the list statistics "
                         "are computed on the host in between: a 20-step measurement right after would run entirely on the clock ramp)")
    ap.add_argument("--no-prune", action="store_true")
    ap.add_argument("--timed-step", choices=["force", "energy", "dhdl"], default="force",
                    help="profiling runs: the step of the timed loop — force-only (the contract's measurement), energy + virial, or dH/dlambda with 11 "
                         "foreign lambdas; anything but force makes the line a diagnostics line (it says so)")
    ap.add_argument("--primary-only", action="store_true",
                    help="skip the secondary figures (energy / dH/dl / virial steps, MD loops): the profile of such a run holds "
                         "the force-only kernels of the timed loop and nothing else")
    ap.add_argument("--perturbed-molecules", type=int, default=-1, help="override the ligand size (diagnostics)")
    ap.add_argument("--extra-types", type=int, default=0,
                    help="diagnostics: give the water oxygens this many extra atom types (the LJ table in LDS grows as ntype^2)")
    ap.add_argument("--dd", action="store_true",
                    help="N > 1: one box decomposed over the ranks with a halo exchange (config 5, strong scaling) "
                         "instead of the default independent lambda replicas (config 4, weak scaling)")
    ap.add_argument("--dd-grid", default="", help="domain grid as AxBxC (default: 2x2x2 for 8 ranks, else the most cubic factorisation)")
    ap.add_argument("--dd-atoms", choices=["24k", "96k", "768k", "1m"], default="1m",
                    help="box of the domain-decomposition leg that follows the replica measurement when N > 1 (configs[4]: 1M atoms)")
    ap.add_argument("--dd-steps", type=int, default=200)
    ap.add_argument("--no-dd-leg", action="store_true", help="N > 1: replicas only, skip the domain-decomposition leg")
    ap.add_argument("--dd-timeout", type=float, default=150.0, help="seconds after which the domain-decomposition leg is given up")
    return ap.parse_args(argv)


# ---- N > 1 from one command: the parent starts the ranks and never touches a GPU itself ----------------------------

def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes under torch.distributed.run (which sets
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), rank 0's JSON line forwarded, exit code of the job returned.  Runs before
    torch is imported: this process never initialises a GPU (a process that has must not exec or fork ranks)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        s = out.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        rec = json.loads(line)
        if rec.get("n_gpus") != args.gpus:
            sys.stderr.write("bench.py: the ranks report n_gpus %r, asked for %d\n" % (rec.get("n_gpus"), args.gpus))
            rc = rc or 1
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks printed no result line\n")
        rc = 1
    return rc


# ---- list statistics and byte counts ---------------------------------------------------------------------------------

def list_statistics(sci, cjPacked):
    """Counts used for the algorithmic-byte and pair figures."""
    import numpy as np
    imask = np.ascontiguousarray(cjPacked["imei"][:, 0]["imask"])
    cluster_pairs = int(np.unpackbits(imask.view(np.uint8)).sum())
    cj_slots = 0
    for sh in range(4):
        cj_slots += int(np.count_nonzero((imask >> (8 * sh)) & 0xFF))
    used_excl = np.unique(cjPacked["imei"]["excl_ind"])
    return dict(nsci=len(sci), ncjPacked=len(cjPacked), cj_slots=cj_slots, cluster_pairs=cluster_pairs, nexcl=int(len(used_excl)))


def host_cores():
    """CPU threads this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a
    16-CPU share of a 256-thread host to one GPU) and by 64."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // period))
        except Exception:
            pass
    if n > 64:
        n = 16   # no quota visible on a large host: the documented share of a one-GPU box
    return n


def algorithmic_bytes(stats, fused, fep_nri=0, fep_nrj=0):
    """360 B per (sci, cj) entry + 2,832 B per sci entry + 128 B per exclusion-mask entry;
    the atom-pair FEP kernel: 64 B per pair + 68 B per i-entry."""
    # (2,832 B per sci for the fused kernel too: its main pass loads no A/B parameters per i atom — the perturbed cluster pairs are
    # masked out of it —, so SURVEY's 3,600 B figure does not apply)
    b = 360 * stats["cj_slots"] + 2832 * stats["nsci"] + 128 * stats["nexcl"]
    b_fep = 64 * fep_nrj + 68 * fep_nri
    return b, b_fep


def committed_counters(fused, args):
    """PMC-derived figures of the dominant kernel from the committed rocprofv3 summary (separate --pmc passes of this same
    command, tools/gpu_pmc.sh + tools/gpu_traffic.sh -> tools/summarize_counters.py).  NOT measured in this run: the record
    says so (`source`, with the commit the profile was taken at)."""
    path = os.path.join(ROOT, COUNTERS_FILE)
    if not (fused and args.atoms == "96k" and args.elec == "ewald" and args.perturbed_molecules < 0 and args.extra_types == 0 and os.path.exists(path)):
        return None
    try:
        rec = json.load(open(path))
    except Exception:
        return None
    rec["source"] = "%s (profiled at commit %s; not measured in this run)" % (COUNTERS_FILE, rec.get("commit", "?"))
    return rec


def other_roofs(counters, kernel_us, hbm_frac):
    """{"hbm_algorithmic", "valu_issue", "atomic_requests"}: fraction of each roof the dominant kernel reaches.  valu_issue is a LOWER
    bound (2 cycles per instruction at the peak clock: the chip holds ~2.0 GHz under this load and a fifth of the instructions —
    compares, selects, DPP adds, reciprocal square roots — issue at half or quarter rate; with the measured issue times the same
    count gives ~0.8 at 96k atoms and ~0.95 at 1M atoms)."""
    out = {"hbm_algorithmic": hbm_frac, "valu_issue": None, "atomic_requests": None, "binding": None}
    if not counters or kernel_us <= 0:
        return out   # (no counter summary for this workload: which roof binds is not known from this run)
    c = counters.get("counters", {})
    t = kernel_us * 1e-6
    if c.get("SQ_INSTS_VALU"):
        out["valu_issue"] = c["SQ_INSTS_VALU"] * VALU_CYCLES_PER_WAVE64_INSTRUCTION / (NUM_SIMDS * PEAK_CLOCK_GHZ * 1e9 * t)
        out["valu_instructions_per_launch"] = c["SQ_INSTS_VALU"]
    if c.get("TCC_EA0_ATOMIC_sum"):
        out["atomic_requests"] = c["TCC_EA0_ATOMIC_sum"] / (ATOMIC_REQUESTS_PER_S * t)
        out["atomic_requests_per_launch"] = c["TCC_EA0_ATOMIC_sum"]
    out["source"] = counters.get("source")
    # the roof the kernel is closest to, from the three fractions of this run (not a constant)
    fracs = {k: out[k] for k in ("hbm_algorithmic", "valu_issue", "atomic_requests") if out[k] is not None}
    out["binding"] = max(fracs, key=fracs.get) if fracs else None
    return out


# ---- the ranks -------------------------------------------------------------------------------------------------------

def dry_run(args, rank, world, dist):
    """BENCH_DRY_RUN=1: the launcher, the rendezvous, the barriers and the JSON contract without any GPU work (CPU test
    of the N > 1 start-up path).  The numbers of such a line mean nothing and it says so."""
    import importlib
    replica = importlib.import_module("gromacs_fep_gpu_amd.replica")
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
    elapsed = replica.max_over_ranks(time.perf_counter() - t0, dist if world > 1 else None, device="cpu")
    out = {"metric": METRIC, "value": 0.0, "unit": "pair-interactions/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * elapsed, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "none", "dry_run": True,
           "rccl_ranks": dist.get_world_size() if world > 1 else 1,
           "config": {"workload": "dry run of the launch path: no GPU work"}}
    simulated = os.environ.get("BENCH_DRY_RUN_DD_LEG")      # "error" / "hang" / "wrong": the end of a run whose decomposition leg went wrong
    if world > 1 and simulated:
        if simulated == "hang":
            args.dd_timeout = 1.0
            def stuck(*a, **k):
                time.sleep(600)
            sys.modules["gromacs_fep_gpu_amd.bench_dd"] = type(sys)("bench_dd")
            sys.modules["gromacs_fep_gpu_amd.bench_dd"].measure = stuck
            guarded_dd_leg(args, rank, world, dist, None, out, "cpu")          # the watchdog prints the line and ends the rank
        rec = {"error": "simulated failure"} if simulated == "error" else {"ms_per_step": 0.1, "parity_of_first_step": {"ok": False}}
        if rank == 0:
            mark_dd_leg(out, rec)
            print(json.dumps(out), flush=True)
        os._exit(dd_leg_failure_exit_code())
    if rank == 0:
        print(json.dumps(out), flush=True)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args, argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    # BENCH_REHEARSAL_GLOO=1: all ranks on cuda:0 with the gloo backend, to walk through the N > 1 code path on a one-GPU box
    # (the numbers of such a run mean nothing)
    rehearsal = os.environ.get("BENCH_REHEARSAL_GLOO") == "1"
    dry = os.environ.get("BENCH_DRY_RUN") == "1"
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if dry:
        if world > 1:
            dist.init_process_group("gloo")
        from __graft_entry__ import load_package
        load_package()
        dry_run(args, rank, world, dist)
        if world > 1:
            dist.destroy_process_group()
        return 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(0 if rehearsal else local_rank)
    reduce_device = "cpu" if rehearsal else "cuda"
    if world > 1:
        dist.init_process_group("gloo" if rehearsal else "nccl")   # nccl = RCCL
    rccl_ranks = dist.get_world_size() if world > 1 else 1

    import importlib
    import numpy as np
    from __graft_entry__ import load_package
    pkg = load_package()
    wl = importlib.import_module("gromacs_fep_gpu_amd.workload")
    replica = importlib.import_module("gromacs_fep_gpu_amd.replica")

    nm = BOXES[args.atoms]
    npert = 3 if args.atoms == "24k" else 16
    if args.perturbed_molecules >= 0:
        npert = args.perturbed_molecules
    if args.dd:
        # --dd: the decomposed box IS the measurement (strong scaling; one rank: a 1 x 1 x 1 grid without a halo, a rehearsal of the leg)
        bench_dd = importlib.import_module("gromacs_fep_gpu_amd.bench_dd")
        rec = bench_dd.measure(args, rank, world, dist, torch, nm, npert, reduce_device, args.steps, args.warmup, check_parity=True)
        if rank == 0:
            parity = rec.get("parity_of_first_step") or {}
            print(json.dumps({
                "dd_leg_ok": bool(parity.get("ok", False)),
                "metric": METRIC, "value": rec["pair_interactions_per_s"], "unit": "pair-interactions/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32", "rccl_ranks": rccl_ranks, "data": "synthetic (seeded SPC/E-like water box + 48-atom decoupled ligand)",
                "config": {"workload": "configs[4]-style: %d-atom box on a %s domain grid, halo exchange over RCCL" % (rec["atoms"], rec["domain_grid"]),
                           "mode": "fused", "atoms": rec["atoms"], "parallelism": "dd " + rec["domain_grid"]},
                "ns_per_day_kernel_bound": 86400.0 / (rec["ms_per_step"] * 1e-3) * DT_FS * 1e-6, "roofline": None,
                "domain_decomposition": rec}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 0
    lam = replica.replica_lambda(rank, world)   # 0.5 on one GPU; window rank mod 11 in the replica set (config 4)
    t0 = time.time()
    case = wl.make_case(nm=nm, num_perturbed_molecules=npert, elec=args.elec, seed=2026, n_lambda=11,
                        lambda_coul=lam, lambda_vdw=lam, max_cjpacked_per_sci=args.max_cjpacked_per_sci, num_extra_types=args.extra_types)
    t_build = time.time() - t0
    fused = args.mode == "fused"
    nb = wl.setup_gpu(case, fused=fused, use_dynamic_pruning=not args.no_prune)
    nb.set_timing(False)   # no per-launch HIP-event regions anywhere in this file: they cost ~15 us per step in stream bubbles
    pl = case.plist_fused if fused else case.plist
    sw_f = pkg.step_workload(energy=False, virial=False, dhdl=False)
    kernel_stream = torch.cuda.ExternalStream(nb.stream())   # the stream the library launches on (not torch's current one)

    sw_timed = {"force": sw_f, "energy": pkg.step_workload(energy=True, virial=True, dhdl=False),
                "dhdl": pkg.step_workload(energy=True, virial=True, dhdl=True)}[args.timed_step]
    timed_virial = args.timed_step != "force"

    def one_step():
        nb.clear_outputs(timed_virial)
        nb.launch_kernel(sw_timed)

    # first step prunes the fresh list (not timed), then warm up
    one_step()
    torch.cuda.synchronize()
    cj_dev = pkg.download_cjpacked(nb, len(pl.cjPacked))
    stats = list_statistics(pl.sci, cj_dev)
    # device conditioning (untimed, NOT the contract's warm-up): the host has just spent ~0.1 s on the list statistics with the GPU idle,
    # and the device's clock takes a few hundred ms of work to settle — a 5 + 20-step measurement (1.5 ms) right away reads 0.0609 - 0.0616 ms
    # per step, after 400 ... 10,000 conditioning steps 0.0553 - 0.0566 ms, and 5,000 timed steps average 0.0536 ms (same kernel, same inputs)
    # ... so the driver's protocol as it stands — exactly W warm-up + K timed steps, nothing ahead of them — is measured first and
    # reported as `ms_per_step_cold` (rank-local; the conditioned figure below is the line's `ms_per_step`)
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    t_cold = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    ms_per_step_cold = 1e3 * (time.perf_counter() - t_cold) / args.steps
    for _ in range(max(0, args.condition_steps)):
        one_step()
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()

    # the timed region: K steps between barrier + synchronize on both sides; ONE HIP-event pair on the kernel's own stream
    # around the same K launches gives the device-side time of the loop (kernel durations + launch gaps): it cannot exceed
    # the wall time of the loop
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    ev0.record(kernel_stream)
    for _ in range(args.steps):
        one_step()
    ev1.record(kernel_stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        elapsed = replica.max_over_ranks(elapsed, dist, device=reduce_device)
    nb_k_us = 1e3 * ev0.elapsed_time(ev1) / args.steps   # per launch, device side, inside the timed loop
    fep_k_us = 0.0                                       # fused: the perturbed pairs ride in the same launch

    # secondary figures (not part of `value`): an energy+virial step and a dH/dlambda step with 11 foreign lambdas
    def timed(sw, n=20):
        for _ in range(3):
            nb.clear_outputs(True)
            nb.launch_kernel(sw)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            nb.clear_outputs(True)
            nb.launch_kernel(sw)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t1) / n
    ms_energy_step = ms_dhdl_step = ms_virial_only = ms_energy_only = None
    if not args.primary_only:
        ms_energy_step = timed(pkg.step_workload(energy=True, virial=True, dhdl=False))
        ms_dhdl_step = timed(pkg.step_workload(energy=True, virial=True, dhdl=True))
        ms_virial_only = timed(pkg.step_workload(energy=False, virial=True, dhdl=False))
        ms_energy_only = timed(pkg.step_workload(energy=True, virial=False, dhdl=False))

    # what a search step costs on the GPU side (secondary figure): atom data, list (from page-locked memory, as the reference keeps it),
    # perturbed-atom bits and coordinates uploaded again, then the first launch on the fresh list (first-pass prune, work partition,
    # force kernel) -- the part of every nstlist-th step that this module owns; the host's list building is the caller's
    ms_search_step = None
    if fused and world == 1 and not args.primary_only:
        g = case.grid
        p_sci, p_cj, p_excl = pkg.pinned_copy(pl.sci), pkg.pinned_copy(pl.cjPacked), pkg.pinned_copy(pl.excl)
        p_xq = pkg.pinned_copy(g.xq)               # (nbnxn_atomdata_t keeps its coordinates in page-locked memory for GPU runs too)
        samples = []
        for _ in range(4):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nb.init_atomdata(g.num_atoms, g.type, qA=g.qA, qB=g.qB, typeA=g.typeA, typeB=g.typeB)
            nb.init_pairlist(p_sci, p_cj, p_excl)
            nb.init_fep_cluster_bits(g.fepBits)
            nb.copy_xq_to_gpu(p_xq)
            one_step()
            torch.cuda.synchronize()
            samples.append(1e3 * (time.perf_counter() - t1))
            for _ in range(20):
                one_step()
        ms_search_step = sorted(samples[1:])[1]      # median of the last three (the first re-allocates nothing either, but warms the path)

    # the same force step in the reference's shape (secondary figure): the carved cluster list + the atom-pair list of make_fep_list through
    # gpu_init_feppairlist, no nbnxm_gpu_set_fep_mode -- a second object on the same box, conditioned by the loop above
    ms_reference_shape = None
    if fused and world == 1 and not args.primary_only:
        nb_split = wl.setup_gpu(case, fused=False, use_dynamic_pruning=not args.no_prune)
        nb_split.set_timing(False)
        for _ in range(200):
            nb_split.clear_outputs(False)
            nb_split.launch_kernel(sw_f)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(500):
            nb_split.clear_outputs(False)
            nb_split.launch_kernel(sw_f)
        torch.cuda.synchronize()
        ms_reference_shape = 1e3 * (time.perf_counter() - t1) / 500
        nb_split.free()

    # GPU-resident MD steps between two searches (secondary figure): x -> xq, clear, kernels, force reduction, leap-frog and
    # SETTLE with coordinates, velocities and forces staying in HBM; 0.5 fs steps so that the list stays valid over the run
    ms_md_step = ms_md_step_sequence = ms_md_step_prune = None
    if fused and world == 1 and not args.primary_only:
        mdloop = importlib.import_module("gromacs_fep_gpu_amd.mdloop")
        nat = case.natoms
        im = np.tile([1 / 15.9994, 1 / 1.008, 1 / 1.008], nat // 3)
        loop = mdloop.ShortRangeMdLoop(nb, case.grid, case.grid.x_wrapped, np.zeros((nat, 3)), im, 0.0005, np.diag(case.grid.box),
                                       settles=np.arange(nat, dtype=np.int32).reshape(-1, 3), settle_params=(15.9994, 1.008, 0.1, 0.16330))

        def time_md(lp, first):
            for i in range(10):
                lp.step(first + i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(50):
                lp.step(first + 10 + i)
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t1) / 50

        ms_md_step_sequence = time_md(loop, 0)       # the reference's kernel sequence
        x_now, v_now = loop.d_x.cpu().numpy(), loop.d_v.cpu().numpy()
        loop.free()
        loop = mdloop.ShortRangeMdLoop(nb, case.grid, x_now, v_now, im, 0.0005, np.diag(case.grid.box), fused_update=True,
                                       settles=np.arange(nat, dtype=np.int32).reshape(-1, 3), settle_params=(15.9994, 1.008, 0.1, 0.16330))
        ms_md_step = time_md(loop, 60)               # one fused update kernel
        loop.rolling_prune_parts = 8                 # + dynamic pruning: one eighth of the list per step
        ms_md_step_prune = time_md(loop, 120)
        loop.rolling_prune_parts = 0
        md_finite = bool(torch.isfinite(loop.d_x).all().item())
        loop.free()
        if not md_finite:
            ms_md_step = ms_md_step_sequence = ms_md_step_prune = None
        # put the object back into the state of the timed loop (coordinates of the search)
        nb.copy_xq_to_gpu(case.grid.xq)

    # lambda windows batched into one object (secondary figure; BASELINE configs[3]: 11 windows — more than the GPUs of a node):
    # one list over R x N slots, per-window lambdas in the perturbed-pair kernel (nbnxm_gpu_set_window_lambdas)
    def batched_windows_steps(lam_w):
        """force step and energy + virial step (ms) of len(lam_w) windows of this box in ONE object"""
        b = replica.batch_windows(case.grid, pl, len(lam_w))
        nbw = pkg.NbnxmGpu(wl.gpu_interaction_params(case, not args.no_prune), case.grid.num_types, case.grid.nbat_nbfp(case.sys["nbfp"]),
                           fep=True, n_lambda=0)
        sig6 = case.sc_sigma ** 6
        nbw.copy_fepparams(case.sc_alpha if case.sc_coul else 0.0, case.sc_alpha, case.sc_power, sig6, sig6 if case.sc_coul else 0.0, 0.5, 0.5)
        nbw.init_atomdata(len(b["type"]), b["type"], qA=b["qA"], qB=b["qB"], typeA=b["typeA"], typeB=b["typeB"])
        nbw.init_pairlist(b["sci"], b["cjPacked"], b["excl"])
        nbw.init_fep_cluster_bits(b["fepBits"])
        nbw.set_fep_mode(True)
        nbw.set_window_lambdas(b["clusters_per_window"], lam_w, lam_w)
        nbw.upload_shiftvec(case.grid.shift_vec)
        nbw.copy_xq_to_gpu(b["xq"])
        nbw.set_timing(False)
        out_ms = []
        for sw_, virial, warm, n in ((sw_f, False, 5, 30), (pkg.step_workload(energy=True, virial=True, dhdl=False), True, 3, 20)):
            for _ in range(warm):
                nbw.clear_outputs(virial)
                nbw.launch_kernel(sw_)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                nbw.clear_outputs(virial)
                nbw.launch_kernel(sw_)
            torch.cuda.synchronize()
            out_ms.append(1e3 * (time.perf_counter() - t1) / n)
        nbw.free()
        return out_ms

    batched = None
    if fused and world == 1 and not args.primary_only and args.atoms in ("3k", "24k", "96k"):
        R = 11
        ms_batched, ms_batched_energy = batched_windows_steps(replica.lambda_schedule(R))
        batched = {"windows": R, "ms_per_step_all_windows": ms_batched,
                   "pair_interactions_per_s": R * 64 * stats["cluster_pairs"] / (ms_batched * 1e-3),
                   "speedup_over_one_window_at_a_time": R * (elapsed / args.steps) / (ms_batched * 1e-3),
                   "ms_per_energy_step_all_windows": ms_batched_energy,
                   "energy_step_speedup_over_one_window_at_a_time": (R * ms_energy_step / ms_batched_energy) if ms_energy_step else None}
    # N > 1: the WHOLE set of 11 windows spread over the ranks (configs[3] names 11 windows on 8 GPUs), a rank's windows in one object —
    # rank r holds windows r, r + N, ... (replica.windows_of_rank).  Secondary figure: the contract's `value` above is the weak-scaling
    # measurement with one window per rank.  A failure here is recorded in the line and changes nothing else.
    window_set = None
    if fused and world > 1 and not args.primary_only and args.atoms in ("3k", "24k", "96k"):
        try:
            mine = replica.windows_of_rank(rank, world, 11)
            ms_set = batched_windows_steps(replica.lambda_schedule(11)[mine]) if mine else [0.0, 0.0]
            dist.barrier()
            ms_set_f = replica.max_over_ranks(ms_set[0], dist, device=reduce_device)
            ms_set_e = replica.max_over_ranks(ms_set[1], dist, device=reduce_device)
            window_set = {"windows": 11, "windows_per_rank_max": -(-11 // world), "ms_per_step_whole_set": ms_set_f,
                          "ms_per_energy_step_whole_set": ms_set_e,
                          "pair_interactions_per_s": 11 * 64 * stats["cluster_pairs"] / (ms_set_f * 1e-3),
                          "one_window_per_rank_in_rounds_ms": -(-11 // world) * 1e3 * elapsed / args.steps}
        except Exception as e:   # noqa: BLE001 (a secondary figure must not cost the line)
            window_set = {"error": "%s: %s" % (type(e).__name__, e)}

    # the same step for a caller that keeps the device pointer of the forces (gpu_get_f: GPU update, GPU force reduction glue): the
    # buffer is pinned from then on, nbnxm_gpu_clear_outputs launches its clear kernel again instead of swapping (secondary figure;
    # measured last, because the pin lasts for the life of the object)
    ms_pinned = None
    if fused and not args.primary_only:
        nb.f_device_pointer()
        for _ in range(10):
            one_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(100):
            one_step()
        torch.cuda.synchronize()
        ms_pinned = 1e3 * (time.perf_counter() - t1) / 100

    ms_per_step = 1e3 * elapsed / args.steps
    pair_evals = 64 * stats["cluster_pairs"]          # atom pairs in the (pruned) list, SURVEY §8d
    fep_pairs = len(case.plist.fep["jjnr"])
    pairs_per_step = pair_evals + (0 if fused else fep_pairs)
    value = replica.aggregate_throughput(pairs_per_step, args.steps, elapsed, world)
    useful_pairs = wl.count_pairs_within(case) if rank == 0 else 0   # r < rc, not excluded, each pair once (host count)
    ns_per_day = world * 86400.0 / (elapsed / args.steps) * DT_FS * 1e-6
    bytes_nb, bytes_fep = algorithmic_bytes(stats, fused, len(case.plist.fep["iinr"]), fep_pairs)
    achieved = bytes_nb / (nb_k_us * 1e-6) / 1e9 if nb_k_us > 0 else 0.0
    counters = committed_counters(fused, args)

    out = {
        "metric": METRIC, "value": value, "unit": "pair-interactions/s",
        "timed_step": args.timed_step if args.timed_step == "force" else args.timed_step + " (a profiling run: NOT the contract's force-only measurement)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "device_conditioning_steps_before_warmup": max(0, args.condition_steps),
        # the same W + K steps with NO conditioning ahead of them (the device's clock is still ramping up): what the driver's protocol reads
        # by itself, with its own fraction of the HBM roofline on the algorithmic bytes
        "ms_per_step_cold": ms_per_step_cold, "roofline_frac_cold": bytes_nb / (ms_per_step_cold * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (seeded SPC/E-like water box + 48-atom decoupled ligand)",
        "rccl_ranks": rccl_ranks,
        "config": {"workload": "configs[2]: 96k-atom water + 48 perturbed atoms, Ewald(analytical) + LJ cut, rc 1.0, rlist 1.1, lambda 0.5"
                   if (args.atoms == "96k" and args.elec == "ewald") else "%s-atom box, %s" % (args.atoms, args.elec),
                   "mode": args.mode, "atoms": int(case.natoms), "atom_types": int(case.ntype), "perturbed_atoms": int(case.perturbed.sum()),
                   "nsci": stats["nsci"], "cj_slots": stats["cj_slots"], "cluster_pairs": stats["cluster_pairs"],
                   "fep_pairs": fep_pairs, "max_cjpacked_per_sci": args.max_cjpacked_per_sci,
                   "parallelism": "1 lambda replica per GPU" if world > 1 else "single GPU"},
        "ns_per_day_kernel_bound": ns_per_day,
        # pairs with r < rc (what the physics needs) next to the pairs the list makes the kernel evaluate (`value`)
        "useful_pairs_per_step": useful_pairs,
        "useful_pair_interactions_per_s": world * useful_pairs * args.steps / elapsed,
        # device-side time per launch of the timed loop itself: one HIP-event pair on the kernel's stream around the K launches
        "kernel_us": {"k_calc_nb": nb_k_us, "k_calc_nb_fep": fep_k_us, "how": "HIP events around the timed loop on the kernel's stream / steps"},
        "ms_per_energy_step": ms_energy_step, "ms_per_dhdl_step_11_foreign_lambdas": ms_dhdl_step,
        "ms_per_virial_only_step": ms_virial_only, "ms_per_energy_only_step": ms_energy_only,
        "ms_per_step_with_pinned_force_buffer": ms_pinned,
        "ms_per_step_reference_shape_atom_pair_list": ms_reference_shape,
        "ms_per_search_step_gpu_side": ms_search_step,
        "ns_per_day_kernel_bound_with_a_search_every_100_steps": (86400.0 * 2e-6 / ((100 * elapsed / args.steps + 1e-3 * ms_search_step) / 100)
                                                                  if ms_search_step is not None else None),
        "ms_per_gpu_resident_md_step": ms_md_step, "ms_per_gpu_resident_md_step_unfused_update": ms_md_step_sequence,
        "ms_per_gpu_resident_md_step_with_rolling_prune_8": ms_md_step_prune,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": counters.get("hbm_bytes_per_launch_corrected") if counters else None,
                     "traffic_source": counters["source"] if counters else None,
                     "kernel": "nbnxmKernel<%s,LJcut,F,%s>" % ("EwaldAna" if args.elec == "ewald" else "RF", "fused" if fused else "plain"),
                     "algorithmic_bytes_per_launch": bytes_nb,
                     "fp32_valu_frac_estimate": (pair_evals * 45.0 / (nb_k_us * 1e-6) / 1e12 / FP32_PEAK_TFLOPS)
                     if nb_k_us > 0 else None,
                     # the kernel's three roofs side by side: `frac` above (algorithmic bytes over the HBM peak, the contract's figure),
                     # VALU issue and memory-side atomic requests — instruction and request counts per launch from the committed PMC
                     # summary, divided by THIS run's kernel time
                     "fractions": other_roofs(counters, nb_k_us, achieved / HBM_PEAK_GBS)},
        "counters": ({"lds_bank_conflict_frac": counters.get("lds_bank_conflict_frac"), "valu_issue_frac": counters.get("valu_issue_frac"),
                      "active_lanes_per_valu_instruction": counters.get("active_lanes_per_valu_instruction"),
                      "source": counters["source"]} if counters else None),
        "lambda_windows_batched": batched, "lambda_window_set_over_the_ranks": window_set,
        "host_list_build_s": t_build,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(case, fep_pairs)
    nb.free()
    if world > 1 and not args.no_dd_leg:
        # second leg, after the contract's measurement is complete: ONE box decomposed over the same ranks (configs[4]).  It cannot
        # cost the line above: a watchdog prints the line without it and ends the ranks if the leg hangs or fails.
        dd_rec = guarded_dd_leg(args, rank, world, dist, torch, out, reduce_device)
        out["domain_decomposition"] = dd_rec
        failed = dd_rec is not None and "error" in dd_rec
        if rank == 0:
            mark_dd_leg(out, dd_rec)
            print(json.dumps(out), flush=True)
        # every rank has to agree on the exit code: rank 0 knows whether the first step reproduced the single-domain forces
        leg_ok = [bool(out.get("dd_leg_ok", False)) if rank == 0 else None]
        if not failed:
            dist.broadcast_object_list(leg_ok, src=0)
        if failed or not leg_ok[0]:
            # a failed leg may have left ranks inside a collective: no orderly shutdown.  The contract's line above is complete, but a
            # leg that failed, hung or computed other forces than the single domain must not look like a green run: exit code 3
            # (BENCH_DD_LENIENT=1: 0, the line's "dd_leg_ok": false and stderr are then the only trace)
            os._exit(dd_leg_failure_exit_code())
        dist.destroy_process_group()
        return 0
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def dd_leg_failure_exit_code():
    """Exit code of every rank when the decomposition leg failed, hung or gave wrong forces: 3; BENCH_DD_LENIENT=1 restores the old 0."""
    return 0 if os.environ.get("BENCH_DD_LENIENT") == "1" else 3


def mark_dd_leg(out, dd_rec):
    """The state of the decomposition leg at the TOP level of the line: ran, and its first step reproduced the single-domain forces."""
    failed = dd_rec is None or "error" in dd_rec
    parity = None if failed else dd_rec.get("parity_of_first_step")
    out["dd_leg_ok"] = bool(not failed and (parity is None or parity.get("ok", False)))
    out["dd_leg_error"] = (dd_rec or {}).get("error") if failed else (None if out["dd_leg_ok"] else "first-step forces differ from the single-domain forces")
    if not out["dd_leg_ok"]:
        sys.stderr.write("bench.py: DOMAIN-DECOMPOSITION LEG FAILED: %s\n" % out["dd_leg_error"])


def guarded_dd_leg(args, rank, world, dist, torch, out, reduce_device):
    import importlib
    import threading

    state = {"primary": None}    # the RCCL leg's record once it is complete: what follows it (the one-sided transport) cannot cost it

    def give_up():
        primary = state["primary"]
        if primary is not None:
            # the decomposition leg itself is done; what did not finish is the additional measurement over the one-sided transport
            primary["one_sided_transport"] = {"error": "no result within %.0f s of the leg's start" % args.dd_timeout}
            if rank == 0:
                out["domain_decomposition"] = primary
                mark_dd_leg(out, primary)
                print(json.dumps(out), flush=True)
            os._exit(0 if (rank != 0 or out.get("dd_leg_ok")) else dd_leg_failure_exit_code())
        if rank == 0:
            out["domain_decomposition"] = {"error": "no result within %.0f s" % args.dd_timeout}
            mark_dd_leg(out, out["domain_decomposition"])
            print(json.dumps(out), flush=True)
        os._exit(dd_leg_failure_exit_code())

    timer = threading.Timer(args.dd_timeout, give_up)
    timer.daemon = True
    timer.start()
    nm = BOXES[args.dd_atoms]
    npert = 3 if args.dd_atoms == "24k" else 16
    try:
        bench_dd = importlib.import_module("gromacs_fep_gpu_amd.bench_dd")
        rec = bench_dd.measure(args, rank, world, dist, torch, nm, npert, reduce_device, args.dd_steps, 10, check_parity=True)
        if world == 8 and not args.dd_grid:
            # the grid BASELINE configs[4] names, next to the one with the smallest halo for this box
            rec2 = bench_dd.measure(args, rank, world, dist, torch, nm, npert, reduce_device, args.dd_steps, 10, grid_text="2x2x2")
            if rank == 0 and rec2["domain_grid"] != rec["domain_grid"]:
                rec["grid_2x2x2"] = rec2
        state["primary"] = rec if rank == 0 else {}
        # the same decomposition over the ONE-SIDED transport (stores into the peers' buffers over hipIpc, sequence flags, no transfer
        # kernel; halo_hip.h): an additional measurement — its failure is recorded inside the leg's record and changes nothing else
        if os.environ.get("BENCH_DD_ONE_SIDED", "1") != "0" and os.environ.get("BENCH_DD_MERGED", "1") != "0":
            try:
                rec3 = bench_dd.measure(args, rank, world, dist, torch, nm, npert, reduce_device, args.dd_steps, 10, check_parity=True, transport="push")
            except Exception as e:      # noqa: BLE001
                rec3 = {"error": repr(e)[:400]}
            if rank == 0:
                rec["one_sided_transport"] = rec3
        if rank != 0:
            rec = None
    except BaseException as e:      # noqa: BLE001 — whatever happens here must not cost the line
        rec = {"error": repr(e)[:400]}
    timer.cancel()
    return rec


def cpu_baseline(case, fep_pairs):
    """CPU baseline (kind "port") of the same step on the host cores this process may use, bounded to ~10 s: the cluster-pair
    part through the SIMD port of the oracle's kernel (oracle/nbnxm_simd.c: 8-wide, rational Ewald correction — the shape
    of the reference's CPU kernels), the perturbed pairs through the scalar FEP oracle; the all-scalar oracle where the
    CPU has no AVX2 + FMA.  The ONLY leg of this file that loads the oracle (test infrastructure)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fep_testlib as tl
    import oracle_binding as ob
    cj_pruned_carved = case.plist.cjPacked.copy()
    ob.nbnxm_prune(case.plist.sci, cj_pruned_carved, case.grid.xq, case.grid.shift_vec, case.rlist)
    cstats = list_statistics(case.plist.sci, cj_pruned_carved)
    cores = host_cores()
    g = case.grid
    ref_p, nbfp_grid, fep_p = tl.oracle_ref_params(case), g.nbat_nbfp(case.sys["nbfp"]), tl.oracle_fep_params(case)

    def cpu_step_simd():
        f = ob.nbnxm_simd(case.plist.sci, cj_pruned_carved, case.plist.excl, g.xq, g.type, g.num_types, nbfp_grid, ref_p, g.shift_vec,
                          num_threads=cores)
        if f is None:
            return None
        ob.fep_kernel(case.plist.fep, g.x_wrapped, case.ntype, fep_p, g.shift_vec, case.sys["nbfp"], None, case.sys["qA"], case.sys["qB"],
                      case.sys["typeA"], case.sys["typeB"], ob.DO_FORCE, case.lambda_coul, case.lambda_vdw, "f32")
        return f

    def cpu_step_scalar():
        return tl.run_oracle(case, energy=False, precision="f32", cjPacked=cj_pruned_carved, num_threads=cores)

    simd = cpu_step_simd() is not None      # also the warm-up
    cpu_step = cpu_step_simd if simd else cpu_step_scalar
    cpu_step()
    n_pass, t_cpu = 0, 0.0
    while t_cpu < 10.0 and n_pass < 1000:
        t1 = time.perf_counter()
        cpu_step()
        t_cpu += time.perf_counter() - t1
        n_pass += 1
    cpu_pairs = 64 * cstats["cluster_pairs"] + fep_pairs
    return {"value": cpu_pairs * n_pass / t_cpu, "unit": "pair-interactions/s", "cores": cores, "kind": "port",
            "sample": "%d full passes of the same %d-atom step (pruned cluster list on %d OpenMP threads, %s; FEP list scalar), f32, %.1f s"
                      % (n_pass, case.natoms, cores, "8-wide SIMD port of the oracle kernel" if simd else "scalar C oracle", t_cpu)}


if __name__ == "__main__":
    sys.exit(main())
