"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on identical
inputs (same lists, same float coordinates).

Tolerance: 1e-4 relative — the bar the reference's own GPU-vs-CPU acceptance test uses
(src/programs/mdrun/tests/freeenergy.cpp:115-135) and the one BASELINE.json states.  Forces are
compared against the RMS force, sums that cancel against the magnitude of their terms (fep_testlib.assert_parity).
"""
import ctypes as C

import math

import numpy as np
import pytest

import fep_testlib as tl

pytestmark = pytest.mark.gpu
pkg = tl.pkg

SMALL = dict(nm=(10, 10, 10), num_perturbed_molecules=3)      # 3000 atoms, 9 perturbed


@pytest.mark.parametrize("elec", ["rf", "cut", "ewald", "ewald_tab", "ewald_tab_kept"])
@pytest.mark.parametrize("energy", [False, True])
def test_split_path_matches_oracle(elec, energy):
    c = tl.make_case(elec=elec, seed=21, **SMALL)
    got = tl.run_gpu(c, energy=energy, fused=False)
    want = tl.run_oracle(c, energy=True)
    tl.assert_parity(got, want, rel=1e-4, energy=energy, label=elec)


@pytest.mark.parametrize("elec,vdw", [("rf", "cut"), ("ewald", "cut"), ("ewald", "pswitch")])
@pytest.mark.parametrize("energy", [False, True])
def test_split_path_with_the_atom_pair_kernel_of_its_own(elec, vdw, energy, monkeypatch):
    # by default the atom-pair list is regrouped by cluster pair and evaluated in the cluster kernel's trailing workgroups
    # (fepListClusterItem); NBNXM_HIP_FEP_LIST_MERGED=0 (read when the object is created) keeps the reference's shape: nbnxmFepKernel
    # on the flattened list, on its own stream.  Both against the oracle, and against each other.
    c = tl.make_case(elec=elec, vdw=vdw, seed=21, nm=(10, 10, 10), num_perturbed_molecules=16)
    want = tl.run_oracle(c, energy=True)
    merged = tl.run_gpu(c, energy=energy, fused=False)
    tl.assert_parity(merged, want, rel=1e-4, energy=energy, label="list in the tail " + elec)
    monkeypatch.setenv("NBNXM_HIP_DIAGNOSTICS", "1")
    monkeypatch.setenv("NBNXM_HIP_FEP_LIST_MERGED", "0")
    own = tl.run_gpu(c, energy=energy, fused=False)
    tl.assert_parity(own, want, rel=1e-4, energy=energy, label="own kernel " + elec)
    frms = math.sqrt(float(np.mean(np.sum(np.asarray(want["f"], np.float64) ** 2, axis=1))))
    assert np.max(np.abs(np.asarray(own["f"], np.float64) - np.asarray(merged["f"], np.float64))) <= 1e-4 * frms


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_atom_pair_lists_of_any_shape(seed, monkeypatch):
    # what gpu_init_feppairlist regroups by cluster pair must not depend on how the caller cut the list: a random subset of the pairs,
    # i-entries split at random places and put in random order, exclusion flags flipped at random — the same list goes to the oracle,
    # to the trailing workgroups of the cluster kernel and to the atom-pair kernel of its own
    c = tl.make_case(elec="ewald", seed=30 + seed, nm=(10, 10, 10), num_perturbed_molecules=12)
    rng = np.random.default_rng(seed)
    fep = c.plist.fep
    iinr, shift, jindex, jjnr, excl = [], [], [0], [], []
    order = rng.permutation(len(fep["iinr"]))
    for e in order:
        b, e1 = int(fep["jindex"][e]), int(fep["jindex"][e + 1])
        keep = np.nonzero(rng.random(e1 - b) < 0.7)[0] + b
        cuts = np.sort(rng.choice(np.arange(1, max(2, len(keep))), size=min(2, max(0, len(keep) - 1)), replace=False)) if len(keep) > 2 else []
        for part in np.split(keep, cuts):
            if len(part) == 0:
                continue
            iinr.append(fep["iinr"][e]); shift.append(fep["shift"][e])
            jjnr.extend(fep["jjnr"][part]); excl.extend(np.where(rng.random(len(part)) < 0.1, 1 - fep["excl_fep"][part], fep["excl_fep"][part]))
            jindex.append(len(jjnr))
    # (a self pair keeps its flag: the reference's list flags it excluded)
    iin, jj, ex = np.repeat(np.asarray(iinr), np.diff(jindex)), np.asarray(jjnr), np.asarray(excl)
    ex[iin == jj] = 0
    c.plist.fep = dict(iinr=np.asarray(iinr, np.int32), shift=np.asarray(shift, np.int32), jindex=np.asarray(jindex, np.int32),
                       jjnr=jj.astype(np.int32), excl_fep=ex.astype(np.int32))
    want = tl.run_oracle(c, energy=True)
    merged = tl.run_gpu(c, energy=True, fused=False)
    tl.assert_parity(merged, want, rel=1e-4, label="regrouped list %d" % seed)
    monkeypatch.setenv("NBNXM_HIP_DIAGNOSTICS", "1")
    monkeypatch.setenv("NBNXM_HIP_FEP_LIST_MERGED", "0")
    own = tl.run_gpu(c, energy=True, fused=False)
    tl.assert_parity(own, want, rel=1e-4, label="flattened list %d" % seed)


@pytest.mark.parametrize("elec", ["rf", "cut", "ewald", "ewald_tab", "ewald_tab_kept"])
@pytest.mark.parametrize("energy", [False, True])
def test_fused_path_matches_oracle(elec, energy):
    c = tl.make_case(elec=elec, seed=22, **SMALL)
    got = tl.run_gpu(c, energy=energy, fused=True)
    want = tl.run_oracle(c, energy=True)
    tl.assert_parity(got, want, rel=1e-4, energy=energy, label="fused " + elec)


@pytest.mark.parametrize("vdw", ["pswitch", "fswitch"])
@pytest.mark.parametrize("fused", [False, True])
def test_vdw_modifiers(vdw, fused):
    # force switch: perturbed pairs use plain shifted LJ on both sides (reference behaviour)
    c = tl.make_case(elec="rf", vdw=vdw, seed=23, **SMALL)
    got = tl.run_gpu(c, energy=True, fused=fused)
    want = tl.run_oracle(c, energy=True)
    tl.assert_parity(got, want, rel=1e-4, label=vdw)


@pytest.mark.parametrize("sc_alpha,sc_power,sc_coul,lam", [(0.0, 1, True, 0.5), (0.5, 2, True, 0.3), (0.5, 1, False, 0.7),
                                                            (0.5, 1, True, 0.0), (0.5, 1, True, 1.0)])
@pytest.mark.parametrize("fused", [False, True])
def test_softcore_settings(sc_alpha, sc_power, sc_coul, lam, fused):
    c = tl.make_case(elec="ewald", seed=24, sc_alpha=sc_alpha, sc_power=sc_power, sc_coul=sc_coul,
                     lambda_coul=lam, lambda_vdw=min(1.0, lam + 0.2) if 0 < lam < 1 else lam, **SMALL)
    got = tl.run_gpu(c, energy=True, fused=fused)
    want = tl.run_oracle(c, energy=True)
    tl.assert_parity(got, want, rel=1e-4, label="sc")


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("elec,vdw", [("rf", "cut"), ("ewald", "cut"), ("ewald", "pswitch"), ("cut", "cut")])
def test_foreign_lambda_energies(elec, vdw, fused):
    # split: nbnxmFepForeignKernel on the atom-pair list; fused: nbnxmFepClusterKernel's FOREIGN flavour, no atom-pair list
    c = tl.make_case(elec=elec, vdw=vdw, seed=25, n_lambda=11, **SMALL)
    got = tl.run_gpu(c, energy=True, fused=fused, dhdl=True)
    want = tl.run_oracle(c, energy=True, foreign=True)
    tl.assert_foreign(got, want, rel=1e-4)      # per lambda, against the magnitude of that lambda's own terms
    # index 0 is the current lambda: must agree with the FEP part of the force/energy kernels
    fep, fw = want["parts"]["fep"], want["foreign"]
    scale = max(1.0, fw["eVdwAbs"][0] + fw["eCoulAbs"][0])
    assert abs(got["foreign"]["energies"][0] - (fep["Vv"] + fep["Vc"])) <= 1e-4 * scale


@pytest.mark.parametrize("own_kernel", [False, True])
@pytest.mark.parametrize("elec", ["rf", "ewald"])
def test_foreign_lambda_energies_of_heavy_cluster_pairs(elec, own_kernel, monkeypatch):
    """dH/dlambda step of a box with a compact region of 40 perturbed molecules: whole clusters are perturbed, their pairs hold 64 perturbed
    atom pairs (evaluated index by index, not dealt out over the lanes), and the ones with more than 32 stand at the front of the
    slow-pair list and are split over three waves by lambda index — in the cluster kernel's trailing workgroups and (own_kernel) in
    nbnxmFepClusterKernel's FOREIGN flavour."""
    if own_kernel:
        monkeypatch.setenv("NBNXM_HIP_DIAGNOSTICS", "1")
        monkeypatch.setenv("NBNXM_HIP_ENERGY_TAIL", "1")     # the perturbed pairs of energy steps in the kernel of their own
    c = tl.make_case(elec=elec, seed=29, n_lambda=11, nm=(10, 10, 10), num_perturbed_molecules=40)
    bits = np.asarray(c.grid.fepBits).reshape(-1)
    assert np.count_nonzero(np.array([bin(int(b)).count("1") for b in bits]) >= 5) >= 10    # clusters that make heavy pairs
    got = tl.run_gpu(c, energy=True, fused=True, dhdl=True)
    want = tl.run_oracle(c, energy=True, foreign=True)
    tl.assert_parity(got, want, rel=1e-4, label="heavy")
    tl.assert_foreign(got, want, rel=1e-4)


def test_empty_and_ragged_lists():
    # no perturbed atoms at all: the FEP list is empty, only the cluster kernel runs
    c = tl.make_case(elec="rf", seed=26, nm=(10, 10, 10), num_perturbed_molecules=0)
    assert len(c.plist.fep["jjnr"]) == 0
    for fused in (False, True):
        got = tl.run_gpu(c, energy=True, fused=fused)
        want = tl.run_oracle(c, energy=True)
        tl.assert_parity(got, want, rel=1e-4, label="no-fep")
        assert got["dvdl_coul"] == 0.0 and got["dvdl_vdw"] == 0.0
    # split i-entries (list balancing) and a larger ligand: ragged FEP entries up to the 64-j cap
    c = tl.make_case(elec="ewald", seed=27, nm=(10, 10, 10), num_perturbed_molecules=16, max_cjpacked_per_sci=4)
    nj = np.diff(c.plist.fep["jindex"])
    assert nj.max() == 64 and nj.min() <= 3
    for fused in (False, True):
        got = tl.run_gpu(c, energy=True, fused=fused)
        want = tl.run_oracle(c, energy=True)
        tl.assert_parity(got, want, rel=1e-4, label="ragged")


def test_outputs_accumulate_and_clear():
    c = tl.make_case(elec="rf", seed=28, **SMALL)
    nb = tl.setup_gpu(c)
    a = tl.run_gpu(c, energy=True, nb=nb)
    b = tl.run_gpu(c, energy=True, nb=nb)   # clear_outputs in between: same result, not doubled
    assert abs(a["e_el"] - b["e_el"]) <= 1e-5 * abs(a["e_el"])
    frms = np.sqrt(np.mean(a["f"] ** 2))
    assert np.max(np.abs(a["f"] - b["f"])) <= 1e-4 * frms   # float atomics: order-dependent round-off only
    # without clearing, a second launch accumulates (+=) like the reference's kernels
    sw = pkg.step_workload(energy=True, virial=True)
    nb.clear_outputs(True)
    nb.launch_kernel(sw)
    nb.launch_kernel(sw)
    f = np.zeros((c.grid.num_atoms, 3), np.float32)
    nb.launch_cpyback(f, sw)
    res = nb.wait_finish_task(sw, c.have_soft_core)
    assert abs(res["e_el"] - 2 * a["e_el"]) <= 1e-4 * abs(2 * a["e_el"])
    assert np.max(np.abs(f - 2 * a["f"])) <= 2e-4 * frms
    nb.free()


@pytest.mark.parametrize("fused", [False, True])
def test_small_outputs_are_swapped_not_stale(fused):
    """The scalar outputs and the shift forces exist twice (round 4): nbnxm_gpu_clear_outputs swaps to the copy a kernel's trailing workgroups
    have zeroed.  A sequence that mixes energy, virial-only, dH/dlambda and force-only steps on ONE object must give every step the oracle's
    energies, dV/dlambda, shift forces and foreign-lambda terms — nothing left over from a step two swaps back, nothing missing; and after
    nbnxm_gpu_get_fshift (a caller that keeps the pointer) the same without swaps."""
    c = tl.make_case(elec="ewald", seed=33, n_lambda=11, **SMALL)
    want = tl.run_oracle(c, energy=True, foreign=True)
    nb = tl.setup_gpu(c, fused=fused, use_dynamic_pruning=True)
    f0 = None
    for round_ in range(2):
        for kind in ("energy", "force", "energy", "dhdl", "virial", "energy", "force", "force", "dhdl", "energy"):
            if kind == "virial":
                sw = pkg.step_workload(energy=False, virial=True, dhdl=False)
                nb.clear_outputs(True)
                nb.launch_kernel(sw)
                f = np.zeros((c.grid.num_atoms, 3), np.float32)
                nb.launch_cpyback(f, sw)
                res = nb.wait_finish_task(sw, c.have_soft_core)
                got = dict(f=f.astype(np.float64), fshift=res["fshift"].astype(np.float64))
                tl.assert_parity(dict(got, e_lj=0, e_el=0, dvdl_coul=0, dvdl_vdw=0), want, rel=1e-4, energy=False, label="virial-only step")   # forces and shift forces
                continue
            got = tl.run_gpu(c, energy=(kind != "force"), fused=fused, dhdl=(kind == "dhdl"), nb=nb)
            tl.assert_parity(got, want, rel=1e-4, energy=(kind != "force"), label="%s step, round %d" % (kind, round_))
            if kind == "dhdl":
                tl.assert_foreign(got, want, rel=1e-4)
            if f0 is None:
                f0 = got["f"]
        if round_ == 0:
            # from here on the caller holds the pointer of the shift forces: no more swaps of the output blocks, same results
            lib = pkg.hip_lib()
            lib.nbnxm_gpu_get_fshift.restype = C.c_void_p
            assert lib.nbnxm_gpu_get_fshift(nb.h)
    nb.free()


def test_prune_kernel_matches_oracle_and_keeps_forces():
    import oracle_binding as ob
    c = tl.make_case(elec="rf", seed=29, **SMALL)
    want = tl.run_oracle(c, energy=True)
    nb = tl.setup_gpu(c, use_dynamic_pruning=True)
    got = tl.run_gpu(c, energy=True, nb=nb)          # first launch prunes the fresh list, then computes
    tl.assert_parity(got, want, rel=1e-4, label="pruned")
    # the pruned masks equal the oracle's pruning of the same list
    cj = c.plist.cjPacked.copy()
    nleft = ob.nbnxm_prune(c.plist.sci, cj, c.grid.xq, c.grid.shift_vec, c.rlist)
    assert nleft < c.plist.num_cluster_pairs
    dev = pkg.download_cjpacked(nb, len(cj))
    assert np.array_equal(dev["imei"]["imask"], cj["imei"]["imask"])
    # rolling prune in 4 parts over the already pruned list leaves it unchanged (rlistInner == rlistOuter)
    for _ in range(4):
        nb.launch_kernel_pruneonly(num_parts=4)
    dev2 = pkg.download_cjpacked(nb, len(cj))
    assert np.array_equal(dev2["imei"]["imask"], cj["imei"]["imask"])
    nb.free()


@pytest.mark.parametrize("split", [0, 3])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("merged", [True, False])
def test_rolling_prune_with_moving_atoms(fused, merged, split, monkeypatch):
    """Dynamic pruning as mdrun drives it: the first launch prunes the fresh list to the outer radius (kept in gpu_plist::imask) and
    to the inner radius (the working masks); afterwards every step re-checks one of numParts parts of the list for cluster pairs that
    have come inside the inner radius (nbnxm_cuda_kernel_pruneonly.cuh:100-316).  merged: the part runs in trailing workgroups of the
    force-only cluster kernel; otherwise in the prune kernel, at once.  Masks against the oracle's pruning, forces against the oracle.
    split 3: the i-entries arrive cut into pieces of at most 3 groups (a list builder balancing for GPUs); the cluster kernel walks them
    joined again (gpu_plist::sciSorted) while the pruning works on the caller's entries."""
    import oracle_binding as ob
    monkeypatch.setenv("NBNXM_HIP_DIAGNOSTICS", "1")
    monkeypatch.setenv("NBNXM_HIP_PRUNE_MERGED", "1" if merged else "0")
    c = tl.make_case(elec="ewald", seed=33, max_cjpacked_per_sci=split, **SMALL)
    c.rlist_inner = 1.03
    g = c.grid
    pl = c.plist_fused if fused else c.plist
    nb = tl.setup_gpu(c, fused=fused, use_dynamic_pruning=True)
    tl.run_gpu(c, energy=False, fused=fused, nb=nb)            # first launch: prunes the fresh list
    outer = pl.cjPacked.copy()
    ob.nbnxm_prune(pl.sci, outer, g.xq, g.shift_vec, c.rlist)
    inner = outer.copy()
    n_inner = ob.nbnxm_prune(pl.sci, inner, g.xq, g.shift_vec, c.rlist_inner)
    dev = pkg.download_cjpacked(nb, len(outer))
    assert np.array_equal(dev["imei"]["imask"], inner["imei"]["imask"])
    # the atoms move (the list stays valid: displacements far below the buffer)
    rng = np.random.default_rng(5)
    xq_new = g.xq.reshape(-1, 4).copy()
    xq_new[:, :3] += rng.normal(0.0, 0.012, size=(len(xq_new), 3)).astype(np.float32) * (g.atomIndices >= 0)[:, None]
    nb.copy_xq_to_gpu(xq_new)
    came_in = outer.copy()
    ob.nbnxm_prune(pl.sci, came_in, xq_new, g.shift_vec, c.rlist_inner)
    want_mask = inner["imei"]["imask"] | came_in["imei"]["imask"]
    assert (want_mask != inner["imei"]["imask"]).any()          # the test moves something across the inner radius
    # reference forces at the new coordinates (the oracle on the full outer list: pruned pairs add nothing)
    xq_old = g.xq
    g.xq = xq_new
    x_wrapped_old = g.x_wrapped
    xw = g.x_wrapped.copy()
    real = g.atomIndices >= 0
    xw[g.atomIndices[real]] = xq_new[real, :3]
    g.x_wrapped = xw
    try:
        want = tl.run_oracle(c, energy=False)
    finally:
        g.xq, g.x_wrapped = xq_old, x_wrapped_old
    for _ in range(4):
        nb.launch_kernel_pruneonly(num_parts=4)
        got = tl.run_gpu(c, energy=False, fused=fused, nb=nb)
        tl.assert_parity(got, want, rel=1e-4, energy=False, label="rolling prune")
    dev = pkg.download_cjpacked(nb, len(outer))
    assert np.array_equal(dev["imei"]["imask"], want_mask)
    nb.free()


@pytest.mark.parametrize("fused", [False, True])
def test_local_launch_in_two_parts(fused):
    """nbnxm_gpu_set_local_launch_parts / nbnxm_gpu_launch_kernel_part (the schedule of a decomposed step: most of the local list beside
    the coordinate halo, the rest behind the non-local kernel): on the 96k box the list is long enough for two sets of ranges.  Part 1 +
    part 2 must give what one launch gives — forces against the oracle —, a caller that does not ask for parts gets both sets from
    nbnxm_gpu_launch_kernel, energy steps work on the two-set partition too, and the first part alone must NOT be the whole result."""
    c = tl.make_case(elec="ewald", seed=2026, nm=(40, 40, 20), num_perturbed_molecules=16, max_cjpacked_per_sci=16)
    want = tl.run_oracle(c, energy=True, num_threads=8)
    nb = tl.setup_gpu(c, fused=fused, use_dynamic_pruning=True)
    nb.set_local_launch_parts(2, 0.65)
    f = np.zeros((c.grid.num_atoms, 3), np.float32)
    frms = math.sqrt(float(np.mean(np.sum(np.asarray(want["f"]) ** 2, axis=1))))

    def forces(launch):
        sw = pkg.step_workload(energy=False, virial=True)
        nb.clear_outputs(True)
        launch(sw)
        nb.launch_cpyback(f, sw)
        res = nb.wait_finish_task(sw, c.have_soft_core)
        return dict(f=f.astype(np.float64).copy(), fshift=res["fshift"].astype(np.float64))

    def both_parts(sw):
        nb.launch_kernel_part(sw, 1)
        nb.launch_kernel_part(sw, 2)

    for step in range(3):       # fresh list (first prune inside part 1), then steady state with swapped force buffers
        got = forces(both_parts)
        tl.assert_parity(got, want, rel=1e-4, energy=False, label="two parts, step %d" % step)
    first_only = forces(lambda sw: nb.launch_kernel_part(sw, 1))
    assert np.max(np.abs(first_only["f"] - want["f"])) > frms              # the second set is real work
    nb.clear_outputs(True)
    got = forces(lambda sw: nb.launch_kernel(sw))                            # no parts asked for: both sets, back to back
    tl.assert_parity(got, want, rel=1e-4, energy=False, label="two sets, one call")
    got = tl.run_gpu(c, energy=True, fused=fused, nb=nb)                     # energy flavour on its own two-set partition
    tl.assert_parity(got, want, rel=1e-4, label="two sets, energy step")
    nb.set_local_launch_parts(1)
    got = forces(lambda sw: nb.launch_kernel(sw))
    tl.assert_parity(got, want, rel=1e-4, energy=False, label="back to one set")
    nb.free()


@pytest.mark.parametrize("fused", [False, True])
def test_two_part_launch_on_a_list_too_short_for_two_sets(fused):
    """The two-part local launch is the default of a decomposed run (csrc/halo_exchange.hip), and a small domain's local list is too
    short for two sets of ranges: part 1 is then the whole launch and part 2 must queue nothing AND take nothing.  Before commit
    c622397 the empty part 2 dropped a pending rolling-prune part and marked the spare force buffer as zeroed, so the next
    nbnxm_gpu_clear_outputs swapped in a buffer nobody had cleared (forces doubled).  Steps F, F, VF, F with a rolling-prune part
    pending before each: forces (and energies on the VF step) against the oracle EVERY step, and the pruning masks must have
    advanced exactly as in a run without parts."""
    import ctypes as C
    import oracle_binding as ob
    c = tl.make_case(elec="ewald", seed=33, **SMALL)
    c.rlist_inner = 1.03
    g = c.grid
    pl = c.plist_fused if fused else c.plist
    nb = tl.setup_gpu(c, fused=fused, use_dynamic_pruning=True)
    nb.set_local_launch_parts(2, 0.65)

    def both_parts(nb_, sw):
        nb_.launch_kernel_part(sw, 1)
        nb_.launch_kernel_part(sw, 2)

    want0 = tl.run_oracle(c, energy=True)
    got = tl.run_gpu(c, energy=False, fused=fused, nb=nb, launch=both_parts)        # fresh list: first prune inside part 1
    tl.assert_parity(got, want0, rel=1e-4, energy=False, label="short list, fresh")
    # the list really is too short for two sets: one set of ranges in both partitions
    lib = pkg.hip_lib()
    lib.nbnxm_gpu_debug_get_work_ranges.restype = C.c_void_p
    for p in (0, 1):
        n = C.c_int()
        lib.nbnxm_gpu_debug_get_work_ranges(nb.h, C.c_int(pkg.LOCAL), C.c_int(p), C.byref(n))
        assert 0 < n.value <= 1024 * (4 + p), "the 3k box must not be partitioned in two sets"
    outer = pl.cjPacked.copy()
    ob.nbnxm_prune(pl.sci, outer, g.xq, g.shift_vec, c.rlist)
    inner = outer.copy()
    ob.nbnxm_prune(pl.sci, inner, g.xq, g.shift_vec, c.rlist_inner)
    # the atoms move (the list stays valid), so that the rolling passes have cluster pairs to bring back
    rng = np.random.default_rng(5)
    xq_new = g.xq.reshape(-1, 4).copy()
    xq_new[:, :3] += rng.normal(0.0, 0.012, size=(len(xq_new), 3)).astype(np.float32) * (g.atomIndices >= 0)[:, None]
    nb.copy_xq_to_gpu(xq_new)
    came_in = outer.copy()
    ob.nbnxm_prune(pl.sci, came_in, xq_new, g.shift_vec, c.rlist_inner)
    want_mask = inner["imei"]["imask"] | came_in["imei"]["imask"]
    assert (want_mask != inner["imei"]["imask"]).any()
    xq_old, xw_old = g.xq, g.x_wrapped
    xw = g.x_wrapped.copy()
    real = g.atomIndices >= 0
    xw[g.atomIndices[real]] = xq_new[real, :3]
    g.xq, g.x_wrapped = xq_new, xw
    try:
        want = tl.run_oracle(c, energy=True)
    finally:
        g.xq, g.x_wrapped = xq_old, xw_old
    for step, energy in enumerate((False, False, True, False)):
        nb.launch_kernel_pruneonly(num_parts=4)                                   # a rolling-prune part is pending at every launch
        got = tl.run_gpu(c, energy=energy, fused=fused, nb=nb, launch=both_parts)
        tl.assert_parity(got, want, rel=1e-4, energy=energy, label="short list in two parts, step %d" % step)
    dev = pkg.download_cjpacked(nb, len(outer))
    assert np.array_equal(dev["imei"]["imask"], want_mask), "a rolling-prune part was dropped"
    nb.free()


@pytest.mark.parametrize("elec,vdw", [("ewald", "cut"), ("rf", "cut"), ("ewald", "fswitch"), ("ewald", "pswitch"), ("ewald_tab", "cut"), ("ewald_tab_kept", "cut"),
                                      ("ewald", "ewald_geom")])
@pytest.mark.parametrize("fused", [False, True])
def test_excluded_atoms_on_top_of_each_other(elec, vdw, fused):
    """Two EXCLUDED atoms at zero distance (a shell on its core; here hydrogens moved onto their oxygen, and one perturbed molecule
    among them): the reference clamps r^2 to 3.82e-7 so that such pairs give no NaN (pairlist.h:166).  The force-only flavours of the
    headline configurations have no clamp — their exclusion mask is a bit-wise AND that returns +0 for any masked value —, the others
    keep it: every flavour must give finite forces equal to the oracle's, force-only and energy steps."""
    c = tl.make_case(elec=elec, vdw=vdw, seed=91, **SMALL)
    g = c.grid
    xq = g.xq.reshape(-1, 4).copy()
    xw = g.x_wrapped.copy()
    real = np.flatnonzero(g.atomIndices >= 0)
    slot_of = {int(g.atomIndices[s]): int(s) for s in real}
    moved = 0
    for mol in range(0, 60, 3):                      # 20 molecules incl. the perturbed ones (the first molecules of the box)
        o, h = 3 * mol, 3 * mol + 1
        if o in slot_of and h in slot_of:
            xq[slot_of[h], :3] = xq[slot_of[o], :3]
            xw[h] = xw[o]
            moved += 1
    assert moved >= 10
    g.xq, g.x_wrapped = xq, xw
    nb = tl.setup_gpu(c, fused=fused)
    for energy in (False, True):
        want = tl.run_oracle(c, energy=energy)
        got = tl.run_gpu(c, energy=energy, fused=fused, nb=nb)
        assert np.isfinite(got["f"]).all() and np.isfinite(want["f"]).all()
        tl.assert_parity(got, want, rel=1e-4, energy=energy, label="coincident excluded atoms, energy=%s" % energy)
    nb.free()


@pytest.mark.parametrize("elec,vdw", [("ewald", "cut"), ("rf", "cut"), ("ewald", "fswitch"), ("rf", "comb_geom"), ("ewald_tab", "cut"), ("ewald_tab_kept", "cut")])
@pytest.mark.parametrize("fused", [False, True])
def test_filler_atoms_on_one_point(elec, vdw, fused):
    """The reference parks ALL filler atoms of its grid on one point (atomdata.cpp:148-184: x = y = z = -1e6) and the shim hands
    nbat->x() over unchanged, so filler-filler pairs — listed, NOT excluded, zero charge, zero LJ type — sit at r = 0 (this repo's own
    grid builder spreads its fillers, so no other test sees them).  The force-only one-mask block has no r^2 clamp: without the floor
    the kernel's sum of squares starts from (c_r2Floor) such a pair is 0 x inf = NaN in the fillers' force slots and, through the
    i-force sum, in the shift forces of a virial step.  Forces and shift forces must be finite and equal to the oracle's (which
    clamps like the reference), on force-only, force + virial and energy steps."""
    c = tl.make_case(elec=elec, vdw=vdw, seed=93, **SMALL)
    g = c.grid
    xq = g.xq.reshape(-1, 4).copy()
    fillers = np.flatnonzero(g.atomIndices < 0)
    assert len(fillers) >= 8, "the test box has to end in a partly filled cluster"
    xq[fillers, :3] = -1.0e6
    g.xq = xq
    nb = tl.setup_gpu(c, fused=fused)
    for energy in (False, True):
        want = tl.run_oracle(c, energy=energy)
        got = tl.run_gpu(c, energy=energy, fused=fused, nb=nb)       # run_gpu asks for the virial too (shift forces)
        assert np.isfinite(got["f"]).all(), "NaN / inf in the forces (%d filler slots)" % len(fillers)
        assert np.isfinite(np.asarray(got["fshift"])).all(), "NaN / inf in the shift forces"
        assert np.all(np.asarray(got["f"])[fillers] == 0.0), "a filler atom received a force"
        tl.assert_parity(got, want, rel=1e-4, energy=energy, label="fillers on one point, energy=%s" % energy)
    nb.free()


def test_force_buffer_swap_and_pinning():
    """nbnxm_gpu_clear_outputs swaps to the force buffer that the last force-only kernel zeroed in its tail; an energy-step kernel has
    no tail (the next clear is a kernel again); nbnxm_gpu_get_f pins the buffer.  Every step of a mixed sequence must give the
    oracle's forces — a stale or doubly used buffer would show as doubled or missing forces."""
    c = tl.make_case(elec="ewald", seed=51, **SMALL)
    want = tl.run_oracle(c, energy=True)
    nb = tl.setup_gpu(c, fused=True)
    for energy in (False, False, False, True, False, True, True, False, False):
        got = tl.run_gpu(c, energy=energy, fused=True, nb=nb)
        tl.assert_parity(got, want, rel=1e-4, energy=energy, label="swap sequence")
    p1 = nb.f_device_pointer()
    for energy in (False, False, True, False):
        got = tl.run_gpu(c, energy=energy, fused=True, nb=nb)
        tl.assert_parity(got, want, rel=1e-4, energy=energy, label="pinned")
        assert nb.f_device_pointer() == p1
    nb.free()


def test_full_size_properties_100k():
    """BASELINE size (96k atoms, 48 perturbed): properties that need no full oracle pass —
    Newton's third law (zero net force incl. shift bookkeeping), fused == split, F-only == VF forces."""
    c = tl.make_case(elec="ewald", seed=2026, nm=(40, 40, 20), num_perturbed_molecules=16, max_cjpacked_per_sci=16)
    split = tl.run_gpu(c, energy=True, fused=False)
    fused = tl.run_gpu(c, energy=True, fused=True)
    fonly = tl.run_gpu(c, energy=False, fused=True)
    frms = np.sqrt(np.mean(split["f"] ** 2))
    assert np.max(np.abs(split["f"].sum(axis=0))) <= 1e-3 * frms * np.sqrt(c.grid.num_atoms)
    assert np.max(np.abs(split["f"] - fused["f"])) <= 1e-4 * frms
    assert np.max(np.abs(fonly["f"] - fused["f"])) <= 1e-4 * frms
    for k in ("e_lj", "e_el", "dvdl_coul", "dvdl_vdw"):
        assert abs(split[k] - fused[k]) <= 1e-4 * max(abs(split[k]), 1e-3 * abs(split["e_el"]))
    # and the oracle on the full box (1.6 s of CPU)
    want = tl.run_oracle(c, energy=True)
    tl.assert_parity(fused, want, rel=1e-4, label="96k fused")


@pytest.mark.parametrize("elec", ["rf", "ewald", "ewald_tab", "ewald_tab_kept", "cut"])
@pytest.mark.parametrize("vdw", ["cut", "fswitch", "pswitch", "comb_geom", "comb_lb", "ewald_geom"])
def test_force_only_flavours(elec, vdw):
    """The force-only instances are code of their own (one-mask pair block, 5 waves per SIMD for cut-off and switch flavours, the Ewald
    tables read first): every electrostatics x VdW combination as a force-only step, both modes, against the forces of the oracle's
    energy pass.  With few atom types the combination-rule flavours run the table kernel (nbnxm_gpu_launch_kernel); their own force
    instances, which systems with more than 28 types get, are switched back on for a second pass."""
    c = tl.make_case(elec=elec, vdw=vdw, seed=57, **SMALL)
    want = tl.run_oracle(c, energy=True)
    for keep_comb in ((False, True) if vdw.startswith("comb") else (False,)):
        for fused in (False, True):
            nb = tl.setup_gpu(c, fused=fused, keep_combination_kernels=keep_comb)
            got = tl.run_gpu(c, energy=False, fused=fused, nb=nb)
            nb.free()
            tl.assert_parity(got, want, rel=1e-4, energy=False, label="F %s %s fused %d own comb kernel %d" % (elec, vdw, fused, keep_comb))


@pytest.mark.parametrize("elec", ["ewald", "ewald_tab", "ewald_tab_kept"])
@pytest.mark.parametrize("vdw", ["cut", "pswitch", "fswitch", "comb_lb"])
@pytest.mark.parametrize("fused", [False, True])
def test_twin_range_kernels(elec, vdw, fused):
    # rvdw < rcoulomb: ElecType::EwaldAnaTwin / EwaldTabTwin (nbnxm_cuda_kernel.cuh VDW_CUTOFF_CHECK); the perturbed pairs
    # follow the CPU kernel's separate rvdw / rcoulomb
    c = tl.make_case(elec=elec, vdw=vdw, seed=44, rvdw=0.85, rvdw_switch=0.7, **SMALL)
    want = tl.run_oracle(c, energy=True)
    got = tl.run_gpu(c, energy=True, fused=fused)
    tl.assert_parity(got, want, rel=1e-4, label="twin %s %s" % (elec, vdw))
    got_f = tl.run_gpu(c, energy=False, fused=fused)
    tl.assert_parity(got_f, want, rel=1e-4, energy=False, label="twin F %s %s" % (elec, vdw))


@pytest.mark.parametrize("elec,vdw", [("rf", "cut"), ("ewald", "cut"), ("cut", "cut"), ("ewald_tab", "cut"), ("ewald_tab_kept", "cut"), ("ewald", "pswitch"),
                                      ("rf", "pswitch")])
@pytest.mark.parametrize("fused", [False, True])
def test_gapsys_softcore(elec, vdw, fused):
    # the CPU kernel's second soft-core function on the GPU (the reference's GPU kernels stop at Beutler); the oracle's Gapsys
    # branch is pinned by 36 of the reference's known answers (tests/test_oracle_golden.py)
    c = tl.make_case(elec=elec, vdw=vdw, seed=46, softcore="gapsys", n_lambda=5, lambda_coul=0.4, lambda_vdw=0.6, **SMALL)
    want = tl.run_oracle(c, energy=True, foreign=True)
    got = tl.run_gpu(c, energy=True, fused=fused, dhdl=True)
    tl.assert_parity(got, want, rel=1e-4, label="gapsys %s %s" % (elec, vdw))
    tl.assert_foreign(got, want, rel=1e-4)
    got_f = tl.run_gpu(c, energy=False, fused=fused)
    tl.assert_parity(got_f, want, rel=1e-4, energy=False, label="gapsys F %s %s" % (elec, vdw))


def test_gapsys_softcore_changes_the_result_and_zero_linpoints_switch_it_off():
    c = tl.make_case(elec="rf", seed=46, softcore="gapsys", **SMALL)
    b = tl.make_case(elec="rf", seed=46, softcore="beutler", sc_alpha=0.0, **SMALL)      # no soft-core at all
    z = tl.make_case(elec="rf", seed=46, softcore="gapsys", gapsys=(0.0, 0.0, 0.3), **SMALL)
    g, gb, gz = (tl.run_gpu(x, energy=True, fused=True) for x in (c, b, z))
    assert abs(g["dvdl_vdw"] - gb["dvdl_vdw"]) > 1e-3 * abs(gb["dvdl_vdw"])
    # (float atomics: equal up to the order of the sums)
    assert np.max(np.abs(gz["f"] - gb["f"])) <= 1e-5 * np.max(np.abs(gb["f"]))
    assert abs(gz["e_lj"] - gb["e_lj"]) <= 1e-5 * abs(gb["e_lj"]) and abs(gz["dvdl_vdw"] - gb["dvdl_vdw"]) <= 1e-5 * abs(gb["dvdl_vdw"])
    tl.assert_parity(gz, tl.run_oracle(z, energy=True), rel=1e-4, label="gapsys off")


@pytest.mark.parametrize("elec", ["ewald", "rf", "cut", "ewald_tab", "ewald_tab_kept"])
@pytest.mark.parametrize("vdw", ["ewald_geom", "ewald_lb"])
@pytest.mark.parametrize("fused", [False, True])
def test_lj_pme_kernels(elec, vdw, fused):
    # VdwType::EwaldGeom / EwaldLB: the real-space part of the LJ-PME grid term in the cluster kernel (several oxygen types: with
    # Lorentz-Berthelot the grid C6 of a pair is not the table's C6); perturbed pairs follow the CPU kernel (grid correction with
    # the A / B grid C6, nb_free_energy.cpp:1103-1136) — the reference's own GPU kernels evaluate plain shifted LJ there
    c = tl.make_case(elec=elec, vdw=vdw, seed=43, num_extra_types=3, **SMALL)
    want = tl.run_oracle(c, energy=True)
    got = tl.run_gpu(c, energy=True, fused=fused)
    tl.assert_parity(got, want, rel=1e-4, label="LJ-PME %s %s" % (elec, vdw))
    got_f = tl.run_gpu(c, energy=False, fused=fused)
    tl.assert_parity(got_f, want, rel=1e-4, energy=False, label="LJ-PME F %s %s" % (elec, vdw))
    if fused and elec == "ewald":
        # dH/dlambda step: foreign-lambda energies of the perturbed cluster pairs
        cf = tl.make_case(elec=elec, vdw=vdw, seed=43, num_extra_types=3, n_lambda=5, **SMALL)
        wf = tl.run_oracle(cf, energy=True, foreign=True)
        gf = tl.run_gpu(cf, energy=True, fused=True, dhdl=True)
        tl.assert_parity(gf, wf, rel=1e-4, label="LJ-PME dhdl")
        tl.assert_foreign(gf, wf, rel=1e-4)


def test_lj_pme_twin_range_kernel():
    c = tl.make_case(elec="ewald", vdw="ewald_geom", seed=45, num_extra_types=2, rvdw=0.85, **SMALL)
    want = tl.run_oracle(c, energy=True)
    for fused in (False, True):
        got = tl.run_gpu(c, energy=True, fused=fused)
        tl.assert_parity(got, want, rel=1e-4, label="LJ-PME twin")
        got = tl.run_gpu(c, energy=False, fused=fused)          # the force-only instance: grid term cut at rvdw too
        tl.assert_parity(got, want, rel=1e-4, energy=False, label="LJ-PME twin F")


@pytest.mark.parametrize("vdw", ["comb_geom", "comb_lb"])
@pytest.mark.parametrize("fused", [False, True])
def test_combination_rule_kernels(vdw, fused):
    # the cluster kernel takes per-atom LJ parameters; perturbed pairs still use the type table (both agree)
    c = tl.make_case(elec="ewald", vdw=vdw, seed=41, **SMALL)
    got = tl.run_gpu(c, energy=True, fused=fused)
    want = tl.run_oracle(c, energy=True)
    tl.assert_parity(got, want, rel=1e-4, label=vdw)


@pytest.mark.parametrize("extra", [13, 37])
def test_many_atom_types_table_in_lds(extra):
    c = tl.make_case(elec="rf", seed=42, num_extra_types=extra, **SMALL)
    assert c.grid.num_types == 4 + extra
    for fused in (False, True):
        got = tl.run_gpu(c, energy=True, fused=fused)
        want = tl.run_oracle(c, energy=True)
        tl.assert_parity(got, want, rel=1e-4, label="types %d" % extra)


@pytest.mark.parametrize("extra", [26, 60, 96, 119])
@pytest.mark.parametrize("elec", ["ewald", "rf"])
def test_many_atom_types_larger_workgroups(extra, elec):
    """30 / 64 / 100 / 123 atom types: the LJ table (8 ntype^2 bytes) no longer fits the LDS once per 4-wave workgroup at full occupancy,
    so the launch switches to the 4-waves-per-SIMD partition and to workgroups of 8 or 16 waves that share one copy; at 123 types only
    one 8-wave workgroup fits a CU and the partition runs in rounds (nbnxm_gpu_launch_kernel, tests/test_launch_shape.py).  Force-only
    (the 5-waves-per-SIMD flavour) and energy steps, both modes."""
    c = tl.make_case(elec=elec, seed=44, num_extra_types=extra, **SMALL)
    assert c.grid.num_types == 4 + extra
    for fused in (False, True):
        for energy in (False, True):
            got = tl.run_gpu(c, energy=energy, fused=fused)
            want = tl.run_oracle(c, energy=energy)
            tl.assert_parity(got, want, rel=1e-4, energy=energy, label="types %d %s fused %d energy %d" % (4 + extra, elec, fused, energy))


def test_local_and_nonlocal_streams():
    """Two localities as with domain decomposition: the sci entries are dealt to a Local and a NonLocal list,
    both kernels accumulate into the same force buffer, copy-back order NonLocal then Local (sim_util.cpp:1914-1924)."""
    c = tl.make_case(elec="rf", seed=43, **SMALL)
    g = c.grid
    nb = pkg.NbnxmGpu(tl.gpu_interaction_params(c), g.num_types, g.nbat_nbfp(c.sys["nbfp"]), local_and_nonlocal=True,
                      fep=True, n_lambda=0)
    sig6 = c.sc_sigma ** 6
    nb.copy_fepparams(c.sc_alpha, c.sc_alpha, c.sc_power, sig6, sig6, c.lambda_coul, c.lambda_vdw)
    nb.init_atomdata(g.num_atoms, g.type, qA=g.qA, qB=g.qB, typeA=g.typeA, typeB=g.typeB)
    sci = c.plist.sci
    nb.init_pairlist(sci[0::2], c.plist.cjPacked, c.plist.excl, iloc=pkg.LOCAL)
    nb.init_pairlist(sci[1::2], c.plist.cjPacked, c.plist.excl, iloc=pkg.NONLOCAL)
    fep = c.plist.fep

    def entries(which):
        """the i-entries `which` of the atom-pair list as a list of their own (the atom-pair list is dealt to the localities too:
        each locality's part rides in the tail of that locality's cluster kernel)"""
        ji = fep["jindex"]
        lens = (ji[1:] - ji[:-1])[which]
        take = np.concatenate([np.arange(ji[e], ji[e + 1]) for e in which]) if len(which) else np.zeros(0, np.int64)
        return dict(iinr=fep["iinr"][which], shift=fep["shift"][which], jindex=np.concatenate([[0], np.cumsum(lens)]).astype(np.int32),
                    jjnr=fep["jjnr"][take], excl_fep=fep["excl_fep"][take])

    nri = len(fep["iinr"])
    assert nri > 4
    nb.init_feppairlist(entries(np.arange(0, nri, 2)), g.atomIndices, iloc=pkg.LOCAL)
    nb.init_feppairlist(entries(np.arange(1, nri, 2)), g.atomIndices, iloc=pkg.NONLOCAL)
    nb.upload_shiftvec(g.shift_vec)
    sw = pkg.step_workload(energy=True, virial=True)
    nb.clear_outputs(True)
    nb.copy_xq_to_gpu(g.xq, pkg.LOCAL)
    nb.launch_kernel(sw, pkg.LOCAL)
    nb.copy_xq_to_gpu(g.xq, pkg.NONLOCAL)
    nb.launch_kernel(sw, pkg.NONLOCAL)
    f = np.zeros((g.num_atoms, 3), np.float32)
    nb.launch_cpyback(f, sw, pkg.NONLOCAL)
    nb.launch_cpyback(f, sw, pkg.LOCAL)
    nb.wait_finish_task(sw, c.have_soft_core, pkg.NONLOCAL)
    res = nb.wait_finish_task(sw, c.have_soft_core, pkg.LOCAL)
    dv = res["dvdl_nonlin"]
    got = dict(f=f.astype(np.float64), fshift=res["fshift"].astype(np.float64), e_lj=res["e_lj"], e_el=res["e_el"],
               dvdl_coul=dv[0], dvdl_vdw=dv[1])
    want = tl.run_oracle(c, energy=True)
    tl.assert_parity(got, want, rel=1e-4, label="two localities")
    nb.free()


@pytest.mark.parametrize("rvdw", [None, 0.9])
def test_a_tabulated_ewald_pick_runs_the_analytical_kernels(rvdw):
    # the reference picks the tabulated flavours by default on AMD devices; here the analytical ones are faster and run instead
    # (nbnxm_gpu_set_kernel_routing keeps the caller's pick: elec="ewald_tab_kept" in these tests)
    lib = pkg.hip_lib()
    got = {}
    for elec in ("ewald_tab", "ewald"):
        c = tl.make_case(elec=elec, rvdw=rvdw, seed=45, **SMALL)
        nb = tl.setup_gpu(c, fused=True)
        assert lib.nbnxm_gpu_is_kernel_ewald_analytical(nb.h) == 1
        got[elec] = tl.run_gpu(c, energy=True, nb=nb)
        nb.free()
        tl.assert_parity(got[elec], tl.run_oracle(c, energy=True), rel=1e-4, label=elec)
    for key in ("e_el", "e_lj", "dvdl_coul", "dvdl_vdw"):
        # (same kernels, same inputs: what differs is the order in which the waves' sums arrive at the accumulator slots)
        assert abs(got["ewald_tab"][key] - got["ewald"][key]) <= 1e-5 * max(1.0, abs(got["ewald"][key])), key
    frms = math.sqrt(float(np.mean(np.sum(got["ewald"]["f"] ** 2, axis=1))))
    assert np.max(np.abs(got["ewald_tab"]["f"] - got["ewald"]["f"])) <= 2e-5 * frms     # same kernels; the adds arrive in a different order
    nb = tl.setup_gpu(tl.make_case(elec="ewald_tab_kept", rvdw=rvdw, seed=45, **SMALL), fused=True)
    assert lib.nbnxm_gpu_is_kernel_ewald_analytical(nb.h) == 0
    nb.set_kernel_routing(keep_tabulated_kernels=False)
    assert lib.nbnxm_gpu_is_kernel_ewald_analytical(nb.h) == 1
    nb.free()


def test_many_search_steps_in_a_row_on_one_object():
    # a run is searches and steps in turn: twelve times the same atom data, list (from page-locked memory, no wait), perturbed-atom bits
    # and coordinates, each followed by forty force-only steps with a rolling-prune part pending and an energy step — the count of
    # perturbed cluster pairs of each new list is picked up a step or two late (the launches stride over the device's count), the
    # ranges' shares survive, the first-pass prune runs in chunks.  Same forces and energies every time, equal to the oracle's.
    c = tl.make_case(elec="ewald", seed=48, nm=(20, 10, 10), num_perturbed_molecules=6)
    want = tl.run_oracle(c, energy=True)
    pl, g = c.plist_fused, c.grid
    lists = (pkg.pinned_copy(pl.sci), pkg.pinned_copy(pl.cjPacked), pkg.pinned_copy(pl.excl))
    nb = tl.setup_gpu(c, fused=True, use_dynamic_pruning=True)
    sw_f = pkg.step_workload(energy=False, virial=False, dhdl=False)
    first = None
    for search in range(12):
        nb.init_atomdata(g.num_atoms, g.type, qA=g.qA, qB=g.qB, typeA=g.typeA, typeB=g.typeB)
        nb.init_pairlist(*lists)
        nb.init_fep_cluster_bits(g.fepBits)
        nb.copy_xq_to_gpu(g.xq)
        for step in range(40):
            if step > 0:
                nb.launch_kernel_pruneonly(num_parts=8)
            nb.clear_outputs(False)
            nb.launch_kernel(sw_f)
        got = tl.run_gpu(c, energy=True, fused=True, nb=nb)
        tl.assert_parity(got, want, rel=1e-4, label="search %d" % search)
        if first is None:
            first = got
        else:
            assert abs(got["e_el"] - first["e_el"]) <= 1e-5 * abs(first["e_el"]) and abs(got["e_lj"] - first["e_lj"]) <= 1e-5 * abs(first["e_lj"]) + 1e-3
    nb.free()


def test_lists_in_page_locked_memory_are_read_in_place():
    # the reference keeps its pair lists in pinned HostVectors and gpu_init_pairlist reads them without a wait; so does this library for
    # arrays in page-locked memory (other memory is staged and complete on return): same forces either way, twice in a row on one object
    c = tl.make_case(elec="ewald", seed=47, **SMALL)
    want = tl.run_oracle(c, energy=True)
    pl = c.plist_fused
    pinned = (pkg.pinned_copy(pl.sci), pkg.pinned_copy(pl.cjPacked), pkg.pinned_copy(pl.excl))
    nb = tl.setup_gpu(c, fused=True)
    for lists in (pinned, (pl.sci, pl.cjPacked, pl.excl), pinned):
        nb.init_pairlist(*lists)
        nb.init_fep_cluster_bits(c.grid.fepBits)
        got = tl.run_gpu(c, energy=True, fused=True, nb=nb)
        tl.assert_parity(got, want, rel=1e-4, label="pinned lists")
    nb.free()


def test_a_malformed_list_is_caught_on_the_device(tmp_path):
    # gpu_init_pairlist checks the i-entries on the host and the packed groups on the device (nbnxmValidateListKernel): a j-cluster
    # outside the atom range is replaced by cluster 0 — nothing faults — and ends the process at the next launch or finish, with a message
    import subprocess
    import sys
    code = """
import resource, sys
resource.setrlimit(resource.RLIMIT_CORE, (0, 0))     # the process ends in abort(): no core file
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import fep_testlib as tl
pkg = tl.pkg
c = tl.make_case(elec="rf", seed=46, nm=(10, 10, 10), num_perturbed_molecules=3)
nb = tl.setup_gpu(c, fused=True)
cj = c.plist_fused.cjPacked.copy()
cj["cj"][7, 2] = c.grid.num_atoms // 8 + 5
nb.init_pairlist(c.plist_fused.sci, cj, c.plist_fused.excl)
sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
nb.clear_outputs(False)
nb.launch_kernel(sw)
f = np.zeros((c.grid.num_atoms, 3), np.float32)
nb.launch_cpyback(f, sw)
nb.wait_finish_task(sw, c.have_soft_core)
print("NOT CAUGHT")
""" % (tl.ROOT if hasattr(tl, "ROOT") else ".", "tests")
    log = tmp_path / "fatal.log"
    import os
    env = dict(os.environ, NBNXM_HIP_FATAL_LOG=str(log))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode != 0 and "NOT CAUGHT" not in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    assert "j-cluster outside the atom range" in (r.stderr + (log.read_text() if log.exists() else ""))


def test_timing_and_query_entry_points():
    c = tl.make_case(elec="ewald", seed=44, n_lambda=3, **SMALL)
    nb = tl.setup_gpu(c)
    nb.set_timing(True)
    tl.run_gpu(c, energy=True, dhdl=True, nb=nb)
    t = nb.get_timings()
    assert t.nb_k_count == 1 and t.fep_k_count == 1 and t.nb_k_ms > 0 and t.fep_k_ms > 0
    nb.reset_timings()
    assert nb.get_timings().nb_k_count == 0
    lib = pkg.hip_lib()
    assert lib.nbnxm_gpu_is_kernel_ewald_analytical(nb.h) == 1
    assert lib.nbnxm_gpu_min_ci_balanced(nb.h) == 0      # no i-entry splitting wanted: the kernel balances by wave-slot ranges
    assert lib.nbnxm_gpu_have_short_range_work(nb.h, pkg.LOCAL) == 1
    assert nb.stream() is not None
    nb.free()


# ---- coordinate / force buffer operations and the halo pack / unpack kernels ----
def _dev(arr):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda()


def test_x_to_nbat_x_and_force_reduction():
    """nbnxn_gpu_x_to_nbat_x and GpuForceReduction against their index definitions: bit-exact."""
    import torch
    c = tl.make_case(elec="rf", seed=41, **SMALL)
    g = c.grid
    nb = tl.setup_gpu(c, fused=True)
    ai = g.atomIndices
    nb.init_x_to_nbat_x(ai)
    rng = np.random.default_rng(5)
    x_new = (c.sys["x"] + rng.normal(0, 0.01, c.sys["x"].shape)).astype(np.float32)   # atom order
    d_x = _dev(x_new)
    nslots = g.num_atoms
    # two calls, as the reference does per grid / locality
    nb.x_to_nbat_x(d_x.data_ptr(), 0, nslots // 2)
    nb.x_to_nbat_x(d_x.data_ptr(), nslots // 2, nslots)
    xq_dev = np.zeros((nslots, 4), np.float32)
    pkg.hip_lib().nbnxm_gpu_debug_download(nb.h, C.c_void_p(pkg.hip_lib().nbnxm_gpu_get_xq(nb.h)), xq_dev.ctypes.data_as(C.c_void_p),
                                           C.c_size_t(xq_dev.nbytes))
    want = g.xq.reshape(-1, 4).copy()
    real = ai >= 0
    want[real, :3] = x_new[ai[real]]
    assert np.array_equal(xq_dev, want)          # fillers and charges untouched, real atoms replaced

    # force reduction: f_total[a] (+)= f_nbnxm[cell[a]] (+ f_rvec[a])
    sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
    nb.clear_outputs(False)
    nb.launch_kernel(sw)
    f_grid = np.zeros((nslots, 3), np.float32)
    nb.launch_cpyback(f_grid, sw)
    nb.wait_finish_task(sw, c.have_soft_core)
    natoms = c.natoms
    cell = np.full(natoms, -1, np.int32)
    cell[ai[real]] = np.nonzero(real)[0]
    assert (cell >= 0).all()
    base = rng.normal(0, 1, (natoms, 3)).astype(np.float32)
    rvec = rng.normal(0, 1, (natoms, 3)).astype(np.float32)
    for accumulate in (False, True):
        for add_rvec in (False, True):
            nb.force_reduction_reinit(cell, atom_start=0, accumulate=accumulate)
            d_base, d_rvec = _dev(base), _dev(rvec)
            nb.force_reduction_execute(d_base.data_ptr(), d_rvec.data_ptr() if add_rvec else None)
            torch.cuda.synchronize()
            want_f = f_grid[cell]
            if accumulate:
                want_f = base + want_f
            if add_rvec:
                want_f = want_f + rvec
            assert np.array_equal(d_base.cpu().numpy(), want_f.astype(np.float32))
    # a sub-range (atomStart) as used for the non-local atoms
    start = natoms // 3
    nb.force_reduction_reinit(cell[start:], atom_start=start, accumulate=True)
    d_base = _dev(base)
    nb.force_reduction_execute(d_base.data_ptr(), None)
    torch.cuda.synchronize()
    want_f = base.copy()
    want_f[start:] += f_grid[cell[start:]]
    assert np.array_equal(d_base.cpu().numpy(), want_f)
    nb.free()


def test_halo_pack_unpack_kernels():
    import torch
    rng = np.random.default_rng(9)
    n, m = 5000, 1777
    x = rng.normal(0, 1, (n, 3)).astype(np.float32)
    idx = rng.choice(n, m, replace=False).astype(np.int32)     # a halo atom is sent once per pulse
    shift = np.array([1.5, -2.25, 0.125], np.float32)
    d_x, d_map = _dev(x), _dev(idx)
    d_send = torch.zeros((m, 3), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    pkg.halo_pack_x(s, d_x.data_ptr(), d_map.data_ptr(), m, d_send.data_ptr(), shift)
    assert np.array_equal(d_send.cpu().numpy(), x[idx] + shift)
    pkg.halo_pack_x(s, d_x.data_ptr(), d_map.data_ptr(), m, d_send.data_ptr(), None)
    assert np.array_equal(d_send.cpu().numpy(), x[idx])
    f = rng.normal(0, 1, (n, 3)).astype(np.float32)
    recv = rng.normal(0, 1, (m, 3)).astype(np.float32)
    d_f, d_recv = _dev(f), _dev(recv)
    pkg.halo_unpack_f(s, d_f.data_ptr(), d_map.data_ptr(), m, d_recv.data_ptr(), True)
    want = f.copy()
    want[idx] += recv
    assert np.array_equal(d_f.cpu().numpy(), want)
    pkg.halo_unpack_f(s, d_f.data_ptr(), d_map.data_ptr(), m, d_recv.data_ptr(), False)
    want[idx] = recv
    assert np.array_equal(d_f.cpu().numpy(), want)
    # empty map: no launch
    pkg.halo_pack_x(s, d_x.data_ptr(), d_map.data_ptr(), 0, d_send.data_ptr(), None)


def _check_virtual_rank_decomposition(c, ncells, oracle_threads=1, self_links=(False, False, False), rccl=False, energy=True, peer_copy=False,
                                      repeats=2, merged=False, reinit_between=False, push=False):
    """All ranks of a decomposition in one process on one GPU: per rank its own grid over home + halo atoms, local and non-local
    list, two streams, x -> xq per locality, fused cluster kernels, force reduction per locality.  The halo moves
      * peer_copy=True: through the library's in-process peer-copy transport, every rank on its own host thread calling the C++ step
        halo_gpu_domain_force_step `repeats` times back to back with no host synchronisation in between (include/halo_hip.h);
      * push=True (with peer_copy=True): the one-sided transport instead — the senders' kernels store into the receivers' rows and set
        sequence flags, the receivers' kernels wait for them (HALO_GPU_TRANSPORT_PEER_PUSH; merged localities);
      * rccl=True, ONE rank that is its own neighbour: through the real RCCL transport, same C++ step;
      * else: through the tensor-index test double and the schedule spelled out in Python.
    merged (C++ step only): the two lists of every rank as one device list, one cluster-kernel launch per step
    (nbnxm_gpu_set_merged_localities).
    Owners must end up with the forces of the single-domain oracle; energies and dV/dlambda summed over the ranks must equal the
    single-domain ones."""
    import importlib
    import torch
    domdec = importlib.import_module("gromacs_fep_gpu_amd.domdec")
    wl = importlib.import_module("gromacs_fep_gpu_amd.workload")
    dd = domdec.DomainDecomposition(c.sys["x"], c.sys["box"], c.sys["molId"], ncells, c.rlist, self_links=self_links)
    sw = pkg.step_workload(energy=energy, virial=False, dhdl=False)
    steps, halos = [], []
    for r in range(dd.num_ranks):
        plan = dd.plan(r)
        system = domdec.RankSystem(pkg, plan, c.sys["box"], c.sys["qA"], c.sys["qB"], c.sys["typeA"], c.sys["typeB"], c.ntype,
                                   c.sys["molId"], c.rlist, perturbed=c.perturbed)
        assert not merged or peer_copy or rccl, "the Python schedule launches the two localities separately"
        nb = domdec.make_rank_gpu(pkg, wl, c, system, use_dynamic_pruning=True, merged=merged)
        # garbage in the xq buffer and in the halo rows of x: x -> xq and the halo exchange must supply everything
        nb.copy_xq_to_gpu(np.full((system.grid.num_atoms, 4), 1.0e5, np.float32) * np.array([1, 1, 1, 0], np.float32)
                          + system.grid.xq.reshape(-1, 4) * np.array([0, 0, 0, 1], np.float32), pkg.LOCAL)
        nb.copy_xq_to_gpu(np.full((system.grid.num_atoms, 4), 1.0e5, np.float32) * np.array([1, 1, 1, 0], np.float32)
                          + system.grid.xq.reshape(-1, 4) * np.array([0, 0, 0, 1], np.float32), pkg.NONLOCAL)
        if peer_copy:
            if r == 0:
                peer_id = domdec.new_halo_id(pkg, domdec.TRANSPORT_PEER_PUSH if push else domdec.TRANSPORT_PEER_COPY)
            halo = domdec.RcclHalo(pkg, None, r, dd.num_ranks, nb.stream(pkg.LOCAL if merged else pkg.NONLOCAL), unique_id=peer_id)
        elif rccl:
            halo = domdec.RcclHalo(pkg, None, 0, 1, nb.stream(pkg.LOCAL if merged else pkg.NONLOCAL))
        else:
            halo = domdec.TensorHalo(peers={})
        if push:
            steps.append((nb, system, halo, plan))     # halo_gpu_reinit is a rendezvous of the ranks here: every rank on its own thread, below
        else:
            st = domdec.DomainStep(pkg, nb, system, halo)
            st.d_x[plan.num_home:] = 1.0e5
            steps.append(st)
        halos.append(halo)
    if push:
        made = [None] * len(steps)

        def make(i):
            def run():
                nb_, system_, halo_, plan_ = steps[i]
                made[i] = domdec.DomainStep(pkg, nb_, system_, halo_, poison_halo_rows=1.0e5)
            return run
        domdec.run_ranks_in_threads([make(i) for i in range(len(steps))])
        steps = made
    torch.cuda.synchronize()
    if peer_copy:
        def rank_thread(st):
            def run():
                for _ in range(repeats):        # back to back: no event, no host synchronisation between the steps
                    st.step(sw)
                if reinit_between:
                    # a search step in the middle of the run: every rank re-registers its maps and buffers (halo_gpu_reinit) from its
                    # own thread, at its own pace, while peers may still be copying out of its buffers; then more steps
                    st.halo.reinit(st.sys.plan, st.d_x, st.d_f)
                    for _ in range(repeats):
                        st.step(sw)
            return run
        domdec.run_ranks_in_threads([rank_thread(st) for st in steps])
        torch.cuda.synchronize()
    for repeat in range(0 if peer_copy else repeats):     # the second pass runs on the pruned lists and the swapped force buffers
        if rccl:
            steps[0].step(sw)
        else:
            domdec.loopback_exchange_coordinates(halos)
            torch.cuda.synchronize()
            for s in steps:
                s.launch(sw)
                s.reduce_halo_forces()
            torch.cuda.synchronize()
            domdec.loopback_exchange_forces(halos)
            torch.cuda.synchronize()
            for s in steps:
                s.reduce_home_forces()
        torch.cuda.synchronize()
    f_dd = np.zeros((c.natoms, 3), np.float32)
    e_lj = e_el = dvdl_c = dvdl_v = 0.0
    for s in steps:
        f_dd[s.sys.plan.home] = s.home_forces()
        fbuf = np.zeros((s.sys.grid.num_atoms, 3), np.float32)
        s.nb.launch_cpyback(fbuf, sw, pkg.NONLOCAL)
        s.nb.launch_cpyback(fbuf, sw, pkg.LOCAL)
        s.nb.wait_finish_task(sw, c.have_soft_core, pkg.NONLOCAL)
        res = s.nb.wait_finish_task(sw, c.have_soft_core, pkg.LOCAL)
        e_lj, e_el = e_lj + res["e_lj"], e_el + res["e_el"]
        dv = res["dvdl_nonlin"] if c.have_soft_core else res["dvdl_lin"]
        dvdl_c, dvdl_v = dvdl_c + dv[0], dvdl_v + dv[1]
    want = tl.run_oracle(c, energy=True, num_threads=oracle_threads)
    g = c.grid
    real = g.atomIndices >= 0
    f_ref = np.zeros((c.natoms, 3))
    f_ref[g.atomIndices[real]] = want["f"][real]
    got = dict(f=f_dd.astype(np.float64), fshift=want["fshift"], e_lj=e_lj, e_el=e_el, dvdl_coul=dvdl_c, dvdl_vdw=dvdl_v)
    tl.assert_parity(got, dict(want, f=f_ref), rel=1e-4, energy=energy, label="decomposition %s" % (ncells,))
    for s, h in zip(steps, halos):
        h.free()
        s.nb.free()


@pytest.mark.parametrize("ncells", [(2, 1, 1), (3, 1, 1), (2, 2, 1), (2, 2, 2)])
def test_domain_decomposition_virtual_ranks(ncells):
    nm = (14, 8, 8) if ncells[0] == 3 else (10, 10, 10)
    _check_virtual_rank_decomposition(tl.make_case(nm=nm, num_perturbed_molecules=3, elec="ewald", seed=78), ncells)


@pytest.mark.parametrize("ncells", [(2, 1, 1), (3, 1, 1), (2, 2, 1), (2, 2, 2)])
@pytest.mark.parametrize("energy", [False, True])
def test_cpp_domain_step_merged_localities_with_real_peers(ncells, energy):
    """The merged-localities schedule of halo_gpu_domain_force_step (one device list and one cluster-kernel launch per rank and step,
    behind the coordinate halo) with 2 to 8 ranks over the peer-copy transport: three steps back to back, force-only (trailing
    workgroups: perturbed pairs, rolling prune, buffer clear) and energy steps."""
    nm = (14, 8, 8) if ncells[0] == 3 else (10, 10, 10)
    _check_virtual_rank_decomposition(tl.make_case(nm=nm, num_perturbed_molecules=3, elec="ewald", seed=78), ncells, peer_copy=True, repeats=3,
                                      merged=True, energy=energy)


@pytest.mark.parametrize("ncells", [(1, 1, 1), (2, 1, 1), (3, 1, 1), (2, 2, 1), (2, 2, 2)])
@pytest.mark.parametrize("energy", [False, True])
def test_cpp_domain_step_one_sided_transport(ncells, energy):
    """halo_gpu_domain_force_step over the one-sided transport (HALO_GPU_TRANSPORT_PEER_PUSH): coordinates stored straight into the
    receivers' halo rows by the senders' pack kernels, forces into the owners' receive buffers by the forces-to-atom-order kernels, sequence
    flags instead of transfer kernels and events, waits inside the consuming kernels; 1 (its own neighbour), 2, 3, 4 and 8 ranks as host
    threads on this one GPU, three steps back to back with no host synchronisation, home forces and summed energies against the oracle."""
    nm = (14, 8, 8) if ncells[0] == 3 else (10, 10, 10)
    links = (True, True, True) if ncells == (1, 1, 1) else (False, False, False)
    c = tl.make_case(nm=nm, num_perturbed_molecules=3, elec="ewald", seed=79)
    _check_virtual_rank_decomposition(c, ncells, self_links=links, peer_copy=True, push=True, merged=True, repeats=3, energy=energy)


def test_cpp_domain_step_one_sided_transport_with_a_reinit_between_steps_and_at_size():
    """a search step in the middle (halo_gpu_reinit is a rendezvous of the ranks in this transport), and 2 x 2 x 2 ranks of the 96k box"""
    c = tl.make_case(nm=(10, 10, 10), num_perturbed_molecules=3, elec="ewald", seed=88)
    _check_virtual_rank_decomposition(c, (2, 2, 1), peer_copy=True, push=True, merged=True, repeats=2, reinit_between=True)
    c = tl.make_case(nm=(40, 40, 20), num_perturbed_molecules=16, elec="ewald", seed=2026, max_cjpacked_per_sci=16)
    _check_virtual_rank_decomposition(c, (2, 2, 2), oracle_threads=8, peer_copy=True, push=True, merged=True, repeats=3)


@pytest.mark.parametrize("merged", [False, True])
def test_cpp_domain_step_peer_copy_with_a_reinit_between_steps(merged):
    """halo_gpu_reinit in the middle of a run (what every search step does), each rank from its own thread without any barrier: the
    posts of the steps before carry their own snapshot of the sender's layout, so a rank that is still pulling step n is not confused
    by a peer that has already re-registered for step n + 1."""
    c = tl.make_case(nm=(10, 10, 10), num_perturbed_molecules=3, elec="ewald", seed=78)
    _check_virtual_rank_decomposition(c, (2, 2, 1), peer_copy=True, repeats=2, merged=merged, reinit_between=True)


def test_cpp_domain_step_merged_localities_over_rccl_and_at_size():
    """merged localities over the real RCCL groups (one rank that is its own neighbour along x, y, z), and 2 x 2 x 2 ranks of the 96k
    box over the peer-copy transport"""
    c = tl.make_case(nm=(10, 10, 10), num_perturbed_molecules=3, elec="ewald", seed=79)
    _check_virtual_rank_decomposition(c, (1, 1, 1), self_links=(True, True, True), rccl=True, repeats=3, merged=True)
    c = tl.make_case(nm=(40, 40, 20), num_perturbed_molecules=16, elec="ewald", seed=2026, max_cjpacked_per_sci=16)
    _check_virtual_rank_decomposition(c, (2, 2, 2), oracle_threads=8, peer_copy=True, repeats=3, merged=True)


@pytest.mark.parametrize("ncells", [(2, 1, 1), (3, 1, 1), (2, 2, 1), (2, 2, 2)])
@pytest.mark.parametrize("parts", [1, 2])
def test_cpp_domain_step_with_real_peers_over_the_peer_copy_transport(ncells, parts, monkeypatch):
    """halo_gpu_domain_force_step — the C++ step, not the Python schedule — with peers other than itself: 2 to 8 ranks on this one
    GPU, each on its own host thread, exchanging through the library's in-process peer-copy transport (per-peer offsets, the
    event handshake, both streams of every rank, the local launch whole and in two parts: small domains have lists too short for
    two sets, so part 2 is empty there).  Three steps back to back without host synchronisation."""
    monkeypatch.setenv("NBNXM_HIP_DIAGNOSTICS", "1")
    monkeypatch.setenv("HALO_GPU_LOCAL_PARTS", str(parts))
    nm = (14, 8, 8) if ncells[0] == 3 else (10, 10, 10)
    _check_virtual_rank_decomposition(tl.make_case(nm=nm, num_perturbed_molecules=3, elec="ewald", seed=78), ncells, peer_copy=True, repeats=3)


@pytest.mark.parametrize("energy", [False, True])
def test_cpp_domain_step_peer_copy_96k_box_two_part_launch(energy, monkeypatch):
    """2 x 2 x 2 ranks of the 96k box over the peer-copy transport: 12k home atoms per rank; and 2 x 1 x 1, where the local lists are
    long enough for the two-part launch to really cut them in two."""
    monkeypatch.setenv("NBNXM_HIP_DIAGNOSTICS", "1")
    monkeypatch.setenv("HALO_GPU_LOCAL_PARTS", "2")
    c = tl.make_case(nm=(40, 40, 20), num_perturbed_molecules=16, elec="ewald", seed=2026, max_cjpacked_per_sci=16)
    _check_virtual_rank_decomposition(c, (2, 1, 1), oracle_threads=8, peer_copy=True, repeats=3, energy=energy)
    if not energy:
        _check_virtual_rank_decomposition(c, (2, 2, 2), oracle_threads=8, peer_copy=True, repeats=2, energy=energy)


@pytest.mark.parametrize("self_links", [(True, True, True)])
def test_cpp_domain_step_back_to_back_without_host_synchronisation(self_links):
    """Three halo_gpu_domain_force_step calls with coordinatesReadyEvent NULL and nothing between them, over the real RCCL groups (one
    rank that is its own neighbour) and over the peer-copy transport: round 2 received the force halo in the buffer the next step's
    pack kernel writes (on another stream), so the home rows could pick up packed coordinates; the forces now arrive in a buffer of
    their own.  Home forces against the oracle after the third step."""
    c = tl.make_case(nm=(10, 10, 10), num_perturbed_molecules=3, elec="ewald", seed=79)
    _check_virtual_rank_decomposition(c, (1, 1, 1), self_links=self_links, rccl=True, repeats=3)
    _check_virtual_rank_decomposition(c, (1, 1, 1), self_links=self_links, peer_copy=True, repeats=3)


@pytest.mark.parametrize("self_links", [(True, False, False), (True, True, True)])
def test_halo_exchange_over_rccl_with_a_rank_that_is_its_own_neighbour(self_links):
    """The RCCL transport of include/halo_hip.h on ONE GPU: a single rank whose periodic images are its halo (ncclSend / ncclRecv
    to itself inside a group), through the whole two-locality step.  More than one rank per GPU is refused by RCCL, so this is
    the only way to run that code path on a one-GPU box; the multi-rank schedule is covered by the gloo tests and the
    virtual-rank test above."""
    _check_virtual_rank_decomposition(tl.make_case(nm=(10, 10, 10), num_perturbed_molecules=3, elec="ewald", seed=79), (1, 1, 1),
                                      self_links=self_links, rccl=True)


@pytest.mark.parametrize("energy", [False, True])
def test_domain_step_over_rccl_with_the_local_launch_in_two_parts(energy, monkeypatch):
    """halo_gpu_domain_force_step with the local launch in two parts (the default with more than one rank: part 1 beside the coordinate
    halo, the non-local kernel, part 2 beside the force halo).  The 96k box is long enough for two sets of ranges; one rank that is its
    own neighbour along x, y and z runs the real RCCL groups.  Force-only steps (trailing workgroups ride with part 2) and energy steps."""
    monkeypatch.setenv("NBNXM_HIP_DIAGNOSTICS", "1")
    monkeypatch.setenv("HALO_GPU_LOCAL_PARTS", "2")
    c = tl.make_case(nm=(40, 40, 20), num_perturbed_molecules=16, elec="ewald", seed=2026, max_cjpacked_per_sci=16)
    _check_virtual_rank_decomposition(c, (1, 1, 1), oracle_threads=8, self_links=(True, True, True), rccl=True, energy=energy)


def test_full_size_properties_1m():
    """BASELINE configs[4] size (1.05 M atoms): Newton's third law, fused == split, parity with the (threaded) oracle on the
    whole box, and the 8-domain decomposition with all ranks on this one GPU."""
    c = tl.make_case(elec="ewald", seed=404, nm=(88, 88, 44), num_perturbed_molecules=16, max_cjpacked_per_sci=16)
    split = tl.run_gpu(c, energy=True, fused=False)
    fused = tl.run_gpu(c, energy=True, fused=True)
    frms = np.sqrt(np.mean(split["f"] ** 2))
    assert np.max(np.abs(split["f"].sum(axis=0))) <= 1e-3 * frms * np.sqrt(c.grid.num_atoms)
    assert np.max(np.abs(split["f"] - fused["f"])) <= 1e-4 * frms
    want = tl.run_oracle(c, energy=True, num_threads=8)
    tl.assert_parity(fused, want, rel=1e-4, label="1M fused")
    tl.assert_parity(split, want, rel=1e-4, label="1M split")
    _check_virtual_rank_decomposition(c, (2, 2, 2), oracle_threads=8)
    # and through the C++ step with real peers (in-process peer-copy transport, one host thread per rank, two-part local launch)
    _check_virtual_rank_decomposition(c, (2, 2, 2), oracle_threads=8, peer_copy=True, repeats=2, energy=False)


def test_work_partition_with_unequal_shares_covers_the_list_once():
    """The ranges of the work partition get unequal shares of the weight (by default: by the age of the wave on its SIMD).  Whatever
    the shares, the ranges must tile the list: same forces as the oracle, borders monotone from 0 to the number of groups."""
    import torch
    c = tl.make_case(elec="ewald", seed=2026, nm=(40, 40, 20), num_perturbed_molecules=16, max_cjpacked_per_sci=16)
    want = tl.run_oracle(c, energy=True, num_threads=8)
    lib = pkg.hip_lib()
    lib.nbnxm_gpu_debug_get_work_ranges.restype = C.c_void_p
    rng = np.random.default_rng(3)
    for fused in (True, False):
        nb = tl.setup_gpu(c, fused=fused)
        for p, energy in ((1, False), (0, True)):
            nr = C.c_int(0)
            lib.nbnxm_gpu_debug_get_work_ranges(nb.h, 0, p, C.byref(nr))
            n = nr.value
            assert n == 1024 * (4 + p)
            shares = rng.uniform(0.3, 1.7, n).astype(np.float32)
            shares[rng.integers(0, n, 50)] = 1e-4                      # nearly empty ranges
            lib.nbnxm_gpu_debug_set_work_shares(nb.h, 0, p, shares.ctypes.data_as(C.c_void_p), n)
            ptr = lib.nbnxm_gpu_debug_get_work_ranges(nb.h, 0, p, C.byref(nr))
            ranges = np.zeros(n + 1, np.int32)
            lib.nbnxm_gpu_debug_download(nb.h, C.c_void_p(ptr), ranges.ctypes.data_as(C.c_void_p), C.c_size_t(ranges.nbytes))
            ncj = len((c.plist_fused if fused else c.plist).cjPacked)
            assert ranges[0] == 0 and ranges[-1] == ncj and (np.diff(ranges) >= 0).all()
            sizes = np.diff(ranges).astype(np.float64)
            big, small = shares > 1.4, (shares < 0.6) & (shares > 1e-3)
            assert sizes[big].mean() > 1.8 * sizes[small].mean()       # the shares steer the sizes
        got = tl.run_gpu(c, energy=True, fused=fused, nb=nb)
        tl.assert_parity(got, want, rel=1e-4, label="unequal shares, fused=%s" % fused)
        got = tl.run_gpu(c, energy=False, fused=fused, nb=nb)
        tl.assert_parity(got, want, rel=1e-4, energy=False, label="unequal shares, F only, fused=%s" % fused)
        nb.free()


@pytest.mark.parametrize("elec,vdw", [("rf", "cut"), ("ewald", "cut"), ("ewald", "pswitch")])
@pytest.mark.parametrize("fused", [False, True])
def test_forces_are_the_gradient_of_the_energy(elec, vdw, fused):
    """Central difference of the kernel's own energy along a random displacement of all atoms against its own forces: the
    energy and force flavours (cluster-pair and perturbed-pair kernels, soft-core at lambda = 0.5, exclusion corrections) must
    describe one Hamiltonian.  fp32 energies resolve the difference to ~1e-3 of its size."""
    c = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=3, elec=elec, vdw=vdw, seed=77)
    g = c.grid
    nb = tl.setup_gpu(c, fused=fused)
    base = tl.run_gpu(c, energy=True, fused=fused, nb=nb)
    rng = np.random.default_rng(1)
    real = g.atomIndices >= 0
    d = np.zeros((g.num_atoms, 3))
    d[real] = rng.normal(0, 1, (int(real.sum()), 3))
    eps = 2e-4
    xq0 = g.xq.reshape(-1, 4).copy()
    energies = []
    for sign in (+1, -1):
        xq = xq0.copy()
        xq[:, :3] += sign * eps * d
        nb.copy_xq_to_gpu(xq)
        r = tl.run_gpu(c, energy=True, fused=fused, nb=nb)
        energies.append(r["e_lj"] + r["e_el"])
    nb.free()
    de = energies[0] - energies[1]
    want = -2 * eps * float(np.sum(base["f"] * d))
    scale = 2 * eps * float(np.sqrt(np.sum(base["f"] ** 2) * np.sum(d * d) / d.size))     # size of a typical projection
    assert abs(de - want) <= 0.02 * max(abs(want), scale), (de, want, scale)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("elec", ["ewald", "ewald_tab", "ewald_tab_kept"])
def test_pme_loadbal_update_param(elec, fused):
    """PME load balancing moves the Coulomb cut-off and the Ewald coefficient on a living object (gpu_pme_loadbal_update_param):
    scalars, the tabulated force and the LDS correction table must all follow.  Same box, same list (rlist unchanged)."""
    c1 = tl.make_case(elec=elec, seed=91, rc=1.0, **SMALL)
    c2 = tl.make_case(elec=elec, seed=91, rc=0.9, **SMALL)
    assert abs(c1.beta - c2.beta) > 0.1 and np.array_equal(c1.plist.cjPacked["cj"], c2.plist.cjPacked["cj"])
    nb = tl.setup_gpu(c1, fused=fused)
    tl.assert_parity(tl.run_gpu(c1, energy=True, fused=fused, nb=nb), tl.run_oracle(c1, energy=True), rel=1e-4, label="before")
    nb.pme_loadbal_update_param(tl.gpu_interaction_params(c2))
    tl.assert_parity(tl.run_gpu(c2, energy=True, fused=fused, nb=nb), tl.run_oracle(c2, energy=True), rel=1e-4, label="after")
    tl.assert_parity(tl.run_gpu(c2, energy=False, fused=fused, nb=nb), tl.run_oracle(c2, energy=False), rel=1e-4, energy=False,
                     label="after, F only")
    nb.free()


def test_polling_finish_and_short_range_work_flags():
    """gpu_try_finish_task polls without blocking and reduces exactly once; setupGpuShortRangeWork / haveGpuShortRangeWork follow
    the lists and the listed-forces flag; the non-local dependency calls are accepted on both localities; gpu_get_fshift points at
    the shift forces the virial step accumulated."""
    import torch
    c = tl.make_case(elec="rf", seed=45, **SMALL)
    g = c.grid
    lib = pkg.hip_lib()
    nb = pkg.NbnxmGpu(tl.gpu_interaction_params(c), g.num_types, g.nbat_nbfp(c.sys["nbfp"]), local_and_nonlocal=True, fep=True, n_lambda=0)
    sig6 = c.sc_sigma ** 6
    nb.copy_fepparams(c.sc_alpha, c.sc_alpha, c.sc_power, sig6, sig6, c.lambda_coul, c.lambda_vdw)
    nb.init_atomdata(g.num_atoms, g.type, qA=g.qA, qB=g.qB, typeA=g.typeA, typeB=g.typeB)
    nb.init_pairlist(c.plist.sci, c.plist.cjPacked, c.plist.excl, iloc=pkg.LOCAL)
    nb.init_pairlist(c.plist.sci[:0], c.plist.cjPacked[:0], c.plist.excl[:1], iloc=pkg.NONLOCAL)     # nothing non-local
    empty = dict(iinr=np.zeros(0, np.int32), shift=np.zeros(0, np.int32), jindex=np.zeros(1, np.int32), jjnr=np.zeros(0, np.int32),
                 excl_fep=np.zeros(0, np.int32))
    nb.init_feppairlist(c.plist.fep, g.atomIndices, iloc=pkg.LOCAL)
    nb.init_feppairlist(empty, g.atomIndices, iloc=pkg.NONLOCAL)
    nb.upload_shiftvec(g.shift_vec)
    for iloc, want in ((pkg.LOCAL, 1), (pkg.NONLOCAL, 0)):
        nb.setup_short_range_work(False, iloc)
        assert lib.nbnxm_gpu_have_short_range_work(nb.h, iloc) == want
    nb.setup_short_range_work(True, pkg.NONLOCAL)          # listed forces on the GPU count as short-range work
    assert lib.nbnxm_gpu_have_short_range_work(nb.h, pkg.NONLOCAL) == 1
    nb.setup_short_range_work(False, pkg.NONLOCAL)
    sw = pkg.step_workload(energy=True, virial=True)
    nb.clear_outputs(True)
    nb.copy_xq_to_gpu(g.xq, pkg.LOCAL)
    nb.insert_nonlocal_dependency(pkg.LOCAL)
    nb.launch_kernel(sw, pkg.LOCAL)
    nb.insert_nonlocal_dependency(pkg.NONLOCAL)
    nb.launch_kernel(sw, pkg.NONLOCAL)                     # no work: returns without a launch
    f = np.zeros((g.num_atoms, 3), np.float32)
    nb.launch_cpyback(f, sw, pkg.NONLOCAL)
    nb.launch_cpyback(f, sw, pkg.LOCAL)
    polls, res = 0, None
    while res is None:
        res = nb.try_finish_task(sw, c.have_soft_core, pkg.LOCAL)
        polls += 1
        assert polls < 10_000_000
    dv = res["dvdl_nonlin"]
    got = dict(f=f.astype(np.float64), fshift=res["fshift"].astype(np.float64), e_lj=res["e_lj"], e_el=res["e_el"], dvdl_coul=dv[0],
               dvdl_vdw=dv[1])
    tl.assert_parity(got, tl.run_oracle(c, energy=True), rel=1e-4, label="polled")
    # gpu_get_fshift: the primary shift-force array (what the atom-pair and listed kernels add to); the cluster kernel's share
    # lies in accumulator slots behind it, and the host reduction above took the sum of all of them
    lib.nbnxm_gpu_get_fshift.restype = C.c_void_p
    block = np.zeros((33, 256), np.float32)
    lib.nbnxm_gpu_debug_download(nb.h, C.c_void_p(lib.nbnxm_gpu_get_fshift(nb.h)), block.ctypes.data_as(C.c_void_p), C.c_size_t(block.nbytes))
    assert np.allclose(block[:, :135].sum(axis=0).reshape(45, 3), res["fshift"], rtol=1e-5, atol=1e-3)
    assert np.abs(block[1:, :135]).sum() > 0 and not block[:, 135:].any()
    nb.free()


def _set_lambdas(nb, c, lam_q, lam_v):
    sig6 = c.sc_sigma ** 6
    nb.copy_fepparams(c.sc_alpha if c.sc_coul else 0.0, c.sc_alpha, c.sc_power, sig6, sig6 if c.sc_coul else 0.0, lam_q, lam_v, c.all_lambda,
                      c.all_lambda)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("elec", ["rf", "ewald"])
def test_free_energy_identities(elec, fused):
    """What the free-energy estimators rest on, from the kernels' own outputs (no oracle):
    thermodynamic integration — dV/dlambda (Coulomb and van der Waals separately) is the derivative of the energy the same
    kernels report when lambda moves; and BAR / MBAR — the foreign-lambda energy differences equal the change of the total energy
    when the simulation lambda itself is set to the foreign value."""
    c = tl.make_case(elec=elec, seed=61, n_lambda=11, nm=(8, 8, 8), num_perturbed_molecules=3)
    nb = tl.setup_gpu(c, fused=fused)

    def energy(lam_q, lam_v, dhdl=False):
        _set_lambdas(nb, c, lam_q, lam_v)
        r = tl.run_gpu(c, energy=True, fused=fused, dhdl=dhdl, nb=nb)
        return r["e_lj"] + r["e_el"], r

    lam, d = 0.5, 0.01
    _, base = energy(lam, lam, dhdl=True)
    for which in ("coul", "vdw"):
        ep, _ = energy(lam + d if which == "coul" else lam, lam + d if which == "vdw" else lam)
        em, _ = energy(lam - d if which == "coul" else lam, lam - d if which == "vdw" else lam)
        fd = (ep - em) / (2 * d)
        want = base["dvdl_" + which]
        assert abs(fd - want) <= 0.01 * max(abs(want), 1.0) + 0.5, (which, fd, want)     # 0.5: fp32 resolution of E / (2 d)
    e0 = base["e_lj"] + base["e_el"]
    fe = base["foreign"]["energies"]
    for k, lk in enumerate(c.all_lambda):
        ek, _ = energy(float(lk), float(lk))
        assert abs((fe[k + 1] - fe[0]) - (ek - e0)) <= 2e-3 * max(abs(fe[k + 1] - fe[0]), 1.0) + 0.02, (k, fe[k + 1] - fe[0], ek - e0)
    nb.free()


@pytest.mark.parametrize("seed,nm,npert", [(101, (7, 9, 11), 2), (102, (11, 7, 8), 5), (103, (6, 6, 13), 1), (104, (9, 12, 7), 8),
                                           (105, (8, 8, 8), 0), (106, (13, 6, 6), 3)])
def test_seed_and_shape_sweep(seed, nm, npert):
    """Boxes that are not cubes and not multiples of the cluster size (filler atoms, ragged last super-clusters, every shift
    pattern), 0 to 8 perturbed molecules, both perturbed-pair paths, Ewald and reaction field: forces, shift forces, energies and
    dV/dlambda against the oracle"""
    for elec in ("ewald", "rf"):
        c = tl.make_case(elec=elec, seed=seed, nm=nm, num_perturbed_molecules=npert)
        want = tl.run_oracle(c, energy=True)
        for fused in (False, True):
            tl.assert_parity(tl.run_gpu(c, energy=True, fused=fused), want, rel=1e-4, label="%s fused=%s VF" % (elec, fused))
            tl.assert_parity(tl.run_gpu(c, energy=False, fused=fused), want, rel=1e-4, energy=False, label="%s fused=%s F" % (elec, fused))


def test_two_lambda_windows_of_the_24k_box_in_one_object():
    """BASELINE configs[1] / [3] at size: two lambda windows of the 24,000-atom reaction-field box as ONE object (what a rank of the
    11-window set holds when there are fewer GPUs than windows) against the two windows run alone — forces per atom, energies and
    dV/dlambda per window."""
    import importlib
    replica = importlib.import_module("gromacs_fep_gpu_amd.replica")
    c = tl.make_case(elec="rf", seed=2026, nm=(20, 20, 20), num_perturbed_molecules=3, n_lambda=0, max_cjpacked_per_sci=16)
    g = c.grid
    lambdas = [(0.3, 0.3), (0.8, 0.8)]
    sw_e = pkg.step_workload(energy=True, virial=True, dhdl=False)

    def run(nb, natoms):
        nb.clear_outputs(True)
        nb.launch_kernel(sw_e)
        f = np.zeros((natoms, 3), np.float32)
        nb.launch_cpyback(f, sw_e)
        return f, nb.wait_finish_task(sw_e, c.have_soft_core)

    alone = []
    for lq, lv in lambdas:
        nb = tl.setup_gpu(c, fused=True)
        _set_lambdas(nb, c, lq, lv)
        alone.append(run(nb, g.num_atoms))
        nb.free()
    b = replica.batch_windows(g, c.plist_fused, len(lambdas))
    nb = pkg.NbnxmGpu(tl.gpu_interaction_params(c), g.num_types, g.nbat_nbfp(c.sys["nbfp"]), fep=True, n_lambda=0)
    _set_lambdas(nb, c, 0.0, 0.0)
    nb.init_atomdata(len(b["type"]), b["type"], qA=b["qA"], qB=b["qB"], typeA=b["typeA"], typeB=b["typeB"])
    nb.init_pairlist(b["sci"], b["cjPacked"], b["excl"])
    nb.init_fep_cluster_bits(b["fepBits"])
    nb.set_fep_mode(True)
    nb.set_window_lambdas(b["clusters_per_window"], [l[0] for l in lambdas], [l[1] for l in lambdas])
    nb.upload_shiftvec(g.shift_vec)
    nb.copy_xq_to_gpu(b["xq"])
    f, tot = run(nb, len(b["type"]))
    ns = b["slots_per_window"]
    for w, (fw, want) in enumerate(alone):
        got = f[w * ns:(w + 1) * ns].astype(np.float64)
        frms = np.sqrt(np.mean(np.sum(fw.astype(np.float64) ** 2, axis=1)))
        err = np.linalg.norm(got - fw, axis=1)
        assert (err <= 1e-4 * np.maximum(np.linalg.norm(fw, axis=1), frms)).all(), "window %d" % w
        res = nb.get_window_energies(w, c.have_soft_core)
        for k in ("e_lj", "e_el"):
            assert abs(res[k] - want[k]) <= 1e-4 * max(abs(want[k]), 1.0), (w, k, res[k], want[k])
        for k in range(2):
            scale = max(max(abs(v) for v in want["dvdl_nonlin"]), 1.0)
            assert abs(res["dvdl_nonlin"][k] - want["dvdl_nonlin"][k]) <= 1e-4 * scale, (w, k)
    assert np.max(np.abs(alone[0][0] - alone[1][0])) > 1.0          # the windows do differ
    assert abs(tot["e_el"] - (alone[0][1]["e_el"] + alone[1][1]["e_el"])) <= 1e-4 * abs(tot["e_el"])
    nb.free()


def test_lambda_windows_batched_into_one_object():
    """Three lambda windows of one system as ONE object (one list over 3 x N slots, per-window lambdas from a table): forces,
    energies, dV/dlambda and foreign-lambda terms of every window equal those of the window run alone with its own lambda and
    coordinates; the object's own accumulators hold the sums over the windows."""
    import importlib
    replica = importlib.import_module("gromacs_fep_gpu_amd.replica")
    c = tl.make_case(elec="ewald", seed=71, n_lambda=11, **SMALL)
    g = c.grid
    lambdas = [(0.15, 0.3), (0.5, 0.5), (0.9, 0.75)]              # (coulomb, van der Waals) per window
    rng = np.random.default_rng(12)
    real = g.atomIndices >= 0
    xqs = []
    for _ in lambdas:
        xq = g.xq.reshape(-1, 4).copy()
        xq[real, :3] += rng.normal(0, 0.004, (int(real.sum()), 3)).astype(np.float32)     # the windows have drifted apart
        xqs.append(xq)

    def run(nb, natoms, sw):
        nb.clear_outputs(True)
        nb.launch_kernel(sw)
        f = np.zeros((natoms, 3), np.float32)
        nb.launch_cpyback(f, sw)
        return f, nb.wait_finish_task(sw, c.have_soft_core)

    sw_f = pkg.step_workload(energy=False, virial=True, dhdl=False)
    sw_e = pkg.step_workload(energy=True, virial=True, dhdl=True)
    # every window alone
    alone = []
    for (lq, lv), xq in zip(lambdas, xqs):
        nb = tl.setup_gpu(c, fused=True)
        _set_lambdas(nb, c, lq, lv)
        nb.copy_xq_to_gpu(xq)
        f, res = run(nb, g.num_atoms, sw_f)
        _, res_e = run(nb, g.num_atoms, sw_e)
        alone.append((f, res["fshift"], res_e))
        nb.free()
    # all windows in one object
    b = replica.batch_windows(g, c.plist_fused, len(lambdas))
    nb = pkg.NbnxmGpu(tl.gpu_interaction_params(c), g.num_types, g.nbat_nbfp(c.sys["nbfp"]), fep=True, n_lambda=c.n_lambda)
    _set_lambdas(nb, c, 0.0, 0.0)                                 # the scalar lambdas are not used once the table is set
    nb.init_atomdata(len(b["type"]), b["type"], qA=b["qA"], qB=b["qB"], typeA=b["typeA"], typeB=b["typeB"])
    nb.init_pairlist(b["sci"], b["cjPacked"], b["excl"])
    nb.init_fep_cluster_bits(b["fepBits"])
    nb.set_fep_mode(True)
    nb.set_window_lambdas(b["clusters_per_window"], [l[0] for l in lambdas], [l[1] for l in lambdas])
    nb.upload_shiftvec(g.shift_vec)
    nb.copy_xq_to_gpu(np.concatenate(xqs))
    f, res = run(nb, len(b["type"]), sw_f)
    ns = b["slots_per_window"]
    for w, (fw, _, _) in enumerate(alone):
        got = f[w * ns:(w + 1) * ns].astype(np.float64)
        frms = np.sqrt(np.mean(np.sum(fw.astype(np.float64) ** 2, axis=1)))
        err = np.linalg.norm(got - fw, axis=1)
        assert (err <= 1e-4 * np.maximum(np.linalg.norm(fw, axis=1), frms)).all(), "window %d" % w
    fs_sum = sum(a[1].astype(np.float64) for a in alone)
    assert np.max(np.abs(res["fshift"] - fs_sum)) <= 1e-3 * max(1.0, np.abs(fs_sum).max())
    # the windows do differ: the same coordinates with another window's lambdas give other forces
    assert np.max(np.abs(alone[0][0] - alone[2][0])) > 1.0
    # energy + dH/dlambda step: per-window shares, and their sum in the object's accumulators
    _, tot = run(nb, len(b["type"]), sw_e)
    close = lambda a, b_, scale: abs(a - b_) <= 1e-4 * max(abs(scale), 1.0)
    sums = dict(e_lj=0.0, e_el=0.0)
    for w, (_, _, want) in enumerate(alone):
        got = nb.get_window_energies(w, c.have_soft_core)
        escale = abs(want["e_lj"]) + abs(want["e_el"])
        assert close(got["e_lj"], want["e_lj"], escale) and close(got["e_el"], want["e_el"], escale), w
        for k in range(2):
            assert close(got["dvdl_nonlin"][k], want["dvdl_nonlin"][k], max(abs(v) for v in want["dvdl_nonlin"])), (w, k)
        fscale = np.max(np.abs(want["foreign_energies"]))
        assert np.max(np.abs(got["foreign_energies"] - want["foreign_energies"])) <= 1e-4 * max(fscale, 1.0), w
        assert np.max(np.abs(got["foreign_dhdl_coul"] - want["foreign_dhdl_coul"])) <= 1e-4 * max(np.max(np.abs(want["foreign_dhdl_coul"])), 1.0)
        assert np.max(np.abs(got["foreign_dhdl_vdw"] - want["foreign_dhdl_vdw"])) <= 1e-4 * max(np.max(np.abs(want["foreign_dhdl_vdw"])), 1.0)
        sums["e_lj"] += want["e_lj"]
        sums["e_el"] += want["e_el"]
    assert close(tot["e_lj"], sums["e_lj"], abs(sums["e_lj"]) + abs(sums["e_el"])) and close(tot["e_el"], sums["e_el"], abs(sums["e_lj"]) + abs(sums["e_el"]))
    with pytest.raises(IndexError):
        nb.get_window_energies(len(lambdas), c.have_soft_core)
    nb.free()


# ---- round 2: BASELINE configs[1] at its size, two localities on force-only steps, slow-pair list of a live list ----
@pytest.mark.parametrize("fused", [False, True])
def test_config1_24k_atom_box_reaction_field(fused):
    """BASELINE configs[1] at full size: 24,000-atom box (2 x 2 x 2 stack of the 3,000-atom cell), 9 perturbed atoms,
    reaction-field electrostatics, lambda 0.5 — forces, energies, dV/dlambda and the 11 foreign lambdas against the oracle,
    force-only and energy flavours, on the dynamically pruned list the bench uses."""
    c = tl.make_case(elec="rf", seed=2026, nm=(20, 20, 20), num_perturbed_molecules=3, n_lambda=11, max_cjpacked_per_sci=16)
    assert c.natoms == 24000 and int(c.perturbed.sum()) == 9
    want = tl.run_oracle(c, energy=True, foreign=True)
    nb = tl.setup_gpu(c, fused=fused, use_dynamic_pruning=True)
    got_f = tl.run_gpu(c, energy=False, fused=fused, nb=nb)
    tl.assert_parity(got_f, want, rel=1e-4, energy=False, label="24k rf F")
    got = tl.run_gpu(c, energy=True, fused=fused, dhdl=True, nb=nb)
    tl.assert_parity(got, want, rel=1e-4, label="24k rf VF")
    tl.assert_foreign(got, want, rel=1e-4)
    got_f2 = tl.run_gpu(c, energy=False, fused=fused, nb=nb)      # after an energy step: the clear kernel instead of the swap
    tl.assert_parity(got_f2, want, rel=1e-4, energy=False, label="24k rf F again")
    got_e = tl.run_gpu(c, energy=True, fused=fused, nb=nb)          # an energy step without foreign lambdas: the energy flavours' own partition
    tl.assert_parity(got_e, want, rel=1e-4, label="24k rf VF, energy partition")
    # the three work partitions of this list (round 4): the force flavours' short-list one (four ranges per SIMD), and in fused mode the
    # energy flavours' with 15/16, on dH/dlambda steps with 3/4 of the wave slots — the trailing workgroups run beside the ranges there
    lib = pkg.hip_lib()
    lib.nbnxm_gpu_debug_get_work_ranges.restype = C.c_void_p
    counts = []
    for p in (0, 1, 2):
        n = C.c_int()
        assert lib.nbnxm_gpu_debug_get_work_ranges(nb.h, C.c_int(pkg.LOCAL), C.c_int(p), C.byref(n))
        counts.append(n.value)
    slots4 = counts[1]      # the force partition of a short list: four ranges on each of the 1,024 SIMDs = the energy flavours' wave slots
    assert slots4 % 1024 == 0
    assert counts == ([slots4 * 15 // 16, slots4, slots4 * 3 // 4] if fused else [slots4, slots4, slots4]), counts
    nb.free()


@pytest.mark.parametrize("elec", ["rf", "ewald"])
def test_the_same_perturbed_pairs_through_both_forms_24k(elec, monkeypatch):
    """What pins the fused perturbed path: the pair math, fepPair, is shared with the atom-pair kernels, which the
    reference's 72 known answers pin (test_gpu_fep_golden.py); the fused form's own part — the walk over the perturbed cluster pairs of
    the list, fepClusterPair — cannot be fed that 4-atom system (it is not a cluster list).  So here the SAME perturbed pairs of the 24k
    box go through all three forms — the fused cluster-pair form, the atom-pair list regrouped by cluster pair in the cluster kernel's
    tail (fepListClusterItem) and the atom-pair kernel of its own (nbnxmFepKernel, the reference's shape) — and have to agree at 2e-5
    per atom, five times tighter than the parity bar against the oracle; energies and dV/dlambda at 2e-5 of their terms' magnitude."""
    c = tl.make_case(elec=elec, seed=2026, nm=(20, 20, 20), num_perturbed_molecules=3, n_lambda=11, max_cjpacked_per_sci=16)
    want = tl.run_oracle(c, energy=True)
    res = {}
    res["fused"] = tl.run_gpu(c, energy=True, fused=True)
    res["list in the tail"] = tl.run_gpu(c, energy=True, fused=False)
    monkeypatch.setenv("NBNXM_HIP_DIAGNOSTICS", "1")
    monkeypatch.setenv("NBNXM_HIP_FEP_LIST_MERGED", "0")
    res["atom-pair kernel"] = tl.run_gpu(c, energy=True, fused=False)
    ref = res["atom-pair kernel"]
    frms = math.sqrt(float(np.mean(np.sum(ref["f"] ** 2, axis=1))))
    pert_slots = np.flatnonzero(np.isin(c.grid.atomIndices, np.flatnonzero(c.perturbed)))
    assert len(pert_slots) == 9
    for name in ("fused", "list in the tail"):
        got = res[name]
        err = np.sqrt(np.sum((got["f"] - ref["f"]) ** 2, axis=1))
        tol = 2e-5 * np.maximum(np.sqrt(np.sum(ref["f"] ** 2, axis=1)), frms)
        worst = int(np.argmax(err / tol))
        assert err[worst] <= tol[worst], "%s vs atom-pair kernel: atom slot %d differs by %.3e (|f| %.3e)" % (name, worst, err[worst], np.linalg.norm(ref["f"][worst]))
        # the perturbed atoms themselves: every one of their pairs went through the form under test
        assert np.all(err[pert_slots] <= tol[pert_slots])
        for k in ("e_lj", "e_el"):
            assert abs(got[k] - ref[k]) <= 2e-5 * max(abs(ref[k]), 1.0), (name, k, got[k], ref[k])
        for k in ("dvdl_coul", "dvdl_vdw"):
            assert abs(got[k] - ref[k]) <= 2e-5 * max(abs(ref[k]), want["fep_abs_sums"][k], 1e-3), (name, k, got[k], ref[k])
    tl.assert_parity(res["fused"], want, rel=1e-4, label="24k %s fused" % elec)


@pytest.mark.parametrize("fused", [False, True])
def test_two_localities_force_only_sequence(fused):
    """F, F, VF, F, F with a Local and a NonLocal list on their two streams, copy-back and clear every step as mdrun does
    (sim_util.cpp:1914-1924, 2303-2320): the force-only kernels zero the spare force buffer in their trailing workgroups and
    nbnxm_gpu_clear_outputs swaps — a buffer zeroed twice, too early or not at all shows as missing or doubled forces."""
    c = tl.make_case(elec="ewald", seed=47, **SMALL)
    g = c.grid
    nb = pkg.NbnxmGpu(tl.gpu_interaction_params(c), g.num_types, g.nbat_nbfp(c.sys["nbfp"]), local_and_nonlocal=True,
                      fep=True, n_lambda=0)
    sig6 = c.sc_sigma ** 6
    nb.copy_fepparams(c.sc_alpha, c.sc_alpha, c.sc_power, sig6, sig6, c.lambda_coul, c.lambda_vdw)
    nb.init_atomdata(g.num_atoms, g.type, qA=g.qA, qB=g.qB, typeA=g.typeA, typeB=g.typeB)
    pl = c.plist_fused if fused else c.plist
    # the i-entries dealt to the two localities in blocks, so that both have perturbed cluster pairs
    half = len(pl.sci) // 2
    nb.init_pairlist(pl.sci[:half], pl.cjPacked, pl.excl, iloc=pkg.LOCAL)
    nb.init_pairlist(pl.sci[half:], pl.cjPacked, pl.excl, iloc=pkg.NONLOCAL)
    if fused:
        nb.init_fep_cluster_bits(g.fepBits)
        nb.set_fep_mode(True)
    else:
        empty = dict(iinr=np.zeros(0, np.int32), shift=np.zeros(0, np.int32), jindex=np.zeros(1, np.int32),
                     jjnr=np.zeros(0, np.int32), excl_fep=np.zeros(0, np.int32))
        nb.init_feppairlist(c.plist.fep, g.atomIndices, iloc=pkg.LOCAL)
        nb.init_feppairlist(empty, g.atomIndices, iloc=pkg.NONLOCAL)
    nb.upload_shiftvec(g.shift_vec)
    want = tl.run_oracle(c, energy=True)
    nb.clear_outputs(True)
    for step, energy in enumerate((False, False, True, False, False)):
        sw = pkg.step_workload(energy=energy, virial=energy)
        nb.copy_xq_to_gpu(g.xq, pkg.LOCAL)
        nb.launch_kernel(sw, pkg.LOCAL)
        nb.copy_xq_to_gpu(g.xq, pkg.NONLOCAL)
        nb.launch_kernel(sw, pkg.NONLOCAL)
        f = np.zeros((g.num_atoms, 3), np.float32)
        nb.launch_cpyback(f, sw, pkg.NONLOCAL)
        nb.launch_cpyback(f, sw, pkg.LOCAL)
        nb.wait_finish_task(sw, c.have_soft_core, pkg.NONLOCAL)
        res = nb.wait_finish_task(sw, c.have_soft_core, pkg.LOCAL)
        nb.clear_outputs(energy)
        dv = res["dvdl_nonlin"]
        got = dict(f=f.astype(np.float64), fshift=res["fshift"].astype(np.float64), e_lj=res["e_lj"], e_el=res["e_el"],
                   dvdl_coul=dv[0], dvdl_vdw=dv[1])
        if not energy:
            got["fshift"] = want["fshift"]        # shift forces are only produced on virial steps
        tl.assert_parity(got, want, rel=1e-4, energy=energy, label="two localities, step %d" % step)
    nb.free()


def test_merged_localities_with_the_plain_call_sequence_return_the_halo_forces():
    """nbnxm_gpu_set_merged_localities with the calls of the reference's sequence (advisor finding of round 3): the non-local list is
    appended to the local device list, ONE launch evaluates both, and nbnxm_gpu_launch_cpyback(LOCAL) has to return the forces of ALL
    atoms — with round 3's code it copied [0, numAtomsLocal) only and the non-local copy-back returned early, so the forces on the halo
    atoms never reached the caller.  Local = the first half of the i-entries, NonLocal = the second, numAtomsLocal = half the atoms."""
    c = tl.make_case(elec="ewald", seed=49, **SMALL)
    g = c.grid
    nb = pkg.NbnxmGpu(tl.gpu_interaction_params(c), g.num_types, g.nbat_nbfp(c.sys["nbfp"]), local_and_nonlocal=True, fep=True, n_lambda=0)
    sig6 = c.sc_sigma ** 6
    nb.copy_fepparams(c.sc_alpha, c.sc_alpha, c.sc_power, sig6, sig6, c.lambda_coul, c.lambda_vdw)
    nb.set_merged_localities(True)
    num_local = (g.num_atoms // 2) // 64 * 64
    nb.init_atomdata(g.num_atoms, g.type, qA=g.qA, qB=g.qB, typeA=g.typeA, typeB=g.typeB, num_atoms_local=num_local)
    pl = c.plist_fused
    half = len(pl.sci) // 2
    nb.init_pairlist(pl.sci[:half], pl.cjPacked, pl.excl, iloc=pkg.LOCAL)
    nb.init_pairlist(pl.sci[half:], pl.cjPacked, pl.excl, iloc=pkg.NONLOCAL)
    nb.init_fep_cluster_bits(g.fepBits)
    nb.set_fep_mode(True)
    nb.upload_shiftvec(g.shift_vec)
    want = tl.run_oracle(c, energy=True)
    nb.clear_outputs(True)
    for step, energy in enumerate((False, True, False)):
        sw = pkg.step_workload(energy=energy, virial=energy)
        nb.copy_xq_to_gpu(g.xq, pkg.LOCAL)
        nb.copy_xq_to_gpu(g.xq, pkg.NONLOCAL)
        import torch
        torch.cuda.synchronize()                   # (the halo coordinates arrive on the non-local stream: the caller orders the launch behind them)
        nb.launch_kernel(sw, pkg.LOCAL)            # the one kernel of the step (all coordinates are in place)
        nb.launch_kernel(sw, pkg.NONLOCAL)         # empty device list: nothing
        f = np.zeros((g.num_atoms, 3), np.float32)
        nb.launch_cpyback(f, sw, pkg.NONLOCAL)     # nothing
        nb.launch_cpyback(f, sw, pkg.LOCAL)        # all atoms
        nb.wait_finish_task(sw, c.have_soft_core, pkg.NONLOCAL)
        res = nb.wait_finish_task(sw, c.have_soft_core, pkg.LOCAL)
        nb.clear_outputs(energy)
        assert np.any(f[num_local:] != 0.0), "no force on any atom behind numAtomsLocal"
        dv = res["dvdl_nonlin"]
        got = dict(f=f.astype(np.float64), fshift=res["fshift"].astype(np.float64) if energy else want["fshift"], e_lj=res["e_lj"], e_el=res["e_el"],
                   dvdl_coul=dv[0], dvdl_vdw=dv[1])
        tl.assert_parity(got, want, rel=1e-4, energy=energy, label="merged localities, plain sequence, step %d" % step)
    nb.free()


def test_slow_pair_list_rebuilt_on_a_pruned_list_keeps_pairs_the_rolling_prune_brings_back():
    """nbnxm_gpu_init_fep_cluster_bits / nbnxm_gpu_set_fep_mode on a LIVE list (after its first prune) rebuild the list of
    perturbed cluster pairs; it has to come from the outer-pruned masks, because the rolling prune re-adds pairs from those.
    Atoms move so that perturbed cluster pairs come inside the inner radius after the rebuild; forces against the oracle."""
    import oracle_binding as ob
    c = tl.make_case(elec="ewald", seed=35, nm=(10, 10, 10), num_perturbed_molecules=40)
    c.rlist_inner = c.rc
    g = c.grid
    pl = c.plist_fused
    nb = tl.setup_gpu(c, fused=True, use_dynamic_pruning=True)
    tl.run_gpu(c, energy=False, fused=True, nb=nb)              # first launch: first-pass prune (outer and inner masks)
    nb.init_fep_cluster_bits(g.fepBits)                         # on the pruned list: the slow-pair list is rebuilt
    nb.set_fep_mode(True)
    outer = pl.cjPacked.copy()
    ob.nbnxm_prune(pl.sci, outer, g.xq, g.shift_vec, c.rlist)
    inner = outer.copy()
    ob.nbnxm_prune(pl.sci, inner, g.xq, g.shift_vec, c.rlist_inner)
    rng = np.random.default_rng(6)
    xq_new = g.xq.reshape(-1, 4).copy()
    xq_new[:, :3] += rng.normal(0.0, 0.02, size=(len(xq_new), 3)).astype(np.float32) * (g.atomIndices >= 0)[:, None]
    came_in = outer.copy()
    ob.nbnxm_prune(pl.sci, came_in, xq_new, g.shift_vec, c.rlist_inner)
    new_bits = came_in["imei"]["imask"][:, 0] & ~inner["imei"]["imask"][:, 0]
    # how many of the cluster pairs that come back touch a perturbed cluster
    group_sci = np.zeros(len(pl.cjPacked), np.int64)
    for e in pl.sci:
        group_sci[e["cjPackedBegin"]:e["cjPackedEnd"]] = e["sci"]
    n_slow_back = 0
    for gi in np.flatnonzero(new_bits):
        for bit in range(32):
            if (int(new_bits[gi]) >> bit) & 1:
                ci = group_sci[gi] * 8 + (bit & 7)
                cj = pl.cjPacked["cj"][gi][bit >> 3]
                n_slow_back += int(g.fepBits[ci] != 0 or g.fepBits[cj] != 0)
    assert n_slow_back >= 5, "the test does not move perturbed cluster pairs across the inner radius"
    nb.copy_xq_to_gpu(xq_new)
    xq_old, xw_old = g.xq, g.x_wrapped
    xw = g.x_wrapped.copy()
    real = g.atomIndices >= 0
    xw[g.atomIndices[real]] = xq_new[real, :3]
    g.xq, g.x_wrapped = xq_new, xw
    try:
        want = tl.run_oracle(c, energy=True)
    finally:
        g.xq, g.x_wrapped = xq_old, xw_old
    for _ in range(4):
        nb.launch_kernel_pruneonly(num_parts=4)
        tl.run_gpu(c, energy=False, fused=True, nb=nb)
    got = tl.run_gpu(c, energy=True, fused=True, nb=nb)          # every part has been through the rolling pass
    tl.assert_parity(got, want, rel=1e-4, label="slow pairs after rolling prune")
    dev = pkg.download_cjpacked(nb, len(outer))
    assert np.array_equal(dev["imei"]["imask"], inner["imei"]["imask"] | came_in["imei"]["imask"])
    nb.free()
