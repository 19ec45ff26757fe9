"""CPU tests of the host-side producers and of the oracle's internal consistency.

 * the cluster pair list covers every atom pair within rlist exactly once;
 * nbnxm_ref (GPU-layout list walk) == O(N^2) minimum-image evaluation;
 * carved list + FEP list reproduce the uncarved list when A == B (this transfers the golden
   pin of fep_oracle.c to nbnxm_ref.c for cut-off/RF/Ewald + LJ cut/pot-switch).
"""
import numpy as np
import pytest

import fep_testlib as tl
import oracle_binding as ob

pkg = tl.pkg


def expand_pairs(c, pl):
    """All (grid i, shift, grid j) atom pairs a list evaluates (imask bit set, exclusion bit set)."""
    pairs = []
    for e in pl.sci:
        for jp in range(e["cjPackedBegin"], e["cjPackedEnd"]):
            grp = pl.cjPacked[jp]
            for jm in range(4):
                for im in range(8):
                    if not (int(grp["imei"][0]["imask"]) >> (jm * 8 + im)) & 1:
                        continue
                    ci, cj = e["sci"] * 8 + im, int(grp["cj"][jm])
                    for jc in range(8):
                        w = pl.excl[grp["imei"][jc // 4]["excl_ind"]]["pair"]
                        for ic in range(8):
                            if (int(w[(jc & 3) * 8 + ic]) >> (jm * 8 + im)) & 1:
                                pairs.append((ci * 8 + ic, int(e["shift"]), cj * 8 + jc))
    return pairs


@pytest.fixture(scope="module")
def tiny():
    # 8x8x8 molecules = 1536 atoms, box 2.49 nm: the smallest box that holds rlist 1.1 twice
    return tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=0, elec="rf", seed=11)


def test_grid_is_a_permutation(tiny):
    g = tiny.grid
    real = g.atomIndices[g.atomIndices >= 0]
    assert sorted(real.tolist()) == list(range(tiny.natoms))
    assert g.num_atoms % 64 == 0
    # fillers are far away and carry no charge / the zero type
    fill = g.atomIndices < 0
    assert np.all(g.xq[fill, 3] == 0) and np.all(g.type[fill] == tiny.ntype)
    assert np.all(g.xq[fill, 0] < -1e4)


def test_cluster_list_covers_every_pair_within_rlist_once(tiny):
    c = tiny
    g = c.grid
    pairs = expand_pairs(c, c.plist_fused)
    seen = {}
    xq = g.xq[:, :3].astype(np.float64)
    sv = g.shift_vec.astype(np.float64)
    for gi, s, gj in pairs:
        ai, aj = g.atomIndices[gi], g.atomIndices[gj]
        if ai < 0 or aj < 0:
            continue
        d = xq[gi] + sv[s] - xq[gj]
        r2 = float(d @ d)
        if r2 < c.rlist ** 2:
            key = (min(ai, aj), max(ai, aj))
            seen[key] = seen.get(key, 0) + 1
    # brute force set of non-excluded pairs within rlist
    x = g.x_wrapped.astype(np.float64)
    box = g.box.astype(np.float64)
    want = set()
    for i in range(c.natoms - 1):
        d = x[i] - x[i + 1:]
        d -= box * np.rint(d / box)
        r2 = np.einsum("ij,ij->i", d, d)
        for j in np.flatnonzero(r2 < c.rlist ** 2) + i + 1:
            if c.sys["molId"][i] != c.sys["molId"][j]:
                want.add((i, int(j)))
    assert set(seen) == want
    assert max(seen.values()) == 1


@pytest.mark.parametrize("elec,vdw", [("rf", "cut"), ("ewald", "cut"), ("cut", "cut"), ("rf", "pswitch"),
                                      ("ewald", "fswitch"), ("ewald", "ewald_geom"), ("ewald", "ewald_lb"),
                                      ("rf", "ewald_geom")])
def test_nbnxm_ref_matches_all_pairs(elec, vdw):
    # (LJ-PME cases: several oxygen types, so that the grid C6 of a pair is not the table's C6 with Lorentz-Berthelot)
    c = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=0, elec=elec, vdw=vdw, seed=5,
                     num_extra_types=3 if vdw.startswith("ewald") else 0)
    g = c.grid
    ref = ob.nbnxm_ref(c.plist_fused.sci, c.plist_fused.cjPacked, c.plist_fused.excl, g.xq, g.type, g.num_types,
                       g.nbat_nbfp(c.sys["nbfp"]), tl.oracle_ref_params(c), g.shift_vec,
                       nbfp_comb=tl.lj_type_params(c) if vdw.startswith("ewald") else None)
    bf = tl.brute_force(c)
    real = g.atomIndices >= 0
    f = np.zeros((c.natoms, 3))
    f[g.atomIndices[real]] = ref["f"][real]
    scale = np.sqrt(np.mean(bf["f"] ** 2))
    # float32 coordinates, double arithmetic on both sides: agreement to round-off of the sums
    assert np.max(np.abs(f - bf["f"])) < 1e-9 * scale
    assert abs(ref["Vv"] - bf["e_lj"]) < 1e-9 * max(1.0, abs(bf["e_lj"]))
    assert abs(ref["Vc"] - bf["e_el"]) < 1e-9 * max(1.0, abs(bf["e_el"]))
    assert np.all(ref["f"][~real] == 0)


@pytest.mark.parametrize("elec,vdw", [("ewald", "cut"), ("ewald_tab", "pswitch"), ("ewald", "fswitch"), ("ewald", "ewald_lb")])
def test_nbnxm_ref_twin_range_matches_all_pairs(elec, vdw):
    # rvdw < rcoulomb: the *_TWIN electrostatics flavours (LJ only inside rvdw, its shift / switch constants built on rvdw)
    c = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=0, elec=elec, vdw=vdw, seed=6, rvdw=0.85, rvdw_switch=0.7,
                     num_extra_types=3 if vdw.startswith("ewald") else 0)
    g = c.grid
    ref = ob.nbnxm_ref(c.plist_fused.sci, c.plist_fused.cjPacked, c.plist_fused.excl, g.xq, g.type, g.num_types,
                       g.nbat_nbfp(c.sys["nbfp"]), tl.oracle_ref_params(c, analytical_ewald=True), g.shift_vec,
                       nbfp_comb=tl.lj_type_params(c) if vdw.startswith("ewald") else None)
    bf = tl.brute_force(c)
    real = g.atomIndices >= 0
    f = np.zeros((c.natoms, 3))
    f[g.atomIndices[real]] = ref["f"][real]
    scale = np.sqrt(np.mean(bf["f"] ** 2))
    assert np.max(np.abs(f - bf["f"])) < 1e-9 * scale
    assert abs(ref["Vv"] - bf["e_lj"]) < 1e-9 * max(1.0, abs(bf["e_lj"]))
    assert abs(ref["Vc"] - bf["e_el"]) < 1e-9 * max(1.0, abs(bf["e_el"]))
    if elec == "ewald_tab":
        # the force table of the tabulated flavours (2000 points per nm, linear interpolation as kernel_gpu_ref.cpp:265-271)
        # stays within 3e-4 of the closed form; energies do not use it
        tab = ob.nbnxm_ref(c.plist_fused.sci, c.plist_fused.cjPacked, c.plist_fused.excl, g.xq, g.type, g.num_types,
                           g.nbat_nbfp(c.sys["nbfp"]), tl.oracle_ref_params(c), g.shift_vec)
        err = np.max(np.abs(tab["f"] - ref["f"]))
        assert 0 < err < 3e-4 * scale
        assert tab["Vc"] == ref["Vc"]


@pytest.mark.parametrize("vdw", ["ewald_geom", "ewald_lb"])
def test_lj_pme_carved_plus_fep_list_equals_uncarved_when_states_are_identical(vdw):
    # the LJ-PME real-space term of the cluster kernel (nbnxm_ref.c) and the grid correction of the perturbed-pair kernel
    # (fep_oracle.c, pinned by the reference's LJ-PME known answers) are two restatements of one function
    c = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=4, elec="ewald", vdw=vdw, seed=3, sc_alpha=0.0,
                     identical_states=True, lambda_coul=0.3, lambda_vdw=0.7, num_extra_types=2)
    g = c.grid
    split = tl.run_oracle(c, energy=True)
    xq_full = g.xq.copy()
    xq_full[:, 3] = g.qA
    full = ob.nbnxm_ref(c.plist_fused.sci, c.plist_fused.cjPacked, c.plist_fused.excl, xq_full, g.typeA,
                        g.num_types, g.nbat_nbfp(c.sys["nbfp"]), tl.oracle_ref_params(c), g.shift_vec,
                        nbfp_comb=tl.lj_type_params(c))
    scale = np.sqrt(np.mean(full["f"] ** 2))
    assert np.max(np.abs(split["f"] - full["f"])) < 1e-9 * scale
    assert abs(split["e_lj"] - full["Vv"]) < 1e-9 * max(1.0, abs(full["Vv"]))
    assert abs(split["e_el"] - full["Vc"]) < 1e-8 * max(1.0, abs(full["Vc"]))
    assert abs(split["dvdl_vdw"]) < 1e-8 * max(1.0, abs(full["Vv"]))


def test_twin_range_carved_plus_fep_list_equals_uncarved_when_states_are_identical():
    c = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=4, elec="ewald", vdw="cut", seed=3, sc_alpha=0.0,
                     identical_states=True, lambda_coul=0.3, lambda_vdw=0.7, rvdw=0.85)
    g = c.grid
    split = tl.run_oracle(c, energy=True)
    xq_full = g.xq.copy()
    xq_full[:, 3] = g.qA
    full = ob.nbnxm_ref(c.plist_fused.sci, c.plist_fused.cjPacked, c.plist_fused.excl, xq_full, g.typeA,
                        g.num_types, g.nbat_nbfp(c.sys["nbfp"]), tl.oracle_ref_params(c), g.shift_vec)
    scale = np.sqrt(np.mean(full["f"] ** 2))
    assert np.max(np.abs(split["f"] - full["f"])) < 1e-9 * scale
    assert abs(split["e_lj"] - full["Vv"]) < 1e-9 * max(1.0, abs(full["Vv"]))
    assert abs(split["e_el"] - full["Vc"]) < 1e-9 * max(1.0, abs(full["Vc"]))


@pytest.mark.parametrize("elec,vdw", [("rf", "cut"), ("ewald", "cut"), ("cut", "cut"), ("rf", "pswitch")])
def test_carved_plus_fep_list_equals_uncarved_when_states_are_identical(elec, vdw):
    # perturbed flags set on 4 molecules, but B == A and no soft-core: the split evaluation must
    # reproduce the plain cluster kernel on the full list
    c = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=4, elec=elec, vdw=vdw, seed=3, sc_alpha=0.0,
                     identical_states=True, lambda_coul=0.3, lambda_vdw=0.7)
    g = c.grid
    assert len(c.plist.fep["jjnr"]) > 0
    split = tl.run_oracle(c, energy=True)
    # full list with UNMASKED parameters
    xq_full = g.xq.copy()
    xq_full[:, 3] = g.qA
    full = ob.nbnxm_ref(c.plist_fused.sci, c.plist_fused.cjPacked, c.plist_fused.excl, xq_full, g.typeA,
                        g.num_types, g.nbat_nbfp(c.sys["nbfp"]), tl.oracle_ref_params(c), g.shift_vec)
    scale = np.sqrt(np.mean(full["f"] ** 2))
    assert np.max(np.abs(split["f"] - full["f"])) < 1e-9 * scale
    assert abs(split["e_lj"] - full["Vv"]) < 1e-9 * max(1.0, abs(full["Vv"]))
    assert abs(split["e_el"] - full["Vc"]) < 1e-8 * max(1.0, abs(full["Vc"]))
    assert np.max(np.abs(split["fshift"] - full["fshift"])) < 1e-8 * max(1.0, np.max(np.abs(full["fshift"])))
    # B == A: dV/dlambda vanishes identically
    assert abs(split["dvdl_coul"]) < 1e-8 * max(1.0, abs(full["Vc"]))
    assert abs(split["dvdl_vdw"]) < 1e-8 * max(1.0, abs(full["Vv"]))


def test_fep_list_shape_follows_the_reference_rules():
    c = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=3, elec="rf", seed=9)
    fep = c.plist.fep
    nj = np.diff(fep["jindex"])
    assert nj.min() >= 1 and nj.max() <= 64              # pairlist.cpp:1509 cap
    # every pair holds at least one perturbed atom; self pairs are present and flagged excluded
    pert = c.perturbed
    ii = np.repeat(fep["iinr"], nj)
    assert np.all(pert[ii] | pert[fep["jjnr"]])
    self_pairs = ii == fep["jjnr"]
    assert self_pairs.sum() == pert.sum()
    assert np.all(fep["excl_fep"][self_pairs] == 0)
    # carving removed the perturbed pairs from the cluster list: no pair appears in both
    g = c.grid
    for a, s, b in expand_pairs(c, c.plist):
        ai, aj = g.atomIndices[a], g.atomIndices[b]
        if ai >= 0 and aj >= 0:
            assert not (pert[ai] or pert[aj])


def test_sci_splitting_keeps_the_list_content():
    a = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=2, elec="rf", seed=4)
    b = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=2, elec="rf", seed=4, max_cjpacked_per_sci=4)
    assert len(b.plist.sci) > len(a.plist.sci)
    assert np.max(b.plist.sci["cjPackedEnd"] - b.plist.sci["cjPackedBegin"]) <= 4
    ra, rb = tl.run_oracle(a), tl.run_oracle(b)
    assert np.allclose(ra["f"], rb["f"], rtol=0, atol=1e-9 * np.sqrt(np.mean(ra["f"] ** 2)))
    assert abs(ra["e_el"] - rb["e_el"]) < 1e-9 * abs(ra["e_el"])


@pytest.mark.parametrize("elec", ["ewald", "rf", "cut"])
def test_simd_port_of_the_cluster_kernel_matches_the_scalar_oracle(elec):
    """oracle/nbnxm_simd.c (the CPU baseline of bench.py) against the scalar parity oracle on the same list: forces to fp32
    round-off plus the 8e-7 relative error of the rational Ewald correction it shares with the GPU kernels"""
    import oracle_binding as ob
    c = tl.make_case(nm=(10, 10, 10), num_perturbed_molecules=3, elec=elec, seed=33)
    g = c.grid
    p = tl.oracle_ref_params(c)
    nbfp = g.nbat_nbfp(c.sys["nbfp"])
    want = ob.nbnxm_ref(c.plist.sci, c.plist.cjPacked, c.plist.excl, g.xq, g.type, g.num_types, nbfp, p, g.shift_vec, compute_energy=False,
                        compute_fshift=False, precision="f64")["f"]
    for threads in (1, 3):
        got = ob.nbnxm_simd(c.plist.sci, c.plist.cjPacked, c.plist.excl, g.xq, g.type, g.num_types, nbfp, p, g.shift_vec, num_threads=threads)
        if got is None:
            pytest.skip("no AVX2 + FMA on this CPU")
        frms = np.sqrt(np.mean(np.sum(want ** 2, axis=1)))
        err = np.linalg.norm(got - want, axis=1)
        assert (err <= 2e-5 * np.maximum(np.linalg.norm(want, axis=1), frms)).all()
