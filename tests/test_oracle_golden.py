"""Pins the CPU oracle (oracle/fep_oracle.c) against the reference's own 72 known answers.

Fixture: tests/golden/nb_fep_refdata.json, transcribed from
/root/reference/src/gromacs/gmxlib/nonbonded/tests/refdata/NBInteraction_NonbondedFepTest_testKernel_*.xml
by tests/golden/make_nb_fep_golden.py.  Inputs as in gmxlib/nonbonded/tests/nb_free_energy.cpp.
Tolerance: the reference's own (1e-6 relative in float, 1e-8 in double, nb_free_energy.cpp:504-506)
with the same absolute floor idea as its FloatingPointTolerance (1e-6 / 1e-11 absolute).
"""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "nb_fep_refdata.json")) as fh:
    GOLDEN = json.load(fh)
SYS = GOLDEN["system"]


def build_inputs(case):
    s = SYS
    ntype = s["ntype"]
    lj = np.array(s["lj_c6_c12"], dtype=np.float64)
    nbfp = np.empty(2 * ntype * ntype)
    nbfp[0::2] = 6.0 * lj[:, 0]    # makeNonBondedParameterLists, mdlib/forcerec.cpp:115-152
    nbfp[1::2] = 12.0 * lj[:, 1]
    grid = np.zeros(2 * ntype * ntype)  # makeLJPmeC6GridCorrectionParameters, forcerec.cpp:154-190 (Geom)
    for i in range(ntype):
        for j in range(ntype):
            c6i = lj[i * (ntype + 1), 0]
            c6j = lj[j * (ntype + 1), 0]
            grid[2 * (ntype * i + j)] = 6.0 * np.sqrt(c6i * c6j)
    p = ob.FepParams()
    inter = case["interaction"]
    p.elecIsEwald = 1 if inter["coulomb"] == "Pme" else 0
    p.vdwIsEwald = 1 if inter["vdw"] == "Pme" else 0
    p.vdwPotSwitch = 1 if inter["vdw_modifier"] == "PotSwitch" else 0
    p.epsfac = ob.ONE_4PI_EPS0 * s["epsfac_factor_of_one4pieps0"]
    p.rcoulomb, p.rvdw, p.rvdw_switch = s["rcoulomb"], s["rvdw"], s["rvdw_switch"]
    p.k_rf, p.c_rf = s["k_rf"], s["c_rf"]
    p.ewaldcoeff_q = ob.lib().oracle_calc_ewaldcoeff_q(s["ewald_rc"], s["ewald_rtol"])
    p.ewaldcoeff_lj = ob.lib().oracle_calc_ewaldcoeff_lj(s["ewald_rc"], s["ewald_rtol"])
    p.sh_ewald, p.sh_lj_ewald = s["sh_ewald"], s["sh_lj_ewald"]
    p.dispersion_shift_cpot = s["dispersion_shift_cpot"]
    p.repulsion_shift_cpot = s["repulsion_shift_cpot"]
    sc_type = ob.SOFTCORE_BEUTLER if case["softcore"] == "Beutler" else ob.SOFTCORE_GAPSYS
    # ForcerecHelper::setSoftcoreAlpha sets sc_alpha and both Gapsys linpoint scalings (nb_free_energy.cpp:254-259)
    a = case["sc_alpha"]
    ob.softcore_params(p, a, s["sc_power"], s["sc_sigma"], s["sc_sigma_min"], case["sc_coul"], sc_type,
                       gapsys_lj=a, gapsys_q=a, gapsys_sigma=s["gapsys_sigma_lj"])
    nbl = dict(iinr=s["iinr"], jindex=s["jindex"], jjnr=s["jjnr"], shift=s["shift"], excl_fep=s["excl_fep"])
    return nbl, nbfp, grid, p


def run_case(case, precision):
    nbl, nbfp, grid, p = build_inputs(case)
    lam = case["lambda"]
    return ob.fep_kernel(nbl, SYS["x"], SYS["ntype"], p, SYS["shiftvec"], nbfp, grid, SYS["chargeA"],
                         SYS["chargeB"], SYS["typeA"], SYS["typeB"],
                         ob.DO_FORCE | ob.DO_SHIFTFORCE | ob.DO_POTENTIAL, lam, lam, precision)


def check(got, want, rel, abs_):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    tol = np.maximum(abs_, rel * np.abs(want))
    assert np.all(err <= tol), "got %s want %s (err %s tol %s)" % (got, want, err, tol)


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: "k%02d" % c["index"])
def test_oracle_f64_matches_reference_known_answers(case):
    out = run_case(case, "f64")
    exp = case["expected"]
    # double build of the reference: 1e-8 relative (nb_free_energy.cpp:374,504); the Ewald cases were
    # generated with the rational erf approximations (accuracy 1e-11..4e-11 of libm erf, simd_math.h).
    rel, abs_ = 1e-8, 1e-9
    check(out["Vv"], exp["EVdw"], rel, abs_)
    check(out["Vc"], exp["ECoul"], rel, abs_)
    check(out["dvdl_coul"], exp["dVdlCoul"], rel, abs_)
    check(out["dvdl_vdw"], exp["dVdlVdw"], rel, abs_)
    fscale = max(1.0, float(np.max(np.abs(exp["Forces"]))))
    check(out["f"], exp["Forces"], rel, abs_ * fscale)
    check(out["fshift"][0], exp["ShiftForceCentral"], rel, abs_ * fscale)


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: "k%02d" % c["index"])
def test_oracle_f32_matches_reference_known_answers(case):
    out = run_case(case, "f32")
    exp = case["expected"]
    # float build of the reference: FloatingPointTolerance(1e-6 rel, ..., 10000 ULP) (nb_free_energy.cpp:433-435),
    # i.e. up to ~6e-4 relative in float; the inputs (0.1 nm separations in float, r^-12) limit an
    # independent float evaluation to ~1e-5, which is what is asserted here.
    scale = max(1.0, abs(exp["EVdw"]), abs(exp["ECoul"]), abs(exp["dVdlCoul"]), abs(exp["dVdlVdw"]))
    rel, abs_ = 1e-5, 1e-5 * scale
    check(out["Vv"], exp["EVdw"], rel, abs_)
    check(out["Vc"], exp["ECoul"], rel, abs_)
    check(out["dvdl_coul"], exp["dVdlCoul"], rel, abs_)
    check(out["dvdl_vdw"], exp["dVdlVdw"], rel, abs_)
    fscale = max(1.0, float(np.max(np.abs(exp["Forces"]))))
    check(out["f"], exp["Forces"], rel, 1e-5 * fscale)
    check(out["fshift"][0], exp["ShiftForceCentral"], rel, 1e-5 * fscale)


# ---- listed (bonded) interactions: oracle/listed_ref.c against the reference's known answers ----------------
def _listed_cases():
    import json
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "listed_refdata.json")))
    return d, [pytest.param(c, id="%s-%d-%s" % (c["suite"], c["index"], c["pbc"])) for c in d["cases"]]


_LISTED, _LISTED_PARAMS = _listed_cases()


@pytest.mark.parametrize("case", _LISTED_PARAMS)
def test_listed_oracle_reproduces_reference_known_answers(case):
    """Epot, dV/dlambda and forces of bonds, angles, Urey-Bradley, proper / improper / Ryckaert-Bellemans dihedrals at
    lambda 0, 0.5, 1 (listed_forces/tests/refdata).  Tolerances: the reference test's own relative tolerance per type
    is 2e-6 ... 2e-2 in mixed precision; the double oracle is held to 1e-6 relative of the largest force."""
    import oracle_binding as ob
    x = np.array(_LISTED["coordinates"], np.float64)
    box = np.full(3, _LISTED["box"])
    npbc = {"none": 0, "xy": 2, "xyz": 3}[case["pbc"]]
    ia = np.array([[0] + t for t in case["iatoms"]], np.int32)
    prm = ob.listed_iparams(case["type"], case["params"])
    for lam_name, want in case["results"].items():
        got = ob.listed(case["type"], ia, prm, x, box, npbc, float(lam_name))
        fw = np.array(want["forces"])
        scale = max(1.0, np.abs(fw).max())
        assert np.abs(got["f"] - fw).max() <= 1e-6 * scale, (case["type"], lam_name)
        assert abs(got["epot"] - want["epot"]) <= 1e-6 * max(1.0, abs(want["epot"]))
        assert abs(got["dvdl"] - want["dvdlambda"]) <= 1e-6 * max(1.0, abs(want["dvdlambda"]))
        assert np.abs(got["fshift"].sum(axis=0)).max() <= 1e-9 * scale   # shift forces sum to zero


def _pairs_fep(P, lam):
    import oracle_binding as ob
    s6 = P["sc_sigma"] ** 6
    return ob.ListedPairsFep(P["sc_alpha"], P["sc_alpha"], P["sc_power"], 0, s6, P["sc_sigma_min"] ** 6, lam, lam)


@pytest.mark.parametrize("case", [pytest.param(c, id="pairs-%d-%s" % (c["index"], c["pbc"])) for c in _LISTED["pairs"]["cases"]])
def test_listed_pairs_oracle_reproduces_reference_known_answers(case):
    """Perturbed 1-4 pairs (Beutler soft-core) at lambda 0, 0.5, 1 against listed_forces/tests/refdata/14Interaction_*.
    The reference evaluates them through cubic-spline tables (its own tolerance: 1e-5 float / 1e-7 double); the analytical
    oracle agrees to 2e-6 relative."""
    import oracle_binding as ob
    P = _LISTED["pairs"]
    x = np.array(P["coordinates"], np.float64)
    npbc = {"none": 0, "xy": 2, "xyz": 3}[case["pbc"]]
    ia = np.array([[0] + t for t in P["iatoms"]], np.int32)
    prm = np.zeros(1, ob.LISTED_IPARAMS)
    prm["p"][0, :4] = [case["params"][k] for k in ("c6A", "c12A", "c6B", "c12B")]
    for lam_name, want in case["results"].items():
        # the reference runs its unperturbed inputs with free-energy perturbation switched off: only the A charges count
        qB = P["chargeB"] if case["fep"] else P["chargeA"]
        got = ob.listed_pairs(ia, prm, x, P["chargeA"], qB, np.full(3, P["box"]), npbc, _pairs_fep(P, float(lam_name)),
                              P["epsfac"] * P["fudgeQQ"])
        fw = np.array(want["forces"])
        scale = max(1e-3, np.abs(fw).max())
        assert np.abs(got["f"] - fw).max() <= 2e-6 * scale, lam_name
        for k in ("eLJ", "eCoul", "dvdlVdw", "dvdlCoul"):
            assert abs(got[k] - want[k]) <= 2e-6 * max(1e-3, abs(want[k])), (k, lam_name)


# ---- random engine + tabulated normal distribution of the Langevin update ---------------------------------------
def test_threefry_and_tabulated_normal_reproduce_reference_known_answers():
    import json
    import oracle_binding as ob
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "langevin_refdata.json")))
    for kat in d["threefry"]:
        got = ob.threefry2x64(int(kat["key"][0]), int(kat["key"][1]), int(kat["ctr"][0]), int(kat["ctr"][1]))
        assert [str(g) for g in got] == kat["out"]                       # bit-exact
    t = d["tabulated"]
    got = ob.tabulated_normal(t["key0"], t["domain"], t["internalCounterBits"], 0, 0, t["mean"], t["stddev"], len(t["values"]))
    assert np.allclose(got, np.array(t["values"], np.float32), rtol=1e-6, atol=0)   # 1-2 float ulp: erfinv implementation, mean + v * stddev rounding
    tab = ob.normal_table(14)
    assert abs(float((tab.astype(np.float64) ** 2).mean()) - 1.0) < 1e-6 and tab[0] == -tab[-1] and (np.diff(tab) > 0).all()


# ---- coordinate update: leap-frog, SETTLE, LINCS (oracle/update_ref.c) against mdlib/tests/refdata ---------------------------
import update_cases as uc


@pytest.mark.parametrize("idx", range(16))
def test_leapfrog_oracle_reproduces_reference_known_answers(idx):
    c = uc.leapfrog_cases()[idx]
    x, v = c.x0, c.v0
    for step in range(c.num_steps):
        pr = c.dt_pc * c.pr_diag if c.do_pressure_couple(step) else None
        x, xp, v = ob.leapfrog(x, v, c.f, c.invmass, c.dt, lambdas=c.lambdas if c.num_tc > 0 else None, groups=c.groups, pr_diag=pr)
    assert np.max(np.abs(x - c.final_x)) <= c.tolerance
    assert np.max(np.abs(v - c.final_v)) <= c.tolerance
    if c.num_tc == 0 and c.nstpcouple == 0:
        t = c.num_steps * c.dt  # the analytical solution for a constant force (leapfrog.cpp:150-176)
        assert np.max(np.abs(x - (c.x0 + c.v0 * t + 0.5 * c.f * t * t * c.invmass[:, None]))) <= c.tolerance


@pytest.mark.parametrize("idx", range(13))
def test_settle_oracle_reproduces_reference_known_answers(idx):
    c = uc.settle_cases()[idx]
    xp, v, vir = ob.settle(c.atoms, c.mO, c.mH, c.dOH, c.dHH, c.x, c.xp, v=c.v if c.update_velocities else None, invdt=c.invdt,
                           compute_virial=c.calc_virial, pbc_type=c.pbc_type, box=c.box)
    n = 3 * c.num_settles
    assert np.max(np.abs(xp[:n] - c.final_x)) <= 1e-6  # settle.cpp:363
    assert np.array_equal(xp[n:], c.xp[n:])
    w = xp[:n].reshape(-1, 3, 3)
    for a, b, d in ((0, 1, c.dOH), (0, 2, c.dOH), (1, 2, c.dHH)):
        assert np.max(np.abs(np.sum((w[:, a] - w[:, b]) ** 2, axis=1) - d * d)) <= 1e-12
    if c.update_velocities:
        assert np.max(np.abs(v[:n] - c.final_v)) <= 1e-4
    if c.calc_virial:
        assert np.max(np.abs(vir - c.virial)) <= 1e-6
        assert np.max(np.abs(vir - vir.T)) <= 1e-6


@pytest.mark.parametrize("idx", range(14))
def test_lincs_oracle_reproduces_reference_known_answers(idx):
    c = uc.constraints_cases()[idx]
    xp, v, vir = ob.lincs(c.iatoms, c.lengths, c.invmass, c.n_iter, c.order, c.x, c.xp, v=c.v, invdt=c.invdt, compute_virial=True,
                          pbc_type=c.pbc_type, box=c.box)
    assert np.max(np.abs(xp - c.final_x)) <= c.tol_x, c.title
    assert np.max(np.abs(v - c.final_v)) <= c.tol_v, c.title
    assert np.max(np.abs(vir - c.virial)) <= c.tol_virial, c.title
    # constr.cpp:668-676: lengths, direction, centre of mass and its velocity
    for t, i, j in c.iatoms:
        d0, d1 = c.x[i] - c.x[j], xp[i] - xp[j]
        if c.pbc_type == 3:
            d0 -= np.rint(d0 / np.diag(c.box)) * np.diag(c.box)
            d1 -= np.rint(d1 / np.diag(c.box)) * np.diag(c.box)
        assert abs(np.linalg.norm(d1) - c.lengths[t]) <= 0.002 * c.lengths[t] + 1e-12
        assert d0 @ d1 >= 0
    assert np.max(np.abs((c.masses[:, None] * (xp - c.xp)).sum(axis=0) / len(c.masses))) <= c.tol_x
    assert np.max(np.abs((c.masses[:, None] * (v - c.v)).sum(axis=0) / len(c.masses))) <= c.tol_v


# ---- listed restraints without known answers in the reference (F_RESTRBONDS, F_DIHRES) and the simple pair types: the oracle's
# forces and dV/dlambda must be the derivatives of its own energy (every branch of the flat-bottomed potentials is visited) ------
def _restraint_systems():
    rng = np.random.default_rng(11)
    x = rng.uniform(0.2, 1.3, (4, 3))
    out = []
    for low, up1, up2 in ((0.9, 1.0, 1.1), (0.2, 0.3, 0.35), (0.05, 0.1, 0.6), (0.0, 0.05, 0.08)):
        prm = dict(lowA=low, up1A=up1, up2A=up2, kA=800.0, lowB=low * 1.1, up1B=up1 * 1.05, up2B=up2 * 1.2, kB=500.0)
        out.append(("restrbonds", np.array([[0, 0, 1], [0, 1, 3], [0, 2, 3]], np.int32), prm, x))
    for phi, dphi in ((60.0, 10.0), (-150.0, 5.0), (170.0, 30.0), (0.0, 0.0)):
        prm = dict(phiA=phi, dphiA=dphi, kfacA=40.0, phiB=phi + 7.0, dphiB=dphi + 3.0, kfacB=55.0)
        out.append(("dihres", np.array([[0, 0, 1, 2, 3], [0, 3, 0, 2, 1]], np.int32), prm, x))
    return out


@pytest.mark.parametrize("idx", range(8))
@pytest.mark.parametrize("npbc", [0, 3])
def test_restraint_oracle_is_self_consistent(idx, npbc):
    import oracle_binding as ob
    name, ia, prm, x = _restraint_systems()[idx]
    ip = ob.listed_iparams(name, prm)
    box = np.full(3, 1.5)
    rng = np.random.default_rng(idx)
    d = rng.normal(0, 1, x.shape)
    for lam in (0.0, 0.3, 1.0):
        r = ob.listed(name, ia, ip, x, box, npbc, lam)
        eps = 1e-6
        ep = ob.listed(name, ia, ip, x + eps * d, box, npbc, lam)["epot"]
        em = ob.listed(name, ia, ip, x - eps * d, box, npbc, lam)["epot"]
        assert abs((ep - em) / (2 * eps) + np.sum(r["f"] * d)) <= 1e-5 * max(1.0, np.abs(r["f"]).max())
        lp = ob.listed(name, ia, ip, x, box, npbc, lam + 1e-6)["epot"]
        lm = ob.listed(name, ia, ip, x, box, npbc, lam - 1e-6)["epot"]
        assert abs((lp - lm) / 2e-6 - r["dvdl"]) <= 1e-5 * max(1.0, abs(r["dvdl"]))
        assert np.abs(r["f"].sum(axis=0)).max() <= 1e-9 * max(1.0, np.abs(r["f"]).max())


@pytest.mark.parametrize("kind", [1, 2])
def test_simple_pairs_oracle_is_self_consistent(kind):
    import oracle_binding as ob
    rng = np.random.default_rng(kind)
    x = rng.uniform(0.2, 1.3, (4, 3))
    ip = np.zeros(2, ob.LISTED_IPARAMS)
    ip["p"][0, :5] = [0.4, -0.6, 0.8, 2.5e-3, 2.0e-6] if kind == 1 else [0.4, -0.6, 2.5e-3, 2.0e-6, 0]
    ip["p"][1, :5] = [-0.3, -0.2, 0.5, 1.0e-3, 1.0e-6] if kind == 1 else [-0.3, -0.2, 1.0e-3, 1.0e-6, 0]
    ia = np.array([[0, 0, 3], [1, 1, 2], [0, 2, 0]], np.int32)
    box, d = np.full(3, 1.5), rng.normal(0, 1, x.shape)
    for npbc in (0, 3):
        r = ob.listed_simple_pairs(kind, ia, ip, x, box, npbc, 138.935)
        e = [sum(ob.listed_simple_pairs(kind, ia, ip, x + s * 1e-6 * d, box, npbc, 138.935)[k] for k in ("e_lj", "e_coul")) for s in (1, -1)]
        assert abs((e[0] - e[1]) / 2e-6 + np.sum(r["f"] * d)) <= 1e-5 * np.abs(r["f"]).max()
