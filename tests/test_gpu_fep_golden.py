"""The reference's 72 known answers of its free-energy kernel, through the HIP perturbed-pair kernels themselves (C ABI).

tests/test_oracle_golden.py pins the ORACLE with tests/golden/nb_fep_refdata.json (transcribed from
/root/reference/src/gromacs/gmxlib/nonbonded/tests/refdata/NBInteraction_NonbondedFepTest_testKernel_*.xml by make_nb_fep_golden.py);
here the same 4-atom system, the same one-i-entry t_nblist {i = 0; j = 0 (self, excluded), 1, 2, 3} and the same non-default constants
(gmxlib/nonbonded/tests/nb_free_energy.cpp:193-201,231-240,306-362) go through nbnxm_gpu_init_feppairlist / nbnxm_gpu_launch_kernel:
nbnxmFepKernel (energy + virial flavour) for EVdw, ECoul, dVdlCoul, dVdlVdw, the forces and the shift force, and — where the case has
soft-core, which is when mdrun asks for foreign-lambda energies (nbnxm_cuda.cu:817-856) — nbnxmFepForeignKernel with the three lambda
values of the test as foreign lambdas: its energies and dV/dl at every lambda must equal the known answers of the SIBLING cases (same
interaction, soft-core function, alpha and sc-coul at that lambda), and index 0 the case's own.

Tolerance: the reference compares its float build with relativeToleranceAsPrecisionDependentFloatingPoint(1, 1e-6, 1e-8) and 10,000 ULP
headroom (nb_free_energy.cpp:433-435, 504-506), i.e. up to 6e-4 relative in float; here 2e-5 relative with an absolute floor of 2e-5 of
the case's largest term — the float oracle meets 1e-5 on these inputs (0.17 nm separations, r^-12).

Nothing is skipped: Beutler and Gapsys, reaction-field-shaped cut-off (k_rf 0, c_rf 1), potential switch from 0 to rvdw, Ewald + LJ-PME
(the GPU path follows the CPU kernel there), alpha 0 and 0.3, soft-core Coulomb on and off."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

import fep_testlib as tl

pkg = tl.pkg
pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "nb_fep_refdata.json")) as fh:
    GOLDEN = json.load(fh)
SYS = GOLDEN["system"]
LAMBDAS = [0.0, 0.5, 1.0]
REL = 2e-5


def _workload():
    import importlib
    return importlib.import_module("gromacs_fep_gpu_amd.workload")


def case_key(case):
    i = case["interaction"]
    return (i["coulomb"], i["vdw"], i["vdw_modifier"], case["softcore"], case["sc_alpha"], case["sc_coul"])


SIBLINGS = {}
for _c in GOLDEN["cases"]:
    SIBLINGS.setdefault(case_key(_c), {})[_c["lambda"]] = _c


def make_gpu_case(case):
    """The golden system as a workload case: interaction constants as the reference test sets them, a 5 nm box around the four atoms."""
    w = _workload()
    s = SYS
    ntype = s["ntype"]
    lj = np.array(s["lj_c6_c12"], np.float64)
    nbfp = np.empty(2 * ntype * ntype, np.float32)
    nbfp[0::2] = 6.0 * lj[:, 0]       # makeNonBondedParameterLists, mdlib/forcerec.cpp:115-152
    nbfp[1::2] = 12.0 * lj[:, 1]
    x = np.array(s["x"], np.float32)
    box = np.array([5.0, 5.0, 5.0], np.float32)       # rectangular box lengths
    qA, qB = np.array(s["chargeA"], np.float32), np.array(s["chargeB"], np.float32)
    tA, tB = np.array(s["typeA"], np.int32), np.array(s["typeB"], np.int32)
    pert = ((qA != qB) | (tA != tB)).astype(np.uint8)
    c = SimpleNamespace()
    c.sys = dict(nbfp=nbfp, ntype=ntype)
    c.ntype, c.natoms = ntype, 4
    c.grid = pkg.Grid(x, box, qA, qB, tA, tB, ntype, perturbed=pert)
    inter = case["interaction"]
    pme = inter["coulomb"] == "Pme"
    c.rc = c.rvdw = s["rcoulomb"]
    c.rlist = c.rlist_inner = s["rcoulomb"]
    c.epsfac = w.ONE_4PI_EPS0 * s["epsfac_factor_of_one4pieps0"]
    c.k_rf, c.c_rf = s["k_rf"], s["c_rf"]
    c.beta = w.calc_ewaldcoeff_q(s["ewald_rc"], s["ewald_rtol"]) if pme else 0.0
    c.sh_ewald = s["sh_ewald"]
    c.elec = "ewald" if pme else "rf"
    c.elec_type = pkg.ELEC_EWALD_ANA if pme else pkg.ELEC_RF
    c.disp_shift = (0.0, 0.0, s["dispersion_shift_cpot"])
    c.rep_shift = (0.0, 0.0, s["repulsion_shift_cpot"])
    c.vdw_switch, c.rvdw_switch = (0.0, 0.0, 0.0), 0.0
    c.beta_lj = c.sh_lj_ewald = 0.0
    c.vdw, c.vdw_type = "cut", pkg.VDW_CUT
    if inter["vdw_modifier"] == "PotSwitch":
        c.vdw, c.vdw_type = "pswitch", pkg.VDW_PSWITCH
        c.rvdw_switch = s["rvdw_switch"]
        d = s["rvdw"] - s["rvdw_switch"]
        c.vdw_switch = (-10.0 / d ** 3, 15.0 / d ** 4, -6.0 / d ** 5)
    if inter["vdw"] == "Pme":
        c.vdw, c.vdw_type = "ewald_geom", pkg.VDW_EWALD_GEOM   # makeLJPmeC6GridCorrectionParameters(..., LongRangeVdW::Geom)
        c.beta_lj = w.calc_ewaldcoeff_lj(s["ewald_rc"], s["ewald_rtol"])
        c.sh_lj_ewald = s["sh_lj_ewald"]
    a = case["sc_alpha"]
    c.sc_alpha, c.sc_power, c.sc_sigma, c.sc_coul = a, s["sc_power"], s["sc_sigma"], case["sc_coul"]
    c.sc_sigma_min = s["sc_sigma_min"]
    c.softcore = "gapsys" if case["softcore"] == "Gapsys" else "beutler"
    c.gapsys = (a, a, s["gapsys_sigma_lj"])        # ForcerecHelper::setSoftcoreAlpha sets both linpoint scalings (nb_free_energy.cpp:254-259)
    c.lambda_coul = c.lambda_vdw = case["lambda"]
    c.have_soft_core = a != 0
    c.n_lambda = len(LAMBDAS)
    c.all_lambda = np.array(LAMBDAS)
    return c


def run_case(case, dhdl):
    w = _workload()
    c = make_gpu_case(case)
    g = c.grid
    nb = pkg.NbnxmGpu(w.gpu_interaction_params(c), g.num_types, g.nbat_nbfp(c.sys["nbfp"]),
                      nbfp_comb=w.lj_type_params(c) if c.vdw == "ewald_geom" else None, fep=True, n_lambda=c.n_lambda)
    sig6 = c.sc_sigma ** 6
    # interaction_const_t::SoftCoreParameters (mdtypes/interaction_const.cpp): alphaCoulomb and sigma6Minimum vanish without sc-coul
    nb.copy_fepparams(c.sc_alpha if c.sc_coul else 0.0, c.sc_alpha, c.sc_power, sig6, c.sc_sigma_min ** 6 if c.sc_coul else 0.0,
                      c.lambda_coul, c.lambda_vdw, c.all_lambda, c.all_lambda)
    if c.softcore == "gapsys":
        nb.set_softcore(pkg.SOFTCORE_GAPSYS, *c.gapsys)
    nb.init_atomdata(g.num_atoms, g.type, qA=g.qA, qB=g.qB, typeA=g.typeA, typeB=g.typeB)
    # every interaction of the test lives in the atom-pair list: the cluster list is empty (the FEP kernels launch regardless)
    nb.init_pairlist(np.zeros(0, pkg.SCI_DTYPE), np.zeros(0, pkg.CJ_PACKED_DTYPE), np.zeros(1, pkg.EXCL_DTYPE))
    fep = dict(iinr=SYS["iinr"], shift=SYS["shift"], jindex=SYS["jindex"], jjnr=SYS["jjnr"], excl_fep=SYS["excl_fep"])
    nb.init_feppairlist(fep, g.atomIndices)
    # the test's only shift vector is zero and its index (0) is not the central one: the shift force is booked (nb_free_energy.cpp:343-346)
    nb.upload_shiftvec(np.zeros((pkg.NUM_SHIFT_VECTORS, 3), np.float32))
    nb.copy_xq_to_gpu(g.xq)
    sw = pkg.step_workload(energy=True, virial=True, dhdl=dhdl)
    nb.clear_outputs(True)
    nb.launch_kernel(sw)
    f = np.zeros((g.num_atoms, 3), np.float32)
    nb.launch_cpyback(f, sw)
    res = nb.wait_finish_task(sw, c.have_soft_core)
    nb.free()
    real = g.atomIndices >= 0
    f_atoms = np.zeros((4, 3))
    f_atoms[g.atomIndices[real]] = f[real]
    dv = res["dvdl_nonlin"] if c.have_soft_core else res["dvdl_lin"]
    return dict(f=f_atoms, fshift0=res["fshift"][0].astype(np.float64), e_lj=res["e_lj"], e_el=res["e_el"], dvdl_coul=dv[0], dvdl_vdw=dv[1],
                raw=res)


def check(got, want, abs_, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want)
    tol = np.maximum(abs_, REL * np.abs(want))
    assert np.all(err <= tol), "%s: got %s want %s (err %s, tol %s)" % (what, got, want, err, tol)


def term_scale(exp):
    return max(1.0, abs(exp["EVdw"]), abs(exp["ECoul"]), abs(exp["dVdlCoul"]), abs(exp["dVdlVdw"]))


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: "k%02d" % c["index"])
def test_hip_fep_kernel_reproduces_reference_known_answers(case):
    got = run_case(case, dhdl=False)
    exp = case["expected"]
    abs_ = REL * term_scale(exp)
    check(got["e_lj"], exp["EVdw"], abs_, "EVdw")
    check(got["e_el"], exp["ECoul"], abs_, "ECoul")
    check(got["dvdl_coul"], exp["dVdlCoul"], abs_, "dVdlCoul")
    check(got["dvdl_vdw"], exp["dVdlVdw"], abs_, "dVdlVdw")
    fscale = max(1.0, float(np.max(np.abs(exp["Forces"]))))
    check(got["f"], exp["Forces"], REL * fscale, "forces")
    check(got["fshift0"], exp["ShiftForceCentral"], REL * fscale, "shift force")


@pytest.mark.parametrize("case", [c for c in GOLDEN["cases"] if c["sc_alpha"] != 0], ids=lambda c: "k%02d" % c["index"])
def test_hip_foreign_kernel_reproduces_the_sibling_known_answers(case):
    """dH/dl step: index 0 of the foreign arrays is the case's own lambda, 1 + k the k-th foreign lambda; the perturbed-pair force
    kernel of the same step must still give the case's own answers."""
    got = run_case(case, dhdl=True)
    exp = case["expected"]
    abs_ = REL * term_scale(exp)
    check(got["e_lj"], exp["EVdw"], abs_, "EVdw on the dH/dl step")
    check(got["e_el"], exp["ECoul"], abs_, "ECoul on the dH/dl step")
    raw = got["raw"]
    sib = SIBLINGS[case_key(case)]
    assert sorted(sib) == LAMBDAS
    for idx, lam in enumerate([case["lambda"]] + LAMBDAS):
        e = sib[lam]["expected"]
        a = REL * term_scale(e)
        check(raw["foreign_energies"][idx], e["EVdw"] + e["ECoul"], a, "foreign energy, index %d (lambda %g)" % (idx, lam))
        check(raw["foreign_dhdl_coul"][idx], e["dVdlCoul"], a, "foreign dV/dl Coulomb, index %d" % idx)
        check(raw["foreign_dhdl_vdw"][idx], e["dVdlVdw"], a, "foreign dV/dl VdW, index %d" % idx)
