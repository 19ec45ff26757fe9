"""Shared helpers for the parity tests, smoke() and bench.py.

Builds a synthetic solvated FEP case (SURVEY §8d recipe), runs it through
  * the CPU oracle (oracle/: nbnxm_ref on the carved cluster list + fep_oracle on the FEP list),
  * the HIP path through the C ABI (gromacs-fep-gpu_amd),
  * an O(N^2) numpy evaluation (small N only),
and compares.  Only test code may import the oracle.
"""
import math
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for _p in (HERE, ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import oracle_binding as ob  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()

ONE_4PI_EPS0 = ob.ONE_4PI_EPS0


def make_case(nm=(8, 8, 8), num_perturbed_molecules=3, elec="rf", vdw="cut", seed=2026, rc=1.0, rlist=1.1,
              rlist_fep=None, sc_alpha=0.5, sc_power=1, sc_sigma=0.3, sc_coul=True, lambda_coul=0.5,
              lambda_vdw=0.5, n_lambda=0, max_cjpacked_per_sci=0, identical_states=False, rvdw_switch=0.8,
              spacing=0.310736, jitter=0.03, num_extra_types=0, rvdw=None, softcore="beutler",
              gapsys=(0.85, 0.3, 0.3)):
    """elec: 'rf' | 'cut' | 'ewald' | 'ewald_tab';  vdw: 'cut' | 'pswitch' | 'fswitch' | 'comb_geom' | 'comb_lb' |
    'ewald_geom' | 'ewald_lb' (LJ-PME real-space part; perturbed pairs as in the CPU kernel, with the grid correction)."""
    sysd = pkg.make_water_box(nm[0], nm[1], nm[2], spacing=spacing, jitter=jitter, seed=seed,
                              num_perturbed_molecules=num_perturbed_molecules)
    n = len(sysd["qA"])
    perturbed = (sysd["qA"] != sysd["qB"]) | (sysd["typeA"] != sysd["typeB"])
    if identical_states:
        # keep the perturbed flags but make B == A (oracle cross-check in the non-perturbed limit)
        sysd["qB"] = sysd["qA"].copy()
        sysd["typeB"] = sysd["typeA"].copy()
    if num_extra_types > 0:
        add_oxygen_type_variants(sysd, num_extra_types, seed)
    ntype = sysd["ntype"]
    grid = pkg.Grid(sysd["x"], sysd["box"], sysd["qA"], sysd["qB"], sysd["typeA"], sysd["typeB"], ntype,
                    perturbed=perturbed.astype(np.uint8))
    excl_index, excl_atoms = pkg.exclusions_from_groups(sysd["molId"])
    if rlist_fep is None:
        rlist_fep = rlist + 0.0886  # rlist_fep 1.1886 for rlist 1.1
    plist = grid.build_pairlist(excl_index, excl_atoms, rlist, max_cjpacked_per_sci, carve_fep=True,
                                rlist_fep=rlist_fep)
    plist_fused = grid.build_pairlist(excl_index, excl_atoms, rlist, max_cjpacked_per_sci, carve_fep=False,
                                      rlist_fep=rlist_fep)

    c = SimpleNamespace()
    c.sys, c.grid, c.plist, c.plist_fused = sysd, grid, plist, plist_fused
    c.natoms, c.ntype = n, ntype
    c.perturbed = perturbed
    c.excl_index, c.excl_atoms = excl_index, excl_atoms
    c.rc, c.rlist, c.rlist_fep = rc, rlist, rlist_fep
    # rvdw < rcoulomb: the twin-range flavours of the Ewald kernels (ElecType::EwaldAnaTwin / EwaldTabTwin)
    c.rvdw = rc if rvdw is None else float(rvdw)
    assert c.rvdw == rc or (c.rvdw < rc and elec in ("ewald", "ewald_tab"))
    c.elec, c.vdw = elec, vdw
    c.epsfac = ONE_4PI_EPS0
    c.k_rf = c.c_rf = c.beta = c.sh_ewald = 0.0
    if elec == "rf":          # epsilon_rf = infinity
        c.k_rf = 1.0 / (2.0 * rc ** 3)
        c.c_rf = 1.0 / rc + c.k_rf * rc * rc
        c.elec_type = pkg.ELEC_RF
    elif elec == "cut":
        c.c_rf = 1.0 / rc
        c.elec_type = pkg.ELEC_CUT
    else:
        c.beta = ob.lib().oracle_calc_ewaldcoeff_q(rc, 1e-5)
        c.sh_ewald = math.erfc(c.beta * rc) / rc
        c.elec_type = pkg.ELEC_EWALD_ANA if elec == "ewald" else pkg.ELEC_EWALD_TAB
        if c.rvdw < rc:
            c.elec_type = pkg.ELEC_EWALD_ANA_TWIN if elec == "ewald" else pkg.ELEC_EWALD_TAB_TWIN
    # potential-shift modifier (GROMACS default): V(rvdw) = 0
    rv = c.rvdw
    c.disp_shift = (0.0, 0.0, -rv ** -6)
    c.rep_shift = (0.0, 0.0, -rv ** -12)
    c.vdw_switch = (0.0, 0.0, 0.0)
    c.rvdw_switch = 0.0
    c.beta_lj = c.sh_lj_ewald = 0.0
    c.vdw_type = pkg.VDW_CUT
    if vdw == "pswitch":
        c.vdw_type = pkg.VDW_PSWITCH
        c.rvdw_switch = rvdw_switch
        d = rv - rvdw_switch
        c.vdw_switch = (-10.0 / d ** 3, 15.0 / d ** 4, -6.0 / d ** 5)   # potential_switch_constants, forcerec/interaction_const
        c.disp_shift = (0.0, 0.0, 0.0)
        c.rep_shift = (0.0, 0.0, 0.0)
    elif vdw in ("comb_geom", "comb_lb"):
        c.vdw_type = pkg.VDW_CUT_COMB_GEOM if vdw == "comb_geom" else pkg.VDW_CUT_COMB_LB
    elif vdw in ("ewald_geom", "ewald_lb"):
        c.vdw_type = pkg.VDW_EWALD_GEOM if vdw == "ewald_geom" else pkg.VDW_EWALD_LB
        c.beta_lj = ob.lib().oracle_calc_ewaldcoeff_lj(rv, 1e-3)
        crc2 = (c.beta_lj * rv) ** 2
        # forcerec: the grid part's potential shift, sh_lj_ewald = (exp(-b^2 rc^2) (1 + b^2 rc^2 + b^4 rc^4 / 2) - 1) / rc^6
        c.sh_lj_ewald = (math.exp(-crc2) * (1 + crc2 + 0.5 * crc2 * crc2) - 1) / rv ** 6
    elif vdw == "fswitch":
        c.vdw_type = pkg.VDW_FSWITCH
        c.rvdw_switch = rvdw_switch
        c.disp_shift = force_switch_constants(6.0, rvdw_switch, rv)
        c.rep_shift = force_switch_constants(12.0, rvdw_switch, rv)
    c.sc_alpha, c.sc_power, c.sc_sigma, c.sc_coul = sc_alpha, sc_power, sc_sigma, sc_coul
    # softcore "gapsys": (sc-gapsys-scale-linpoint-lj, -q, sc-gapsys-sigma-lj); the alphas are not used then
    c.softcore, c.gapsys = softcore, tuple(gapsys)
    c.lambda_coul, c.lambda_vdw = lambda_coul, lambda_vdw
    c.n_lambda = n_lambda
    c.all_lambda = np.linspace(0.0, 1.0, n_lambda) if n_lambda > 0 else np.zeros(0)
    c.have_soft_core = (sc_alpha != 0) if softcore == "beutler" else (gapsys[0] != 0 or gapsys[1] != 0)
    return c


def add_oxygen_type_variants(sysd, k, seed):
    """Gives the water oxygens k extra atom types with slightly different LJ parameters (geometric mixing),
    to exercise the type-pair table with a realistic number of types (ntype = 3 + k)."""
    rng = np.random.default_rng(seed)
    nt_old, nt = sysd["ntype"], sysd["ntype"] + k
    old = np.asarray(sysd["nbfp"], np.float64).reshape(nt_old, nt_old, 2)
    c6 = np.zeros(nt)
    c12 = np.zeros(nt)
    c6[0], c12[0] = old[0, 0]
    for t in range(nt_old, nt):
        c6[t] = old[0, 0, 0] * (1.0 + 0.03 * (t - nt_old + 1))
        c12[t] = old[0, 0, 1] * (1.0 + 0.05 * (t - nt_old + 1))
    nbfp = np.zeros((nt, nt, 2))
    nbfp[..., 0] = np.sqrt(np.outer(c6, c6))
    nbfp[..., 1] = np.sqrt(np.outer(c12, c12))
    sysd["nbfp"] = nbfp.reshape(-1).astype(np.float32)
    sysd["ntype"] = nt
    ox = np.flatnonzero(sysd["typeA"] == 0)
    newt = rng.integers(0, k + 1, size=len(ox))
    newt = np.where(newt == 0, 0, newt + nt_old - 1).astype(np.int32)
    same = sysd["typeB"][ox] == sysd["typeA"][ox]
    sysd["typeA"][ox] = newt
    sysd["typeB"][ox] = np.where(same, newt, sysd["typeB"][ox])


def lj_grid_table(c):
    """ntype x ntype x 2 table of the grid C6 (the CPU kernel's fr->ljpme_c6grid, nb_free_energy.cpp:560-566; second component unused)"""
    if c.vdw not in ("ewald_geom", "ewald_lb"):
        return None
    t = lj_type_params(c).astype(np.float64)[:c.ntype]
    if c.vdw == "ewald_geom":
        c6 = np.outer(t[:, 0], t[:, 0])
    else:
        c6 = np.outer(t[:, 1], t[:, 1]) * (t[:, 0][:, None] + t[:, 0][None, :]) ** 6
    out = np.zeros((c.ntype, c.ntype, 2))
    out[..., 0] = c6
    return out.reshape(-1)


def lj_type_params(c):
    """The per-TYPE table of the same parameters (numTypes = ntype + 1 rows): NBParamGpu::nbfp_comb of the LJ-PME flavours"""
    return lj_comb_params(c, np.arange(c.ntype + 1))


def lj_comb_params(c, atype):
    """Per-atom combination-rule parameters in the kernels' convention (nbnxm_cuda_kernel.cuh:504-516):
    geometric: (sqrt(6 C6), sqrt(12 C12)); Lorentz-Berthelot: (sigma/2, sqrt(eps)) with
    c6 = eps sigma^6 = 6 C6 and c12 = c6 sigma^6 = 12 C12.  atype: grid-order (masked) types, numTypes = ntype + 1."""
    nt = c.ntype
    tab = np.zeros((nt + 1, 2))
    diag = np.asarray(c.sys["nbfp"], np.float64).reshape(nt, nt, 2)[np.arange(nt), np.arange(nt)]
    tab[:nt] = diag
    out = np.zeros((len(atype), 2), np.float32)
    c6, c12 = tab[atype, 0], tab[atype, 1]
    has = (c6 > 0) & (c12 > 0)
    if c.vdw in ("comb_geom", "ewald_geom"):
        out[:, 0] = np.sqrt(c6)
        out[:, 1] = np.sqrt(c12)
    else:
        sig6 = np.where(has, c12 / np.where(has, c6, 1.0), 0.0)
        out[:, 0] = 0.5 * sig6 ** (1.0 / 6.0)
        out[:, 1] = np.where(has, c6 / np.sqrt(np.where(has, c12, 1.0)), 0.0)
    return out


def force_switch_constants(p, rsw, rc):
    """force_switch_constants() of mdtypes/interaction_const.cpp: (c2, c3, cpot)."""
    c2 = ((p + 1) * rsw - (p + 4) * rc) / (rc ** (p + 2) * (rc - rsw) ** 2)
    c3 = -((p + 1) * rsw - (p + 3) * rc) / (rc ** (p + 2) * (rc - rsw) ** 3)
    cpot = -rc ** -p + p * c2 / 3 * (rc - rsw) ** 3 + p * c3 / 4 * (rc - rsw) ** 4
    return (c2, c3, cpot)


# ---- oracle -----------------------------------------------------------------------------------------

def oracle_fep_params(c):
    p = ob.FepParams()
    p.elecIsEwald = 1 if c.elec in ("ewald", "ewald_tab") else 0
    p.vdwIsEwald = 1 if c.vdw in ("ewald_geom", "ewald_lb") else 0
    p.ewaldcoeff_lj, p.sh_lj_ewald = c.beta_lj, c.sh_lj_ewald
    p.vdwPotSwitch = 1 if c.vdw == "pswitch" else 0
    p.epsfac = c.epsfac
    p.rcoulomb, p.rvdw = c.rc, c.rvdw
    p.rvdw_switch = c.rvdw_switch
    p.k_rf, p.c_rf = c.k_rf, c.c_rf
    p.ewaldcoeff_q, p.sh_ewald = c.beta, c.sh_ewald
    p.dispersion_shift_cpot, p.repulsion_shift_cpot = c.disp_shift[2], c.rep_shift[2]
    if c.softcore == "gapsys":
        ob.softcore_params(p, c.sc_alpha, c.sc_power, c.sc_sigma, c.sc_sigma, c.sc_coul, softcore_type=ob.SOFTCORE_GAPSYS,
                           gapsys_lj=c.gapsys[0], gapsys_q=c.gapsys[1], gapsys_sigma=c.gapsys[2])
    else:
        ob.softcore_params(p, c.sc_alpha, c.sc_power, c.sc_sigma, c.sc_sigma, c.sc_coul)
    return p


def oracle_ref_params(c):
    p = ob.RefParams()
    p.elecType, p.vdwType = c.elec_type, c.vdw_type
    p.epsfac, p.c_rf, p.k_rf = c.epsfac, c.c_rf, c.k_rf
    p.ewaldcoeff_q, p.sh_ewald = c.beta, c.sh_ewald
    p.rcoulomb, p.rvdw = c.rc, c.rvdw
    p.rvdw_switch, p.rlist = c.rvdw_switch, c.rlist
    p.disp_c2, p.disp_c3, p.disp_cpot = c.disp_shift
    p.rep_c2, p.rep_c3, p.rep_cpot = c.rep_shift
    p.sw_c3, p.sw_c4, p.sw_c5 = c.vdw_switch
    p.ewaldcoeff_lj, p.sh_lj_ewald = c.beta_lj, c.sh_lj_ewald
    return p


def run_oracle(c, energy=True, precision="f64", foreign=False, cjPacked=None, num_threads=1):
    """Reference result of one step: cluster kernel on the carved list + FEP kernel on the FEP list.
    Forces are returned in GRID order (like nbat->out[0].f)."""
    g = c.grid
    flags = ob.DO_FORCE | ob.DO_SHIFTFORCE | (ob.DO_POTENTIAL if energy else 0)
    ljc = lj_comb_params(c, g.type) if c.vdw in ("comb_geom", "comb_lb") else None
    ref = ob.nbnxm_ref(c.plist.sci, c.plist.cjPacked if cjPacked is None else cjPacked, c.plist.excl, g.xq,
                       g.type, g.num_types, g.nbat_nbfp(c.sys["nbfp"]), oracle_ref_params(c), g.shift_vec,
                       compute_energy=energy, compute_fshift=True, lj_comb=ljc,
                       nbfp_comb=lj_type_params(c) if c.vdw in ("ewald_geom", "ewald_lb") else None, precision=precision,
                       num_threads=num_threads)
    fp = oracle_fep_params(c)
    fep = ob.fep_kernel(c.plist.fep, g.x_wrapped, c.ntype, fp, g.shift_vec, c.sys["nbfp"], lj_grid_table(c), c.sys["qA"],
                        c.sys["qB"], c.sys["typeA"], c.sys["typeB"], flags, c.lambda_coul, c.lambda_vdw, precision)
    f = np.array(ref["f"], dtype=np.float64)
    real = g.atomIndices >= 0
    f[real] += fep["f"][g.atomIndices[real]]
    fshift = np.array(ref["fshift"], dtype=np.float64)
    fs_fep = np.array(fep["fshift"], dtype=np.float64)
    fs_fep[pkg.CENTRAL_SHIFT_INDEX] = 0.0  # the GPU path does not accumulate the central shift
    fshift += fs_fep
    out = dict(f=f, fshift=fshift, e_lj=ref["Vv"] + fep["Vv"], e_el=ref["Vc"] + fep["Vc"],
               dvdl_coul=fep["dvdl_coul"], dvdl_vdw=fep["dvdl_vdw"], npairs=ref["npairs"],
               parts=dict(ref=ref, fep=fep))
    if foreign and c.n_lambda > 0:
        out["foreign"] = ob.fep_foreign(c.plist.fep, g.x_wrapped, c.ntype, fp, g.shift_vec, c.sys["nbfp"], lj_grid_table(c),
                                        c.sys["qA"], c.sys["qB"], c.sys["typeA"], c.sys["typeB"], c.lambda_coul,
                                        c.lambda_vdw, c.all_lambda, c.all_lambda, precision)
    return out


# ---- GPU through the C ABI -----------------------------------------------------------------------------

def gpu_interaction_params(c, use_dynamic_pruning=False):
    tab, scale = None, 0.0
    if c.elec == "ewald_tab":
        tab, scale = ewald_force_table(c.beta, c.rc + 0.3)
    return pkg.make_interaction_params(c.elec_type, c.vdw_type, c.epsfac, c.rc, c.rvdw, c.rlist, getattr(c, "rlist_inner", c.rlist),
                                       k_rf=c.k_rf, c_rf=c.c_rf, ewaldcoeff_q=c.beta, sh_ewald=c.sh_ewald,
                                       rvdw_switch=c.rvdw_switch, dispersion_shift=c.disp_shift,
                                       repulsion_shift=c.rep_shift, vdw_switch=c.vdw_switch,
                                       use_dynamic_pruning=use_dynamic_pruning, coulomb_tab=tab,
                                       coulomb_tab_scale=scale, ewaldcoeff_lj=c.beta_lj, sh_lj_ewald=c.sh_lj_ewald)


def ewald_force_table(beta, rmax, scale=2000.0):
    """Ewald correction force table with the meaning the kernels give coulombEwaldTables->tableF
    (nbnxm_cuda_kernel.cuh:624-628: F/r += qq (int_bit/r^2 - T(r)) / r):
    T(r) = -d/dr(erf(beta r)/r) = erf(beta r)/r^2 - 2 beta/sqrt(pi) exp(-beta^2 r^2)/r, at r = i/scale."""
    n = int(rmax * scale) + 2
    r = np.arange(n) / scale
    tab = np.zeros(n)
    r1 = r[1:]
    tab[1:] = (np.vectorize(math.erf)(beta * r1) / r1 ** 2
               - 2 * beta / math.sqrt(math.pi) * np.exp(-(beta * r1) ** 2) / r1)
    return tab.astype(np.float32), scale


def setup_gpu(c, fused=False, use_dynamic_pruning=False, list_override=None):
    """list_override: (sci, cjPacked, excl) to upload instead of the case's own list (domain decomposition)."""
    g = c.grid
    ic = gpu_interaction_params(c, use_dynamic_pruning)
    nb = pkg.NbnxmGpu(ic, g.num_types, g.nbat_nbfp(c.sys["nbfp"]),
                      nbfp_comb=lj_type_params(c) if c.vdw in ("ewald_geom", "ewald_lb") else None, fep=True, n_lambda=c.n_lambda)
    sig6 = c.sc_sigma ** 6
    alpha_coul = c.sc_alpha if c.sc_coul else 0.0
    sig6_min = sig6 if c.sc_coul else 0.0
    nb.copy_fepparams(alpha_coul, c.sc_alpha, c.sc_power, sig6, sig6_min, c.lambda_coul, c.lambda_vdw,
                      c.all_lambda, c.all_lambda)
    if c.softcore == "gapsys":
        nb.set_softcore(pkg.SOFTCORE_GAPSYS, *c.gapsys)
    ljc = lj_comb_params(c, g.type) if c.vdw in ("comb_geom", "comb_lb") else None
    nb.init_atomdata(g.num_atoms, g.type, lj_comb=ljc, qA=g.qA, qB=g.qB, typeA=g.typeA, typeB=g.typeB)
    pl = c.plist_fused if fused else c.plist
    if list_override is not None:
        nb.init_pairlist(*list_override)
    else:
        nb.init_pairlist(pl.sci, pl.cjPacked, pl.excl)
    if fused:
        # no atom-pair list at all: nbnxmFepClusterKernel covers forces, energies and foreign lambdas of the perturbed cluster pairs
        nb.init_fep_cluster_bits(g.fepBits)
        nb.set_fep_mode(True)
    else:
        nb.init_feppairlist(c.plist.fep, g.atomIndices)
    nb.upload_shiftvec(g.shift_vec)
    nb.copy_xq_to_gpu(g.xq)
    return nb


def run_gpu(c, energy=True, fused=False, dhdl=False, nb=None, prune=False):
    own = nb is None
    if own:
        nb = setup_gpu(c, fused=fused, use_dynamic_pruning=prune)
    sw = pkg.step_workload(energy=energy, virial=True, dhdl=dhdl)
    nb.clear_outputs(True)
    nb.launch_kernel(sw)
    f = np.zeros((c.grid.num_atoms, 3), np.float32)
    nb.launch_cpyback(f, sw)
    res = nb.wait_finish_task(sw, c.have_soft_core)
    dv = res["dvdl_nonlin"] if c.have_soft_core else res["dvdl_lin"]
    out = dict(f=f.astype(np.float64), fshift=res["fshift"].astype(np.float64), e_lj=res["e_lj"], e_el=res["e_el"],
               dvdl_coul=dv[0], dvdl_vdw=dv[1], raw=res)
    if dhdl:
        out["foreign"] = dict(energies=res["foreign_energies"], dvdlCoul=res["foreign_dhdl_coul"],
                              dvdlVdw=res["foreign_dhdl_vdw"])
    if own:
        nb.free()
    return out


# ---- comparison -------------------------------------------------------------------------------------------

def assert_parity(got, want, rel=1e-4, energy=True, label=""):
    """Forces: per-atom error relative to max(|f_i|, RMS |f|); sums that cancel (energies, dV/dl, fshift):
    relative to the larger of |value| and a scale of the summed magnitudes (SURVEY §7 'hard parts')."""
    f_got, f_want = np.asarray(got["f"]), np.asarray(want["f"])
    frms = math.sqrt(float(np.mean(np.sum(f_want ** 2, axis=1)))) + 1e-30
    # per atom: |delta f_i| <= rel * max(|f_i|, rms |f|)
    ferr = np.sqrt(np.sum((f_got - f_want) ** 2, axis=1))
    ftol = rel * np.maximum(np.sqrt(np.sum(f_want ** 2, axis=1)), frms)
    worst = int(np.argmax(ferr / ftol))
    assert ferr[worst] <= ftol[worst], \
        "%s force err %.3e on atom %d (|f| %.3e, rms |f| %.3e)" % (label, ferr[worst], worst,
                                                                 math.sqrt(float(np.sum(f_want[worst] ** 2))), frms)
    fs_scale = max(float(np.max(np.abs(want["fshift"]))), frms)
    fserr = float(np.max(np.abs(np.asarray(got["fshift"]) - np.asarray(want["fshift"]))))
    assert fserr <= 10 * rel * fs_scale, "%s fshift max err %.3e (scale %.3e)" % (label, fserr, fs_scale)
    if energy:
        for k in ("e_lj", "e_el", "dvdl_coul", "dvdl_vdw"):
            scale = max(abs(want[k]), 1e-3 * (abs(want["e_lj"]) + abs(want["e_el"])), 1.0)
            assert abs(got[k] - want[k]) <= rel * scale, "%s %s got %.8g want %.8g" % (label, k, got[k], want[k])


def assert_foreign(got, want, rel=1e-4):
    """foreign-lambda energies and dV/dlambda of a dH/dlambda step against run_oracle(..., foreign=True)"""
    fw = want["foreign"]
    e_want = fw["eVdw"] + fw["eCoul"]
    scale = max(1.0, float(np.max(np.abs(fw["eVdw"])) + np.max(np.abs(fw["eCoul"]))))
    assert np.max(np.abs(got["foreign"]["energies"] - e_want)) <= rel * scale
    assert np.max(np.abs(got["foreign"]["dvdlCoul"] - fw["dvdlCoul"])) <= rel * max(1.0, float(np.max(np.abs(fw["dvdlCoul"]))))
    assert np.max(np.abs(got["foreign"]["dvdlVdw"] - fw["dvdlVdw"])) <= rel * max(1.0, float(np.max(np.abs(fw["dvdlVdw"]))))


# ---- O(N^2) evaluation (tiny systems) ------------------------------------------------------------------------

def brute_force(c, use_state="A"):
    """Minimum-image all-pairs evaluation of the non-perturbed Hamiltonian (state A parameters)
    with the same functional forms (cut-off/RF/Ewald real space + exclusion corrections + self term,
    LJ with potential shift / switches).  Returns forces in TOPOLOGY order."""
    x = c.grid.x_wrapped.astype(np.float64)
    n = len(x)
    box = c.grid.box.astype(np.float64)
    q = c.sys["qA" if use_state == "A" else "qB"].astype(np.float64)
    t = c.sys["typeA" if use_state == "A" else "typeB"]
    nbfp = np.asarray(c.sys["nbfp"], np.float64).reshape(c.ntype, c.ntype, 2)
    mol = c.sys["molId"]
    f = np.zeros((n, 3))
    e_lj = e_el = 0.0
    ljt = lj_type_params(c).astype(np.float64) if c.vdw in ("ewald_geom", "ewald_lb") else None
    erf = np.vectorize(math.erf)
    rc2 = c.rc * c.rc
    for i in range(n - 1):
        d = x[i] - x[i + 1:]
        d -= box * np.rint(d / box)
        r2 = np.einsum("ij,ij->i", d, d)
        m = r2 < rc2
        if not m.any():
            continue
        idx = np.flatnonzero(m) + i + 1
        d, r2 = d[m], r2[m]
        r = np.sqrt(r2)
        inc = (mol[idx] != mol[i]).astype(np.float64)
        qq = c.epsfac * q[i] * q[idx]
        c6 = nbfp[t[i], t[idx], 0]
        c12 = nbfp[t[i], t[idx], 1]
        rinv = 1.0 / r
        rinv6 = rinv ** 6 * inc
        F = rinv6 * (c12 * rinv6 - c6) * rinv ** 2
        E = inc * (c12 * (rinv6 ** 2 + c.rep_shift[2]) / 12 - c6 * (rinv6 + c.disp_shift[2]) / 6)
        if c.vdw in ("pswitch", "fswitch"):
            rsw = np.maximum(r - c.rvdw_switch, 0.0)
            if c.vdw == "fswitch":
                F += (-c6 * (c.disp_shift[0] + c.disp_shift[1] * rsw) + c12 * (c.rep_shift[0] + c.rep_shift[1] * rsw)) * rsw ** 2 * rinv
                E += (c6 * (c.disp_shift[0] / 3 + c.disp_shift[1] / 4 * rsw) - c12 * (c.rep_shift[0] / 3 + c.rep_shift[1] / 4 * rsw)) * rsw ** 3
            else:
                sw = 1 + (c.vdw_switch[0] + (c.vdw_switch[1] + c.vdw_switch[2] * rsw) * rsw) * rsw ** 3
                dsw = (3 * c.vdw_switch[0] + (4 * c.vdw_switch[1] + 5 * c.vdw_switch[2] * rsw) * rsw) * rsw ** 2
                F = F * sw - rinv * E * dsw
                E = E * sw
        if c.vdw in ("ewald_geom", "ewald_lb"):
            # real-space part of the LJ-PME grid term, E = c6grid / 6 (g(r) + shift), g = (1 - exp(-b^2 r^2) P(b^2 r^2)) / r^6,
            # for excluded pairs too (without the shift); the force as -dE/dr / r with dg/dr = -6 g / r + exp(-b^2 r^2) b^6 / r
            ca, cb = ljt[t[i]], ljt[t[idx]]
            if c.vdw == "ewald_geom":
                c6g = ca[0] * cb[:, 0]
            else:
                c6g = ca[1] * cb[:, 1] * (ca[0] + cb[:, 0]) ** 6
            b2 = c.beta_lj ** 2
            ex = np.exp(-b2 * r2)
            gr = (1 - ex * (1 + b2 * r2 + 0.5 * (b2 * r2) ** 2)) / r2 ** 3
            E += c6g / 6 * (gr + c.sh_lj_ewald * inc)
            F += c6g / 6 * (6 * gr / r2 - ex * b2 ** 3 / r2)
        if c.rvdw < c.rc:
            inside = (r2 < c.rvdw ** 2).astype(np.float64)
            F, E = F * inside, E * inside
        if c.elec in ("ewald", "ewald_tab"):
            br = c.beta * r
            F += qq * (inc * rinv ** 3 + (2 / math.sqrt(math.pi) * br * np.exp(-br * br) - erf(br)) / (br ** 3) * c.beta ** 3)
            e_el += float(np.sum(qq * (rinv * (inc - erf(br)) - inc * c.sh_ewald)))
        elif c.elec == "rf":
            F += qq * (inc * rinv ** 3 - 2 * c.k_rf)
            e_el += float(np.sum(qq * (inc * rinv + c.k_rf * r2 - c.c_rf)))
        else:
            F += qq * inc * rinv ** 3
            e_el += float(np.sum(qq * (inc * rinv - c.c_rf)))
        e_lj += float(np.sum(E))
        fv = d * F[:, None]
        f[i] += fv.sum(axis=0)
        np.subtract.at(f, idx, fv)
    if c.vdw in ("ewald_geom", "ewald_lb"):
        # every atom's pair with itself on the grid: g(0) = b^6 / 6
        e_lj += float(np.sum(nbfp[t, t, 0])) * 0.5 / 6 * c.beta_lj ** 6 / 6
    q2 = float(np.sum(q * q)) * c.epsfac
    if c.elec in ("ewald", "ewald_tab"):
        e_el += -q2 * c.beta / math.sqrt(math.pi)
    else:
        e_el += -0.5 * q2 * c.c_rf
    return dict(f=f, e_lj=e_lj, e_el=e_el)
