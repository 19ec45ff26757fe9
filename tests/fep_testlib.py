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
import importlib  # noqa: E402

workload = importlib.import_module("gromacs_fep_gpu_amd.workload")
# the product-side half (case construction, GPU call sequence) lives in the package; the oracle side is below
from gromacs_fep_gpu_amd.workload import (ONE_4PI_EPS0, make_case, add_oxygen_type_variants, lj_grid_table, lj_type_params,  # noqa: E402,F401
                                          lj_comb_params, force_switch_constants, gpu_interaction_params, ewald_force_table,
                                          setup_gpu, run_gpu)


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


def oracle_ref_params(c, analytical_ewald=False):
    """analytical_ewald: evaluate the Ewald correction of the tabulated flavours in closed form (comparisons with all-pairs sums)"""
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
    if c.elec == "ewald_tab" and not analytical_ewald:
        # the table the GPU is handed, interpolated the same way (kernel_gpu_ref.cpp:265-271)
        tab, scale = ewald_force_table(c.beta, c.rc + 0.3)
        p._tab_keepalive = tab
        p.coulomb_tab = tab.ctypes.data_as(ob.C.POINTER(ob.C.c_float))
        p.coulomb_tab_size, p.coulomb_tab_scale = len(tab), scale
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
               parts=dict(ref=ref, fep=fep), fep_abs_sums=fep["abs_sums"], fshift_abs=ref["fshift_abs"] + fep["fshift_abs"])
    if foreign and c.n_lambda > 0:
        out["foreign"] = ob.fep_foreign(c.plist.fep, g.x_wrapped, c.ntype, fp, g.shift_vec, c.sys["nbfp"], lj_grid_table(c),
                                        c.sys["qA"], c.sys["qB"], c.sys["typeA"], c.sys["typeB"], c.lambda_coul,
                                        c.lambda_vdw, c.all_lambda, c.all_lambda, precision)
    return out


# ---- GPU through the C ABI -----------------------------------------------------------------------------

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
    # shift forces: each component of each shift vector against the larger of its value and the sum of |f_i| over the i-atoms booked
    # to it — the magnitude of ITS terms, returned by the oracles (oracle_nbnxm_fshift_abs, oracle_fep_last_fshift_abs) — at rel
    fs_want = np.asarray(want["fshift"], np.float64)
    fs_err = np.abs(np.asarray(got["fshift"], np.float64) - fs_want)
    if want.get("fshift_abs") is not None:
        fs_tol = rel * np.maximum(np.maximum(np.abs(fs_want), np.asarray(want["fshift_abs"])), 1e-3 * frms)
    else:   # (a reference without the term sums, e.g. a summed multi-domain result: the largest shift force and the rms force)
        fs_tol = np.full_like(fs_err, rel * max(float(np.max(np.abs(fs_want))), frms))
    w = np.unravel_index(int(np.argmax(fs_err / fs_tol)), fs_err.shape)
    assert fs_err[w] <= fs_tol[w], "%s fshift[%d][%d] err %.3e (tolerance %.3e, value %.3e)" % (label, w[0], w[1], fs_err[w], fs_tol[w], fs_want[w])
    if energy:
        # dV/dlambda is a sum over the perturbed pairs only: measured against the magnitude of ITS terms (the oracle's sum of
        # |dV/dl| per pair), never against the energy of the whole box; energies against their own value
        sums = want.get("fep_abs_sums")
        for k in ("e_lj", "e_el", "dvdl_coul", "dvdl_vdw"):
            scale = max(abs(want[k]), sums[k] if (sums and k.startswith("dvdl")) else 0.0, 1e-3 if k.startswith("dvdl") else 1.0)
            assert abs(got[k] - want[k]) <= rel * scale, "%s %s got %.8g want %.8g (scale %.3g)" % (label, k, got[k], want[k], scale)


def assert_foreign(got, want, rel=1e-4):
    """foreign-lambda energies and dV/dlambda of a dH/dlambda step against run_oracle(..., foreign=True)"""
    fw = want["foreign"]
    e_want = fw["eVdw"] + fw["eCoul"]
    # per lambda: the energy against the magnitude of ITS two terms at THAT lambda, dV/dl against its own value at that lambda (with
    # the floors of assert_parity) — not against the largest value over all lambda
    for k in range(len(e_want)):
        scale = max(1.0, abs(e_want[k]), fw["eVdwAbs"][k] + fw["eCoulAbs"][k])      # sum of |V| over the perturbed pairs at lambda k
        assert abs(got["foreign"]["energies"][k] - e_want[k]) <= rel * scale, \
            "foreign energy, index %d: got %.8g want %.8g" % (k, got["foreign"]["energies"][k], e_want[k])
        for name in ("dvdlCoul", "dvdlVdw"):
            sums = fw.get(name + "Abs")
            scale = max(abs(fw[name][k]), sums[k] if sums is not None else 1.0, 1e-3)
            assert abs(got["foreign"][name][k] - fw[name][k]) <= rel * scale, \
                "foreign %s, index %d: got %.8g want %.8g (scale %.3g)" % (name, k, got["foreign"][name][k], fw[name][k], scale)


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
