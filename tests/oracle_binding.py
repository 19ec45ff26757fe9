"""ctypes binding of the CPU oracle (oracle/build/liboracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (gromacs-fep-gpu_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "build", "liboracle.so")

ONE_4PI_EPS0 = 138.93545764438198  # gmx::c_one4PiEps0, api/legacy/include/gromacs/math/units.h:110

DO_FORCE, DO_SHIFTFORCE, DO_POTENTIAL = 1, 2, 4
SOFTCORE_BEUTLER, SOFTCORE_GAPSYS = 0, 1


class FepParams(C.Structure):
    _fields_ = [
        ("elecIsEwald", C.c_int), ("vdwIsEwald", C.c_int), ("vdwPotSwitch", C.c_int),
        ("epsfac", C.c_double),
        ("rcoulomb", C.c_double), ("rvdw", C.c_double), ("rvdw_switch", C.c_double),
        ("k_rf", C.c_double), ("c_rf", C.c_double),
        ("ewaldcoeff_q", C.c_double), ("ewaldcoeff_lj", C.c_double),
        ("sh_ewald", C.c_double), ("sh_lj_ewald", C.c_double),
        ("dispersion_shift_cpot", C.c_double), ("repulsion_shift_cpot", C.c_double),
        ("softcoreType", C.c_int),
        ("alphaVdw", C.c_double), ("alphaCoulomb", C.c_double),
        ("lambdaPower", C.c_int),
        ("sigma6WithInvalidSigma", C.c_double), ("sigma6Minimum", C.c_double),
        ("gapsysScaleLinpointVdW", C.c_double), ("gapsysScaleLinpointCoul", C.c_double),
        ("gapsysSigma6VdW", C.c_double),
    ]


class RefParams(C.Structure):
    _fields_ = [
        ("elecType", C.c_int), ("vdwType", C.c_int),
        ("epsfac", C.c_double), ("c_rf", C.c_double), ("k_rf", C.c_double),
        ("ewaldcoeff_q", C.c_double), ("sh_ewald", C.c_double), ("sh_lj_ewald", C.c_double),
        ("ewaldcoeff_lj", C.c_double),
        ("rcoulomb", C.c_double), ("rvdw", C.c_double), ("rvdw_switch", C.c_double),
        ("rlist", C.c_double),
        ("disp_c2", C.c_double), ("disp_c3", C.c_double), ("disp_cpot", C.c_double),
        ("rep_c2", C.c_double), ("rep_c3", C.c_double), ("rep_cpot", C.c_double),
        ("sw_c3", C.c_double), ("sw_c4", C.c_double), ("sw_c5", C.c_double),
        ("coulomb_tab_scale", C.c_double), ("coulomb_tab", C.POINTER(C.c_float)), ("coulomb_tab_size", C.c_int),
    ]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.oracle_calc_ewaldcoeff_q.restype = C.c_double
        _lib.oracle_calc_ewaldcoeff_q.argtypes = [C.c_double, C.c_double]
        _lib.oracle_calc_ewaldcoeff_lj.restype = C.c_double
        _lib.oracle_calc_ewaldcoeff_lj.argtypes = [C.c_double, C.c_double]
        _lib.oracle_nbnxm_prune.restype = C.c_longlong
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def softcore_params(p, sc_alpha, sc_power, sc_sigma, sc_sigma_min, sc_coul, softcore_type=SOFTCORE_BEUTLER,
                    gapsys_lj=0.85, gapsys_q=0.3, gapsys_sigma=0.3):
    lib().oracle_softcore_from_fepvals(C.byref(p), C.c_double(sc_alpha), C.c_int(sc_power),
                                       C.c_double(sc_sigma), C.c_double(sc_sigma_min),
                                       C.c_int(1 if sc_coul else 0), C.c_int(softcore_type),
                                       C.c_double(gapsys_lj), C.c_double(gapsys_q),
                                       C.c_double(gapsys_sigma))
    return p


def fep_kernel(nbl, x, ntype, p, shiftvec, nbfp, nbfp_grid, qA, qB, typeA, typeB, flags,
               lambda_coul, lambda_vdw, precision="f64"):
    """nbl: dict(iinr, jindex, jjnr, shift, excl_fep|None) with atom ids indexing x (N x 3).
    Returns dict(f, fshift, Vc, Vv, dvdl_coul, dvdl_vdw)."""
    rt = np.float64 if precision == "f64" else np.float32
    fn = getattr(lib(), "oracle_nb_free_energy_kernel_" + precision)
    iinr = _arr(nbl["iinr"], np.int32)
    jindex = _arr(nbl["jindex"], np.int32)
    jjnr = _arr(nbl["jjnr"], np.int32)
    shift = _arr(nbl["shift"], np.int32)
    excl = _arr(nbl.get("excl_fep"), np.int32)
    x = _arr(x, rt)
    sv = _arr(shiftvec, rt)
    nbfp_ = _arr(nbfp, rt)
    grid_ = _arr(nbfp_grid if nbfp_grid is not None else np.zeros_like(nbfp_), rt)
    qA_, qB_ = _arr(qA, rt), _arr(qB, rt)
    tA, tB = _arr(typeA, np.int32), _arr(typeB, np.int32)
    n = x.shape[0]
    f = np.zeros((n, 3), rt)
    fshift = np.zeros((sv.reshape(-1, 3).shape[0], 3), rt)
    Vc, Vv = C.c_double(0), C.c_double(0)
    dvdl = (C.c_double * 2)(0, 0)
    fn(C.c_int(len(iinr)), _ptr(iinr), _ptr(jindex), _ptr(jjnr), _ptr(shift), _ptr(excl), _ptr(x),
       C.c_int(ntype), C.byref(p), _ptr(sv), _ptr(nbfp_), _ptr(grid_), _ptr(qA_), _ptr(qB_),
       _ptr(tA), _ptr(tB), C.c_int(flags), C.c_double(lambda_coul), C.c_double(lambda_vdw),
       _ptr(f), _ptr(fshift), C.byref(Vc), C.byref(Vv), dvdl)
    sums = (C.c_double * 4)()
    getattr(lib(), "oracle_fep_last_abs_sums_" + precision)(sums)
    fsabs = np.zeros((45, 3))
    getattr(lib(), "oracle_fep_last_fshift_abs_" + precision)(_ptr(fsabs))
    return dict(f=f, fshift=fshift, Vc=Vc.value, Vv=Vv.value, dvdl_coul=dvdl[0], dvdl_vdw=dvdl[1],
                abs_sums=dict(e_el=sums[0], e_lj=sums[1], dvdl_coul=sums[2], dvdl_vdw=sums[3]), fshift_abs=fsabs)


def fep_foreign(nbl, x, ntype, p, shiftvec, nbfp, nbfp_grid, qA, qB, typeA, typeB, lambda_coul,
                lambda_vdw, all_lambda_coul, all_lambda_vdw, precision="f64"):
    rt = np.float64 if precision == "f64" else np.float32
    fn = getattr(lib(), "oracle_fep_foreign_" + precision)
    iinr = _arr(nbl["iinr"], np.int32)
    jindex = _arr(nbl["jindex"], np.int32)
    jjnr = _arr(nbl["jjnr"], np.int32)
    shift = _arr(nbl["shift"], np.int32)
    excl = _arr(nbl.get("excl_fep"), np.int32)
    x = _arr(x, rt)
    sv = _arr(shiftvec, rt)
    nbfp_ = _arr(nbfp, rt)
    grid_ = _arr(nbfp_grid if nbfp_grid is not None else np.zeros_like(nbfp_), rt)
    qA_, qB_ = _arr(qA, rt), _arr(qB, rt)
    tA, tB = _arr(typeA, np.int32), _arr(typeB, np.int32)
    alc = _arr(all_lambda_coul, np.float64)
    alv = _arr(all_lambda_vdw, np.float64)
    nl = len(alc)
    out = [np.zeros(nl + 1) for _ in range(4)]
    fn(C.c_int(len(iinr)), _ptr(iinr), _ptr(jindex), _ptr(jjnr), _ptr(shift), _ptr(excl), _ptr(x),
       C.c_int(ntype), C.byref(p), _ptr(sv), _ptr(nbfp_), _ptr(grid_), _ptr(qA_), _ptr(qB_),
       _ptr(tA), _ptr(tB), C.c_double(lambda_coul), C.c_double(lambda_vdw), C.c_int(nl),
       _ptr(alc), _ptr(alv), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]))
    sums = np.zeros((nl + 1, 4))
    getattr(lib(), "oracle_fep_foreign_abs_sums_" + precision)(_ptr(sums), C.c_int(nl + 1))
    return dict(eVdw=out[0], eCoul=out[1], dvdlVdw=out[2], dvdlCoul=out[3], eCoulAbs=sums[:, 0], eVdwAbs=sums[:, 1],
                dvdlCoulAbs=sums[:, 2], dvdlVdwAbs=sums[:, 3])


def nbnxm_ref(sci, cjPacked, excl, xq, atype, ntype, nbfp, p, shiftvec, compute_energy=True,
              compute_fshift=True, lj_comb=None, nbfp_comb=None, precision="f64", num_threads=1):
    """sci/cjPacked/excl: numpy structured or raw int32/uint32 arrays with the ABI layout.
    num_threads > 1: the OpenMP variant (bench.py's CPU baseline)."""
    rt = np.float64 if precision == "f64" else np.float32
    fn = getattr(lib(), ("oracle_nbnxm_ref_mt_" if num_threads > 1 else "oracle_nbnxm_ref_") + precision)
    sci = np.ascontiguousarray(sci)
    cjPacked = np.ascontiguousarray(cjPacked)
    excl = np.ascontiguousarray(excl)
    xq_ = _arr(xq, rt)
    n = xq_.reshape(-1, 4).shape[0]
    t = _arr(atype, np.int32)
    nbfp_ = _arr(nbfp, rt)
    ljc = _arr(lj_comb, rt)
    nbc = _arr(nbfp_comb, rt)
    sv = _arr(shiftvec, rt)
    f = np.zeros((n, 3), rt)
    fshift = np.zeros((45, 3), rt)
    Vc, Vv = C.c_double(0), C.c_double(0)
    npair = C.c_longlong(0)
    nsci = sci.size if sci.dtype.names else sci.reshape(-1, 4).shape[0]
    lead = (C.c_int(num_threads), C.c_int(n)) if num_threads > 1 else ()
    fsabs_fn = getattr(lib(), "oracle_nbnxm_fshift_abs_" + precision)
    fsabs_fn(None, C.c_int(1))
    fn(*lead, C.c_int(nsci), _ptr(sci), _ptr(cjPacked), _ptr(excl), _ptr(xq_), _ptr(t), C.c_int(ntype),
       _ptr(nbfp_), _ptr(ljc), _ptr(nbc), C.byref(p), _ptr(sv), C.c_int(1 if compute_energy else 0),
       C.c_int(1 if compute_fshift else 0), _ptr(f), _ptr(fshift), C.byref(Vc), C.byref(Vv),
       C.byref(npair))
    fsabs = np.zeros((45, 3))
    fsabs_fn(_ptr(fsabs), C.c_int(0))
    return dict(f=f, fshift=fshift, Vc=Vc.value, Vv=Vv.value, npairs=npair.value, fshift_abs=fsabs)


def nbnxm_simd(sci, cjPacked, excl, xq, atype, ntype, nbfp, p, shiftvec, num_threads=1):
    """Throughput port of the force-only cluster kernel (oracle/nbnxm_simd.c, CPU baseline of bench.py); returns the forces in
    grid order, or None when the CPU or the flavour is not supported."""
    sci = np.ascontiguousarray(sci)
    cjPacked = np.ascontiguousarray(cjPacked)
    excl = np.ascontiguousarray(excl)
    xq_ = _arr(xq, np.float32)
    n = xq_.reshape(-1, 4).shape[0]
    f = np.zeros((n, 3), np.float32)
    nsci = sci.size if sci.dtype.names else sci.reshape(-1, 4).shape[0]
    rc = lib().oracle_nbnxm_simd_f32(C.c_int(num_threads), C.c_int(n), C.c_int(nsci), _ptr(sci), _ptr(cjPacked), _ptr(excl), _ptr(xq_),
                                     _ptr(_arr(atype, np.int32)), C.c_int(ntype), _ptr(_arr(nbfp, np.float32)), C.byref(p),
                                     _ptr(_arr(shiftvec, np.float32)), _ptr(f))
    return f if rc == 0 else None


def nbnxm_prune(sci, cjPacked, xq, shiftvec, rlist):
    """Prunes cjPacked IN PLACE (imask bits); returns the number of cluster pairs left."""
    sci = np.ascontiguousarray(sci)
    assert cjPacked.flags["C_CONTIGUOUS"]
    xq_ = _arr(xq, np.float32)
    sv = _arr(shiftvec, np.float32)
    nsci = sci.size if sci.dtype.names else sci.reshape(-1, 4).shape[0]
    return lib().oracle_nbnxm_prune(C.c_int(nsci), _ptr(sci), _ptr(cjPacked), _ptr(xq_), _ptr(sv),
                                    C.c_double(rlist))


# ---- listed (bonded) interactions with A/B parameters: oracle/listed_ref.h --------------------------------
LISTED_TYPES = {"bonds": 0, "angles": 1, "urey_bradley": 2, "pdihs": 3, "rbdihs": 4, "idihs": 5, "restrbonds": 6, "angres": 7, "dihres": 8}
LISTED_NRAL = {"bonds": 2, "angles": 3, "urey_bradley": 3, "pdihs": 4, "rbdihs": 4, "idihs": 4, "restrbonds": 2, "angres": 4, "dihres": 4}
LISTED_IPARAMS = np.dtype([("p", np.float64, 12), ("mult", np.int32), ("pad", np.int32)])


def listed_iparams(type_name, prm):
    """One listed_iparams_t from the reference test's parameter names (tests/golden/make_listed_golden.py)."""
    ip = np.zeros(1, LISTED_IPARAMS)
    if type_name in ("bonds", "angles", "idihs"):
        ip["p"][0, :4] = [prm["rA"], prm["krA"], prm["rB"], prm["krB"]]
    elif type_name == "urey_bradley":
        ip["p"][0, :8] = [prm[k] for k in ("thetaA", "kthetaA", "r13A", "kUBA", "thetaB", "kthetaB", "r13B", "kUBB")]
    elif type_name in ("pdihs", "angres"):
        ip["p"][0, :4] = [prm["phiA"], prm["cpA"], prm["phiB"], prm["cpB"]]
        ip["mult"][0] = prm["mult"]
    elif type_name == "rbdihs":
        ip["p"][0, :6] = prm["rbcA"]
        ip["p"][0, 6:] = prm["rbcB"]
    elif type_name == "restrbonds":
        ip["p"][0, :8] = [prm[k] for k in ("lowA", "up1A", "up2A", "kA", "lowB", "up1B", "up2B", "kB")]
    elif type_name == "dihres":
        ip["p"][0, :6] = [prm[k] for k in ("phiA", "dphiA", "kfacA", "phiB", "dphiB", "kfacB")]
    else:
        raise ValueError(type_name)
    return ip


def listed(type_name, iatoms, params, x, box, npbcdim, lam, want_fshift=True):
    """iatoms: (n, 1 + nral) int32 rows [parameter index, atoms...]; returns dict(f, fshift, epot, dvdl)."""
    ia = _arr(iatoms, np.int32)
    prm = np.ascontiguousarray(params)
    x_ = _arr(x, np.float64)
    n = x_.reshape(-1, 3).shape[0]
    f = np.zeros((n, 3), np.float64)
    fshift = np.zeros((45, 3), np.float64)
    box_ = _arr(box, np.float64)
    epot, dvdl = C.c_double(0), C.c_double(0)
    lib().oracle_listed(C.c_int(LISTED_TYPES[type_name]), C.c_int(ia.reshape(-1, 1 + LISTED_NRAL[type_name]).shape[0]), _ptr(ia),
                        _ptr(prm), _ptr(x_), _ptr(box_), C.c_int(npbcdim), C.c_double(lam), _ptr(f),
                        _ptr(fshift) if want_fshift else None, C.byref(epot), C.byref(dvdl))
    return dict(f=f, fshift=fshift, epot=epot.value, dvdl=dvdl.value)


def listed_simple_pairs(kind, iatoms, params, x, box, npbcdim, epsfac):
    """kind 1 = F_LJC14_Q (p: qi qj fqq c6 c12), 2 = F_LJC_PAIRS_NB (p: qi qj c6 c12); returns dict(f, fshift, e_lj, e_coul)"""
    ia = _arr(iatoms, np.int32)
    prm = np.ascontiguousarray(params)
    x_ = _arr(x, np.float64)
    f = np.zeros((x_.reshape(-1, 3).shape[0], 3), np.float64)
    fshift = np.zeros((45, 3), np.float64)
    e_lj, e_coul = C.c_double(0), C.c_double(0)
    lib().oracle_listed_simple_pairs(C.c_int(kind), C.c_int(ia.reshape(-1, 3).shape[0]), _ptr(ia), _ptr(prm), _ptr(x_),
                                     _ptr(_arr(box, np.float64)), C.c_int(npbcdim), C.c_double(epsfac), _ptr(f), _ptr(fshift),
                                     C.byref(e_lj), C.byref(e_coul))
    return dict(f=f, fshift=fshift, e_lj=e_lj.value, e_coul=e_coul.value)


class ListedPairsFep(C.Structure):
    _fields_ = [("alphaCoul", C.c_double), ("alphaVdw", C.c_double), ("lambdaPower", C.c_int), ("pad", C.c_int),
                ("sc_sigma6", C.c_double), ("sc_sigma6_min", C.c_double), ("lambdaCoul", C.c_double), ("lambdaVdw", C.c_double)]


def listed_pairs(iatoms, params, x, qA, qB, box, npbcdim, fep, elec_scale):
    """iatoms: (n, 3) rows [parameter index, ai, aj]; params: LISTED_IPARAMS with p[:4] = c6A c12A c6B c12B"""
    ia = _arr(iatoms, np.int32)
    prm = np.ascontiguousarray(params)
    x_ = _arr(x, np.float64)
    n = x_.reshape(-1, 3).shape[0]
    f = np.zeros((n, 3), np.float64)
    fshift = np.zeros((45, 3), np.float64)
    out = [C.c_double(0) for _ in range(4)]
    lib().oracle_listed_pairs(C.c_int(ia.reshape(-1, 3).shape[0]), _ptr(ia), _ptr(prm), _ptr(x_), _ptr(_arr(qA, np.float64)),
                              _ptr(_arr(qB, np.float64)), _ptr(_arr(box, np.float64)), C.c_int(npbcdim), C.byref(fep),
                              C.c_double(elec_scale), _ptr(f), _ptr(fshift), *[C.byref(o) for o in out])
    return dict(f=f, fshift=fshift, eLJ=out[0].value, eCoul=out[1].value, dvdlVdw=out[2].value, dvdlCoul=out[3].value)


# ---- Langevin update: oracle/langevin_ref.h -----------------------------------------------------------------
def threefry2x64(key0, key1, ctr0, ctr1):
    out = (C.c_uint64 * 2)()
    lib().oracle_threefry2x64(C.c_uint64(key0), C.c_uint64(key1), C.c_uint64(ctr0), C.c_uint64(ctr1), out)
    return int(out[0]), int(out[1])


def normal_table(bits=14):
    t = np.zeros(1 << bits, np.float32)
    lib().oracle_normal_table(C.c_int(bits), _ptr(t))
    return t


def tabulated_normal(key0, domain, internal_counter_bits, ctr0, ctr1, mean, stddev, n):
    out = np.zeros(n, np.float32)
    lib().oracle_tabulated_normal(C.c_uint64(key0), C.c_uint64(domain), C.c_int(internal_counter_bits), C.c_uint64(ctr0),
                                  C.c_uint64(ctr1), C.c_float(mean), C.c_float(stddev), C.c_int(n), _ptr(out))
    return out


def langevin_update(update_type, x, v, f, inverse_masses, tc_groups, ref_t, tau_t, dt, seed, step):
    """returns (x, xp, v) after the update; update_type 0 = forces only, 1 = friction and noise only"""
    x_ = np.array(x, np.float32).copy()
    v_ = np.array(v, np.float32).copy()
    xp = np.zeros_like(x_)
    n = x_.shape[0]
    lib().oracle_langevin_update(C.c_int(update_type), C.c_int(n), _ptr(x_), _ptr(xp), _ptr(v_), _ptr(_arr(f, np.float32)),
                                 _ptr(_arr(inverse_masses, np.float32)), _ptr(_arr(tc_groups, np.uint16)), C.c_int(len(ref_t)),
                                 _ptr(_arr(ref_t, np.float32)), _ptr(_arr(tau_t, np.float32)), C.c_float(dt), C.c_int(seed), C.c_int(step))
    return x_, xp, v_


def _dptr(a):
    return a.ctypes.data_as(C.c_void_p)


def leapfrog(x, v, f, inverse_masses, dt, lambdas=None, groups=None, pr_diag=None):
    """one leap-frog step (double); lambdas: None, [l] or per-group factors with groups; pr_diag = dtPressureCouple * diag(M).
    Returns (x, xp, v)."""
    x_ = np.array(x, np.float64).copy()
    v_ = np.array(v, np.float64).copy()
    xp = np.zeros_like(x_)
    n = x_.shape[0]
    lam = _arr(lambdas if lambdas is not None else [1.0], np.float64)
    grp = _arr(groups if groups is not None else np.zeros(n), np.uint16)
    prd = _arr(pr_diag, np.float64) if pr_diag is not None else None
    lib().oracle_leapfrog(C.c_int(n), _dptr(x_), _dptr(xp), _dptr(v_), _dptr(_arr(f, np.float64)), _dptr(_arr(inverse_masses, np.float64)),
                          C.c_double(dt), C.c_int(0 if lambdas is None else len(lam)), _dptr(lam), _dptr(grp),
                          _dptr(prd) if prd is not None else None)
    return x_, xp, v_


def settle(atoms, mO, mH, dOH, dHH, x, xp, v=None, invdt=0.0, compute_virial=False, pbc_type=3, box=None):
    """returns (xp, v, virial[3,3])"""
    at = _arr(atoms, np.int32).reshape(-1)
    xp_ = np.array(xp, np.float64).copy()
    v_ = np.array(v, np.float64).copy() if v is not None else None
    vir = np.zeros((3, 3))
    bx = _arr(box if box is not None else np.zeros((3, 3)), np.float64)
    lib().oracle_settle(C.c_int(len(at) // 3), _dptr(at), C.c_double(mO), C.c_double(mH), C.c_double(dOH), C.c_double(dHH),
                        _dptr(_arr(x, np.float64)), _dptr(xp_), _dptr(v_) if v_ is not None else None, C.c_double(invdt),
                        _dptr(vir) if compute_virial else None, C.c_int(pbc_type), _dptr(bx))
    return xp_, v_, vir


def lincs(iatoms, lengths, inverse_masses, num_iterations, expansion_order, x, xp, v=None, invdt=0.0, compute_virial=False, pbc_type=3,
          box=None):
    """returns (xp, v, virial[3,3])"""
    ia = _arr(iatoms, np.int32).reshape(-1)
    xp_ = np.array(xp, np.float64).copy()
    v_ = np.array(v, np.float64).copy() if v is not None else None
    vir = np.zeros((3, 3))
    im = _arr(inverse_masses, np.float64)
    bx = _arr(box if box is not None else np.zeros((3, 3)), np.float64)
    lib().oracle_lincs(C.c_int(len(ia) // 3), _dptr(ia), _dptr(_arr(lengths, np.float64)), C.c_int(len(im)), _dptr(im),
                       C.c_int(num_iterations), C.c_int(expansion_order), _dptr(_arr(x, np.float64)), _dptr(xp_),
                       _dptr(v_) if v_ is not None else None, C.c_double(invdt), _dptr(vir) if compute_virial else None,
                       C.c_int(pbc_type), _dptr(bx))
    return xp_, v_, vir
