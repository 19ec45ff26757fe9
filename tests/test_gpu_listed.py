"""GPU parity of the perturbed listed (bonded) interactions: the HIP kernels behind
include/listed_hip.h against the reference's own known answers and against the CPU oracle (oracle/listed_ref.c, itself
pinned by those known answers) on random molecules with periodic wrapping.  Tolerance 1e-4 relative of the largest force /
energy term (fp32 on the GPU, double in the oracle), the bar of the non-bonded path."""
import json
import os

import numpy as np
import pytest

import fep_testlib as tl
import oracle_binding as ob

pytestmark = pytest.mark.gpu
pkg = tl.pkg
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "listed_refdata.json")))
PBC = {"none": 0, "xy": 2, "xyz": 3}


def _gpu_params(prm64):
    out = np.zeros(prm64.size, pkg.LISTED_IPARAMS)
    out["p"] = prm64["p"].astype(np.float32)
    out["mult"] = prm64["mult"]
    return out


def _fep(lam, alpha=0.3, power=1, sigma=0.3, sigma_min=0.3, lam_coul=None, lam_vdw=None, lam_restraint=None):
    return pkg.ListedFepParams(alpha, alpha, power, sigma ** 6, sigma_min ** 6, lam, lam if lam_coul is None else lam_coul,
                               lam if lam_vdw is None else lam_vdw, lam if lam_restraint is None else lam_restraint)


def _run_gpu(lists, params64, x, box3, pbc_type, lam, energy=True, virial=True, qA=None, qB=None, elec_scale=0.0, fep=None, epsfac=0.0):
    """lists: {type name: (n, 1 + nral) int32}; returns f, fshift, energy terms, dvdl (bonded + restraint part), dvdl3 (all four)"""
    import torch
    n = x.shape[0]
    xq = np.zeros((n, 4), np.float32)
    xq[:, :3] = x
    d_xq = torch.from_numpy(xq).cuda()
    d_q4 = None
    if qA is not None:
        q4 = np.zeros((n, 4), np.float32)
        q4[:, 0], q4[:, 1] = qA, qB
        d_q4 = torch.from_numpy(q4).cuda()
    d_f = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    d_fs = torch.zeros((45, 3), dtype=torch.float32, device="cuda")
    lg = pkg.ListedGpu()
    lg.set_force_params(_gpu_params(params64))
    for name, ia in lists.items():
        lg.update_interaction_list(name, ia, n)
    assert lg.have_interactions() == any(len(ia) for ia in lists.values())
    torch.cuda.synchronize()
    box = np.diag(np.asarray(box3, np.float32))
    lg.launch_kernel(d_xq.data_ptr(), d_f.data_ptr(), d_fs.data_ptr(), box, pbc_type, fep or _fep(lam),
                     d_q4=d_q4.data_ptr() if d_q4 is not None else None, elec_scale=elec_scale, compute_energy=energy,
                     compute_virial=virial, epsfac=epsfac)
    epot, dvdl3 = lg.energies()
    torch.cuda.synchronize()
    out = dict(f=d_f.cpu().numpy(), fshift=d_fs.cpu().numpy(), epot=epot, dvdl=dvdl3[0] + dvdl3[3], dvdl3=dvdl3)
    lg.free()
    return out


@pytest.mark.parametrize("case", [pytest.param(c, id="%s-%d-%s" % (c["suite"], c["index"], c["pbc"])) for c in GOLD["cases"]])
def test_listed_gpu_matches_reference_known_answers(case):
    x = np.array(GOLD["coordinates"], np.float64)
    ia = np.array([[0] + t for t in case["iatoms"]], np.int32)
    prm = ob.listed_iparams(case["type"], case["params"])
    for lam_name, want in case["results"].items():
        got = _run_gpu({case["type"]: ia}, prm, x, [GOLD["box"]] * 3, PBC[case["pbc"]], float(lam_name))
        fw = np.array(want["forces"])
        scale = max(1.0, np.abs(fw).max())
        assert np.abs(got["f"] - fw).max() <= 1e-4 * scale
        t = pkg.LISTED_TYPES[case["type"]]
        assert abs(got["epot"][t] - want["epot"]) <= 1e-4 * max(1.0, abs(want["epot"]))
        assert abs(got["dvdl"] - want["dvdlambda"]) <= 1e-4 * max(1.0, abs(want["dvdlambda"]), scale)
        assert np.abs(got["epot"]).sum() == pytest.approx(abs(got["epot"][t]))   # nothing leaks into other types


def _random_system(seed, nmol=400, box=3.0):
    """nmol 6-atom chains with random A/B parameters of every type, coordinates wrapped into the box so that bonds
    cross the periodic boundaries"""
    rng = np.random.default_rng(seed)
    x = np.zeros((6 * nmol, 3))
    for m in range(nmol):
        p = rng.uniform(0, box, 3)
        for a in range(6):
            x[6 * m + a] = p
            p = p + rng.normal(0, 1, 3) * 0.09
    x = x % box
    nprm = 40
    prm = np.zeros(nprm, ob.LISTED_IPARAMS)
    kinds = ["bonds", "angles", "urey_bradley", "pdihs", "rbdihs", "idihs"]
    kind_of = [kinds[i % 6] for i in range(nprm)]
    for i, k in enumerate(kind_of):
        same = rng.random() < 0.3      # some interactions are not perturbed
        if k == "bonds":
            a = [rng.uniform(0.1, 0.2), rng.uniform(100, 500)]
            b = a if same else [rng.uniform(0.1, 0.2), rng.uniform(100, 500)]
            prm["p"][i, :4] = a + list(b)
        elif k in ("angles", "idihs"):
            a = [rng.uniform(60, 140) if k == "angles" else rng.uniform(-170, 170), rng.uniform(20, 80)]
            b = a if same else [a[0] + rng.uniform(-10, 10), rng.uniform(20, 80)]
            prm["p"][i, :4] = a + list(b)
        elif k == "urey_bradley":
            a = [rng.uniform(60, 140), rng.uniform(20, 80), rng.uniform(0.15, 0.3), rng.uniform(1, 10)]
            b = a if same else [a[0] + rng.uniform(-8, 8), rng.uniform(20, 80), rng.uniform(0.15, 0.3), rng.uniform(1, 10)]
            prm["p"][i, :8] = a + list(b)
        elif k == "pdihs":
            a = [rng.uniform(-180, 180), rng.uniform(1, 20)]
            b = a if same else [rng.uniform(-180, 180), rng.uniform(1, 20)]
            prm["p"][i, :4] = a + list(b)
            prm["mult"][i] = rng.integers(1, 5)
        else:
            a = list(rng.uniform(-10, 10, 6))
            b = a if same else list(rng.uniform(-10, 10, 6))
            prm["p"][i, :12] = a + b
    by_kind = {k: [i for i in range(nprm) if kind_of[i] == k] for k in kinds}
    lists = {k: [] for k in kinds}
    for m in range(nmol):
        o = 6 * m
        for a in range(5):
            lists["bonds"].append([rng.choice(by_kind["bonds"]), o + a, o + a + 1])
        for a in range(4):
            k = "angles" if rng.random() < 0.5 else "urey_bradley"
            lists[k].append([rng.choice(by_kind[k]), o + a, o + a + 1, o + a + 2])
        for a in range(3):
            k = ["pdihs", "rbdihs", "idihs"][rng.integers(0, 3)]
            lists[k].append([rng.choice(by_kind[k]), o + a, o + a + 1, o + a + 2, o + a + 3])
    return x, prm, {k: np.array(v, np.int32).reshape(-1, 1 + pkg.LISTED_NRAL[k]) for k, v in lists.items()}


@pytest.mark.parametrize("pbc", ["xyz", "xy", "none"])
@pytest.mark.parametrize("lam", [0.0, 0.35, 1.0])
def test_listed_gpu_random_molecules_against_oracle(pbc, lam):
    box = 3.0
    x, prm, lists = _random_system(seed=3)
    if pbc != "xyz":
        # without full PBC the molecules must be whole: undo the wrapping through the minimum image along the chain
        for m in range(x.shape[0] // 6):
            for a in range(1, 6):
                d = x[6 * m + a] - x[6 * m + a - 1]
                x[6 * m + a] -= box * np.round(d / box)
    got = _run_gpu(lists, prm, x, [box] * 3, PBC[pbc], lam)
    f = np.zeros_like(x)
    fs = np.zeros((45, 3))
    epot = np.zeros(pkg.LISTED_NUM_ENERGY_TERMS)
    dvdl = 0.0
    xf = x.astype(np.float32).astype(np.float64)     # the coordinates the GPU sees
    for k, ia in lists.items():
        r = ob.listed(k, ia, prm, xf, np.full(3, box), PBC[pbc], lam)
        f += r["f"]
        fs += r["fshift"]
        epot[pkg.LISTED_TYPES[k]] += r["epot"]
        dvdl += r["dvdl"]
    rms = np.sqrt((f ** 2).sum(axis=1).mean())
    err = np.abs(got["f"] - f)
    assert (err <= 1e-4 * np.maximum(np.linalg.norm(f, axis=1, keepdims=True), rms)).all()
    assert np.abs(got["epot"] - epot).max() <= 1e-4 * np.abs(epot).max()
    assert abs(got["dvdl"] - dvdl) <= 1e-4 * max(abs(dvdl), np.abs(epot).max())
    # shift forces are sums of +f / -f terms that largely cancel: the fp32 noise scales with the sum of magnitudes
    assert np.abs(got["fshift"] - fs).max() <= max(1e-4 * np.abs(fs).max(), 1e-6 * np.abs(f).sum())
    if pbc == "xyz":
        assert np.abs(fs).max() > 0          # bonds do cross the boundaries in this system


def test_listed_gpu_force_only_and_empty_lists():
    x, prm, lists = _random_system(seed=4, nmol=50)
    got = _run_gpu(lists, prm, x, [3.0] * 3, 3, 0.5, energy=False, virial=False)
    ref = _run_gpu(lists, prm, x, [3.0] * 3, 3, 0.5, energy=True, virial=True)
    assert np.abs(got["f"] - ref["f"]).max() <= 1e-5 * np.abs(ref["f"]).max()   # atomic ordering only
    assert not got["epot"].any() and got["dvdl"] == 0.0 and not got["fshift"].any()
    empty = {k: np.zeros((0, 1 + pkg.LISTED_NRAL[k]), np.int32) for k in lists}
    got = _run_gpu(empty, prm, x, [3.0] * 3, 3, 0.5)
    assert not got["f"].any()


@pytest.mark.parametrize("case", [pytest.param(c, id="pairs-%d-%s" % (c["index"], c["pbc"])) for c in GOLD["pairs"]["cases"]])
def test_listed_gpu_pairs_match_reference_known_answers(case):
    """perturbed 1-4 pairs (Beutler soft-core) at lambda 0, 0.5, 1 against the reference's 14Interaction known answers"""
    P = GOLD["pairs"]
    x = np.array(P["coordinates"], np.float64)
    ia = np.array([[0] + t for t in P["iatoms"]], np.int32)
    prm = np.zeros(1, ob.LISTED_IPARAMS)
    prm["p"][0, :4] = [case["params"][k] for k in ("c6A", "c12A", "c6B", "c12B")]
    qB = P["chargeB"] if case["fep"] else P["chargeA"]
    for lam_name, want in case["results"].items():
        lam = float(lam_name)
        got = _run_gpu({"lj14": ia}, prm, x, [P["box"]] * 3, PBC[case["pbc"]], lam, qA=P["chargeA"], qB=qB,
                       elec_scale=P["epsfac"] * P["fudgeQQ"], fep=_fep(lam, P["sc_alpha"], P["sc_power"], P["sc_sigma"], P["sc_sigma_min"]))
        fw = np.array(want["forces"])
        scale = max(1e-3, np.abs(fw).max())
        assert np.abs(got["f"] - fw).max() <= 1e-4 * scale
        assert abs(got["epot"][pkg.LISTED_TYPES["lj14"]] - want["eLJ"]) <= 1e-4 * max(1e-3, abs(want["eLJ"]))
        assert abs(got["epot"][pkg.LISTED_ENERGY_COULOMB14] - want["eCoul"]) <= 1e-4 * max(1e-3, abs(want["eCoul"]))
        assert abs(got["dvdl3"][pkg.LISTED_DVDL["coul"]] - want["dvdlCoul"]) <= 1e-4 * max(1e-3, abs(want["dvdlCoul"]))
        assert abs(got["dvdl3"][pkg.LISTED_DVDL["vdw"]] - want["dvdlVdw"]) <= 1e-4 * max(1e-3, abs(want["dvdlVdw"]))
        assert got["dvdl3"][pkg.LISTED_DVDL["bonded"]] == 0.0


@pytest.mark.parametrize("power", [1, 2])
def test_listed_gpu_random_pairs_against_oracle(power):
    """many 1-4 pairs with random A/B LJ parameters and charges (some hard-core, some vanishing, some unperturbed),
    different lambda_coul / lambda_vdw, periodic wrapping"""
    rng = np.random.default_rng(12)
    box, n = 3.0, 3000
    x = rng.uniform(0, box, (n, 3))
    nprm = 12
    prm = np.zeros(nprm, ob.LISTED_IPARAMS)
    for i in range(nprm):
        a = [rng.uniform(1e-3, 3e-3), rng.uniform(1e-6, 3e-6)]
        kind = i % 4
        b = a if kind == 0 else ([0.0, 0.0] if kind == 1 else [rng.uniform(1e-3, 3e-3), rng.uniform(1e-6, 3e-6)])
        if kind == 3:
            a = [0.0, 0.0]
        prm["p"][i, :4] = a + list(b)
    qA = rng.uniform(-1, 1, n)
    qB = np.where(rng.random(n) < 0.5, qA, rng.uniform(-1, 1, n) * (rng.random(n) < 0.7))
    ai = rng.integers(0, n, 8000)
    aj = (ai + rng.integers(1, n - 1, 8000)) % n
    # keep realistic 1-4 distances: move j next to i
    x[aj] = (x[ai] + rng.normal(0, 1, (8000, 3)) * 0.12 + 0.15) % box
    ia = np.stack([rng.integers(0, nprm, 8000), ai, aj], axis=1).astype(np.int32)
    fep64 = ob.ListedPairsFep(0.5, 0.3, power, 0, 0.3 ** 6, 0.28 ** 6, 0.35, 0.6)
    fep32 = pkg.ListedFepParams(0.5, 0.3, power, 0.3 ** 6, 0.28 ** 6, 0.5, 0.35, 0.6, 0.0)
    xf = x.astype(np.float32).astype(np.float64)
    qAf, qBf = qA.astype(np.float32).astype(np.float64), qB.astype(np.float32).astype(np.float64)
    want = ob.listed_pairs(ia, prm, xf, qAf, qBf, np.full(3, box), 3, fep64, 138.935 * 0.5)
    got = _run_gpu({"lj14": ia}, prm, x, [box] * 3, 3, 0.5, qA=qA, qB=qB, elec_scale=138.935 * 0.5, fep=fep32)
    f = want["f"]
    rms = np.sqrt((f ** 2).sum(axis=1).mean())
    assert (np.abs(got["f"] - f) <= 1e-4 * np.maximum(np.linalg.norm(f, axis=1, keepdims=True), rms)).all()
    assert abs(got["epot"][pkg.LISTED_TYPES["lj14"]] - want["eLJ"]) <= 1e-4 * max(abs(want["eLJ"]), 1.0)
    assert abs(got["epot"][pkg.LISTED_ENERGY_COULOMB14] - want["eCoul"]) <= 1e-4 * max(abs(want["eCoul"]), 1.0)
    assert abs(got["dvdl3"][1] - want["dvdlCoul"]) <= 2e-4 * max(abs(want["dvdlCoul"]), abs(want["eCoul"]))
    assert abs(got["dvdl3"][2] - want["dvdlVdw"]) <= 2e-4 * max(abs(want["dvdlVdw"]), abs(want["eLJ"]))


def test_restraints_and_simple_pairs_match_oracle():
    """The remaining types of the fork's GPU list: flat-bottomed distance, angle and dihedral restraints (with their own
    lambda and dV/dlambda component) and the unperturbed pair types with per-interaction charges.  F_RESTRBONDS and F_DIHRES have
    no known answers in the reference's tests: the oracle they are compared with is checked by finite differences
    (tests/test_oracle_golden.py), F_ANGRES additionally by the reference's known answers above."""
    rng = np.random.default_rng(21)
    nmol, box = 600, 3.0
    x = np.zeros((4 * nmol, 3))
    for m in range(nmol):
        p = rng.uniform(0, box, 3)
        for a in range(4):
            x[4 * m + a] = p
            p = p + rng.normal(0, 1, 3) * rng.choice([0.08, 0.2, 0.45])
    x = (x % box).astype(np.float32).astype(np.float64)
    nprm = 30
    prm = np.zeros(nprm, ob.LISTED_IPARAMS)
    kinds = ["restrbonds", "angres", "dihres", "ljc14_q", "ljc_pairs_nb"]
    for i in range(nprm):
        k = kinds[i % 5]
        if k == "restrbonds":
            low = rng.uniform(0.05, 0.3)
            a = [low, low + rng.uniform(0.0, 0.1), low + rng.uniform(0.1, 0.3), rng.uniform(100, 900)]
            prm["p"][i, :8] = a + [a[0] * 1.1, a[1] * 1.05, a[2] * 1.2, rng.uniform(100, 900)]
        elif k == "angres":
            prm["p"][i, :4] = [rng.uniform(-180, 180), rng.uniform(1, 20), rng.uniform(-180, 180), rng.uniform(1, 20)]
            prm["mult"][i] = rng.integers(1, 4)
        elif k == "dihres":
            phi, dphi = rng.uniform(-180, 180), rng.uniform(0, 40)
            prm["p"][i, :6] = [phi, dphi, rng.uniform(10, 80), phi + rng.uniform(-10, 10), dphi + rng.uniform(0, 5), rng.uniform(10, 80)]
        elif k == "ljc14_q":
            prm["p"][i, :5] = [rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(0.5, 1.0), rng.uniform(0, 3e-3), rng.uniform(0, 3e-6)]
        else:
            prm["p"][i, :4] = [rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(0, 3e-3), rng.uniform(0, 3e-6)]
    prm["p"] = prm["p"].astype(np.float32).astype(np.float64)
    lists = {k: [] for k in kinds}
    for m in range(nmol):
        b = 4 * m
        t = lambda k: int(rng.choice([i for i in range(nprm) if kinds[i % 5] == k]))
        lists["restrbonds"] += [[t("restrbonds"), b, b + 3], [t("restrbonds"), b + 2, b + 1]]
        lists["angres"].append([t("angres"), b, b + 1, b + 2, b + 3])
        lists["dihres"].append([t("dihres"), b, b + 1, b + 2, b + 3])
        lists["ljc14_q"].append([t("ljc14_q"), b, b + 3])
        lists["ljc_pairs_nb"].append([t("ljc_pairs_nb"), b + 1, b + 3])
    lists = {k: np.array(v, np.int32) for k, v in lists.items()}
    epsfac, lam_r = 138.935, 0.35
    for pbc_type, npbc in ((3, 3), (0, 0)):
        got = _run_gpu(lists, prm, x, [box] * 3, pbc_type, 0.9, fep=_fep(0.9, lam_restraint=lam_r), epsfac=epsfac)
        f = np.zeros_like(x)
        fs = np.zeros((45, 3))
        dvdl_r = 0.0
        for k in ("restrbonds", "angres", "dihres"):
            r = ob.listed(k, lists[k], prm, x, np.full(3, box), npbc, lam_r)
            f += r["f"]
            fs += r["fshift"]
            dvdl_r += r["dvdl"]
            assert abs(got["epot"][pkg.LISTED_TYPES[k]] - r["epot"]) <= 2e-4 * max(1.0, abs(r["epot"])), k
        e_coul = {}
        for kind, k in ((1, "ljc14_q"), (2, "ljc_pairs_nb")):
            r = ob.listed_simple_pairs(kind, lists[k], prm, x, np.full(3, box), npbc, epsfac)
            f += r["f"]
            fs += r["fshift"]
            e_coul[k] = r["e_coul"]
            assert abs(got["epot"][pkg.LISTED_TYPES[k]] - r["e_lj"]) <= 2e-4 * max(1.0, abs(r["e_lj"])), k
        assert abs(got["epot"][pkg.LISTED_ENERGY_COULOMB14] - e_coul["ljc14_q"]) <= 2e-4 * max(1.0, abs(e_coul["ljc14_q"]))
        assert abs(got["epot"][pkg.LISTED_ENERGY_COULOMB_PAIRS_NB] - e_coul["ljc_pairs_nb"]) <= 2e-4 * max(1.0, abs(e_coul["ljc_pairs_nb"]))
        assert abs(got["dvdl3"][3] - dvdl_r) <= 2e-4 * max(1.0, abs(dvdl_r))
        assert got["dvdl3"][0] == 0 and got["dvdl3"][1] == 0 and got["dvdl3"][2] == 0
        scale = np.sqrt(np.mean(np.sum(f * f, axis=1)))
        err = np.linalg.norm(got["f"] - f, axis=1)
        assert (err <= 2e-4 * np.maximum(np.linalg.norm(f, axis=1), scale)).all()
        assert np.abs(got["fshift"] - fs).max() <= 2e-4 * max(np.abs(fs).max(), scale)
