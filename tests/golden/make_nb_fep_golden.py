#!/usr/bin/env python3
"""Transcribe the reference's 72 known-answer XML files for the CPU free-energy kernel
into one JSON fixture (tests/golden/nb_fep_refdata.json).

Source of the vectors (data only, no code):
  /root/reference/src/gromacs/gmxlib/nonbonded/tests/refdata/
      NBInteraction_NonbondedFepTest_testKernel_{0..71}.xml
The INPUTS those answers belong to are fully specified by the reference's test
(gmxlib/nonbonded/tests/nb_free_energy.cpp:193-201,231-240,306-362,503-518); they
are re-stated here as plain numbers so that the fixture is self-contained:
  index = ((((softcore*3 + interaction)*3 + lambda)*2 + alpha)*2 + scCoul)
  (gtest Combine, last parameter fastest; nb_free_energy.cpp:520-527)

Run in the build container only (needs /root/reference); the JSON is committed.
"""
import json
import os
import sys
import xml.etree.ElementTree as ET

REFDIR = "/root/reference/src/gromacs/gmxlib/nonbonded/tests/refdata"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nb_fep_refdata.json")

SOFTCORE = ["Beutler", "Gapsys"]                     # nb_free_energy.cpp:513
INTERACTION = [                                      # nb_free_energy.cpp:503-507
    {"coulomb": "Cut", "vdw": "Cut", "vdw_modifier": "None"},
    {"coulomb": "Cut", "vdw": "Cut", "vdw_modifier": "PotSwitch"},
    {"coulomb": "Pme", "vdw": "Pme", "vdw_modifier": "None"},
]
LAMBDAS = [0.0, 0.5, 1.0]                            # :510
ALPHAS = [0.0, 0.3]                                  # :511
SC_COUL = [True, False]                              # :512


def vec(node):
    return [float(node.find("Real[@Name='%s']" % c).text) for c in "XYZ"]


def parse(path):
    root = ET.parse(path).getroot()
    out = {}
    for r in root.findall("Real"):
        out[r.get("Name").strip()] = float(r.text)
    seq = root.find("Sequence[@Name='Forces']")
    out["Forces"] = [vec(v) for v in seq.findall("Vector")]
    sh = root.find("Shift-Forces")
    out["ShiftForceCentral"] = vec(sh.find("Vector[@Name='Central']"))
    return out


def main():
    cases = []
    for idx in range(72):
        rest = idx
        sc_coul = rest % 2; rest //= 2
        alpha = rest % 2; rest //= 2
        lam = rest % 3; rest //= 3
        inter = rest % 3; rest //= 3
        sc = rest
        path = os.path.join(REFDIR, "NBInteraction_NonbondedFepTest_testKernel_%d.xml" % idx)
        cases.append({
            "index": idx,
            "softcore": SOFTCORE[sc],
            "interaction": INTERACTION[inter],
            "lambda": LAMBDAS[lam],
            "sc_alpha": ALPHAS[alpha],
            "sc_coul": SC_COUL[sc_coul],
            "expected": parse(path),
        })
    system = {
        # nb_free_energy.cpp:306-362, 516-518
        "x": [[1.0, 1.0, 1.0], [1.1, 1.15, 1.2], [0.9, 0.85, 0.8], [1.1, 1.15, 0.8]],
        "chargeA": [1.0, -1.0, -1.0, 1.0],
        "chargeB": [1.0, 0.0, 0.0, 1.0],
        "typeA": [0, 0, 0, 0],
        "typeB": [0, 1, 2, 1],
        "ntype": 3,
        "lj_c6_c12": [[0.001458, 1.0062882e-6], [0, 0], [0.001458, 1.0062882e-6],
                      [0, 0], [0, 0], [0, 0],
                      [0.001458, 1.0062882e-6], [0, 0], [0.001458, 1.0062882e-6]],
        "iinr": [0], "jindex": [0, 4], "jjnr": [0, 1, 2, 3], "shift": [0],
        "excl_fep": [0, 1, 1, 1],
        "shiftvec": [[0.0, 0.0, 0.0]],
        # nb_free_energy.cpp:157-166,193-201 and interaction_const.h:142-156 defaults
        "epsfac_factor_of_one4pieps0": 0.25,
        "k_rf": 0.0, "c_rf": 1.0, "sh_ewald": 1.0e-5, "sh_lj_ewald": -1.0,
        "dispersion_shift_cpot": -1.0, "repulsion_shift_cpot": -1.0,
        "rcoulomb": 1.0, "rvdw": 1.0, "rvdw_switch": 0.0,
        "ewald_rc": 1.0, "ewald_rtol": 1.0e-5,
        # nb_free_energy.cpp:231-240
        "sc_power": 1, "sc_r_power": 6.0, "sc_sigma": 0.3, "sc_sigma_min": 0.3,
        "gapsys_sigma_lj": 0.3,
        # nb_free_energy.cpp:459-462
        "flags": ["FORCE", "SHIFTFORCE", "POTENTIAL"],
        # tolerance the reference itself uses, nb_free_energy.cpp:504-506,433-435
        "tolerance_float_rel": 1e-6, "tolerance_double_rel": 1e-8,
    }
    with open(OUT, "w") as fh:
        json.dump({"source": "gmxlib/nonbonded/tests/refdata (72 XML known answers)",
                   "system": system, "cases": cases}, fh, indent=1)
    print("wrote %s (%d cases)" % (OUT, len(cases)))


if __name__ == "__main__":
    sys.exit(main())
