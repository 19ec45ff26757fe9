#!/usr/bin/env python3
"""Transcribe the reference's nblib known answers for the NON-PERTURBED cluster-pair kernel into one JSON
fixture (tests/golden/nblib_refdata.json).  They pin oracle/nbnxm_ref.c (list semantics, exclusion and
diagonal rules, self terms, LJ potential shift, plain cut-off / Ewald electrostatics) and the HIP cluster kernel.

Source of the vectors (data only, no code): /root/reference/api/nblib/tests/refdata/
    NBlibTest_SpcMethanolForcesAreCorrect.xml        (nbkernelsystem.cpp:63-83, tolerance 5e-5)
    NBlibTest_SpcMethanolForcesAreCorrectOnGpu.xml   (no test body in this tree; the fp32 GPU run of the same system)
    NBlibTest_SpcMethanolEnergiesAreCorrect.xml      (gmxcalculator.cpp:112-130, 5e-5)
    NBlibTest_ArgonOplsaForcesAreCorrect.xml         (nbkernelsystem.cpp:201-220, 1e-7)
    NBlibTest_ArgonGromos43A1ForcesAreCorrect.xml    (nbkernelsystem.cpp:222-241) == NBlibTest_ArgonForcesAreCorrect.xml
    NBlibTest_ArgonVirialsAreCorrect.xml             (gmxcalculator.cpp:74-90, 1e-7)
    NBlibTest_ArgonEnergiesAreCorrect.xml            (gmxcalculator.cpp:92-110, 5e-5)
The INPUTS those answers belong to are specified by the reference's test sources; they are re-stated here as plain
numbers so that the fixture is self-contained:
  * coordinates, box, charges, per-type C6 / C12, exclusions: api/nblib/tests/testsystems.cpp:55-100 (parameters),
    :113-176 (molecules: all intramolecular pairs of water and methanol excluded), :237-256 (argon), :330-344 (SPC-methanol;
    topology order = methanol first, then water, :186-189)
  * pair parameters: geometric mean of the per-type C6 and C12 (api/nblib/interactions.cpp:83-88,164-165), times 6 / 12
    in the kernel table (api/nblib/nbnxmsetuphelpers.cpp:152-178)
  * options (api/nblib/include/nblib/kerneloptions.h:85-107, nbnxmsetuphelpers.cpp:232-292): cut-off 1.0 nm for list and
    interactions, LJ cut with potential shift (cpot = -rc^-6, -rc^-12), epsilon_r = epsilon_rf = 1;
    CoulombType::Cutoff -> eeltype Cut = reaction field with k_rf = 0, c_rf = 1/rc (mdlib/rf_util.cpp:50-62);
    CoulombType::Pme (the default; the energy and virial tests) -> Ewald real space, beta = calc_ewaldcoeff_q(1.0, 1e-5),
    **sh_ewald stays 0** (nblib fills interaction_const_t by hand and "ignores the potential shift",
    mdtypes/interaction_const.h:162), tabulated correction in the plain-C kernel
  * energies: [CoulombSR, LJSR, BuckinghamSR, Coulomb14, LJ14]; virial = -0.5 (sum_shift s (x) fshift + sum_atoms x (x) f)
    (api/nblib/virials.cpp:53-84, mdlib/calcvir.cpp)

Run in the build container only (needs /root/reference); the JSON is committed.
"""
import json
import os
import xml.etree.ElementTree as ET

REFDIR = "/root/reference/api/nblib/tests/refdata"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nblib_refdata.json")

# testsystems.cpp:72-84
C6 = {"Ow": 0.0026173456, "H": 0.0, "OMet": 0.0022619536, "CMet": 0.0088755241, "Ar_gromos": 0.0062647225, "Ar_opls": 0.0058560692}
C12 = {"Ow": 2.634129e-06, "H": 0.0, "OMet": 1.505529e-06, "CMet": 2.0852922e-05, "Ar_gromos": 9.847044e-06, "Ar_opls": 8.203193e-06}
# testsystems.cpp:96-100
Q = {"Ow": -0.82, "Hw": 0.41, "OMet": -0.574, "CMet": 0.176, "HMet": 0.398}

ARGON_X = [  # testsystems.cpp:241-246
    [0.794, 1.439, 0.610], [1.397, 0.673, 1.916], [0.659, 1.080, 0.573], [1.105, 0.090, 3.431], [1.741, 1.291, 3.432],
    [1.936, 1.441, 5.873], [0.960, 2.246, 1.659], [0.382, 3.023, 2.793], [0.053, 4.857, 4.242], [2.655, 5.057, 2.211],
    [4.114, 0.737, 0.614], [5.977, 5.104, 5.217]]
SPC_METHANOL_X = [  # testsystems.cpp:333-340
    [1.970, 1.460, 1.209], [1.978, 1.415, 1.082], [1.905, 1.460, 1.030],   # Me1 O2 H3
    [1.555, 1.511, 0.703], [1.498, 1.495, 0.784], [1.496, 1.521, 0.623]]   # Ow Hw1 Hw2


def vectors(name, seq):
    root = ET.parse(os.path.join(REFDIR, name)).getroot()
    s = root.find("Sequence[@Name='%s']" % seq)
    return [[float(v.find("Real[@Name='%s']" % c).text) for c in "XYZ"] for v in s.findall("Vector")]


def reals(name, seq):
    root = ET.parse(os.path.join(REFDIR, name)).getroot()
    s = root.find("Sequence[@Name='%s']" % seq)
    return [float(r.text) for r in s.findall("Real")]


def main():
    spc = {
        "box": 3.01, "x": SPC_METHANOL_X,
        "type_names": ["CMet", "OMet", "H", "Ow", "H", "H"],
        "q": [Q["CMet"], Q["OMet"], Q["HMet"], Q["Ow"], Q["Hw"], Q["Hw"]],
        "molecule": [0, 0, 0, 1, 1, 1],
    }
    argon = {"box": 6.05449, "x": ARGON_X, "q": [0.0] * 12, "molecule": list(range(12))}
    out = {
        "source": "api/nblib/tests/refdata/NBlibTest_*.xml, inputs from api/nblib/tests/testsystems.cpp (see make_nblib_golden.py)",
        "c6": C6, "c12": C12, "cutoff": 1.0, "ewald_rtol": 1e-5,
        "cases": [
            dict(name="SpcMethanolForcesAreCorrect", system=spc, coulomb="cut", tolerance=5e-5,
                 forces=vectors("NBlibTest_SpcMethanolForcesAreCorrect.xml", "SPC-methanol forces")),
            dict(name="SpcMethanolForcesAreCorrectOnGpu", system=spc, coulomb="cut", tolerance=5e-5,
                 forces=vectors("NBlibTest_SpcMethanolForcesAreCorrectOnGpu.xml", "SPC-methanol forces on GPU")),
            dict(name="SpcMethanolEnergiesAreCorrect", system=spc, coulomb="pme", tolerance=5e-5,
                 energies=reals("NBlibTest_SpcMethanolEnergiesAreCorrect.xml", "SPC-methanol energies")),
            dict(name="ArgonOplsaForcesAreCorrect", system=dict(argon, type_names=["Ar_opls"] * 12), coulomb="cut", tolerance=1e-7,
                 forces=vectors("NBlibTest_ArgonOplsaForcesAreCorrect.xml", "Argon forces")),
            dict(name="ArgonGromos43A1ForcesAreCorrect", system=dict(argon, type_names=["Ar_gromos"] * 12), coulomb="cut", tolerance=1e-8,
                 forces=vectors("NBlibTest_ArgonGromos43A1ForcesAreCorrect.xml", "Argon forces")),
            dict(name="ArgonForcesAreCorrect", system=dict(argon, type_names=["Ar_gromos"] * 12), coulomb="cut", tolerance=1e-8,
                 forces=vectors("NBlibTest_ArgonForcesAreCorrect.xml", "Argon forces")),
            dict(name="ArgonVirialsAreCorrect", system=dict(argon, type_names=["Ar_opls"] * 12), coulomb="pme", tolerance=1e-7,
                 virial=reals("NBlibTest_ArgonVirialsAreCorrect.xml", "Virials")),
            dict(name="ArgonEnergiesAreCorrect", system=dict(argon, type_names=["Ar_opls"] * 12), coulomb="pme", tolerance=5e-5,
                 energies=reals("NBlibTest_ArgonEnergiesAreCorrect.xml", "Argon energies")),
        ],
    }
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", OUT, len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
