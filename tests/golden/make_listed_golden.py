#!/usr/bin/env python3
"""Builds tests/golden/listed_refdata.json from the reference's own known answers for the listed (bonded)
interactions, src/gromacs/listed_forces/tests/refdata/{Bond,Angle,Dihedral}_ListedForcesTest_Ifunc_*.xml.

Only DATA is taken from the reference: the expected outputs (XML) and the inputs its test feeds the kernels
(listed_forces/tests/bonded.cpp:515-525 atom tuples, :655-700 parameters, :752-760 butane coordinates, :539-541 box,
:576 charges), restated below.  Test instances are indexed input * 3 + pbc (pbc in {none, xy, xyz}).
Run once in the build container (the reference is not present on the GPU box); the JSON is committed.
"""
import json
import os
import xml.etree.ElementTree as ET

REF = "/root/reference/src/gromacs/listed_forces/tests/refdata"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "listed_refdata.json")

COORDS = [[1.382, 1.573, 1.482], [1.281, 1.559, 1.596], [1.292, 1.422, 1.663], [1.189, 1.407, 1.775]]
BOX = 1.5
PBC = ["none", "xy", "xyz"]
RBC_A = [-5.35, 13.6, 8.4, -16.7, 0.3, 12.4]
RBC_B = [-6.35, 12.6, 8.1, -10.7, 0.9, 15.4]
RBC = [-7.35, 13.6, 8.4, -16.7, 1.3, 12.4]

# (suite, input index, our type name, reference FunctionType name, parameters)
CASES = [
    ("Bond", 0, "bonds", "BONDS", dict(rA=0.15, krA=500.0, rB=0.15, krB=500.0)),
    ("Bond", 1, "bonds", "BONDS", dict(rA=0.15, krA=500.0, rB=0.17, krB=400.0)),
    ("Angle", 0, "angles", "ANGLES", dict(rA=100.0, krA=50.0, rB=100.0, krB=50.0)),
    ("Angle", 1, "angles", "ANGLES", dict(rA=100.15, krA=50.0, rB=95.0, krB=30.0)),
    ("Angle", 8, "urey_bradley", "UREY_BRADLEY", dict(thetaA=950.0, kthetaA=46.0, r13A=0.3, kUBA=5.0,
                                                     thetaB=950.0, kthetaB=46.0, r13B=0.3, kUBB=5.0)),
    ("Angle", 9, "urey_bradley", "UREY_BRADLEY", dict(thetaA=100.0, kthetaA=45.0, r13A=0.3, kUBA=5.0,
                                                     thetaB=90.0, kthetaB=47.0, r13B=0.32, kUBB=7.0)),
    ("Dihedral", 0, "pdihs", "PDIHS", dict(phiA=-100.0, cpA=10.0, mult=2, phiB=-80.0, cpB=20.0)),
    ("Dihedral", 1, "pdihs", "PDIHS", dict(phiA=-105.0, cpA=15.0, mult=2, phiB=-105.0, cpB=15.0)),
    ("Dihedral", 2, "idihs", "IDIHS", dict(rA=100.0, krA=50.0, rB=100.0, krB=50.0)),
    ("Dihedral", 3, "idihs", "IDIHS", dict(rA=100.15, krA=50.0, rB=95.0, krB=30.0)),
    ("Dihedral", 4, "rbdihs", "RBDIHS", dict(rbcA=RBC_A, rbcB=RBC_B)),
    ("Dihedral", 5, "rbdihs", "RBDIHS", dict(rbcA=RBC, rbcB=RBC)),
    # bonded.cpp:709-716 (c_InputRestraints): angle restraints between the vectors i->j and k->l
    ("Restraints", 0, "angres", "ANGRES", dict(phiA=-100.0, cpA=10.0, mult=2, phiB=-80.0, cpB=20.0)),
    ("Restraints", 1, "angres", "ANGRES", dict(phiA=-105.0, cpA=15.0, mult=2, phiB=-105.0, cpB=15.0)),
]
NRAL = {"bonds": 2, "angles": 3, "urey_bradley": 3, "pdihs": 4, "idihs": 4, "rbdihs": 4, "angres": 4}
IATOMS = {2: [[0, 1], [1, 2], [2, 3]], 3: [[0, 1, 2], [1, 2, 3]], 4: [[0, 1, 2, 3]]}


def parse_lambda_block(node):
    reals = {r.get("Name").strip(): float(r.text) for r in node.findall("Real")}
    forces = [[float(v.find("Real[@Name='%s']" % c).text) for c in "XYZ"] for v in node.find("Sequence[@Name='Forces']").findall("Vector")]
    return dict(epot=reals["Epot"], dvdlambda=reals["dVdlambda"], forces=forces)


def main():
    out = dict(coordinates=COORDS, box=BOX, cases=[])
    for suite, inp, name, refname, params in CASES:
        for ip, pbc in enumerate(PBC):
            fn = os.path.join(REF, "%s_ListedForcesTest_Ifunc_%d.xml" % (suite, inp * 3 + ip))
            root = ET.parse(fn).getroot()
            ft = root.find("FunctionType")
            assert ft.get("Name") == refname, (fn, ft.get("Name"), refname)
            fep = ft.find("FEP")
            results = {}
            if fep.get("Name") == "Yes":
                for lam in fep.findall("Lambda"):
                    results[lam.get("Name")] = parse_lambda_block(lam)
            else:
                results["0"] = parse_lambda_block(fep)
            out["cases"].append(dict(suite=suite, index=inp * 3 + ip, type=name, pbc=pbc, params=params,
                                     iatoms=IATOMS[NRAL[name]], fep=fep.get("Name") == "Yes", results=results))
    # 1-4 pairs (listed_forces/tests/pairs.cpp): 3 atoms, box 1.0, interactions (1,2) and (0,2), charges A = {1, -0.5, -0.5},
    # B = 0, fudgeQQ 0.5, epsfac 1 (default interaction_const_t), sc-alpha 0.3, sc-power 1, sc-sigma = sc-sigma-min = 0.3,
    # sc-coul on; inputs :446-451: LJ14 (c6A, c12A, c6B, c12B).  Only the "beutler" soft-core entries are taken.
    pairs = dict(coordinates=[[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [1.1, 1.2, 1.3]], box=1.0, iatoms=[[1, 2], [0, 2]],
                 chargeA=[1.0, -0.5, -0.5], chargeB=[0.0, 0.0, 0.0], fudgeQQ=0.5, epsfac=1.0, sc_alpha=0.3, sc_power=1,
                 sc_sigma=0.3, sc_sigma_min=0.3, cases=[])
    for inp, prm in enumerate([dict(c6A=0.001458, c12A=1.0062882e-6, c6B=0.0, c12B=0.0),
                               dict(c6A=0.001458, c12A=1.0062882e-6, c6B=0.001458, c12B=1.0062882e-6)]):
        for ip, pbc in enumerate(PBC):
            fn = os.path.join(REF, "14Interaction_ListedForcesPairsTest_Ifunc_%d.xml" % (inp * 3 + ip))
            ft = ET.parse(fn).getroot().find("FunctionType")
            assert ft.get("Name") == "LJ14", fn
            fep = ft.find("FEP")
            results = {}

            def block(node):
                reals = {r.get("Name").strip(): float(r.text) for r in node.findall("Real")}
                forces = [[float(v.find("Real[@Name='%s']" % c).text) for c in "XYZ"]
                          for v in node.find("Sequence[@Name='Forces']").findall("Vector")]
                return dict(eCoul=reals["Epot Coulomb14"], eLJ=reals["Epot LJ14"], dvdlCoul=reals["dVdlCoul"],
                            dvdlVdw=reals["dVdlVdw"], forces=forces)
            if fep.get("Name") == "Yes":
                for lam in fep.findall("Lambda"):
                    results[lam.get("Name")] = block(lam.find("Sofcore[@Name='beutler']"))
            else:
                results["0"] = block(fep)
            pairs["cases"].append(dict(index=inp * 3 + ip, pbc=pbc, params=prm, fep=fep.get("Name") == "Yes", results=results))
    out["pairs"] = pairs
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote %s: %d cases" % (OUT, len(out["cases"])))


if __name__ == "__main__":
    main()
