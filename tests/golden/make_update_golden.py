#!/usr/bin/env python3
"""Builds tests/golden/update_refdata.json from the reference's own known answers for the coordinate update:
src/gromacs/mdlib/tests/refdata/WithParameters_{LeapFrogTest_SimpleIntegration,SettleTest_SatisfiesConstraints,
ConstraintsTest_SatisfiesConstraints}_*.xml.

Only DATA is taken from the reference: the expected outputs (XML) and the inputs its tests feed the integrators, restated
below — leap-frog: mdlib/tests/leapfrog.cpp:109-127 (parameter sets) and leapfrogtestdata.cpp:92-113,152-160,188-197
(positions, masses, coupling factors); SETTLE: settle.cpp:123-137 (parameter sets), settletestdata.h:85-93 (masses, distances,
1/dt), settletestdata.cpp:84-96 (displacements) and the 17 water molecules of watersystem.h:50-62 (numbers parsed from that
file); constraints: constr.cpp:150-362 (seven systems) x :84-98 (two boxes), instance index = system * 2 + pbc.
Run once in the build container (the reference is not present on the GPU box); the JSON is committed.
"""
import json
import math
import os
import re
import xml.etree.ElementTree as ET

REF = "/root/reference/src/gromacs/mdlib/tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "update_refdata.json")

# numAtoms, dt, numSteps, v0, f0, numTCoupleGroups, nstpcouple
LEAPFROG = [
    (1, 0.001, 1, [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], 0, 0),
    (1, 0.001, 1, [0.0, 0.0, 0.0], [-3.0, 2.0, -1.0], 0, 0),
    (1, 0.001, 1, [1.0, -2.0, 3.0], [0.0, 0.0, 0.0], 0, 0),
    (1, 0.001, 1, [1.0, -2.0, 3.0], [-3.0, 2.0, -1.0], 0, 0),
    (10, 0.001, 1, [1.0, -2.0, 3.0], [-3.0, 2.0, -1.0], 0, 0),
    (100, 0.001, 1, [1.0, -2.0, 3.0], [-3.0, 2.0, -1.0], 0, 0),
    (300, 0.001, 1, [1.0, -2.0, 3.0], [-3.0, 2.0, -1.0], 0, 0),
    (1, 0.0005, 1, [1.0, -2.0, 3.0], [-3.0, 2.0, -1.0], 0, 0),
    (1, 0.001, 10, [1.0, -2.0, 3.0], [-3.0, 2.0, -1.0], 0, 0),
    (1, 0.001, 100, [1.0, -2.0, 3.0], [-3.0, 2.0, -1.0], 0, 0),
    (100, 0.001, 1, [1.0, -2.0, 3.0], [-3.0, 2.0, -1.0], 1, 0),
    (100, 0.001, 1, [1.0, -2.0, 3.0], [-3.0, 2.0, -1.0], 2, 0),
    (100, 0.001, 1, [1.0, -2.0, 3.0], [-3.0, 2.0, -1.0], 10, 0),
    (100, 0.001, 10, [1.0, -2.0, 3.0], [-3.0, 2.0, -1.0], 0, 1),
    (100, 0.001, 10, [1.0, -2.0, 3.0], [-3.0, 2.0, -1.0], 2, 1),
    (100, 0.001, 10, [1.0, -2.0, 3.0], [-3.0, 2.0, -1.0], 0, 3),
]
LEAPFROG_PR_DIAGONAL = [1.2, 0.8, 0.9]

# numSettles, updateVelocities, calcVirial, pbc
SETTLE = [(1, False, False, "xyz"), (2, False, False, "xyz"), (4, False, False, "xyz"), (5, False, False, "xyz"),
          (6, False, False, "xyz"), (10, False, False, "xyz"), (12, False, False, "xyz"), (15, False, False, "xyz"),
          (17, True, False, "xyz"), (17, False, True, "xyz"), (17, False, False, "none"), (17, True, True, "none"),
          (17, True, True, "xyz")]
SETTLE_CONST = dict(dOH=0.09572, dHH=0.15139, mO=15.9994, mH=1.008, invdt=1.0 / 0.002, box=1.86206,
                    deltas=[0.01, -0.01, 0.02, -0.02])

S2 = 0.1 / math.sqrt(2.0)
S3 = 0.2 / math.sqrt(3.0)
FOUR_X = [[2.50, -3.10, 15.70], [0.51, -3.02, 15.55], [-0.50, -3.00, 15.20], [-1.51, -2.95, 15.05]]
CH2 = dict(title="three atoms, connected longitudinally (e.g. CH2)", masses=[1.0, 12.0, 16.0], constraints=[0, 0, 1, 1, 1, 2],
           r0=[0.1, 0.2], x=[[S2, S2, 0.0], [0.0, 0.0, 0.0], [S3, S3, S3]],
           xPrime=[[0.08, 0.07, 0.01], [-0.02, 0.01, -0.02], [0.10, 0.12, 0.11]], v=[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])


def many_molecules(mol, count):
    n = len(mol["masses"])
    cons = []
    for m in range(count):
        for c in range(0, len(mol["constraints"]), 3):
            cons += [mol["constraints"][c], mol["constraints"][c + 1] + m * n, mol["constraints"][c + 2] + m * n]
    return dict(title="system of many molecules", masses=mol["masses"] * count, constraints=cons, r0=mol["r0"], x=mol["x"] * count,
                xPrime=mol["xPrime"] * count, v=mol["v"] * count)


CONSTRAINT_SYSTEMS = [
    dict(title="one constraint (e.g. OH)", masses=[1.0, 12.0], constraints=[0, 0, 1], r0=[0.1], x=[[0.0, S2, 0.0], [S2, 0.0, 0.0]],
         xPrime=[[0.01, 0.08, 0.01], [0.06, 0.01, -0.01]], v=[[1.0, 2.0, 3.0], [3.0, 2.0, 1.0]]),
    dict(title="two disjoint constraints", masses=[0.5, 1.0 / 3.0, 0.25, 1.0], constraints=[0, 0, 1, 1, 2, 3], r0=[2.0, 1.0], x=FOUR_X,
         xPrime=FOUR_X, v=[[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, 0.0, 0.0]]),
    CH2,
    dict(title="four atoms, connected longitudinally", masses=[0.5, 1.0 / 3.0, 0.25, 1.0], constraints=[0, 0, 1, 1, 1, 2, 2, 2, 3],
         r0=[2.0, 1.0, 1.0], x=FOUR_X, xPrime=FOUR_X, v=[[0.0, 0.0, 2.0], [0.0, 0.0, 3.0], [0.0, 0.0, -4.0], [0.0, 0.0, -1.0]],
         lincsNIter=4, lincsExpansionOrder=8),
    dict(title="three atoms, connected to the central atom (e.g. CH3)", masses=[12.0, 1.0, 1.0, 1.0],
         constraints=[0, 0, 1, 0, 0, 2, 0, 0, 3], r0=[0.1],
         x=[[0.00, 0.00, 0.00], [0.10, 0.00, 0.00], [0.00, -0.10, 0.00], [0.00, 0.00, 0.10]],
         xPrime=[[0.004, 0.009, -0.010], [0.110, -0.006, 0.003], [-0.007, -0.102, -0.007], [-0.005, 0.011, 0.102]],
         v=[[1.0, 0.0, 0.0]] * 4),
    dict(title="basic triangle (three atoms, connected to each other)", masses=[1.0, 1.0, 1.0], constraints=[0, 0, 1, 2, 0, 2, 1, 1, 2],
         r0=[0.1, 0.1, 0.1], x=[[S2, 0.0, 0.0], [0.0, S2, 0.0], [0.0, 0.0, S2]],
         xPrime=[[0.09, -0.02, 0.01], [-0.02, 0.10, -0.02], [0.03, -0.01, 0.07]], v=[[1.0, 1.0, 1.0], [-2.0, -2.0, -2.0], [1.0, 1.0, 1.0]]),
    many_molecules(CH2, 150),
]
CONSTRAINT_BOXES = [("none", [0.0, 0.0, 0.0]), ("xyz", [10.0, 20.0, 15.0])]
CONSTRAINT_CONST = dict(timestep=0.001, lincsNIter=1, lincsExpansionOrder=4, positionsTolerance=0.001, velocityTolerance=0.02,
                        virialRelativeTolerance=0.02)


def atoms_of(seq):
    return [[float(a.find("Real[@Name='%s']" % c).text) for c in ("XX", "YY", "ZZ")] for a in seq.findall("Atom")]


def tensor_of(node):
    return [[float(node.find("Real[@Name='%s%s']" % (a, b)).text) for b in "XYZ"] for a in "XYZ"]


def main():
    out = {}
    # leap-frog
    cases = []
    for i, (n, dt, steps, v0, f0, ntc, nstpc) in enumerate(LEAPFROG):
        root = ET.parse(os.path.join(REF, "refdata", "WithParameters_LeapFrogTest_SimpleIntegration_%d.xml" % i)).getroot()
        fx = atoms_of(root.find("Sequence[@Name='FinalPositions']"))
        fv = atoms_of(root.find("Sequence[@Name='FinalVelocities']"))
        assert len(fx) == n and len(fv) == n
        cases.append(dict(numAtoms=n, timestep=dt, numSteps=steps, v0=v0, f0=f0, numTCoupleGroups=ntc, nstpcouple=nstpc,
                          finalPositions=fx, finalVelocities=fv))
    out["leapfrog"] = dict(prDiagonal=LEAPFROG_PR_DIAGONAL, cases=cases)

    # SETTLE
    text = open(os.path.join(REF, "watersystem.h")).read()
    nums = [float(t) for t in re.findall(r"-?\d*\.\d+", text[text.index("c_waterPositions"):])]
    assert len(nums) == 17 * 9
    waters = [nums[i:i + 3] for i in range(0, len(nums), 3)]
    cases = []
    for i, (ns, upd, vir, pbc) in enumerate(SETTLE):
        root = ET.parse(os.path.join(REF, "refdata", "WithParameters_SettleTest_SatisfiesConstraints_%d.xml" % i)).getroot()
        settlers = root.find("Sequence[@Name='FinalCoordinates']").findall("Settler")
        assert len(settlers) == ns
        c = dict(numSettles=ns, updateVelocities=upd, calcVirial=vir, pbc=pbc,
                 finalCoordinates=[a for s in settlers for a in atoms_of(s.find("Sequence[@Name='Atoms']"))])
        if upd:
            c["finalVelocities"] = [a for s in root.find("Sequence[@Name='FinalVelocities']").findall("Settler")
                                    for a in atoms_of(s.find("Sequence[@Name='Atoms']"))]
        if vir:
            c["virial"] = tensor_of(root.find("Virial"))
        cases.append(c)
    out["settle"] = dict(SETTLE_CONST, waterPositions=waters, cases=cases)

    # constraints
    cases = []
    for isys, system in enumerate(CONSTRAINT_SYSTEMS):
        for ip, (pbc, box) in enumerate(CONSTRAINT_BOXES):
            root = ET.parse(os.path.join(REF, "refdata", "WithParameters_ConstraintsTest_SatisfiesConstraints_%d.xml" % (isys * 2 + ip))).getroot()
            fx = atoms_of(root.find("Sequence[@Name='FinalPositions']"))
            fv = atoms_of(root.find("Sequence[@Name='FinalVelocities']"))
            assert len(fx) == len(system["masses"])
            cases.append(dict(system=isys, pbc=pbc, box=box, finalPositions=fx, finalVelocities=fv,
                              virialScaled=tensor_of(root.find("VirialScaled"))))
    out["constraints"] = dict(CONSTRAINT_CONST, systems=CONSTRAINT_SYSTEMS, cases=cases)

    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=0)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
