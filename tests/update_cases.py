"""Inputs of the reference's update / constraint tests, rebuilt from tests/golden/update_refdata.json (see
tests/golden/make_update_golden.py for where every number comes from).  Shared by the oracle tests (CPU) and the GPU parity tests."""
import json
import os

import numpy as np

_DATA = None


def data():
    global _DATA
    if _DATA is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "update_refdata.json")) as fh:
            _DATA = json.load(fh)
    return _DATA


class LeapFrogCase:
    """leapfrogtestdata.cpp:92-160"""

    def __init__(self, c):
        n = c["numAtoms"]
        i = np.arange(n)
        self.n, self.dt, self.num_steps = n, c["timestep"], c["numSteps"]
        self.x0 = np.stack([(i % 21) * 1.0, 6.5 - (i % 13) * 1.0, np.zeros(n)], axis=1)
        self.v0 = np.tile(np.array(c["v0"], float), (n, 1))
        self.f = np.tile(np.array(c["f0"], float), (n, 1))
        self.invmass = 1.0 / (1.0 + i % 100)
        self.num_tc = c["numTCoupleGroups"]
        self.groups = (i % self.num_tc).astype(np.uint16) if self.num_tc > 0 else np.zeros(n, np.uint16)
        self.lambdas = np.array([1.0 - (g + 1.0) / 10.0 for g in range(self.num_tc)])
        self.nstpcouple = c["nstpcouple"]
        self.dt_pc = self.nstpcouple * self.dt
        self.pr_diag = np.array(data()["leapfrog"]["prDiagonal"])
        self.final_x = np.array(c["finalPositions"])
        self.final_v = np.array(c["finalVelocities"])
        self.tolerance = self.num_steps * 0.000005  # leapfrog.cpp:261

    def do_pressure_couple(self, step):
        # leapfrogtestrunners_gpu.cpp:99-102: do_per_step(step + nstpcouple - 1, nstpcouple)
        return self.nstpcouple != 0 and (step + self.nstpcouple - 1) % self.nstpcouple == 0


def leapfrog_cases():
    return [LeapFrogCase(c) for c in data()["leapfrog"]["cases"]]


class SettleCase:
    """settletestdata.cpp:75-131"""

    def __init__(self, c):
        d = data()["settle"]
        self.num_settles = c["numSettles"]
        self.update_velocities, self.calc_virial = c["updateVelocities"], c["calcVirial"]
        self.pbc_type = 3 if c["pbc"] == "xyz" else 0
        self.box = np.eye(3) * (d["box"] if c["pbc"] == "xyz" else 0.0)
        self.x = np.array(d["waterPositions"], float)
        delt = np.array(d["deltas"])
        self.xp = self.x + delt[np.arange(self.x.size) % 4].reshape(-1, 3)
        self.v = np.zeros_like(self.x)
        self.atoms = np.arange(3 * self.num_settles, dtype=np.int32).reshape(-1, 3)
        self.mO, self.mH, self.dOH, self.dHH, self.invdt = d["mO"], d["mH"], d["dOH"], d["dHH"], d["invdt"]
        self.final_x = np.array(c["finalCoordinates"])
        self.final_v = np.array(c["finalVelocities"]) if "finalVelocities" in c else None
        self.virial = np.array(c["virial"]) if "virial" in c else None


def settle_cases():
    return [SettleCase(c) for c in data()["settle"]["cases"]]


class ConstraintsCase:
    """constr.cpp:100-362, constrtestdata.cpp:59-190"""

    def __init__(self, c):
        d = data()["constraints"]
        s = d["systems"][c["system"]]
        self.title = s["title"] + " / " + c["pbc"]
        self.masses = np.array(s["masses"], float)
        self.invmass = 1.0 / self.masses
        self.iatoms = np.array(s["constraints"], np.int32).reshape(-1, 3)
        self.lengths = np.array(s["r0"], float)
        self.x = np.array(s["x"], float)
        self.xp = np.array(s["xPrime"], float)
        self.v = np.array(s["v"], float)
        self.n_iter = s.get("lincsNIter", d["lincsNIter"])
        self.order = s.get("lincsExpansionOrder", d["lincsExpansionOrder"])
        self.invdt = 1.0 / d["timestep"]
        self.pbc_type = 3 if c["pbc"] == "xyz" else 0
        self.box = np.diag(np.array(c["box"], float))
        self.final_x = np.array(c["finalPositions"])
        self.final_v = np.array(c["finalVelocities"])
        self.virial = np.array(c["virialScaled"])
        self.tol_x, self.tol_v = d["positionsTolerance"], d["velocityTolerance"]
        self.tol_virial = abs(np.trace(self.virial)) / 3 * d["virialRelativeTolerance"]


def constraints_cases():
    return [ConstraintsCase(c) for c in data()["constraints"]["cases"]]
