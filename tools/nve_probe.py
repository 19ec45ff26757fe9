#!/usr/bin/env python3
"""Energy drift of the GPU-resident loop: against the time step (an integration error falls with dt^2), with and without
perturbed molecules, and the list's validity at the end (all-pairs energy of the final coordinates)."""
import sys, os, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fep_testlib as tl
pkg = tl.pkg
mdloop = importlib.import_module("gromacs_fep_gpu_amd.mdloop")
for npert, dt, nsteps, every in ((0, 0.0001, 150, 5), (0, 0.00005, 300, 10), (3, 0.00005, 300, 10)):
    c = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=npert, elec=os.environ.get("NVE_ELEC", "rf"), seed=5)
    g = c.grid
    nb = tl.setup_gpu(c, fused=True)
    n = c.natoms
    mass = np.tile([15.9994, 1.008, 1.008], n // 3)
    loop = mdloop.ShortRangeMdLoop(nb, g, g.x_wrapped, np.zeros((n, 3)), 1.0 / mass, dt, np.diag(g.box.astype(np.float64)),
                                   settles=np.arange(n, dtype=np.int32).reshape(-1, 3), settle_params=(15.9994, 1.008, 0.1, 0.16330), fused_update=True)
    sw_e = pkg.step_workload(energy=True, virial=False, dhdl=False)
    def kin():
        torch.cuda.synchronize()      # the loop runs on the non-bonded object's own stream
        return 0.5 * float(np.sum(mass[:, None] * loop.d_v.cpu().numpy().astype(np.float64) ** 2))
    et, ek, ep = [], [], []
    x_at_energy = None
    for step in range(nsteps):
        if step % every == 0:
            torch.cuda.synchronize()
            x_at_energy = loop.d_x.cpu().numpy()
            k0 = kin(); loop.step(step, step_work=sw_e)
            f = np.zeros((g.num_atoms, 3), np.float32); nb.launch_cpyback(f, sw_e); r = nb.wait_finish_task(sw_e, c.have_soft_core)
            k1 = kin(); ek.append(0.5 * (k0 + k1)); ep.append(r["e_lj"] + r["e_el"]); et.append(ek[-1] + ep[-1])
        else:
            loop.step(step)
    et, ek = np.array(et), np.array(ek)
    print("npert %d dt %.6f ps: exchange %.1f, max drift %.2f, end drift %.2f; series %s" % (npert, dt, ek.max() - ek.min(), np.max(np.abs(et - et[0])), et[-1] - et[0], np.round(et - et[0], 1).tolist()), flush=True)
    if npert == 0:
        x0 = g.x_wrapped.copy()
        e_first = tl.brute_force(c)
        g.x_wrapped[:] = x_at_energy
        e_last = tl.brute_force(c)
        g.x_wrapped[:] = x0
        print("   all-pairs energy: first %.2f (kernels %.2f), last energy step %.2f (kernels %.2f)" % (e_first["e_lj"] + e_first["e_el"], ep[0], e_last["e_lj"] + e_last["e_el"], ep[-1]))
        disp = x_at_energy - x0
        print("   largest displacement %.3f nm" % np.sqrt((disp ** 2).sum(axis=1)).max())
    loop.free(); nb.free()
