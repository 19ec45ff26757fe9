"""GPU-resident short-range MD steps between two pair searches: the part of mdrun's force schedule this repository covers,
queued the way do_force / do_md queue it when update and buffer operations run on the GPU
(mdlib/sim_util.cpp:1764-1767 x -> xq, :1886 non-bonded launch, :2400-2440 force reduction; mdrun/md.cpp:1500-1560
UpdateConstrainGpu::integrate).  Everything is one stream of C-ABI calls; Python only sequences them, as the C++ schedule does.

    x (atom order, HBM) --x_to_nbat_x--> xq (grid order) --cluster-pair + FEP kernels--> f (grid order)
      --force reduction--> f (atom order) --leap-frog | SD, LINCS, SETTLE--> x, v

With fused_update (MI355X extension; not for stochastic dynamics with LINCS constraints) the four kernels around the non-bonded ones
collapse into one (+ the LINCS kernel when there are such constraints):

    xq, f (grid order) --cluster-pair + FEP kernels--> f --fused update: gather f, integrate, SETTLE, clear f, write x, v AND xq

No long-range (PME) part, no search: coordinates, velocities and forces never leave HBM between two searches."""
import numpy as np

from . import UpdateConstrainGpu, step_workload


class ShortRangeMdLoop:
    def __init__(self, nb, grid, x0, v0, inverse_masses, dt, box, settles=None, settle_params=None, constraints=None,
                 constraint_lengths=None, temp_coupl_groups=None, num_temp_coupl_groups=0, stochastic_dynamics=False, ref_t=None,
                 tau_t=None, n_lincs_iter=1, n_proj_order=4, device="cuda", fused_update=False, listed=None, listed_fep=None,
                 listed_elec_scale=0.0, listed_epsfac=0.0, rolling_prune_parts=0):
        """nb: NbnxmGpu with atom data and pair list uploaded; grid: the host grid the list was built on (atomIndices);
        x0 / v0: atom order, x0 as the grid saw it (inside the unit cell); settle_params: (mO, mH, dOH, dHH);
        listed: a ListedGpu created on nb.stream() whose interaction lists hold GRID-order atom indices, as the reference's
        ListedForcesGpu gets them (convertIlistToNbnxnOrder); it works on the non-bonded xq / f buffers, listed_fep: ListedFepParams"""
        import torch
        self.nb, self.grid, self.dt = nb, grid, float(dt)
        self.natoms, self.nslots = int(grid.natoms), int(grid.num_atoms)
        ai = grid.atomIndices
        real = ai >= 0
        cell = np.full(self.natoms, -1, np.int32)
        cell[ai[real]] = np.nonzero(real)[0]
        assert (cell >= 0).all()
        self.d_x = torch.from_numpy(np.ascontiguousarray(x0, np.float32)).to(device)
        self.d_v = torch.from_numpy(np.ascontiguousarray(v0, np.float32)).to(device)
        self.d_f = torch.zeros_like(self.d_x)
        self.stream = nb.stream()
        nb.init_x_to_nbat_x(ai)
        nb.force_reduction_reinit(cell, atom_start=0, accumulate=False)
        self.update = UpdateConstrainGpu(dt, num_temp_coupl_groups=num_temp_coupl_groups, stochastic_dynamics=stochastic_dynamics,
                                         ref_t=ref_t, tau_t=tau_t, n_lincs_iter=n_lincs_iter, n_proj_order=n_proj_order,
                                         settle=settle_params, stream=self.stream)
        ok = self.update.set(self.d_x.data_ptr(), self.d_v.data_ptr(), self.d_f.data_ptr(), inverse_masses, temp_coupl_groups, constraints,
                             constraint_lengths, settles)
        if not ok:
            raise ValueError("a group of coupled constraints is too large for the GPU LINCS")
        self.update.set_pbc(3, box)
        self.fused_update = False
        if fused_update:
            self.update.set_nbat_coupling(cell, nb.xq_device_pointer(), nb.f_device_pointer())
            if not self.update.can_fuse():
                raise ValueError("stochastic dynamics with LINCS constraints needs the kernel sequence (fused_update=False)")
            self.fused_update = True
            # the state the fused kernel leaves behind: xq = current coordinates, non-bonded forces cleared
            nb.x_to_nbat_x(self.d_x.data_ptr(), 0, self.nslots)
            nb.clear_outputs(True)
        # dynamic pruning as mdrun does it between searches: every step one of rolling_prune_parts parts of the list is pruned to the
        # inner radius (nonbonded_verlet_t::dispatchPruneKernelGpu); 0: the list stays as the first step left it
        self.rolling_prune_parts = int(rolling_prune_parts)
        self.listed, self.listed_fep = listed, listed_fep
        self.listed_scales = (float(listed_elec_scale), float(listed_epsfac))
        self.box9 = np.ascontiguousarray(np.asarray(box, np.float32).reshape(3, 3) if np.asarray(box).size == 9 else np.diag(np.asarray(box, np.float32)))
        self.force_only = step_workload(energy=False, virial=False, dhdl=False)
        torch.cuda.synchronize()

    def _launch_listed(self, sw):
        """ListedForcesGpu::launchKernel on the non-bonded buffers, behind the non-bonded kernels of the same stream"""
        if self.listed is None or not self.listed.have_interactions():
            return
        nb = self.nb
        self.listed.launch_kernel(nb.xq_device_pointer(), nb.f_device_pointer(), nb.fshift_device_pointer(), self.box9, 3, self.listed_fep,
                                  d_q4=nb.q4_device_pointer(), elec_scale=self.listed_scales[0], epsfac=self.listed_scales[1],
                                  compute_energy=bool(sw.computeEnergy), compute_virial=bool(sw.computeVirial))

    def compute_forces(self, step_work=None):
        """x -> xq, clear, kernels, reduction: afterwards d_f holds the short-range forces in atom order (stream-ordered)"""
        sw = self.force_only if step_work is None else step_work
        self.nb.x_to_nbat_x(self.d_x.data_ptr(), 0, self.nslots)
        self.nb.clear_outputs(bool(sw.computeVirial))
        self.nb.launch_kernel(sw)
        self._launch_listed(sw)
        self.nb.force_reduction_execute(self.d_f.data_ptr(), None, self.stream)

    def step(self, step_index=0, step_work=None, seed=0, compute_virial=False, tc_lambdas=None):
        if self.rolling_prune_parts > 0:
            self.nb.launch_kernel_pruneonly(num_parts=self.rolling_prune_parts)
        if self.fused_update:
            sw = self.force_only if step_work is None else step_work
            if sw.computeEnergy or sw.computeVirial:
                self.nb.clear_outputs(bool(sw.computeVirial))     # energies and shift forces; the forces are clear already
            self.nb.launch_kernel(sw)
            self._launch_listed(sw)
            return self.update.integrate_fused(self.dt, compute_virial=compute_virial, tc_lambdas=tc_lambdas, seed=seed, step=step_index)
        self.compute_forces(step_work)
        return self.update.integrate(self.dt, update_velocities=True, compute_virial=compute_virial, tc_lambdas=tc_lambdas, seed=seed,
                                     step=step_index)

    def synchronize(self):
        import torch
        torch.cuda.synchronize()

    def free(self):
        self.update.free()
