"""Slab domain decomposition of the non-bonded path with a GPU halo exchange.

The reference decomposes space into a 3-D grid of domains, each with its own local + non-local grids and lists, and
moves halo coordinates / forces with pack -> MPI or peer copy -> unpack (domdec/gpuhaloexchange_impl_gpu.cpp:120-420,
kernels gpuhaloexchange_impl_gpu.cu:62-116).  MI355X-first this is laid out differently: a GPU has 288 GB of HBM, so
every rank keeps the *whole* system's coordinate and force arrays resident (16 MB for 10^6 atoms) and all ranks share
one slot numbering (the same cluster grid).  What is decomposed is the WORK:

  * rank r owns the i-super-clusters of slab r (equal counts along one box axis) and evaluates exactly the list
    entries whose i-super-cluster it owns (the half-shell list assigns every pair to one i-entry);
  * before a step, rank r needs current coordinates of the j-atoms of those entries that other ranks own (its halo):
    the owners pack them (nbnxm_gpu_halo_pack_x), RCCL sends them over xGMI, rank r unpacks into the same global
    indices;
  * after the kernel, the forces rank r accumulated on halo atoms travel the other way and are added by the owners
    (nbnxm_gpu_halo_unpack_f, accumulate).

Index maps are plain global atom indices, identical on both sides; there is no zone bookkeeping.  Coordinates are the
box's own (images are handled by the list's shift vectors), so no coordinate shift is applied in the halo.
The exchange is neighbour point-to-point (batched isend/irecv = ncclGroupStart/End on ROCm), not a collective.
"""
import numpy as np

CLUSTER = 8
SUPERCLUSTER = 64


def owners_by_slab(grid, num_ranks, axis=0):
    """Owner rank of every super-cluster: equal counts of super-clusters, ordered along `axis`."""
    nsc = grid.num_atoms // SUPERCLUSTER
    real = (grid.atomIndices >= 0).reshape(nsc, SUPERCLUSTER)
    coord = grid.xq.reshape(nsc, SUPERCLUSTER, 4)[:, :, axis].astype(np.float64)
    cnt = real.sum(axis=1)
    mean = np.where(cnt > 0, (coord * real).sum(axis=1) / np.maximum(cnt, 1), 0.0)
    order = np.argsort(mean, kind="stable")
    owner = np.empty(nsc, np.int32)
    owner[order] = (np.arange(nsc, dtype=np.int64) * num_ranks // nsc).astype(np.int32)
    return owner


class RankPlan:
    """What one rank evaluates and what it exchanges.  All atom indices are global (topology order)."""

    def __init__(self, rank, sci, cjPacked, recv_atoms, send_atoms, home_atoms):
        self.rank = rank
        self.sci = sci                # this rank's i-entries, cjPacked ranges renumbered into self.cjPacked
        self.cjPacked = cjPacked      # the packed j-groups of those entries, contiguous
        self.recv_atoms = recv_atoms  # {owner q: int32[] atoms owned by q whose x this rank needs / whose f it returns}
        self.send_atoms = send_atoms  # {rank q: int32[] atoms owned here that q needs}
        self.home_atoms = home_atoms  # int32[] atoms owned by this rank


class SlabDecomposition:
    """molecule_ids (one id per atom, optional): keep molecules whole — every atom is owned by the rank that owns the
    super-cluster of its molecule's first atom (the reference's update groups, domdec/updategroups.cpp, needed as soon as a
    rank constrains and integrates its home atoms).  Without it an atom belongs to the owner of its own super-cluster.  The
    WORK is decomposed by super-cluster either way; ownership only decides who sends coordinates and who collects forces."""

    def __init__(self, grid, plist, num_ranks, axis=0, molecule_ids=None):
        self.grid, self.plist, self.num_ranks = grid, plist, int(num_ranks)
        self.owner_sc = owners_by_slab(grid, num_ranks, axis)
        sci, cj = plist.sci, plist.cjPacked
        ngroups = (sci["cjPackedEnd"] - sci["cjPackedBegin"]).astype(np.int64)
        # owner of the i-entry each packed group belongs to (groups of an entry are contiguous)
        self._entry_owner = self.owner_sc[sci["sci"]]
        group_entry = np.repeat(np.arange(len(sci)), ngroups)
        group_index = np.concatenate([np.arange(b, e) for b, e in zip(sci["cjPackedBegin"], sci["cjPackedEnd"])]) \
            if len(sci) else np.zeros(0, np.int64)
        self._group_entry, self._group_index = group_entry, group_index
        # owner of every atom (topology order)
        ai = grid.atomIndices
        real = ai >= 0
        natoms = int(grid.natoms)
        self.owner_atom = np.zeros(natoms, np.int32)
        self.owner_atom[ai[real]] = self.owner_sc[np.nonzero(real)[0] // SUPERCLUSTER]
        if molecule_ids is not None:
            mol = np.asarray(molecule_ids)
            first = np.full(int(mol.max()) + 1, natoms, np.int64)
            np.minimum.at(first, mol, np.arange(natoms))
            self.owner_atom = self.owner_atom[first[mol]]
        # the clusters a rank's entries touch: the 8 i-clusters of each entry and the j-clusters of its groups
        cjs = cj["cj"][group_index]                                    # (ngroupsTotal, 4) j-cluster indices
        own_g = self._entry_owner[group_entry]
        ai_cl = ai.reshape(-1, CLUSTER)
        self._needs = {}
        for r in range(self.num_ranks):
            jcl = np.unique(cjs[own_g == r]) if len(cjs) else np.zeros(0, np.int64)
            isc = np.unique(sci["sci"][self._entry_owner == r]) if len(sci) else np.zeros(0, np.int64)
            icl = (isc[:, None] * (SUPERCLUSTER // CLUSTER) + np.arange(SUPERCLUSTER // CLUSTER)[None, :]).reshape(-1)
            atoms = ai_cl[np.union1d(jcl, icl).astype(np.int64)].reshape(-1)
            atoms = np.unique(atoms[atoms >= 0])
            own = self.owner_atom[atoms]
            for q in range(self.num_ranks):
                if q != r:
                    sel = atoms[own == q]
                    if len(sel):
                        self._needs[(r, q)] = sel.astype(np.int32)

    def plan(self, rank):
        sci, cj = self.plist.sci, self.plist.cjPacked
        mine = np.nonzero(self._entry_owner == rank)[0]
        new_sci = sci[mine].copy()
        n = (new_sci["cjPackedEnd"] - new_sci["cjPackedBegin"]).astype(np.int64)
        ends = np.cumsum(n)
        idx = np.concatenate([np.arange(b, e) for b, e in zip(new_sci["cjPackedBegin"], new_sci["cjPackedEnd"])]) \
            if len(mine) else np.zeros(0, np.int64)
        new_sci["cjPackedBegin"] = (ends - n).astype(np.int32)
        new_sci["cjPackedEnd"] = ends.astype(np.int32)
        new_cj = np.ascontiguousarray(cj[idx]) if len(idx) else cj[:0].copy()
        recv = {q: a for (r, q), a in self._needs.items() if r == rank}
        send = {r: a for (r, q), a in self._needs.items() if q == rank}
        home = np.nonzero(self.owner_atom == rank)[0].astype(np.int32)
        return RankPlan(rank, new_sci, new_cj, recv, send, home)


class HaloExchange:
    """Device-side pack / unpack of one rank's halo (maps and staging buffers live in HBM) plus the transfers.

    pack_fn / unpack_fn default to the HIP kernels behind the C ABI; the CPU tests of the exchange schedule inject
    index-copy stand-ins (test doubles, not a fallback: nothing in the package selects them).
    """

    def __init__(self, plan, device, pack_fn=None, unpack_fn=None):
        import torch
        self.plan, self.device = plan, device
        self.peers = sorted(set(plan.recv_atoms) | set(plan.send_atoms))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.recv_map = {q: t(a) for q, a in plan.recv_atoms.items()}
        self.send_map = {q: t(a) for q, a in plan.send_atoms.items()}
        z = lambda n: torch.zeros((n, 3), dtype=torch.float32, device=device)
        # x travels owner -> needer (send_map side packs); f travels needer -> owner (recv_map side packs)
        self.x_send = {q: z(len(m)) for q, m in self.send_map.items()}
        self.x_recv = {q: z(len(m)) for q, m in self.recv_map.items()}
        self.f_send = {q: z(len(m)) for q, m in self.recv_map.items()}
        self.f_recv = {q: z(len(m)) for q, m in self.send_map.items()}
        if pack_fn is None or unpack_fn is None:
            from . import halo_pack_x, halo_unpack_f
            pack_fn = pack_fn or (lambda stream, data, imap, out: halo_pack_x(stream, data.data_ptr(), imap.data_ptr(), imap.numel(),
                                                                          out.data_ptr(), None))
            unpack_fn = unpack_fn or (lambda stream, data, imap, buf, acc: halo_unpack_f(stream, data.data_ptr(), imap.data_ptr(),
                                                                                         imap.numel(), buf.data_ptr(), acc))
        self._pack, self._unpack = pack_fn, unpack_fn

    def bytes_per_step(self):
        n = sum(m.numel() for m in self.send_map.values()) + sum(m.numel() for m in self.recv_map.values())
        return 12 * n  # x out + f in on the send side, x in + f out on the recv side: each atom 12 B per direction

    # the four stages; a communicator moves x_send -> peer.x_recv and f_send -> peer.f_recv in between
    def pack_x(self, d_x, stream=None):
        for q, m in self.send_map.items():
            self._pack(stream, d_x, m, self.x_send[q])

    def unpack_x(self, d_x, stream=None):
        for q, m in self.recv_map.items():
            self._unpack(stream, d_x, m, self.x_recv[q], False)

    def pack_f(self, d_f, stream=None):
        for q, m in self.recv_map.items():
            self._pack(stream, d_f, m, self.f_send[q])

    def unpack_f(self, d_f, stream=None):
        for q, m in self.send_map.items():
            self._unpack(stream, d_f, m, self.f_recv[q], True)


class TorchDistComm:
    """Neighbour exchange with torch.distributed point-to-point ops (backend nccl = RCCL over xGMI; gloo in CPU tests)."""

    def __init__(self, dist):
        self.dist = dist

    def exchange(self, send_bufs, recv_bufs):
        ops = []
        for q, buf in sorted(recv_bufs.items()):
            if buf.numel():
                ops.append(self.dist.P2POp(self.dist.irecv, buf, q))
        for q, buf in sorted(send_bufs.items()):
            if buf.numel():
                ops.append(self.dist.P2POp(self.dist.isend, buf, q))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()

    def exchange_x(self, halo):
        self.exchange(halo.x_send, halo.x_recv)

    def exchange_f(self, halo):
        self.exchange(halo.f_send, halo.f_recv)


class LoopbackComm:
    """All ranks in one process (single-GPU tests of the decomposition): buffers are copied between the rank objects."""

    def __init__(self, halos):
        self.halos = halos  # list indexed by rank

    def exchange_all_x(self):
        for r, h in enumerate(self.halos):
            for q, buf in h.x_recv.items():
                buf.copy_(self.halos[q].x_send[r])

    def exchange_all_f(self):
        for r, h in enumerate(self.halos):
            for q, buf in h.f_recv.items():
                buf.copy_(self.halos[q].f_send[r])


class DomainStep:
    """One rank's force step on the GPU: halo x -> x to xq -> cluster kernel -> force reduction -> halo f.

    `nb` is the rank's NbnxmGpu object with its share of the list uploaded (plan.sci / plan.cjPacked and the full
    exclusion array); x and f live in HBM in atom order for the whole system, only home + halo entries are current.
    Everything is queued on the object's local stream; a communicator does the transfers between pack and unpack.
    """

    def __init__(self, nb, grid, plan, halo):
        import torch
        self.nb, self.grid, self.plan, self.halo = nb, grid, plan, halo
        self.natoms = int(grid.natoms)
        self.nslots = int(grid.num_atoms)
        ai = grid.atomIndices
        real = ai >= 0
        cell = np.full(self.natoms, -1, np.int32)
        cell[ai[real]] = np.nonzero(real)[0]
        dev = halo.device
        self.d_x = torch.zeros((self.natoms, 3), dtype=torch.float32, device=dev)
        self.d_f = torch.zeros((self.natoms, 3), dtype=torch.float32, device=dev)
        nb.init_x_to_nbat_x(ai)
        nb.force_reduction_reinit(cell, atom_start=0, accumulate=False)
        self.stream = nb.stream()
        torch.cuda.synchronize()

    def torch_stream(self):
        import torch
        return torch.cuda.ExternalStream(self.stream)

    def pack_x(self):
        self.halo.pack_x(self.d_x, self.stream)

    def compute(self, step_work):
        """after the x transfers: unpack, convert, kernel, reduce to atom order, pack the halo forces"""
        self.halo.unpack_x(self.d_x, self.stream)
        self.nb.x_to_nbat_x(self.d_x.data_ptr(), 0, self.nslots)
        self.nb.clear_outputs(False)
        self.nb.launch_kernel(step_work)
        self.nb.force_reduction_execute(self.d_f.data_ptr(), None, self.stream)
        self.halo.pack_f(self.d_f, self.stream)

    def unpack_f(self):
        self.halo.unpack_f(self.d_f, self.stream)

    def step(self, comm, step_work):
        """the whole step of a rank that has its own process (torch.distributed communicator)"""
        import torch
        with torch.cuda.stream(self.torch_stream()):
            self.pack_x()
            comm.exchange_x(self.halo)
            self.compute(step_work)
            comm.exchange_f(self.halo)
            self.unpack_f()


class DomainMdStep(DomainStep):
    """DomainStep plus the update of the rank's home atoms: halo x, kernels, halo f, then leap-frog | SD, LINCS, SETTLE on the
    molecules this rank owns (the decomposition must keep molecules whole: SlabDecomposition(molecule_ids=...)).  The update
    objects see the whole atom range — the coordinate and force arrays are global — with zero inverse mass and zero velocity
    outside the home atoms, so nothing else moves; settles / constraints are the home molecules' only."""

    def __init__(self, nb, grid, plan, halo, x0, v0, inverse_masses, dt, box, settles=None, settle_params=None, constraints=None,
                 constraint_lengths=None, **update_args):
        import torch
        from . import UpdateConstrainGpu
        super().__init__(nb, grid, plan, halo)
        home = np.zeros(self.natoms, bool)
        home[plan.home_atoms] = True
        im = np.where(home, np.asarray(inverse_masses, np.float64), 0.0)
        self.d_x.copy_(torch.from_numpy(np.ascontiguousarray(x0, np.float32)))
        v = np.where(home[:, None], np.asarray(v0, np.float64), 0.0)
        self.d_v = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(self.d_x.device)
        keep = lambda rows: None if rows is None else np.asarray(rows, np.int32)[home[np.asarray(rows, np.int32)[:, -1]]]
        st, cs = keep(settles), keep(constraints)
        for rows in (st, cs):
            if rows is not None and len(rows):
                atoms = rows[:, 1:] if rows is cs else rows
                assert home[atoms].all(), "a molecule is split over ranks: decompose with molecule_ids"
        self.dt = float(dt)
        self.update = UpdateConstrainGpu(dt, settle=settle_params, stream=self.stream, **update_args)
        ok = self.update.set(self.d_x.data_ptr(), self.d_v.data_ptr(), self.d_f.data_ptr(), im, None, cs, constraint_lengths, st)
        if not ok:
            raise ValueError("a group of coupled constraints is too large for the GPU LINCS")
        self.update.set_pbc(3, box)
        torch.cuda.synchronize()

    def integrate(self, step_index=0, seed=0, tc_lambdas=None):
        return self.update.integrate(self.dt, update_velocities=True, tc_lambdas=tc_lambdas, seed=seed, step=step_index)

    def md_step(self, comm, step_work, step_index=0):
        import torch
        self.step(comm, step_work)
        with torch.cuda.stream(self.torch_stream()):
            self.integrate(step_index)
