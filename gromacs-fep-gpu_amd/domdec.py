"""Domain decomposition of the non-bonded path on a 3-D grid of domains, one domain per GPU (SURVEY §8 rows e / f2,
BASELINE configs[4]: 2 x 2 x 2 over 8 MI355X), with the halo exchange of include/halo_hip.h (RCCL point-to-point).

What the reference does (domdec/domdec.cpp, domdec_setup.cpp; GPU halo: domdec/gpuhaloexchange_impl_gpu.cpp:122-511): space is
cut into nx x ny x nz cells; a rank owns the atoms of its cell (whole update groups), builds a LOCAL grid and pair list over its
home atoms and NON-LOCAL ones over the halo atoms it received (nbnxm/gridset.cpp, nbnxm/pairlist.cpp:3960-4100), computes
local and non-local interactions on two streams so that the halo traffic hides behind the local kernel
(mdlib/sim_util.cpp:1783-1899), and sends the forces on halo atoms back.  All of that is kept.  Two things are MI355X-first:

  * Half shell instead of eighth shell.  The reference receives halo atoms only from the "forward" neighbours (+x, +y, +z:
    7 zones) and therefore also has to evaluate pairs between two halo zones (zone pairs (1,3..5), (2,5), (3,5..6) of
    domdec_setup.cpp dd_zp3).  Here a rank receives from the 13 neighbour directions of the positive half space —
    (+,*,*), (0,+,*), (0,0,+) — and evaluates home x home and home x halo only: a pair of atoms in cells A and A + d is computed
    on A when d lies in the positive half, else on the other rank.  Every rank has the same 13 directions, so the load is
    balanced by construction, there are exactly two localities and no zone bookkeeping in the list builder, and the halo is
    hardly larger (faces dominate: 3 a^2 r in both schemes; numbers in DESIGN.md §6).
  * Direct transfers instead of pulses.  The reference forwards halo atoms dimension by dimension (x pulse, then y, then z),
    which keeps MPI messages few and large.  xGMI connects every GPU of a node with every other one, so each link is one
    direct send: one pack kernel, one ncclGroup with all sends and receives, coordinates land in place.

Coordinates on a rank live in the rank's own frame: along a decomposed dimension a halo atom arrives shifted by the box
vector of the image the receiver sees (applied while packing, packSendBufKernel's usePbc path), molecules are whole, nothing is
wrapped and the pair search looks at no periodic images; along a dimension that is not decomposed, coordinates are wrapped
into the box per atom and the pair search handles the images with the list's shift vectors, as in a single-domain run.

A link is one (peer rank, direction) pair.  With two cells along a dimension the +1 and -1 neighbours are the same rank seen
through different faces: two links, two images, one message per peer (the blocks are concatenated in a fixed order).
"""
import itertools

import numpy as np

CLUSTER = 8
SUPERCLUSTER = 64


def positive_half_directions(decomposed):
    """The neighbour directions a rank receives halo atoms from: the positive half space of {-1,0,1}^3, restricted to the
    decomposed dimensions (first non-zero component positive)."""
    out = []
    for d in itertools.product((-1, 0, 1), repeat=3):
        if any(d[k] != 0 and not decomposed[k] for k in range(3)):
            continue
        nz = [c for c in d if c != 0]
        if nz and nz[0] > 0:
            out.append(d)
    return out


def default_grid(num_ranks, box=None, rcomm=1.2):
    """The domain grid with the smallest halo for this box: halo volume of the positive half shell of an a x b x c cell is
    ((a + 2r)(b + 2r)(c + 2r) - abc) / 2 over the decomposed dimensions.  Without a box: the most cubic factorisation
    (2 x 2 x 2 for 8 ranks, the grid BASELINE configs[4] names); ties go to the more cubic grid."""
    cands = set()
    for nx in range(1, num_ranks + 1):
        if num_ranks % nx:
            continue
        for ny in range(1, num_ranks // nx + 1):
            if (num_ranks // nx) % ny == 0:
                cands.add((nx, ny, num_ranks // nx // ny))
    if box is None:
        cubic = min({tuple(sorted(c, reverse=True)) for c in cands}, key=lambda c: (c[0] - c[2], c))
        return cubic
    best = None
    for c in sorted(cands):
        cell = [float(box[k]) / c[k] for k in range(3)]
        if any(c[k] > 1 and cell[k] < rcomm + 0.2 for k in range(3)):
            continue
        grown = [cell[k] + (2 * rcomm if c[k] > 1 else 0.0) for k in range(3)]
        halo = 0.5 * (grown[0] * grown[1] * grown[2] - cell[0] * cell[1] * cell[2])
        key = (round(halo, 6), max(c) - min(c))
        if best is None or key < best[0]:
            best = (key, c)
    assert best is not None, "the box is too small for %d domains" % num_ranks
    return best[1]


class RankPlan:
    """One rank's share: its atoms in rank order (home, then the halo blocks), what it sends and what it receives."""

    def __init__(self, rank, cell, home, halo_blocks, send_blocks, x_rank, periodic):
        self.rank, self.cell = rank, cell
        self.home = home                    # int64[]: global ids of the home atoms, ascending
        self.halo_blocks = halo_blocks      # [(src rank, direction, shift(3), global ids)] in (src, direction) order
        self.send_blocks = send_blocks      # [(dst rank, direction, shift(3), local home indices)] in (dst, direction) order
        self.x_rank = x_rank                # float32 (nhome + nhalo, 3): coordinates in the rank's frame at the search
        self.periodic = periodic            # per dimension: handled by the pair list's shift vectors (not decomposed)
        self.num_home = len(home)
        self.num_halo = sum(len(b[3]) for b in halo_blocks)
        self.global_ids = np.concatenate([home] + [b[3] for b in halo_blocks]) if halo_blocks else home.copy()

    # ---- the arrays include/halo_hip.h takes -------------------------------------------------------------------------
    def halo_arrays(self):
        """one message per peer: blocks to / from the same rank are concatenated (fixed (rank, direction) order on both sides)"""
        send_peer, send_offset, send_map, send_shift_index, shifts = [], [0], [], [], []
        for dst, _d, sh, idx in self.send_blocks:
            if not send_peer or send_peer[-1] != dst:
                send_peer.append(dst)
                send_offset.append(send_offset[-1])
            shifts.append(sh)
            send_map.append(idx)
            send_shift_index.append(np.full(len(idx), len(shifts) - 1, np.int32))
            send_offset[-1] += len(idx)
        recv_peer, recv_offset, recv_count = [], [], []
        off = self.num_home
        for src, _d, _sh, ids in self.halo_blocks:
            if not recv_peer or recv_peer[-1] != src:
                recv_peer.append(src)
                recv_offset.append(off)
                recv_count.append(0)
            recv_count[-1] += len(ids)
            off += len(ids)
        cat = lambda parts, dt: np.concatenate(parts).astype(dt) if parts else np.zeros(0, dt)
        return dict(send_peer=np.array(send_peer, np.int32), send_offset=np.array(send_offset, np.int32),
                    send_map=cat(send_map, np.int32), send_shift_index=cat(send_shift_index, np.int32),
                    shift_vectors=np.array(shifts, np.float32).reshape(-1, 3) if shifts else np.zeros((1, 3), np.float32),
                    recv_peer=np.array(recv_peer, np.int32), recv_offset=np.array(recv_offset, np.int32),
                    recv_count=np.array(recv_count, np.int32))


class DomainDecomposition:
    """The decomposition at a search step, from the system's coordinates (what dd_partition_system does in the reference).

    Every rank computes the same cheap O(N) plan from the global coordinates (cell of every molecule, send lists by distance to
    the neighbour cells); the expensive parts — grid and pair lists — are built per rank over its home + halo atoms only."""

    def __init__(self, x, box, molecule_ids, ncells, rlist, self_links=(False, False, False)):
        """self_links[k]: treat dimension k as decomposed although it has ONE cell — the rank is then its own neighbour through
        the periodic boundary and exchanges a halo with itself (exercises the whole machinery, transport included, on one GPU)"""
        self.box = np.asarray(box, np.float64)
        self.ncells = tuple(int(n) for n in ncells)
        self.decomposed = tuple(n > 1 or bool(s) for n, s in zip(self.ncells, self_links))
        self.num_ranks = int(np.prod(self.ncells))
        self.rlist = float(rlist)
        x = np.asarray(x, np.float64)
        mol = np.asarray(molecule_ids)
        n = len(x)
        # molecules whole, next to their first atom, which is wrapped into the box: the update groups of the reference
        first = np.full(int(mol.max()) + 1, n, np.int64)
        np.minimum.at(first, mol, np.arange(n))
        xf = x[first[mol]]
        xf_wrapped = xf - self.box * np.floor(xf / self.box)
        d = x - xf
        d -= self.box * np.rint(d / self.box)
        self.x_whole = xf_wrapped + d
        self.x_wrapped = x - self.box * np.floor(x / self.box)
        self.mol_extent = float(np.abs(d).max()) if n else 0.0
        self.cell_size = self.box / np.array(self.ncells)
        cell = np.minimum((xf_wrapped / self.cell_size).astype(np.int64), np.array(self.ncells) - 1)
        self.atom_cell = cell
        self.owner = ((cell[:, 0] * self.ncells[1] + cell[:, 1]) * self.ncells[2] + cell[:, 2]).astype(np.int32)
        self.directions = positive_half_directions(self.decomposed)
        # a halo atom can be needed when it is within rlist of a home atom, and home atoms reach mol_extent out of their cell
        self.rcomm = self.rlist + self.mol_extent
        for k in range(3):
            if self.decomposed[k]:
                assert self.cell_size[k] >= self.rcomm + self.mol_extent, \
                    "cells along dimension %d are smaller than the communication range: fewer domains, please" % k
        self.periodic = tuple(not dk for dk in self.decomposed)
        self._home = [np.flatnonzero(self.owner == r) for r in range(self.num_ranks)]

    def rank_of_cell(self, c):
        return int((c[0] * self.ncells[1] + c[1]) * self.ncells[2] + c[2])

    def cell_of_rank(self, r):
        ny, nz = self.ncells[1], self.ncells[2]
        return (r // (ny * nz), (r // nz) % ny, r % nz)

    def _link(self, recv_cell, d):
        """source rank, the shift the receiver sees its atoms with, and which of the source's home atoms travel"""
        src_cell, shift = [], np.zeros(3)
        for k in range(3):
            c = recv_cell[k] + d[k]
            n = self.ncells[k]
            if c >= n:
                c -= n
                shift[k] = self.box[k]
            elif c < 0:
                c += n
                shift[k] = -self.box[k]
            src_cell.append(c)
        src = self.rank_of_cell(src_cell)
        ids = self._home[src]
        pos = self.x_whole[ids] + shift
        lo = np.array(recv_cell) * self.cell_size
        hi = lo + self.cell_size
        dist2 = np.zeros(len(ids))
        for k in range(3):
            if not self.decomposed[k]:
                continue                              # not decomposed: the cell is the whole box along k
            gap = np.maximum(np.maximum(lo[k] - pos[:, k], pos[:, k] - hi[k]), 0.0)
            dist2 += gap * gap
        sel = dist2 < self.rcomm ** 2
        return src, shift, ids[sel]

    def plan(self, rank):
        cell = self.cell_of_rank(rank)
        home = self._home[rank]
        halo = []
        for di, d in enumerate(self.directions):
            src, shift, ids = self._link(cell, d)
            if len(ids):
                halo.append((src, di, shift, ids))
        halo.sort(key=lambda b: (b[0], b[1]))
        # what this rank sends: it is the source of the link (receiver cell = my cell - d, direction d)
        local_of = np.full(len(self.owner), -1, np.int64)
        local_of[home] = np.arange(len(home))
        send = []
        for di, d in enumerate(self.directions):
            recv_cell = tuple((cell[k] - d[k]) % self.ncells[k] for k in range(3))
            src, shift, ids = self._link(recv_cell, d)
            assert src == rank
            if len(ids):
                send.append((self.rank_of_cell(recv_cell), di, shift, local_of[ids]))
        send.sort(key=lambda b: (b[0], b[1]))
        # coordinates in the rank's frame: whole molecules (+ image shift) along decomposed dimensions, wrapped per atom elsewhere
        parts = [self.x_whole[home]] + [self.x_whole[b[3]] + b[2] for b in halo]
        x_rank = np.concatenate(parts) if parts else np.zeros((0, 3))
        ids_all = np.concatenate([home] + [b[3] for b in halo]) if halo else home
        for k in range(3):
            if self.periodic[k]:
                x_rank[:, k] = self.x_wrapped[ids_all, k]
        return RankPlan(rank, cell, home, halo, send, x_rank.astype(np.float32), self.periodic)

    def halo_statistics(self):
        """halo atoms and bytes per rank and step (coordinates in + forces out of a rank: 12 B per atom each way)"""
        counts = []
        for r in range(self.num_ranks):
            cell = self.cell_of_rank(r)
            counts.append(sum(len(self._link(cell, d)[2]) for d in self.directions))
        home = [len(h) for h in self._home]
        return dict(home_atoms_mean=float(np.mean(home)), home_atoms_max=int(np.max(home)), halo_atoms_mean=float(np.mean(counts)),
                    halo_atoms_max=int(np.max(counts)), halo_over_home=float(np.mean(counts) / max(1.0, np.mean(home))),
                    bytes_sent_per_rank_per_step=int(24 * np.mean(counts)), bytes_sent_and_received_per_rank_per_step=int(48 * np.mean(counts)),
                    links_per_rank=len(self.directions))


class RankSystem:
    """The non-bonded inputs of one rank: cluster grid over home + halo atoms (two zones), local and non-local pair list."""

    def __init__(self, pkg, plan, box, qA, qB, typeA, typeB, ntype, molecule_ids, rlist, perturbed=None, max_cjpacked_per_sci=16):
        ids = plan.global_ids
        self.plan = plan
        pert = None if perturbed is None else np.asarray(perturbed)[ids].astype(np.uint8)
        self.grid = pkg.Grid(plan.x_rank, box, np.asarray(qA)[ids], np.asarray(qB)[ids], np.asarray(typeA)[ids], np.asarray(typeB)[ids],
                             ntype, perturbed=pert, num_home=plan.num_home, periodic=plan.periodic)
        # exclusions inside molecules; a molecule is whole on its home rank, halo atoms take part in no exclusion here
        mol = np.asarray(molecule_ids)[ids].astype(np.int64)
        group = np.concatenate([mol[:plan.num_home], -1 - np.arange(plan.num_halo)])
        self.excl_index, self.excl_atoms = pkg.exclusions_from_groups(group)
        self.local = self.grid.build_pairlist_dd(False, self.excl_index, self.excl_atoms, rlist, max_cjpacked_per_sci)
        self.nonlocal_ = self.grid.build_pairlist_dd(True, self.excl_index, self.excl_atoms, rlist, max_cjpacked_per_sci)
        g = self.grid
        real = g.atomIndices >= 0
        self.cell = np.full(len(ids), -1, np.int32)       # rank atom -> grid slot
        self.cell[g.atomIndices[real]] = np.nonzero(real)[0]
        assert (self.cell >= 0).all()
        assert (self.cell[:plan.num_home] < g.num_atoms_home).all() and (self.cell[plan.num_home:] >= g.num_atoms_home).all()


# ---- transports ---------------------------------------------------------------------------------------------------------

TRANSPORT_RCCL, TRANSPORT_PEER_COPY, TRANSPORT_PEER_PUSH, TRANSPORT_IPC_PUSH = 0, 1, 2, 3      # include/halo_hip.h


def _halo_lib(pkg):
    import ctypes as C
    lib = pkg.hip_lib()
    lib.halo_gpu_create.restype = C.c_void_p
    lib.halo_gpu_last_error.restype = C.c_char_p
    lib.halo_gpu_coordinates_ready_event.restype = C.c_void_p
    lib.halo_gpu_forces_ready_event.restype = C.c_void_p
    lib.halo_gpu_bytes_per_step.restype = C.c_longlong
    return lib


def new_halo_id(pkg, transport=TRANSPORT_RCCL):
    """halo_gpu_get_unique_id_ex: the 128 bytes every rank of one communicator hands to its transport object; the id selects the
    transport (RCCL | in-process peer copies)"""
    import ctypes as C
    lib = _halo_lib(pkg)
    uid = np.zeros(128, np.uint8)
    if lib.halo_gpu_get_unique_id_ex(uid.ctypes.data_as(C.c_void_p), C.c_int(transport)) != 0:
        raise RuntimeError("halo_gpu_get_unique_id_ex: %s" % lib.halo_gpu_last_error().decode())
    return uid


class RcclHalo:
    """include/halo_hip.h: the exchange inside libnbnxm_hip.so.  Default: ncclSend / ncclRecv groups on the non-local stream, one
    process per rank, the id broadcast over `dist`.  unique_id: an id made by new_halo_id (the same bytes for every rank) — with
    TRANSPORT_PEER_COPY the ranks are threads of this process and copy device to device out of each other's buffers; every rank then
    needs its own host thread for the calls that exchange data (run_ranks_in_threads)."""

    def __init__(self, pkg, dist, rank, num_ranks, stream, unique_id=None, ipc_push=False):
        import ctypes as C
        import torch
        self._C, self._lib = C, _halo_lib(pkg)
        lib = self._lib
        # ipc_push: the one-sided transport between processes (HALO_GPU_TRANSPORT_IPC_PUSH): reinit() exchanges the ranks' export
        # records over `dist` (an all-gather) — the only use of the process group; the steps themselves have no collective
        self._ipc_push, self._dist, self._rank, self._num_ranks = bool(ipc_push), dist, int(rank), int(num_ranks)
        if ipc_push:
            unique_id = new_halo_id(pkg, TRANSPORT_IPC_PUSH)
        if unique_id is not None:
            uid = np.ascontiguousarray(unique_id, np.uint8)
        else:
            uid = np.zeros(128, np.uint8)
            if rank == 0:
                if lib.halo_gpu_get_unique_id(uid.ctypes.data_as(C.c_void_p)) != 0:
                    raise RuntimeError("halo_gpu_get_unique_id: %s" % lib.halo_gpu_last_error().decode())
            if num_ranks > 1:
                # the id travels over the process group that exists anyway (any out-of-band channel would do)
                backend = dist.get_backend()
                t = torch.from_numpy(uid).to("cuda" if backend == "nccl" else "cpu")
                dist.broadcast(t, 0)
                uid = t.cpu().numpy()
        self._h = lib.halo_gpu_create(uid.ctypes.data_as(C.c_void_p), C.c_int(rank), C.c_int(num_ranks), C.c_void_p(stream))
        if not self._h:
            raise RuntimeError("halo_gpu_create: %s" % lib.halo_gpu_last_error().decode())

    def reinit(self, plan, d_x, d_f):
        C = self._C
        a = plan.halo_arrays()
        self._keep = a
        p = lambda v: v.ctypes.data_as(C.c_void_p)
        self._lib.halo_gpu_reinit(C.c_void_p(self._h), C.c_void_p(d_x.data_ptr()), C.c_void_p(d_f.data_ptr()), C.c_int(plan.num_home),
                                  C.c_int(len(a["send_peer"])), p(a["send_peer"]), p(a["send_offset"]), p(a["send_map"]),
                                  p(a["send_shift_index"]), C.c_int(len(a["shift_vectors"])), p(a["shift_vectors"]),
                                  C.c_int(len(a["recv_peer"])), p(a["recv_peer"]), p(a["recv_offset"]), p(a["recv_count"]))
        if self._ipc_push:
            self._exchange_push_records()

    def _exchange_push_records(self):
        """halo_gpu_push_export on every rank, an all-gather of the records, halo_gpu_push_import"""
        import torch
        C, lib = self._C, self._lib
        lib.halo_gpu_push_export_bytes.restype = C.c_int
        nbytes = int(lib.halo_gpu_push_export_bytes())
        mine = np.zeros(nbytes, np.uint8)
        if lib.halo_gpu_push_export(C.c_void_p(self._h), mine.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError("halo_gpu_push_export: %s" % lib.halo_gpu_last_error().decode())
        if self._num_ranks > 1:
            dist = self._dist
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = torch.from_numpy(mine).to(dev)
            parts = [torch.empty_like(t) for _ in range(self._num_ranks)]
            dist.all_gather(parts, t)
            everyone = np.concatenate([q.cpu().numpy() for q in parts])
        else:
            everyone = mine
        everyone = np.ascontiguousarray(everyone, np.uint8)
        rc = lib.halo_gpu_push_import(C.c_void_p(self._h), everyone.ctypes.data_as(C.c_void_p), C.c_int(self._num_ranks))
        message = lib.halo_gpu_last_error().decode() if rc != 0 else ""
        if self._num_ranks > 1:
            # every rank has opened the others' buffers before any rank stores into them — and every rank learns whether ALL could:
            # a rank that cannot open a peer's buffer must not leave the others waiting inside the next collective
            ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
            self._dist.all_reduce(ok, op=self._dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                raise RuntimeError("halo_gpu_push_import failed on %s: %s" % ("this rank" if rc != 0 else "another rank", message))
        elif rc != 0:
            raise RuntimeError("halo_gpu_push_import: %s" % message)

    def push_status(self):
        """0, or 1 + the link a kernel of the one-sided transport gave up waiting for"""
        self._lib.halo_gpu_push_status.restype = self._C.c_int
        return int(self._lib.halo_gpu_push_status(self._C.c_void_p(self._h)))

    def communicate_coordinates(self, dependency_event=None):
        self._lib.halo_gpu_communicate_coordinates(self._C.c_void_p(self._h), self._C.c_void_p(dependency_event))

    def communicate_forces(self, accumulate=True, dependency_event=None):
        self._lib.halo_gpu_communicate_forces(self._C.c_void_p(self._h), self._C.c_int(1 if accumulate else 0), self._C.c_void_p(dependency_event))

    def forces_ready_event(self):
        return self._lib.halo_gpu_forces_ready_event(self._C.c_void_p(self._h))

    def coordinates_ready_event(self):
        return self._lib.halo_gpu_coordinates_ready_event(self._C.c_void_p(self._h))

    def bytes_per_step(self):
        return int(self._lib.halo_gpu_bytes_per_step(self._C.c_void_p(self._h)))

    def domain_force_step(self, nb, step_work, num_home_slots, num_slots, num_atoms, coordinates_ready_event=None):
        """halo_gpu_domain_force_step: the whole two-locality force step of the domain, host side in C++"""
        C = self._C
        self._lib.halo_gpu_domain_force_step(C.c_void_p(self._h), nb.h, C.byref(step_work), C.c_int(num_home_slots), C.c_int(num_slots),
                                             C.c_int(num_atoms), C.c_void_p(coordinates_ready_event))

    def free(self):
        if getattr(self, "_h", None):
            self._lib.halo_gpu_free(self._C.c_void_p(self._h))
            self._h = None


def run_ranks_in_threads(calls):
    """calls: one function per in-process rank; each runs on its own host thread (the peer-copy transport blocks a rank's thread until
    its peers have queued their side of an exchange; ctypes releases the interpreter lock inside the library).  Re-raises the first
    exception of any rank."""
    import threading
    errors = [None] * len(calls)

    def run(i):
        try:
            calls[i]()
        except BaseException as e:      # noqa: BLE001 (handed to the caller's thread)
            errors[i] = e

    threads = [threading.Thread(target=run, args=(i,), name="halo-rank-%d" % i) for i in range(len(calls))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e


class ThreadRanks:
    """A stand-in for torch.distributed among the host THREADS of one process (the ranks of the in-process peer-copy transport): just
    what bench_dd.measure uses — broadcast, all_reduce (sum | max), barrier.  One `view(rank)` per thread.  It exists so that the
    decomposition leg of bench.py can be walked through with several ranks on a one-GPU box (tests/test_gpu_bench_dd.py); nothing in
    the product path uses it."""

    class ReduceOp:
        SUM, MAX = "sum", "max"

    def __init__(self, pkg, world):
        import threading
        self.world = int(world)
        self._barrier = threading.Barrier(self.world)
        self._slots = [None] * self.world
        self.peer_copy_id = new_halo_id(pkg, TRANSPORT_PEER_COPY)

    def view(self, rank):
        outer = self

        class View:
            ReduceOp = ThreadRanks.ReduceOp
            peer_copy_id = outer.peer_copy_id

            def is_initialized(self):
                return True

            def get_world_size(self):
                return outer.world

            def get_backend(self):
                return "threads"

            def barrier(self):
                outer._barrier.wait()

            def broadcast(self, t, src):
                outer._slots[rank] = t
                outer._barrier.wait()
                if rank != src:
                    t.copy_(outer._slots[src])
                outer._barrier.wait()

            def all_reduce(self, t, op="sum"):
                import torch
                outer._slots[rank] = t.clone()
                outer._barrier.wait()
                parts = torch.stack([outer._slots[r].to(t.device) for r in range(outer.world)])
                t.copy_(parts.max(dim=0).values if op == "max" else parts.sum(dim=0))
                outer._barrier.wait()

        return View()


class TensorHalo:
    """TEST DOUBLE of the transport (CPU tests with gloo, and all ranks of a decomposition inside one process): the same
    schedule — pack with the image shift, one message per peer, coordinates land in place, forces are added through the send
    map — written with tensor index operations.  Nothing in the package selects it; tests construct it."""

    def __init__(self, dist=None, peers=None):
        self.dist, self.peers = dist, peers      # peers: {rank: TensorHalo} for the in-process loopback

    def reinit(self, plan, d_x, d_f):
        import torch
        a = plan.halo_arrays()
        dev = d_x.device
        self.a, self.d_x, self.d_f, self.rank = a, d_x, d_f, plan.rank
        self.send_map = torch.from_numpy(a["send_map"].astype(np.int64)).to(dev)
        self.send_shift = torch.from_numpy(a["shift_vectors"][a["send_shift_index"]] if len(a["send_map"]) else np.zeros((0, 3), np.float32)).to(dev)
        self.send_buf = torch.zeros((len(a["send_map"]), 3), dtype=torch.float32, device=dev)

    def _segments(self):
        a = self.a
        sends = [(int(p), int(a["send_offset"][k]), int(a["send_offset"][k + 1])) for k, p in enumerate(a["send_peer"])]
        recvs = [(int(p), int(o), int(o + c)) for p, o, c in zip(a["recv_peer"], a["recv_offset"], a["recv_count"])]
        return sends, recvs

    def _exchange(self, out_parts, in_parts):
        """out_parts / in_parts: [(peer, tensor view)]"""
        if self.peers is not None:
            raise RuntimeError("loopback halos are driven by exchange_all_*")
        ops = []
        for peer, t in in_parts:
            ops.append(self.dist.P2POp(self.dist.irecv, t, peer))
        for peer, t in out_parts:
            ops.append(self.dist.P2POp(self.dist.isend, t, peer))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()

    def pack_coordinates(self):
        if len(self.send_map):
            self.send_buf.copy_(self.d_x[self.send_map] + self.send_shift)

    def communicate_coordinates(self, dependency_event=None):
        self.pack_coordinates()
        sends, recvs = self._segments()
        tmp = [(p, self.d_x.new_zeros((e - b, 3))) for p, b, e in recvs]
        self._exchange([(p, self.send_buf[b:e].contiguous()) for p, b, e in sends], tmp)
        for (p, b, e), (_, t) in zip(recvs, tmp):
            self.d_x[b:e] = t

    def communicate_forces(self, accumulate=True, dependency_event=None):
        sends, recvs = self._segments()
        tmp = [(p, self.d_f.new_zeros((e - b, 3))) for p, b, e in sends]
        self._exchange([(p, self.d_f[b:e].contiguous()) for p, b, e in recvs], tmp)
        for (p, b, e), (_, t) in zip(sends, tmp):
            self.send_buf[b:e] = t
        self.unpack_forces(accumulate)

    def unpack_forces(self, accumulate=True):
        if len(self.send_map):
            if accumulate:
                self.d_f.index_add_(0, self.send_map, self.send_buf)
            else:
                self.d_f[self.send_map] = self.send_buf

    def forces_ready_event(self):
        return None

    def bytes_per_step(self):
        return 12 * (len(self.a["send_map"]) + int(self.a["recv_count"].sum()))

    def free(self):
        pass


def loopback_exchange_coordinates(halos):
    """all ranks in one process: every rank packs, then the blocks are copied to where RCCL would deliver them"""
    for h in halos:
        h.pack_coordinates()
    for h in halos:
        sends, _ = h._segments()
        for peer, b, e in sends:
            _, recvs = halos[peer]._segments()
            (rb, re_), = [(rb, re_) for p, rb, re_ in recvs if p == h.rank]
            halos[peer].d_x[rb:re_] = h.send_buf[b:e]


def loopback_exchange_forces(halos, accumulate=True):
    for h in halos:
        _, recvs = h._segments()
        for peer, b, e in recvs:
            sends, _ = halos[peer]._segments()
            (sb, se), = [(sb, se) for p, sb, se in sends if p == h.rank]
            halos[peer].send_buf[sb:se] = h.d_f[b:e]
    for h in halos:
        h.unpack_forces(accumulate)


# ---- one rank's force step on the GPU ---------------------------------------------------------------------------------------

class DomainStep:
    """Force step of one domain with the reference's two-locality schedule (mdlib/sim_util.cpp:1783-1924):

        non-local stream:  halo x (pack, send / receive)  ->  x -> xq (halo slots)  ->  non-local kernel
                           ->  forces of the halo atoms to atom order  ->  halo f (send, receive, add to the home atoms)
        local stream:      x -> xq (home slots)  ->  local kernel   [overlaps all of the above]
                           ->  wait for the non-local kernel and the halo forces  ->  forces of the home atoms (+=)

    d_x / d_f: float3 per atom in rank order (home, then halo) in HBM.  `halo` is an RcclHalo (or a test double)."""

    def __init__(self, pkg, nb, system, halo, device="cuda", poison_halo_rows=None):
        import torch
        self.pkg, self.nb, self.sys, self.halo = pkg, nb, system, halo
        plan, g = system.plan, system.grid
        self.num_home, self.num_all = plan.num_home, plan.num_home + plan.num_halo
        self.home_slots, self.all_slots = g.num_atoms_home, g.num_atoms
        self.d_x = torch.from_numpy(plan.x_rank.copy()).to(device)
        if poison_halo_rows is not None:      # tests: the exchange has to supply every halo row (set BEFORE the peers may store into them)
            self.d_x[plan.num_home:] = poison_halo_rows
            torch.cuda.synchronize()
        self.d_f = torch.zeros((self.num_all, 3), dtype=torch.float32, device=device)
        nb.init_x_to_nbat_x(g.atomIndices)
        nb.force_reduction_reinit(system.cell, atom_start=0, accumulate=False)
        self.s_local, self.s_nonlocal = nb.stream(pkg.LOCAL), nb.stream(pkg.NONLOCAL)
        self.ev_nonlocal_kernel = torch.cuda.Event()
        self.ev_local_done = torch.cuda.Event()
        self.ts_local = torch.cuda.ExternalStream(self.s_local)
        self.ts_nonlocal = torch.cuda.ExternalStream(self.s_nonlocal)
        halo.reinit(plan, self.d_x, self.d_f)
        torch.cuda.synchronize()

    def launch(self, step_work):
        """everything up to and including the non-local kernel (the halo coordinates must have been exchanged: call
        halo.communicate_coordinates() first, or loopback_exchange_coordinates for in-process ranks)"""
        nb, pkg = self.nb, self.pkg
        nb.clear_outputs(bool(step_work.computeVirial))
        nb.x_to_nbat_x(self.d_x.data_ptr(), 0, self.home_slots, pkg.LOCAL, insert_nonlocal_dependency=True)
        nb.launch_kernel(step_work, pkg.LOCAL)
        nb.x_to_nbat_x(self.d_x.data_ptr(), self.home_slots, self.all_slots, pkg.NONLOCAL, insert_nonlocal_dependency=True)
        nb.launch_kernel(step_work, pkg.NONLOCAL)
        self.ev_nonlocal_kernel.record(self.ts_nonlocal)

    def reduce_halo_forces(self):
        """halo atoms' forces (only the non-local kernel writes them) to atom order, on the non-local stream; the home rows are
        zeroed there too, so that the forces arriving from the other ranks can be added before the local kernel is done"""
        import torch
        with torch.cuda.stream(self.ts_nonlocal):
            self.d_f[:self.num_home].zero_()
        self.nb.force_reduction_execute_range(self.d_f.data_ptr(), self.num_home, self.num_all, False, self.s_nonlocal)

    def reduce_home_forces(self):
        """after the local AND the non-local kernel (both add to home slots) and the halo forces — everything queued on the
        non-local stream so far: f_home += nbnxm forces, on the local stream"""
        self.ts_local.wait_stream(self.ts_nonlocal)
        self.nb.force_reduction_execute_range(self.d_f.data_ptr(), 0, self.num_home, True, self.s_local)
        self.ev_local_done.record(self.ts_local)

    def step(self, step_work, coordinates_ready_event=None):
        """the whole step of a rank that has its own process: one call into the library when the transport is the RCCL one
        (halo_gpu_domain_force_step, host side in C++), else the same schedule spelled out"""
        if hasattr(self.halo, "domain_force_step"):
            self.halo.domain_force_step(self.nb, step_work, self.home_slots, self.all_slots, self.num_all, coordinates_ready_event)
            return
        self.halo.communicate_coordinates()
        self.launch(step_work)
        self.reduce_halo_forces()
        self.halo.communicate_forces(True)
        self.reduce_home_forces()

    def home_forces(self):
        import torch
        torch.cuda.synchronize()
        return self.d_f[:self.num_home].cpu().numpy()


class DomainMdStep(DomainStep):
    """DomainStep plus the update of the rank's home atoms (leap-frog | SD, LINCS, SETTLE on whole home molecules): the rows
    [0, num_home) of the rank's coordinate, velocity and force arrays are the update's whole world — no masks, no global arrays.
    inverse_masses / v0: per GLOBAL atom; settles / constraints: rows of GLOBAL atom indices (the home molecules' are picked)."""

    def __init__(self, pkg, nb, system, halo, v0, inverse_masses, dt, box, settles=None, settle_params=None, constraints=None,
                 constraint_lengths=None, device="cuda", **update_args):
        import torch
        super().__init__(pkg, nb, system, halo, device)
        plan = system.plan
        home = plan.home
        local_of = {}
        local_index = np.full(int(max(home.max() if len(home) else 0, 0)) + 1, -1, np.int64)
        local_index[home] = np.arange(len(home))

        def to_local(rows, first_col):
            if rows is None:
                return None
            rows = np.asarray(rows, np.int64)
            keep = np.isin(rows[:, -1], home)
            rows = rows[keep].copy()
            loc = local_index[rows[:, first_col:]]
            assert (loc >= 0).all(), "a molecule is split over ranks"
            rows[:, first_col:] = loc
            return rows.astype(np.int32), keep
        st = to_local(settles, 0)
        cs = to_local(constraints, 1)
        self.d_v = torch.from_numpy(np.ascontiguousarray(np.asarray(v0)[home], np.float32)).to(device)
        self.dt = float(dt)
        self.update = pkg.UpdateConstrainGpu(dt, settle=settle_params, stream=self.s_local, **update_args)
        lengths = None if cs is None else np.asarray(constraint_lengths)
        ok = self.update.set(self.d_x.data_ptr(), self.d_v.data_ptr(), self.d_f.data_ptr(), np.asarray(inverse_masses)[home], None,
                             None if cs is None else cs[0], lengths, None if st is None else st[0])
        if not ok:
            raise ValueError("a group of coupled constraints is too large for the GPU LINCS")
        self.update.set_pbc(3, box)
        self._x_ready = None
        self.update.x_updated_event()       # a consumer exists: the update records the event from its first step on
        torch.cuda.synchronize()
        del local_of

    def integrate(self, step_index=0, seed=0, tc_lambdas=None):
        """on the local stream, behind reduce_home_forces"""
        return self.update.integrate(self.dt, update_velocities=True, tc_lambdas=tc_lambdas, seed=seed, step=step_index)

    def md_step(self, step_work, step_index=0):
        """force step + update of a rank with its own process: the next step's coordinate reads (x -> xq on the local stream, the
        halo pack on the non-local one) wait for this step's update through the update's x-updated event"""
        self.step(step_work, self._x_ready)
        self.integrate(step_index)
        self._x_ready = self.update.x_updated_event()


def make_rank_gpu(pkg, wl, case, system, use_dynamic_pruning=True, merged=False):
    """the NbnxmGpu object of one domain: both localities, fused perturbed pairs, home + halo atom data.  merged: the two lists as ONE
    device list and one launch per step (nbnxm_gpu_set_merged_localities; the C++ step halo_gpu_domain_force_step knows the schedule)"""
    g = system.grid
    nb = pkg.NbnxmGpu(wl.gpu_interaction_params(case, use_dynamic_pruning), g.num_types, g.nbat_nbfp(case.sys["nbfp"]),
                      local_and_nonlocal=True, fep=True, n_lambda=case.n_lambda)
    if merged:
        nb.set_merged_localities(True)
    sig6 = case.sc_sigma ** 6
    nb.copy_fepparams(case.sc_alpha if case.sc_coul else 0.0, case.sc_alpha, case.sc_power, sig6, sig6 if case.sc_coul else 0.0,
                      case.lambda_coul, case.lambda_vdw, case.all_lambda, case.all_lambda)
    nb.init_atomdata(g.num_atoms, g.type, qA=g.qA, qB=g.qB, typeA=g.typeA, typeB=g.typeB, num_atoms_local=g.num_atoms_home)
    nb.init_pairlist(system.local.sci, system.local.cjPacked, system.local.excl, iloc=pkg.LOCAL)
    nb.init_pairlist(system.nonlocal_.sci, system.nonlocal_.cjPacked, system.nonlocal_.excl, iloc=pkg.NONLOCAL)
    nb.init_fep_cluster_bits(g.fepBits)
    nb.set_fep_mode(True)
    nb.upload_shiftvec(g.shift_vec)
    nb.copy_xq_to_gpu(g.xq, pkg.LOCAL)
    nb.copy_xq_to_gpu(g.xq, pkg.NONLOCAL)
    return nb
