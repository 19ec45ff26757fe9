"""Domain decomposition + halo exchange on CPU (SURVEY §8 rows e / f2): the decomposition plan, the per-rank grids and pair
lists of gromacs-fep-gpu_amd/domdec.py, and the exchange schedule with gloo ranks.  The HIP kernels and the RCCL transport need
GPUs; here the CPU oracle evaluates each rank's two lists and a tensor-index test double (domdec.TensorHalo, constructed by the
tests only) moves the halo, so what is under test is: every pair is evaluated on exactly one rank, which atoms travel where
with which image shift, and that the owners end up with the forces of a single-domain evaluation."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _case(nm=(10, 10, 10)):
    import fep_testlib as tl
    return tl, tl.make_case(nm=nm, num_perturbed_molecules=0, elec="rf", seed=77)


def _domdec():
    from __graft_entry__ import load_package
    load_package()
    return importlib.import_module("gromacs_fep_gpu_amd.domdec")


def _single_domain(tl, c):
    import oracle_binding as ob
    g, full = c.grid, c.plist_fused
    r = ob.nbnxm_ref(full.sci, full.cjPacked, full.excl, g.xq, g.type, g.num_types, g.nbat_nbfp(c.sys["nbfp"]), tl.oracle_ref_params(c),
                     g.shift_vec)
    real = g.atomIndices >= 0
    f = np.zeros((c.natoms, 3))
    f[g.atomIndices[real]] = r["f"][real]
    return f, r


def _rank_system(tl, domdec, c, plan):
    return domdec.RankSystem(tl.pkg, plan, c.sys["box"], c.sys["qA"], c.sys["qB"], c.sys["typeA"], c.sys["typeB"], c.ntype, c.sys["molId"],
                             c.rlist, perturbed=c.perturbed)


def _rank_oracle(tl, c, system, xq=None):
    """forces of a rank's local + non-local list in rank atom order (home, then halo), energies, pairs within the cut-off"""
    import oracle_binding as ob
    g = system.grid
    f = np.zeros((g.num_atoms, 3))
    vc = vv = 0.0
    npairs = 0
    for pl in (system.local, system.nonlocal_):
        res = ob.nbnxm_ref(pl.sci, pl.cjPacked, pl.excl, g.xq if xq is None else xq, g.type, g.num_types, g.nbat_nbfp(c.sys["nbfp"]),
                           tl.oracle_ref_params(c), g.shift_vec)
        f += res["f"]
        vc, vv, npairs = vc + res["Vc"], vv + res["Vv"], npairs + res["npairs"]
    return f[system.cell], vc, vv, npairs


@pytest.mark.parametrize("nm,ncells,self_links", [
    ((10, 10, 10), (2, 1, 1), (False, False, False)),
    ((10, 10, 10), (2, 2, 1), (False, False, False)),
    ((10, 10, 10), (2, 2, 2), (False, False, False)),      # BASELINE configs[4]
    ((14, 8, 8), (3, 1, 1), (False, False, False)),
    ((10, 10, 10), (1, 1, 1), (True, False, False)),        # a rank that is its own neighbour
    ((10, 10, 10), (1, 1, 1), (True, True, True)),
])
def test_every_pair_is_evaluated_on_exactly_one_rank(nm, ncells, self_links):
    tl, c = _case(nm)
    domdec = _domdec()
    f0, r0 = _single_domain(tl, c)
    dd = domdec.DomainDecomposition(c.sys["x"], c.sys["box"], c.sys["molId"], ncells, c.rlist, self_links=self_links)
    f = np.zeros((c.natoms, 3))
    vc = vv = 0.0
    npairs = 0
    homes = []
    for r in range(dd.num_ranks):
        plan = dd.plan(r)
        homes.append(plan.home)
        system = _rank_system(tl, domdec, c, plan)
        fr, a, b, n = _rank_oracle(tl, c, system)
        np.add.at(f, plan.global_ids, fr)       # what the force halo does: halo rows go back to their owners
        vc, vv, npairs = vc + a, vv + b, npairs + n
        # molecules are whole on their home rank
        mol = c.sys["molId"]
        assert np.isin(mol[plan.home], mol[np.setdiff1d(np.arange(c.natoms), plan.home)]).sum() == 0
    assert np.array_equal(np.sort(np.concatenate(homes)), np.arange(c.natoms))      # one owner per atom
    assert npairs == r0["npairs"]                # atom pairs within the cut-off: the same count, so none twice and none missing
    rms = np.sqrt((f0 ** 2).sum(axis=1).mean())
    assert np.abs(f - f0).max() <= 5e-5 * rms    # float32 coordinates in shifted frames (x + box rounds differently)
    assert abs(vc - r0["Vc"]) <= 1e-6 * abs(r0["Vc"]) and abs(vv - r0["Vv"]) <= 1e-6 * abs(r0["Vv"])


def test_plans_agree_on_both_sides_of_every_link():
    tl, c = _case()
    domdec = _domdec()
    dd = domdec.DomainDecomposition(c.sys["x"], c.sys["box"], c.sys["molId"], (2, 2, 2), c.rlist)
    plans = [dd.plan(r) for r in range(8)]
    assert len(dd.directions) == 13
    for p in plans:
        a = p.halo_arrays()
        assert len(a["recv_peer"]) == len(set(a["recv_peer"])) and len(a["send_peer"]) == len(set(a["send_peer"]))   # one message per peer
        for k, src in enumerate(a["recv_peer"]):
            b = plans[src].halo_arrays()
            j = list(b["send_peer"]).index(p.rank)
            n = b["send_offset"][j + 1] - b["send_offset"][j]
            assert n == a["recv_count"][k]
            # the same atoms in the same order, and the image the sender packs is the one the receiver gridded
            sent_ids = plans[src].home[b["send_map"][b["send_offset"][j]:b["send_offset"][j + 1]]]
            recv_ids = p.global_ids[a["recv_offset"][k]:a["recv_offset"][k] + n]
            assert np.array_equal(sent_ids, recv_ids)
            sent_x = plans[src].x_rank[b["send_map"][b["send_offset"][j]:b["send_offset"][j + 1]]] \
                + b["shift_vectors"][b["send_shift_index"][b["send_offset"][j]:b["send_offset"][j + 1]]]
            assert np.abs(sent_x - p.x_rank[a["recv_offset"][k]:a["recv_offset"][k] + n]).max() < 2e-6
    st = dd.halo_statistics()
    assert st["links_per_rank"] == 13 and st["halo_atoms_max"] > 0


def test_default_grid():
    domdec = _domdec()
    assert domdec.default_grid(8) == (2, 2, 2) and domdec.default_grid(4) == (2, 2, 1) and domdec.default_grid(2) == (2, 1, 1)
    assert domdec.default_grid(1) == (1, 1, 1) and domdec.default_grid(6) == (3, 2, 1)


def _worker(rank, world, port, out_dir, ncells):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tl, c = _case()
    domdec = _domdec()
    dd = domdec.DomainDecomposition(c.sys["x"], c.sys["box"], c.sys["molId"], ncells, c.rlist)
    plan = dd.plan(rank)
    system = _rank_system(tl, domdec, c, plan)
    # this rank knows the coordinates of its home atoms only; the halo brings the rest
    x = torch.from_numpy(plan.x_rank.copy())
    x[plan.num_home:] = 1.0e6
    f = torch.zeros((plan.num_home + plan.num_halo, 3), dtype=torch.float32)
    halo = domdec.TensorHalo(dist)
    halo.reinit(plan, x, f)
    halo.communicate_coordinates()
    assert np.abs(x.numpy() - plan.x_rank).max() < 2e-6
    # the rank's two lists on the coordinates that arrived
    g = system.grid
    xq = g.xq.reshape(-1, 4).copy()
    real = g.atomIndices >= 0
    xq[real, :3] = x.numpy()[g.atomIndices[real]]
    fr, vc, vv, npairs = _rank_oracle(tl, c, system, xq=xq)
    f.copy_(torch.from_numpy(fr.astype(np.float32)))
    halo.communicate_forces(True)
    np.save(os.path.join(out_dir, "f_home_%d.npy" % rank), f.numpy()[:plan.num_home])
    np.save(os.path.join(out_dir, "home_%d.npy" % rank), plan.home)
    np.save(os.path.join(out_dir, "npairs_%d.npy" % rank), np.array([npairs]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ncells", [(2, 1, 1), (2, 2, 2)])
def test_gloo_ranks_with_halo_exchange_match_single_domain(tmp_path, ncells):
    world = int(np.prod(ncells))
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ncells), nprocs=world, join=True)
    tl, c = _case()
    f_ref, r0 = _single_domain(tl, c)
    f_dd = np.zeros((c.natoms, 3))
    npairs = 0
    for r in range(world):
        f_dd[np.load(os.path.join(str(tmp_path), "home_%d.npy" % r))] = np.load(os.path.join(str(tmp_path), "f_home_%d.npy" % r))
        npairs += int(np.load(os.path.join(str(tmp_path), "npairs_%d.npy" % r))[0])
    assert npairs == r0["npairs"]
    rms = np.sqrt((f_ref ** 2).sum(axis=1).mean())
    assert np.abs(f_dd - f_ref).max() <= 5e-5 * rms      # float32 transport, shifted frames, different summation order
