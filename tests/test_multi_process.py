"""The N>1 path on CPU: two processes (gloo), static history-range split, one reduce of the detector arrays onto
rank 0.  The photon loop stand-in here is the CPU oracle with the engine's per-history streams (the real engine
needs a GPU); what is under test is the sharding rule (pmc_history_range)
and that the detector arrays and the radiation field of the shards add up to those of the undivided segment."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, ski


def _worker(rank, world, port, n, outfile):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from skirt9_amd.engine import history_range
    from skirt9_amd.host import Simulation
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sim = Simulation(ski("cfg2small.ski"), num_packets=n).setup()  # full replica on every rank
    first, count = history_range(n, rank, world)
    frames, _ = O.run_primary(sim, first, count, O.RNG_PHILOX, seed=11)
    t = torch.from_numpy(frames)
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)  # (on the GPU: pmc_reduce_frames, one ncclReduce onto the root)
    if rank == 0:
        np.save(outfile, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_history_range_partition():
    from skirt9_amd.engine import history_range
    for n in (0, 1, 7, 1000, 10 ** 9 + 7):
        for world in (1, 2, 3, 8):
            pieces = [history_range(n, r, world) for r in range(world)]
            assert pieces[0][0] == 0
            assert sum(c for _, c in pieces) == n
            for (f0, c0), (f1, _) in zip(pieces, pieces[1:]):
                assert f0 + c0 == f1


def test_two_ranks_equal_one(tmp_path):
    import oracle_lib as O
    from skirt9_amd.host import Simulation
    n = 4001
    out = str(tmp_path / "frames.npy")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, n, out), nprocs=2, join=True)
    both = np.load(out)
    sim = Simulation(ski("cfg2small.ski"), num_packets=n).setup()
    single, _ = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=11)
    lay = sim.layout(0)
    assert both[lay.wsed_offset] == n  # every history counted once
    assert np.allclose(both, single, rtol=1e-12, atol=0)


def _rf_worker(rank, world, port, n, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from skirt9_amd.engine import history_range
    from skirt9_amd.host import Simulation
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sim = Simulation(ski("cfg3rf.ski"), num_packets=n).setup()
    first, count = history_range(n, rank, world)
    _, rf, _ = O.run_primary_rf(sim, first, count, O.RNG_PHILOX, seed=3)
    t = torch.from_numpy(rf)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)  # (on the GPU: pmc_allreduce_radiation_field)
    np.save(os.path.join(outdir, f"rf{rank}.npy"), t.numpy())  # every rank must hold the whole field (sumToAll)
    dist.barrier()
    dist.destroy_process_group()


def test_radiation_field_allreduce_two_ranks(tmp_path):
    """MediumSystem::communicateRadiationField (MediumSystem.cpp:1304-1313): the per-rank tables add up, on EVERY rank,
    to the table of the undivided segment"""
    import oracle_lib as O
    from skirt9_amd.host import Simulation
    n = 3001
    port = 31500 + os.getpid() % 2000
    mp.spawn(_rf_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    sim = Simulation(ski("cfg3rf.ski"), num_packets=n).setup()
    _, single, _ = O.run_primary_rf(sim, 0, n, O.RNG_PHILOX, seed=3)
    for rank in (0, 1):
        both = np.load(tmp_path / f"rf{rank}.npy")
        assert np.allclose(both, single, rtol=1e-12, atol=0) and both.sum() > 0
