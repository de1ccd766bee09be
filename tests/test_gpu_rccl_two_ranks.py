"""Two MI355X, two processes, one segment: the product's multi-GPU path with a REAL exchange (skipped on a box with one GPU).

Each process owns one device: it sets the scene up, creates its engine, takes its history range (pmc_history_range), joins the
communicator (rank 0 draws the id with pmc_comm_unique_id and hands it over a pipe; pmc_comm_init_rank), runs its range on the
HIP engine and calls pmc_reduce_frames -- ONE ncclReduce (f64, sum) onto rank 0 over xGMI, the counterpart of
ProcessManager::sumToRoot behind FluxRecorder::flush (SKIRT/mpi/ProcessManager.cpp:223-255, SKIRT/core/FluxRecorder.cpp:487-493;
the chunk server it replaces: SKIRT/core/MultiHybridParallel.cpp:26-104).  Rank 0 must then hold the frames of the undivided
single-device segment: totals to 1e-9, elements to 1e-6, and the integer counts wsed[0] (histories per wavelength bin) exactly;
rank 1's frames must be cleared.  The same through the CLI driver: `skirt_mi355x -g 0,1` writes the files of `-g 0`."""
import multiprocessing as mp
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, ski

pytestmark = pytest.mark.gpu

N = 60000
SEED = 11


def _device_count():
    import ctypes as C
    try:
        hip = C.CDLL("libamdhip64.so")
    except OSError:
        return 0
    n = C.c_int(0)
    return n.value if hip.hipGetDeviceCount(C.byref(n)) == 0 else 0


def _rank(rank, world, pipe, out):
    """one process = one device; returns its frames after the reduce (and the communicator's own view of its size)"""
    try:
        from skirt9_amd.engine import Communicator, Engine, history_range
        from skirt9_amd.host import Simulation
        sim = Simulation(ski("cfg2small.ski"), num_packets=N).setup()
        eng = Engine(sim.scene, rank)
        if rank == 0:
            uid = Communicator.unique_id()
            for p in pipe:
                p.send(uid)
        else:
            uid = pipe.recv()
        comm = Communicator.rank(rank, world, rank, uid)
        first, count = history_range(N, rank, world)
        eng.run_primary(first, count, SEED)
        before = eng.download()
        eng.reduce_frames(comm.handles[0], 0)
        after = eng.download()
        size = comm.size()
        comm.close()
        eng.close()
        out.put((rank, before, after, size, None))
    except Exception as exc:  # noqa: BLE001 - reported to the parent
        out.put((rank, None, None, None, repr(exc)))


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs (the test tier's box has one)")
def test_two_ranks_reduce_to_the_single_device_frames():
    from skirt9_amd.engine import Engine
    from skirt9_amd.host import Simulation
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    a, b = ctx.Pipe()
    procs = [ctx.Process(target=_rank, args=(0, 2, [a], out)), ctx.Process(target=_rank, args=(1, 2, b, out))]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        rank, before, after, size, err = out.get(timeout=600)
        assert err is None, f"rank {rank}: {err}"
        got[rank] = (before, after, size)
    for p in procs:
        p.join(timeout=60)
    assert got[0][2] == (2, 0) and got[1][2] == (2, 1)          # what ncclCommCount / ncclCommUserRank say
    sim = Simulation(ski("cfg2small.ski"), num_packets=N).setup()
    eng = Engine(sim.scene, 0)
    eng.run_primary(0, N, SEED)
    expect = eng.download()
    eng.close()
    lay = sim.layout(0)
    summed = got[0][1]
    assert np.all(got[1][1] == 0)                               # the non-root arrays are cleared, as the reference's are
    assert np.array_equal(got[0][0] + got[1][0], summed)        # the reduce IS the sum of what the ranks held (one addition per element)
    w0 = slice(lay.wsed_offset, lay.wsed_offset + lay.num_lambda)
    assert np.array_equal(summed[w0], expect[w0]) and summed[w0].sum() == N
    assert abs(summed.sum() - expect.sum()) <= 1e-9 * np.abs(expect).sum()
    bad = np.abs(summed - expect) > 1e-6 * np.abs(expect) + 1e-12 * np.abs(expect).max()
    assert bad.sum() <= 4


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs (the test tier's box has one)")
def test_cli_driver_on_two_devices(tmp_path):
    exe = os.path.join(ROOT, "skirt9_amd", "lib", "skirt_mi355x")
    a, b = tmp_path / "one", tmp_path / "two"
    a.mkdir(), b.mkdir()
    subprocess.run([exe, "-g", "0", "-o", str(a), "-n", str(N), ski("cfg2small.ski")], check=True, stdout=subprocess.DEVNULL)
    out = subprocess.run([exe, "-g", "0,1", "-o", str(b), "-n", str(N), ski("cfg2small.ski")], check=True, capture_output=True, text=True)
    assert "summed over RCCL" in out.stdout
    files = sorted(f for f in os.listdir(a) if f.endswith((".fits", ".dat")))
    assert files and files == sorted(f for f in os.listdir(b) if f.endswith((".fits", ".dat")))
    from test_gpu_parity import _read_fits
    for f in files:
        if f.endswith(".fits"):
            x, y = _read_fits(str(a / f)), _read_fits(str(b / f))
            assert x.shape == y.shape and np.allclose(x, y, rtol=1e-5, atol=1e-7 * np.abs(x).max())
