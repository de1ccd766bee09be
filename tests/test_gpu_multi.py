"""Several devices on one segment, product path (include/pmc.h: pmc_history_range, pmc_comm_*, pmc_reduce_frames,
pmc_allreduce_radiation_field; CLI driver ``skirt_mi355x -g ...``).  The GPU box of the test tier has ONE MI355X, so the
collective runs over a one-rank RCCL communicator: what is checked is that the library links RCCL, that the reduce runs
on the context's stream behind the segment, and that a segment run as several history ranges plus the reduce equals the
undivided one.  Counterpart of ProcessManager::sumToRoot behind FluxRecorder::flush (FluxRecorder.cpp:487-493)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, ski

pytestmark = pytest.mark.gpu


def test_reduce_frames_over_a_one_rank_communicator():
    from skirt9_amd.engine import Communicator, Engine, history_range
    from skirt9_amd.host import Simulation
    n = 20000
    sim = Simulation(ski("cfg2small.ski"), num_packets=n).setup()
    single = Engine(sim.scene, 0)
    single.run_primary(0, n, 5)
    expect = single.download()
    single.close()
    comm = Communicator.all([0])
    eng = Engine(sim.scene, 0)
    # the ranges of a 3-way split, one after the other on the one device; the reduce after each (root 0 = this rank)
    for rank in range(3):
        first, count = history_range(n, rank, 3)
        eng.run_primary(first, count, 5)
        eng.reduce_frames(comm.handles[0], 0)
    got = eng.download()
    eng.close()
    comm.close()
    assert got.sum() > 0
    assert abs(got.sum() - expect.sum()) <= 1e-9 * np.abs(expect).sum()
    bad = np.abs(got - expect) > 1e-6 * np.abs(expect) + 1e-12 * np.abs(expect).max()
    assert bad.sum() <= 4


def test_allreduce_radiation_field_over_a_one_rank_communicator():
    from skirt9_amd.engine import Communicator, Engine
    from skirt9_amd.host import Simulation
    n = 5000
    sim = Simulation(ski("cfg3rf.ski"), num_packets=n).setup()
    eng = Engine(sim.scene, 0)
    eng.run_primary(0, n, 3)
    before = eng.download_radiation_field()
    comm = Communicator.all([0])
    eng.allreduce_radiation_field(comm.handles[0])
    after = eng.download_radiation_field()
    eng.close()
    comm.close()
    assert before.sum() > 0 and np.array_equal(before, after)


def test_cli_driver_with_the_collective(tmp_path):
    """skirt_mi355x --rccl -g 0: the driver's multi-device code path (threads, history ranges, reduce, root writes) on one
    device must write the same files as the plain single-device run"""
    exe = os.path.join(ROOT, "skirt9_amd", "lib", "skirt_mi355x")
    a, b = tmp_path / "plain", tmp_path / "rccl"
    a.mkdir(), b.mkdir()
    subprocess.run([exe, "-o", str(a), "-n", "20000", ski("cfg2small.ski")], check=True, stdout=subprocess.DEVNULL)
    out = subprocess.run([exe, "--rccl", "-g", "0", "-o", str(b), "-n", "20000", ski("cfg2small.ski")], check=True, capture_output=True, text=True)
    assert "summed over RCCL" in out.stdout
    files = sorted(f for f in os.listdir(a) if f.endswith((".fits", ".dat")))
    assert files and files == sorted(f for f in os.listdir(b) if f.endswith((".fits", ".dat")))
    from test_gpu_parity import _read_fits
    for f in files:
        if f.endswith(".fits"):
            x, y = _read_fits(str(a / f)), _read_fits(str(b / f))
            # (float32 files; the two runs differ in the summation order of the floating-point atomics only)
            assert x.shape == y.shape and np.allclose(x, y, rtol=1e-5, atol=1e-7 * np.abs(x).max())


def test_statistics_overflow_is_an_error(tmp_path):
    """FluxRecorder::recordContributions (FluxRecorder.cpp:962-1014) keeps any number of contributions per history; the
    engine's list holds 48 distinct pixels per instrument.  A scene that exceeds it (120 scattering events per history)
    must make pmc_run_primary FAIL instead of returning statistics computed from a truncated list; the same scene with
    few events must pass (many contributions to the same pixel share one entry)"""
    from skirt9_amd.engine import Engine
    from skirt9_amd.host import Simulation
    text = open(ski("cfg2small.ski")).read()
    assert 'minScattEvents="0"' in text
    many = tmp_path / "many.ski"
    many.write_text(text.replace('minScattEvents="0"', 'minScattEvents="120"'))
    sim = Simulation(str(many), num_packets=500).setup()
    eng = Engine(sim.scene, 0)
    with pytest.raises(RuntimeError, match="distinct pixels"):
        eng.run_primary(0, 500, 1)
    assert eng.counters()["stat_overflows"] > 0
    eng.close()
    few = tmp_path / "few.ski"
    few.write_text(text.replace('minScattEvents="0"', 'minScattEvents="20"'))
    sim = Simulation(str(few), num_packets=500).setup()
    eng = Engine(sim.scene, 0)
    eng.run_primary(0, 500, 1)
    assert eng.counters()["stat_overflows"] == 0
    eng.close()
