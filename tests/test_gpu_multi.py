"""Several devices on one segment, product path (include/pmc.h: pmc_history_range, pmc_comm_*, pmc_reduce_frames,
pmc_allreduce_radiation_field; CLI driver ``skirt_mi355x -g ...``).  The GPU box of the test tier has ONE MI355X, so the
collective runs over a one-rank RCCL communicator: what is checked is that the library links RCCL, that the reduce runs
on the context's stream behind the segment, and that a segment run as several history ranges plus the reduce equals the
undivided one.  Counterpart of ProcessManager::sumToRoot behind FluxRecorder::flush (FluxRecorder.cpp:487-493)."""
import os
import subprocess

import numpy as np
import pytest

from skirt9_amd.engine import set_tuning
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


def _many_events(tmp_path, events):
    text = open(ski("cfg2small.ski")).read()
    assert 'minScattEvents="0"' in text
    path = tmp_path / f"events{events}.ski"
    path.write_text(text.replace('minScattEvents="0"', f'minScattEvents="{events}"'))
    return str(path)


@pytest.mark.parametrize("events,n,pool", [(120, 2000, None), (400, 600, None), (120, 2000, "16")])
def test_statistics_lists_are_unbounded(tmp_path, monkeypatch, events, n, pool):
    """FluxRecorder::recordContributions (FluxRecorder.cpp:962-1014, FluxRecorder.hpp:327-338) keeps any number of
    contributions per history.  A slot's own list holds 48 distinct pixels per instrument; a history with more continues
    in chained blocks from the slot group's pool.  120 / 400 scattering events per history (every history leaves its own list):
    the statistics arrays must agree with the oracle's, which keeps a std::vector per history.  With a pool of 16 blocks
    (PMC_STAT_POOL_BLOCKS) the same segment must complete as well: the pool grows between the generations of a slot group
    (round 6; rounds 1-5 failed such a segment with PMC_ERR_OVERFLOW)."""
    import oracle_lib as O
    from skirt9_amd.engine import Engine
    from skirt9_amd.host import Simulation
    from test_gpu_parity import _compare_frames
    if pool:
        monkeypatch.setenv("PMC_STAT_POOL_BLOCKS", pool)
    sim = Simulation(_many_events(tmp_path, events), num_packets=n).setup()
    eng = Engine(sim.scene, 0)
    eng.run_primary(0, n, 1)
    gpu = eng.download()
    c = eng.counters()
    eng.close()
    # (a packet's weight underflows to zero after some 120 forced scatterings in this scene, which ends the history whatever minScattEvents asks)
    assert c["stat_overflows"] == 0 and c["scatterings"] >= 100 * n
    ref, counters = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=1)
    _compare_frames(sim, gpu, ref, n)
    lay = sim.layout(0)
    npix = lay.npix * lay.num_lambda
    for k in range(5):
        a = gpu[lay.wifu_offset + k * npix:lay.wifu_offset + (k + 1) * npix]
        b = ref[lay.wifu_offset + k * npix:lay.wifu_offset + (k + 1) * npix]
        assert abs(a.sum() - b.sum()) <= 1e-9 * np.abs(b).sum()
    # sum of w^0 over the pixels = number of (history, distinct pixel) pairs: beyond 48 per history on average
    assert ref[lay.wifu_offset:lay.wifu_offset + npix].sum() > 50 * n   # (beyond the 48 entries of a slot's own list)
    assert gpu[lay.wifu_offset:lay.wifu_offset + npix].sum() == ref[lay.wifu_offset:lay.wifu_offset + npix].sum()


def test_statistics_pool_grows_with_several_slot_groups(tmp_path, monkeypatch):
    """the pool grows while the OTHER slot groups have kernels in flight (the growth waits for the device, moves the pool arrays and hands the new
    blocks to the group that asked): 200 000 packets in three groups with 150 scattering events each, once with a pool of 64 blocks -- which has to
    grow several times -- and once with the default pool: the same histories, the same statistics (to the order of the atomic additions)"""
    from skirt9_amd.engine import Engine
    from skirt9_amd.host import Simulation
    n = 200000
    sim = Simulation(_many_events(tmp_path, 150), num_packets=n).setup()

    def run():
        eng = Engine(sim.scene, 0)
        eng.run_primary(0, n, 3)
        frames, c, t = eng.download(), eng.counters(), eng.last_timing()
        eng.close()
        return frames, c, t

    base, cb, tb = run()
    monkeypatch.setenv("PMC_STAT_POOL_BLOCKS", "64")
    grown, cg, tg = run()
    assert cb["stat_overflows"] == 0 and cg["stat_overflows"] == 0
    assert (cb["histories"], cb["scatterings"], cb["detector_updates"]) == (cg["histories"], cg["scatterings"], cg["detector_updates"])
    assert cb["scatterings"] >= 100 * n and tb["generations"] > 100
    lay = sim.layout(0)
    npix = lay.npix * lay.num_lambda
    # (the count of (history, distinct pixel) pairs is an integer sum: exact; more than 48 per history: the pool is in use)
    assert grown[lay.wifu_offset:lay.wifu_offset + npix].sum() == base[lay.wifu_offset:lay.wifu_offset + npix].sum() > 50 * n
    assert np.allclose(grown, base, rtol=1e-9, atol=1e-12 * np.abs(base).max())


def test_statistics_pool_exhaustion_is_an_error(tmp_path, monkeypatch):
    """the pool of list blocks is finite (device memory): a segment that runs out of blocks must FAIL instead of returning
    statistics computed from truncated lists; the next segment starts with a full pool again"""
    from skirt9_amd.engine import Engine, set_tuning, clear_tuning
    from skirt9_amd.host import Simulation
    monkeypatch.setenv("PMC_STAT_POOL_BLOCKS", "16")
    set_tuning("PMC_STAT_POOL_NO_GROWTH", "1")   # (the pool as rounds 1-5 had it: what happens when the device has no memory left for more blocks)
    try:
        sim = Simulation(_many_events(tmp_path, 120), num_packets=500).setup()
        eng = Engine(sim.scene, 0)
        with pytest.raises(RuntimeError, match="PMC_STAT_POOL_BLOCKS"):
            eng.run_primary(0, 500, 1)
        assert eng.counters()["stat_overflows"] > 0
        eng.close()
    finally:
        clear_tuning()
    sim = Simulation(_many_events(tmp_path, 20), num_packets=500).setup()
    eng = Engine(sim.scene, 0)
    eng.run_primary(0, 500, 1)
    assert eng.counters()["stat_overflows"] == 0
    eng.close()


def test_sparse_generations_equal_the_slot_order_generations(monkeypatch):
    """The end of a segment (nothing left to launch) runs over lists of live slots, and the transition kernel retires the
    histories that end itself (pmc_api.hip `sparseLists`; PMC_NO_LIVE_LISTS=1 keeps every generation in slot order with the
    scan and launch kernels).  Both forms run the same histories with the same random streams: every counter is equal, the
    number of contributing histories per pixel is equal, and the sums agree to the order of the atomic additions.  Two
    segments on one context: the lists of the first must not leak into the second."""
    from skirt9_amd.engine import Engine
    from skirt9_amd.host import Simulation
    n = 300000
    sim = Simulation(ski("cfg2small.ski"), num_packets=n).setup()
    lay = sim.layout(0)
    out = []
    for lists in (True, False):
        if lists:
            set_tuning("PMC_NO_LIVE_LISTS", None)
        else:
            set_tuning("PMC_NO_LIVE_LISTS", "1")
        eng = Engine(sim.scene, 0)
        eng.run_primary(0, n // 3, 7)
        eng.run_primary(n // 3, n - n // 3, 7)
        out.append((eng.download(), eng.counters()))
        eng.close()
    (a, ca), (b, cb) = out
    for key in ("histories", "paths", "scatterings", "cell_visits", "rewalk_visits", "detector_updates", "stat_overflows"):
        assert ca[key] == cb[key], key
    assert ca["histories"] == n
    npix = lay.npix * lay.num_lambda
    assert np.array_equal(a[lay.wifu_offset:lay.wifu_offset + npix], b[lay.wifu_offset:lay.wifu_offset + npix])
    assert np.allclose(a, b, rtol=1e-11, atol=0)


def test_one_wavelength_constants_equal_the_per_slot_arrays(monkeypatch):
    """A scene whose sources all emit at one wavelength keeps that wavelength, its bin in every instrument and the dust mix's
    properties at it as constants of the scene (pmc_api.hip `D.mono`, found as launchHistory finds them: DustMix::indexForLambda,
    DisjointWavelengthGrid::bin); PMC_NO_MONO=1 stores and loads them per slot, as panchromatic scenes do.  Same histories, same
    random streams: equal counters, equal numbers of contributing histories per pixel, sums equal to the order of the atomic
    additions."""
    from skirt9_amd.engine import Engine
    from skirt9_amd.host import Simulation
    n = 200000
    for name in ("cfg1.ski", "cfg2small.ski"):
        sim = Simulation(ski(name), num_packets=n).setup()
        lay = sim.layout(0)
        out = []
        for mono in (True, False):
            if mono:
                set_tuning("PMC_NO_MONO", None)
            else:
                set_tuning("PMC_NO_MONO", "1")
            eng = Engine(sim.scene, 0)
            eng.run_primary(0, n, 3)
            out.append((eng.download(), eng.counters()))
            eng.close()
        (a, ca), (b, cb) = out
        assert ca == cb and ca["histories"] == n, name
        npix = lay.npix * lay.num_lambda
        assert np.array_equal(a[lay.wifu_offset:lay.wifu_offset + npix], b[lay.wifu_offset:lay.wifu_offset + npix]), name
        assert np.allclose(a, b, rtol=1e-11, atol=0), name


@pytest.mark.gpu
def test_progress_reports():
    """MonteCarloSimulation::logProgress (MonteCarloSimulation.cpp:522-526, 609): pmc_set_progress reports the histories handed out so far
    from the thread that runs pmc_run_primary, never more than the segment holds, never backwards"""
    from skirt9_amd.engine import Engine
    from skirt9_amd.host import Simulation
    n = 300000
    sim = Simulation(ski("cfg2small.ski"), num_packets=n).setup()
    eng = Engine(sim.scene, 0)
    seen = []
    eng.set_progress(lambda launched, count: seen.append((launched, count)), interval_seconds=0.0)
    eng.run_primary(0, n, 3)
    eng.sync()
    assert len(seen) >= 2
    assert all(c == n for _, c in seen) and all(0 <= a <= n for a, _ in seen)
    assert all(b[0] >= a[0] for a, b in zip(seen, seen[1:])) and seen[-1][0] == n
    eng.set_progress(None)
    count = len(seen)
    eng.run_primary(n, 1000, 3)
    eng.sync()
    assert len(seen) == count and eng.counters()["histories"] == n + 1000
    eng.close()
