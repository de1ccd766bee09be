"""GPU tests at BASELINE.json's full grid size (configs[1]: the 953 688-cell octree of tests/ski/cfg2.ski; configs[2]:
the same octree, panchromatic).

* photon loop: 4e4 histories on the full-size scene, HIP engine against the oracle on the same Philox streams with the
  tolerances of test_gpu_parity (totals 1e-9, elements 1e-6); 3e6 histories in two segments over three slot groups with
  launch-kernel refills and the live-list drain, against the form without live lists (integer counts exact, sums 1e-11);
  1e5 histories against the files the UNMODIFIED reference wrote for this scene (chi^2 on the 8 x 8 rebinned cube);
* traversal: (m, ds) of 300 fixed rays bit-exact against the oracle on the full octree;
* conservation: every history reaches the SED of the FullInstrument exactly once per emission, so the number of
  contributing histories (sum of w^0 over the wavelength bins) equals N, and the transparent component of the SED sums
  to N times the packet luminosity (to summation order, 1e-12);
* ordering: extinction only removes flux: PrimaryDirect <= Transparent in every pixel;
* linearity / partition independence: two half segments accumulate to the arrays of one whole segment (1e-10);
* the counted work per history is in the range the reference's probe gave (V about 480-580).
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from skirt9_amd.engine import set_tuning
from conftest import ski
from skirt9_amd.host import Simulation

pytestmark = pytest.mark.gpu

N = 400000


@pytest.fixture(scope="module")
def full():
    from skirt9_amd.engine import Engine
    sim = Simulation(ski("cfg2.ski"), num_packets=N).setup()
    return sim, Engine(sim.scene, 0)


def test_full_octree_rays_bit_exact(full):
    sim, eng = full
    scale = 4000 * 3.0857e16
    rng = np.random.default_rng(7)
    total = 0
    for i in range(300):
        r = (rng.random(3) - 0.5) * 2 * scale * np.array([5.0, 5.0, 1.0]) * (1.2 if i % 5 == 0 else 0.95)
        k = rng.normal(size=3)
        if i % 3 == 0:
            k[2] *= 0.02  # nearly in the mid-plane: hundreds of cells of the finest levels
        k /= np.linalg.norm(k)
        m_ref, ds_ref = O.trace_ray(sim, r, k)
        m_gpu, ds_gpu = eng.trace_ray(r, k)
        assert np.array_equal(m_ref, m_gpu), (r, k)
        assert np.array_equal(ds_ref.view(np.uint64), ds_gpu.view(np.uint64)), (r, k)
        total += len(m_ref)
    assert total > 5000


def test_full_octree_rays_equal_the_reference(full):
    """the same octree against the UNMODIFIED reference: 312 fixed rays (tests/golden/cfg2_rays.txt: random, mid-plane,
    axis-parallel, and rays through cell corners and edges) whose (m, ds) sequences the reference dumped
    (tests/golden/make_golden.py cfg2, TreeSpatialGrid.cpp:132-217) -- bit for bit from the HIP traversal"""
    from test_oracle_golden import _ray_fixture
    sim, eng = full
    total = 0
    for r, k, m_ref, ds_ref in _ray_fixture("cfg2"):
        m_gpu, ds_gpu = eng.trace_ray(r, k)
        assert np.array_equal(m_ref, m_gpu), (r, k)
        assert np.array_equal(ds_ref.view(np.uint64), ds_gpu.view(np.uint64)), (r, k)
        total += len(m_ref)
    assert total > 10000


def test_full_size_conservation_and_linearity(full):
    sim, eng = full
    lay = sim.layout(0)
    eng.clear()
    eng.reset_counters()
    eng.run_primary(0, N, 2024)
    whole = eng.download()
    c = eng.counters()
    assert c["histories"] == N
    assert c["stat_overflows"] == 0
    assert 400 <= c["cell_visits"] / N <= 650
    nl, npix = lay.num_lambda, lay.npix
    sed = whole[lay.sed_offset:lay.sed_offset + lay.num_components * nl].reshape(lay.num_components, nl)
    wsed = whole[lay.wsed_offset:lay.wsed_offset + 5 * nl].reshape(5, nl)
    assert wsed[0].sum() == N
    # transparent SED = sum of the launched luminosities = N * packet luminosity * oligo weight (one wavelength)
    scene_lum = sim.packet_luminosity(0)
    assert abs(sed[0].sum() - N * scene_lum) <= 1e-12 * N * scene_lum
    ifu = whole[lay.ifu_offset:lay.ifu_offset + lay.num_components * npix * nl].reshape(lay.num_components, nl * npix)
    assert np.all(ifu[1] <= ifu[0] * (1 + 1e-12))
    assert np.all(sed[1] <= sed[0])
    assert sed[2].sum() > 0  # scattered light reaches the detector
    # two half segments accumulate to the same arrays
    eng.clear()
    eng.run_primary(0, N // 2, 2024)
    eng.run_primary(N // 2, N - N // 2, 2024)
    halves = eng.download()
    assert np.allclose(whole, halves, rtol=1e-10, atol=1e-14 * np.abs(whole).max())


def test_full_size_photon_loop_matches_oracle(full):
    """configs[1] at full size, frame for frame: the HIP engine and the oracle follow the same 4e4 Philox histories through the
    953 688-cell octree (propagation workgroups of 768 lanes on the 10-level coordinate table, peel-off queues) -- the oracle needs
    a few seconds for them"""
    from test_gpu_parity import _compare_frames
    sim, eng = full
    n = 40000
    eng.clear()
    eng.reset_counters()
    eng.run_primary(0, n, 99)
    gpu = eng.download()
    ref, counters = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=99)
    c = eng.counters()
    assert c["histories"] == n and c["stat_overflows"] == 0
    assert abs(c["cell_visits"] - counters.cell_visits) <= 1e-4 * counters.cell_visits
    assert abs(c["scatterings"] - counters.scatterings) <= 1e-4 * counters.scatterings + 2
    _compare_frames(sim, gpu, ref, n)


def test_full_size_three_slot_groups_refills_and_drain(full, monkeypatch):
    """3e6 histories through a pool of 2^20 slots in two segments: three slot groups, launch-kernel refills of ended slots
    (the histories outnumber the slots three to one) and, at the end of each segment, the sparse generations over lists of
    live slots -- against ONE segment of the form without live lists (PMC_NO_LIVE_LISTS).  A history's stream depends only on
    its index, so the integer count cubes (sum of w^0 per pixel) must be equal exactly and every array to summation order."""
    sim, eng = full
    lay = sim.layout(0)
    n = 3000000
    eng.set_num_slots(1 << 20)
    try:
        eng.clear()
        eng.reset_counters()
        eng.run_primary(0, n // 2, 5)
        assert eng.last_timing()["generations"] > 40
        eng.run_primary(n // 2, n - n // 2, 5)
        lists = eng.download()
        c = eng.counters()
        assert c["histories"] == n and c["stat_overflows"] == 0
        set_tuning("PMC_NO_LIVE_LISTS", "1")
        eng.clear()
        eng.run_primary(0, n, 5)
        plain = eng.download()
    finally:
        set_tuning("PMC_NO_LIVE_LISTS", None)
        eng.set_num_slots(8 * 1024 * 1024)
    nl, npix = lay.num_lambda, lay.npix
    w0 = slice(lay.wifu_offset, lay.wifu_offset + nl * npix)
    assert np.array_equal(lists[w0], plain[w0]) and lists[w0].sum() > n
    s0 = slice(lay.wsed_offset, lay.wsed_offset + nl)
    assert np.array_equal(lists[s0], plain[s0]) and lists[s0].sum() == n
    assert abs(lists.sum() - plain.sum()) <= 1e-11 * np.abs(plain).sum()
    assert np.allclose(lists, plain, rtol=1e-9, atol=1e-13 * np.abs(plain).max())


def _rebin(a, f=8):
    ny, nx = a.shape
    return a.reshape(ny // f, f, nx // f, f).sum(axis=(1, 3))


def test_full_size_cube_within_noise_of_the_reference(full, tmp_path):
    """north_star: 'FITS output within 1 sigma of the CPU reference at equal packet count', on the headline scene.  1e5
    histories on the GPU (Philox streams) against what the UNMODIFIED reference wrote for tests/ski/cfg2.ski with its own
    generator (tests/golden/cfg2_full_rebinned.npz: every FITS frame summed over 8 x 8 blocks of the 512^2 pixels, made by
    tests/golden/make_golden.py cfg2; FluxRecorder.cpp:58-62, 962-1014).  The GPU frames go through the host layer's
    calibrateAndWrite, so the comparison is in the files' own units (MJy/sr) and covers the calibration.  A block's relative
    error is sqrt(S2 / S1^2 - 1 / N) from the sums of w and w^2 over its pixels (a history that reaches two pixels of one block
    makes this a slight underestimate).  Stated tolerances: reduced chi^2 over the blocks with >= 30 contributions in both
    runs within [0.8, 1.25] (1 expected), no block beyond 5.5 sigma, the integrated flux within 3 sigma; the integrated
    transparent flux (no extinction: source sampling and detector geometry alone) within 0.2 %."""
    from conftest import golden
    from test_gpu_parity import _read_fits
    sim, eng = full
    n = 100000
    eng.clear()
    eng.run_primary(0, n, 20260930)
    gpu = eng.download()
    # (the fixture's simulation divides the luminosity over N = 400 000 packets, the reference's run over 1e5)
    sim.write(gpu * (N / n), str(tmp_path))
    gold = np.load(golden("cfg2_full_rebinned.npz"))
    lay = sim.layout(0)
    assert lay.num_lambda == 1 and lay.npix == 512 * 512

    def blocks(name):
        return _rebin(_read_fits(os.path.join(str(tmp_path), f"cfg2_i0_{name}.fits")).reshape(512, 512))

    # (sums of w^k per pixel straight from the engine's arrays: the relative error is a ratio, free of the rescaling above)
    g_s = [_rebin(gpu[lay.wifu_offset + k * lay.npix:lay.wifu_offset + (k + 1) * lay.npix].reshape(512, 512)) for k in range(3)]
    r_s = [gold[f"stats{k}"] for k in range(3)]
    assert g_s[0].sum() > n and r_s[0].sum() > n
    good = (g_s[0] >= 30) & (r_s[0] >= 30)
    assert good.sum() > 500
    with np.errstate(divide="ignore", invalid="ignore"):
        rel_g = np.sqrt(np.maximum(g_s[2] / g_s[1] ** 2 - 1.0 / n, 0))
        rel_r = np.sqrt(np.maximum(r_s[2] / r_s[1] ** 2 - 1.0 / n, 0))
    a, b = blocks("total"), gold["total"]
    sigma = np.sqrt((rel_g * a) ** 2 + (rel_r * b) ** 2)
    z = (a - b)[good] / sigma[good]
    chi2 = float(np.mean(z ** 2))
    assert 0.8 <= chi2 <= 1.25, chi2
    assert np.abs(z).max() < 5.5, float(np.abs(z).max())
    assert abs(a[good].sum() - b[good].sum()) <= 3 * np.sqrt(np.sum(sigma[good] ** 2))
    # the transparent frame: every history adds its full luminosity at emission unless the pixel lies off the detector
    a, b = blocks("transparent"), gold["transparent"]
    assert abs(a.sum() - b.sum()) <= 2e-3 * b.sum()


def test_config3_full_size_photon_loop_matches_oracle():
    """BASELINE configs[2] at full size (953 688 cells, 50 wavelength bins): 3e4 histories, HIP engine against the oracle on
    the same Philox streams"""
    from skirt9_amd.engine import Engine
    from test_gpu_parity import _compare_frames
    n = 30000
    sim = Simulation(ski("cfg3.ski"), num_packets=n).setup()
    eng = Engine(sim.scene, 0)
    eng.run_primary(0, n, 41)
    gpu = eng.download()
    ref, counters = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=41)
    c = eng.counters()
    assert c["histories"] == n and c["stat_overflows"] == 0
    assert abs(c["cell_visits"] - counters.cell_visits) <= 1e-4 * counters.cell_visits
    _compare_frames(sim, gpu, ref, n)


def test_config3_full_size_panchromatic():
    """BASELINE configs[2] at full size (tests/ski/cfg3.ski: the 953 688-cell octree, 50 wavelength bins, 512^2 pixels,
    839 MB of detector arrays): every history lands in exactly one wavelength bin of the SED statistics, extinction only
    removes flux in every bin, every bin receives packets, and a segment split in two accumulates to the same arrays"""
    from skirt9_amd.engine import Engine
    n = 200000
    sim = Simulation(ski("cfg3.ski"), num_packets=n).setup()
    eng = Engine(sim.scene, 0)
    lay = sim.layout(0)
    assert lay.num_lambda == 50 and lay.npix == 512 * 512
    eng.run_primary(0, n, 77)
    whole = eng.download()
    c = eng.counters()
    assert c["histories"] == n and c["stat_overflows"] == 0
    nl = lay.num_lambda
    sed = whole[lay.sed_offset:lay.sed_offset + lay.num_components * nl].reshape(lay.num_components, nl)
    wsed = whole[lay.wsed_offset:lay.wsed_offset + 5 * nl].reshape(5, nl)
    assert wsed[0].sum() == n
    assert np.all(wsed[0] > 0)                       # the wavelength bias spreads packets over all 50 bins
    assert np.all(sed[1] <= sed[0]) and np.all(sed[0] > 0)
    # the dust is more opaque in the blue: the attenuated fraction falls with wavelength
    frac = sed[1] / sed[0]
    assert frac[:10].mean() < frac[-10:].mean()
    eng.clear()
    eng.run_primary(0, n // 2, 77)
    eng.run_primary(n // 2, n - n // 2, 77)
    halves = eng.download()
    assert np.allclose(whole, halves, rtol=1e-10, atol=1e-14 * np.abs(whole).max())


def test_full_size_radiation_field_log_partition(tmp_path, monkeypatch):
    """configs[1] with storeRadiationField: 953 688 table entries = 117 partitions of the radiation-field log.  The table from the
    logged path (rfHistKernel / rfScanKernel / rfScatterKernel / rfReduceKernel behind every generation) against the oracle on the
    same 3e4 Philox histories (totals 1e-9, elements 1e-6), and 1.5e6 histories (three slot groups, chunks of many waves, long and
    short runs per partition) against the form that adds every contribution atomically (PMC_RF_ATOMICS=1): totals 1e-11, elements
    1e-9 -- the two differ in summation order only."""
    from skirt9_amd.engine import Engine
    text = open(ski("cfg2.ski")).read()
    assert 'storeRadiationField="false"' in text
    path = tmp_path / "cfg2rf.ski"
    path.write_text(text.replace('storeRadiationField="false"', 'storeRadiationField="true"'))
    n = 30000
    sim = Simulation(str(path), num_packets=n).setup()
    assert sim.radiation_field_size == 953688
    eng = Engine(sim.scene, 0)
    eng.run_primary(0, n, 21)
    got = eng.download_radiation_field()
    _, ref, _ = O.run_primary_rf(sim, 0, n, O.RNG_PHILOX, seed=21)
    assert abs(got.sum() - ref.sum()) <= 1e-9 * ref.sum()
    assert np.array_equal(got > 0, ref > 0)
    assert (np.abs(got - ref) > 1e-6 * np.abs(ref) + 1e-13 * ref.max()).sum() == 0
    big = 1500000
    eng.clear_radiation_field()
    eng.run_primary(0, big, 22)
    logged = eng.download_radiation_field()
    eng.close()
    set_tuning("PMC_RF_ATOMICS", "1")
    eng = Engine(sim.scene, 0)
    eng.run_primary(0, big, 22)
    atomics = eng.download_radiation_field()
    eng.close()
    assert abs(logged.sum() - atomics.sum()) <= 1e-11 * atomics.sum()
    assert np.array_equal(logged > 0, atomics > 0)
    assert (np.abs(logged - atomics) > 1e-9 * np.abs(atomics) + 1e-15 * atomics.max()).sum() == 0


def test_full_size_sorted_peel_records_change_nothing(full, monkeypatch):
    """The peel-off walks of a generation run from records sorted by detector tile (pmc_device.h PeelRec; up to four observers, octree); PMC_NO_PEEL_SORT=1
    runs them from the task arrays in slot order as in rounds 1-3.  Same histories, same walks: the counted work is identical, the integer
    counts are identical, the sums agree to summation order."""
    from skirt9_amd.engine import Engine
    sim, eng = full
    n = 1000000
    eng.clear()
    eng.reset_counters()
    eng.run_primary(0, n, 31)
    a, ca = eng.download(), eng.counters()
    set_tuning("PMC_NO_PEEL_SORT", "1")
    plain = Engine(sim.scene, 0)
    plain.run_primary(0, n, 31)
    b, cb = plain.download(), plain.counters()
    plain.close()
    assert ca["cell_visits"] == cb["cell_visits"] and ca["scatterings"] == cb["scatterings"] and ca["histories"] == cb["histories"] == n
    lay = sim.layout(0)
    w0 = slice(lay.wsed_offset, lay.wsed_offset + lay.num_lambda)
    assert np.array_equal(a[w0], b[w0])
    assert abs(a.sum() - b.sum()) <= 1e-11 * np.abs(b).sum()
    assert (np.abs(a - b) > 1e-9 * np.abs(b) + 1e-15 * np.abs(b).max()).sum() == 0


def test_full_size_sorted_peel_records_three_observers(tmp_path, monkeypatch):
    """The same with three observers (three FullInstruments at 0 / 60 / 90 degrees: three sets of sorted records, three peel-off kernels per
    generation) against the slot-order form, and both against the oracle on 2e4 histories."""
    import re
    from skirt9_amd.engine import Engine
    from test_gpu_parity import _compare_frames
    text = open(ski("cfg2.ski")).read()
    ins = re.search(r"<FullInstrument [^>]*/>", text).group(0)
    a = ins.replace('instrumentName="i0"', 'instrumentName="i1"').replace('inclination="60 deg"', 'inclination="0 deg"')
    b = ins.replace('instrumentName="i0"', 'instrumentName="i2"').replace('inclination="60 deg"', 'inclination="90 deg"').replace('azimuth="30 deg"', 'azimuth="0 deg"')
    assert a != ins and b != ins
    path = tmp_path / "cfg2three.ski"
    path.write_text(text.replace(ins, ins + a + b))
    n = 300000
    sim = Simulation(str(path), num_packets=n).setup()
    eng = Engine(sim.scene, 0)
    eng.run_primary(0, n, 5)
    a_frames, ca = eng.download(), eng.counters()
    eng.clear()
    eng.run_primary(0, 20000, 6)
    small = eng.download()
    eng.close()
    ref, _ = O.run_primary(sim, 0, 20000, O.RNG_PHILOX, seed=6)
    _compare_frames(sim, small, ref, 20000)
    set_tuning("PMC_NO_PEEL_SORT", "1")
    plain = Engine(sim.scene, 0)
    plain.run_primary(0, n, 5)
    b_frames, cb = plain.download(), plain.counters()
    plain.close()
    assert ca["cell_visits"] == cb["cell_visits"] and ca["scatterings"] == cb["scatterings"]
    assert abs(a_frames.sum() - b_frames.sum()) <= 1e-11 * np.abs(b_frames).sum()
    assert (np.abs(a_frames - b_frames) > 1e-9 * np.abs(b_frames) + 1e-15 * np.abs(b_frames).max()).sum() == 0
