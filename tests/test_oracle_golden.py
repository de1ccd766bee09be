"""Pins the CPU oracle (oracle/life_cycle.cpp) and the host model layer against the UNMODIFIED reference:
with the reference's own mt19937_64 stream (continued after the setup draws) the oracle must reproduce the golden
output files that oracle/_ref wrote (tests/golden/make_golden.py) BYTE FOR BYTE -- FITS cubes (float32 pixels,
header cards, ASCII wavelength table), SED and statistics tables -- apart from the FITS DATE card."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import golden, ski
from skirt9_amd.host import Simulation


def _same_file(path_a, path_b):
    a = bytearray(open(path_a, "rb").read())
    b = bytearray(open(path_b, "rb").read())
    if path_a.endswith(".fits"):
        assert a[880:888] == b"DATE    " and b[880:888] == b"DATE    "
        a[880:960] = b" " * 80
        b[880:960] = b" " * 80
    return a == b


@pytest.mark.parametrize("name,nfiles", [("cfg1", 11), ("cfg1mesh", 11), ("cfg1mesh2", 11), ("cfg2small", 11), ("cfg2deep", 11), ("cfg2deeper", 11), ("cfg3small", 27), ("cfg3z", 35), ("cfg1nf", 12), ("cfg2nf", 11), ("cfg4small", 11), ("cfg1file", 11), ("cfg1sed", 14), ("cfg3sed", 27), ("cfg2ea", 11), ("cfg1nfea", 12), ("cfg2mm", 11), ("cfg2mmea", 11), ("cfg1mmnf", 12), ("cfg3mm", 27), ("cfg1con", 11), ("cfg1netzer", 11), ("cfg1laser", 11), ("cfg2agn", 11), ("cfg1nomed", 8)])
def test_byte_identical_to_reference(name, nfiles, tmp_path):
    """cfg3small: panchromatic sampling with a wavelength bias, tabulated dust, 20 wavelength bins, and three
    instruments (scattering levels, a FrameInstrument sharing its observer, a second observer); cfg1nf: non-forced
    scattering, box source that sticks out of the grid, isotropic and strongly forward scattering dust; cfg3z: model
    redshift 0.5 (FlatUniverseCosmology), six instruments, three of them in the observer frame; cfg1nomed: a simulation without a
    medium system (total flux only)"""
    sim = Simulation(ski(name + ".ski")).setup()
    frames, counters = O.run_primary(sim, 0, sim.num_packets, O.RNG_MT19937)
    assert counters.histories == sim.num_packets
    sim.write(frames, str(tmp_path))
    expected = sorted(f for f in os.listdir(golden("")) if (f.startswith(name + "_i") or f.startswith(name + "_s"))
                      and (f.endswith(".fits") or f.endswith(".dat")))
    assert len(expected) == nfiles
    for f in expected:
        assert os.path.exists(tmp_path / f), f
        assert _same_file(golden(f), str(tmp_path / f)), f"{f} differs from the reference output"


@pytest.mark.parametrize("name", ["cfg1", "cfg1mesh", "cfg1mesh2", "cfg2small", "cfg2deep", "cfg2deeper", "cfg4small", "cfg4deepest", "cfg5small", "cfg5peak", "cfg5imp"])
def test_ray_segments_bit_exact(name):
    """PathSegmentGenerator (m, ds) sequences dumped from the reference (skirt_ref rays) vs the oracle's generators"""
    sim = Simulation(ski(name + ".ski")).setup()
    rays = [[float.fromhex(t) for t in line.split()] for line in open(golden(name + "_rays.txt"))]
    ref = open(golden(name + "_rays_ref.txt")).read().split("\n")
    pos = 0
    total = 0
    for i, ray in enumerate(rays):
        head = ref[pos].split()
        assert head[0] == "ray" and int(head[1]) == i
        n = int(head[2])
        k = np.array([float.fromhex(v) for v in head[3:6]])  # the direction as normalised by the reference
        m_ref = np.array([int(ref[pos + 1 + j].split()[0]) for j in range(n)], dtype=np.int32)
        ds_ref = np.array([float.fromhex(ref[pos + 1 + j].split()[1]) for j in range(n)])
        pos += 1 + n
        m, ds = O.trace_ray(sim, ray[:3], k)
        assert len(m) == n, (i, len(m), n)
        assert np.array_equal(m, m_ref), i
        assert np.array_equal(ds.view(np.uint64), ds_ref.view(np.uint64)), i
        total += n
    assert total > 300


def test_known_answer_ray_from_survey():
    """config-1 grid, r = (1e15, 2e15, -3e15) m, k = normalised (0.3, 0.5, 0.81): 35 segments"""
    sim = Simulation(ski("cfg1.ski")).setup()
    k = np.array([0.3, 0.5, 0.81])
    k = k / np.sqrt((k * k).sum())
    m, ds = O.trace_ray(sim, [1e15, 2e15, -3e15], k)
    assert len(m) == 35
    assert list(m[:4]) == [16942, 16943, 17967, 17968]
    assert ds[0] == float.fromhex("0x1.2c2d9ecefa492p+50")
    assert ds[3] == float.fromhex("0x1.30b5de21246ebp+43")


def test_philox_partition_independence():
    """per-history streams: any split of the history range gives the same detector arrays"""
    sim = Simulation(ski("cfg2small.ski"), num_packets=3000).setup()
    whole, _ = O.run_primary(sim, 0, 3000, O.RNG_PHILOX, seed=5)
    parts = np.zeros_like(whole)
    O.run_primary(sim, 0, 1000, O.RNG_PHILOX, seed=5, frames=parts)
    O.run_primary(sim, 1000, 2000, O.RNG_PHILOX, seed=5, frames=parts)
    assert np.allclose(whole, parts, rtol=1e-12, atol=0)


def test_philox_agrees_with_reference_stream_statistically():
    """the engine's RNG differs from mt19937_64, so SEDs agree within Monte Carlo noise: R from the reference's
    own sum-of-w^k statistics (R = sqrt(S2/S1^2 - 1/N)); tolerance 5 sigma"""
    n = 20000
    sim = Simulation(ski("cfg2small.ski"), num_packets=n).setup()
    lay = sim.layout(0)
    a, _ = O.run_primary(sim, 0, n, O.RNG_MT19937)
    b, _ = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=99)
    for frames in (a, b):
        assert frames[lay.wsed_offset] == n
    s1, s2 = a[lay.wsed_offset + 1], a[lay.wsed_offset + 2]
    R = np.sqrt(s2 / s1 ** 2 - 1.0 / n)
    ta = a[lay.sed_offset + 1] + a[lay.sed_offset + 2]
    tb = b[lay.sed_offset + 1] + b[lay.sed_offset + 2]
    assert abs(ta - tb) <= 5 * np.sqrt(2) * R * ta
    # the transparent component depends only on the source sampling: relative noise ~ 0 (every packet contributes
    # the same weight), so it must agree to rounding
    assert abs(a[lay.sed_offset] - b[lay.sed_offset]) <= 1e-9 * a[lay.sed_offset]


@pytest.mark.parametrize("name", ["cfg1rf", "cfg3rf", "cfg1rfea", "cfg1mmrf"])
def test_radiation_field_probe_byte_identical(name, tmp_path):
    """storeRadiationField (MonteCarloSimulation.cpp:638-662): with the reference's random stream the oracle's radiation
    field table, written by the host layer as the RadiationFieldProbe / PerCellForm file, equals the reference's file
    byte for byte (cfg1rf: oligochromatic, Cartesian grid; cfg3rf: panchromatic, octree, a 6-bin radiation field grid
    narrower than the source range so that some packets fall outside it); the SED files are unchanged by the option"""
    import gzip
    sim = Simulation(ski(name + ".ski")).setup()
    assert sim.radiation_field_size > 0
    frames, rf, counters = O.run_primary_rf(sim, 0, sim.num_packets, O.RNG_MT19937)
    assert counters.histories == sim.num_packets and rf.min() >= 0 and rf.max() > 0
    sim.write(frames, str(tmp_path))
    sim.write_radiation_field(rf, str(tmp_path))
    produced = open(tmp_path / f"{name}_rf_J.dat", "rb").read()
    assert produced == gzip.open(golden(f"{name}_rf_J.dat.gz"), "rb").read()
    for f in os.listdir(golden("")):
        if f.startswith(name + "_i") and f.endswith("_sed.dat"):
            assert _same_file(golden(f), str(tmp_path / f)), f


def test_voronoi_outputs_follow_the_reference(tmp_path):
    """cfg5small (VoronoiMeshSpatialGrid, 1500 random sites): the host layer builds its own tessellation (the
    reference's comes from Voro++), so bounding boxes, and with them the random positions at which the cell densities
    are sampled, agree with the reference's to rounding only (1e-14); with the reference's random stream the oracle then
    follows the same histories and reproduces the reference's SED files to 1e-9 and its frames to float32 rounding"""
    sim = Simulation(ski("cfg5small.ski")).setup()
    gold = np.load(golden("cfg5small_cells.npz"))
    dens = _densities(sim)
    assert np.allclose(dens, gold["density"], rtol=1e-12, atol=0)
    frames, counters = O.run_primary(sim, 0, sim.num_packets, O.RNG_MT19937)
    sim.write(frames, str(tmp_path))
    checked = 0
    for f in sorted(os.listdir(golden(""))):
        if not f.startswith("cfg5small_i"):
            continue
        assert os.path.exists(tmp_path / f), f
        if f.endswith("_sed.dat"):
            a = np.loadtxt(golden(f))
            b = np.loadtxt(str(tmp_path / f))
            assert a.shape == b.shape and np.allclose(a, b, rtol=1e-8, atol=0), f
            checked += 1
        elif f.endswith("_total.fits"):
            a = np.frombuffer(open(golden(f), "rb").read()[2880:], dtype=">f4")
            b = np.frombuffer(open(tmp_path / f, "rb").read()[2880:], dtype=">f4")
            n = min(a.size, b.size)
            assert a.size == b.size and np.allclose(a[:n], b[:n], rtol=1e-5, atol=1e-6 * np.abs(a).max()), f
            checked += 1
    assert checked >= 4


def _densities(sim):
    import ctypes as C
    from test_host_model import scene_head
    head = scene_head(sim)
    return np.ctypeslib.as_array(head.medium.number_density, shape=(head.grid.num_cells,)).copy()


def _ray_fixture(name):
    rays = [[float.fromhex(t) for t in line.split()] for line in open(golden(name + "_rays.txt"))]
    if os.path.exists(golden(name + "_rays_ref.txt.gz")):
        import gzip
        ref = gzip.open(golden(name + "_rays_ref.txt.gz"), "rt").read().split("\n")
    else:
        ref = open(golden(name + "_rays_ref.txt")).read().split("\n")
    out, pos = [], 0
    for i, ray in enumerate(rays):
        head = ref[pos].split()
        assert head[0] == "ray" and int(head[1]) == i
        n = int(head[2])
        k = np.array([float.fromhex(v) for v in head[3:6]])
        m_ref = np.array([int(ref[pos + 1 + j].split()[0]) for j in range(n)], dtype=np.int32)
        ds_ref = np.array([float.fromhex(ref[pos + 1 + j].split()[1]) for j in range(n)])
        pos += 1 + n
        out.append((np.array(ray[:3]), k, m_ref, ds_ref))
    return out


def config5_simulation(tmp_dir, num_packets=1000):
    """tests/ski/cfg5.ski with its 10^5 Voronoi sites regenerated by tools/make_sites.py (deterministic)"""
    import subprocess
    import sys
    from conftest import ROOT
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_sites.py"), "--n", "100000", "--seed", "1",
                           os.path.join(str(tmp_dir), "cfg5_sites.txt")])
    old = os.environ.get("SKH_INPUT_PATH")
    os.environ["SKH_INPUT_PATH"] = str(tmp_dir)
    try:
        return Simulation(ski("cfg5.ski"), num_packets=num_packets).setup()
    finally:
        if old is None:
            del os.environ["SKH_INPUT_PATH"]
        else:
            os.environ["SKH_INPUT_PATH"] = old


def test_full_size_config2_equals_the_reference(tmp_path):
    """BASELINE configs[1] at FULL size (tests/ski/cfg2.ski, the octree the benchmark runs on): the host layer builds the
    reference's tree (953 688 cells; DensityTreePolicy.cpp:117-231, TreeSpatialGrid.cpp:34-130) with the reference's
    volumes and sampled densities -- SHA-256 of the bit patterns in cell order -- and the oracle's path generator
    reproduces the reference's (m, ds) sequences of 312 rays bit for bit, including rays through cell corners and edges
    (TreeSpatialGrid.cpp:132-217)"""
    import hashlib
    import json
    from test_host_model import scene_head
    sim = Simulation(ski("cfg2.ski")).setup()
    gold = json.load(open(golden("cfg2_cells.json")))
    g = scene_head(sim).grid
    assert g.num_cells == gold["num_cells"] == 953688
    dens = _densities(sim)
    assert hashlib.sha256(dens.tobytes()).hexdigest() == gold["density_sha256"]
    n = g.num_nodes
    box = np.ctypeslib.as_array(g.node_box, shape=(n, 6))
    cell = np.ctypeslib.as_array(g.node_cell, shape=(n,))
    leaf = np.ctypeslib.as_array(g.node_first_child, shape=(n,)) < 0
    vol = ((box[:, 3] - box[:, 0]) * (box[:, 4] - box[:, 1]) * (box[:, 5] - box[:, 2]))[leaf][np.argsort(cell[leaf])]
    assert hashlib.sha256(np.ascontiguousarray(vol).tobytes()).hexdigest() == gold["volume_sha256"]
    total = 0
    fixture = _ray_fixture("cfg2")
    assert len(fixture) == 312
    for r, k, m_ref, ds_ref in fixture:
        m, ds = O.trace_ray(sim, r, k)
        assert np.array_equal(m, m_ref), (r, k)
        assert np.array_equal(ds.view(np.uint64), ds_ref.view(np.uint64)), (r, k)
        total += len(m_ref)
    assert total > 10000
    # the photon loop on this scene: 1e5 histories with the reference's generator continued after the setup draws -- the SED files
    # byte for byte, and every FITS frame (flux components, statistics) equal to the reference's in sums over 8 x 8 pixel blocks
    # (tests/golden/cfg2_full_rebinned.npz: the float32 pixels of the reference's files summed in double precision)
    assert sim.num_packets == 100000
    frames, counters = O.run_primary(sim, 0, sim.num_packets, O.RNG_MT19937)
    assert counters.histories == sim.num_packets
    sim.write(frames, str(tmp_path))
    for f in ("cfg2_i0_sed.dat", "cfg2_i0_sedstats.dat"):
        assert _same_file(golden(f), str(tmp_path / f)), f
    from test_gpu_parity import _read_fits
    blocks = np.load(golden("cfg2_full_rebinned.npz"))
    for name in blocks.files:
        a = _read_fits(str(tmp_path / f"cfg2_i0_{name}.fits")).reshape(512, 512)
        assert np.array_equal(a.reshape(64, 8, 64, 8).sum(axis=(1, 3)), blocks[name]), name


def test_full_size_voronoi_rays_bit_exact(tmp_path):
    """BASELINE configs[4] at full size: the host layer's own tessellation of 10^5 sites carries the reference's
    (Voro++'s) paths -- 208 rays, 7339 segments, cell indices and lengths bit for bit"""
    sim = config5_simulation(tmp_path)
    total = 0
    for r, k, m_ref, ds_ref in _ray_fixture("cfg5"):
        m, ds = O.trace_ray(sim, r, k)
        assert np.array_equal(m, m_ref)
        assert np.array_equal(ds.view(np.uint64), ds_ref.view(np.uint64))
        total += len(m_ref)
    assert total > 7000


def test_specific_luminosity_normalization(tmp_path):
    """SpecificLuminosityNormalization (SpecificLuminosityNormalization.cpp:12-21) with a per-frequency value
    (Units::fromFluxStyle): the SED files equal the reference's byte for byte"""
    sim = Simulation(ski("cfg3norm.ski")).setup()
    frames, _ = O.run_primary(sim, 0, sim.num_packets, O.RNG_MT19937)
    sim.write(frames, str(tmp_path))
    expected = [f for f in os.listdir(golden("")) if f.startswith("cfg3norm_") and f.endswith("_sed.dat")]
    assert len(expected) == 2
    for f in expected:
        assert _same_file(golden(f), str(tmp_path / f)), f


@pytest.mark.parametrize("name", ["cfg3disk", "cfg3plum", "cfg3multi", "cfg3ten", "cfg3flat", "cfg3off"])
def test_disk_and_plummer_sources(tmp_path, name):
    """launch positions drawn from a truncated exponential disk (ExpDiskGeometry.cpp:46-68: Lambert W_-1 for the radius,
    rejection on the truncations) and from a Plummer sphere (PlummerGeometry.cpp:29-33); cfg3multi: a source system of
    three sources (SourceSystem.cpp:14-40,75-107: composite-bias launch weights, history index ranges per source), cfg3ten: of ten; cfg3flat: SpheroidalGeometryDecorator around the Sersic source and around a
    Plummer dust distribution; cfg3off: OffsetGeometryDecorator around source and dust: the SED files equal the reference's byte for byte"""
    sim = Simulation(ski(name + ".ski")).setup()
    frames, _ = O.run_primary(sim, 0, sim.num_packets, O.RNG_MT19937)
    sim.write(frames, str(tmp_path))
    expected = [f for f in os.listdir(golden("")) if f.startswith(name + "_") and f.endswith("_sed.dat")]
    assert len(expected) == 2
    for f in expected:
        assert _same_file(golden(f), str(tmp_path / f)), f
