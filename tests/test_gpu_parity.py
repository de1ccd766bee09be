"""GPU parity tests (run on the MI355X box with `pytest -m gpu`): the HIP engine, called through the C ABI of
include/pmc.h, against the CPU oracle (oracle/life_cycle.cpp) on the same scene.

* grid walk: the (m, ds) sequence of fixed rays must be BIT-EXACT (integer cell indices and IEEE doubles);
* photon loop: with the same per-history Philox streams the oracle follows the same histories as the GPU, so the
  detector arrays agree to floating-point summation order and libm differences -- far tighter than Monte Carlo
  noise.  Tolerances are stated per test.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from skirt9_amd.engine import set_tuning
from conftest import ski
from skirt9_amd.host import Simulation

pytestmark = pytest.mark.gpu


def _engine(sim):
    from skirt9_amd.engine import Engine
    return Engine(sim.scene, 0)


def _rays(sim, n, seed, scale):
    rng = np.random.default_rng(seed)
    rays = []
    for i in range(n):
        r = (rng.random(3) - 0.5) * 2 * scale * (1.3 if i % 4 == 0 else 0.9)   # some start outside the grid
        k = rng.normal(size=3)
        k /= np.linalg.norm(k)
        rays.append((r, k))
    # degenerate rays: axis-parallel through cell boundaries, from the origin, grazing
    rays += [(np.zeros(3), np.array([1.0, 0.0, 0.0])), (np.zeros(3), np.array([0.0, -1.0, 0.0])),
             (np.zeros(3), np.array([0.0, 0.0, 1.0])), (np.zeros(3), np.array([1.0, 1.0, 0.0]) / np.sqrt(2.0)),
             (np.array([scale * 0.25, scale * 0.125, 0.0]), np.array([0.0, 0.0, -1.0])),
             (np.array([-scale * 2, 0.0, 0.0]), np.array([1.0, 0.0, 0.0])),
             (np.array([-scale * 2, 1.0, 1.0]), np.array([-1.0, 0.0, 0.0]))]
    return rays


@pytest.mark.parametrize("name,scale", [("cfg1.ski", 3.0857e16), ("cfg1mesh.ski", 3.0857e16), ("cfg1mesh2.ski", 3.0857e16), ("cfg1long.ski", 3.0857e16), ("cfg2small.ski", 4000 * 3.0857e16), ("cfg4small.ski", 4000 * 3.0857e16), ("cfg5small.ski", 4000 * 3.0857e16), ("cfg5peak.ski", 4000 * 3.0857e16), ("cfg5imp.ski", 4000 * 3.0857e16), ("cfg5relax.ski", 4000 * 3.0857e16),
                                        ("cfg2deep.ski", 300 * 3.0857e16), ("cfg2deeper.ski", 100 * 3.0857e16)])
def test_trace_ray_bit_exact(name, scale):
    sim = Simulation(ski(name)).setup()
    eng = _engine(sim)
    for r, k in _rays(sim, 200, 1, scale):
        m_ref, ds_ref = O.trace_ray(sim, r, k)
        m_gpu, ds_gpu = eng.trace_ray(r, k)
        assert len(m_ref) == len(m_gpu), (r, k)
        assert np.array_equal(m_ref, m_gpu), (r, k)
        # bit-exact: compare the IEEE bit patterns
        assert np.array_equal(ds_ref.view(np.uint64), ds_gpu.view(np.uint64)), (r, k)


def test_deepest_octree_rays_equal_the_reference():
    """tests/ski/cfg4deepest.ski: an octree of EIGHTEEN levels (TreePolicy.hpp:32-35 allows maxLevel up to 99; rounds 1-5 stopped at 15: 20-bit
    byte offsets in the box codes, 4-bit size exponents in links and task records).  The traversal kernel against the REFERENCE's own (m, ds)
    dump (tests/golden/cfg4deepest_rays_ref.txt: 88 rays, half of them aimed at the cusp the tree is refined around, up to 100 segments each),
    bit for bit (TreeSpatialGrid.cpp:140-216)."""
    from conftest import golden
    sim = Simulation(ski("cfg4deepest.ski")).setup()
    eng = _engine(sim)
    rays = [[float.fromhex(t) for t in line.split()] for line in open(golden("cfg4deepest_rays.txt"))]
    ref = open(golden("cfg4deepest_rays_ref.txt")).read().split("\n")
    pos = total = deep = 0
    for i, ray in enumerate(rays):
        head = ref[pos].split()
        assert head[0] == "ray" and int(head[1]) == i
        n = int(head[2])
        k = np.array([float.fromhex(v) for v in head[3:6]])  # the direction as normalised by the reference
        m_ref = np.array([int(ref[pos + 1 + j].split()[0]) for j in range(n)], dtype=np.int32)
        ds_ref = np.array([float.fromhex(ref[pos + 1 + j].split()[1]) for j in range(n)])
        pos += 1 + n
        m, ds = eng.trace_ray(np.array(ray[:3]), k)
        assert len(m) == n, (i, len(m), n)
        assert np.array_equal(m, m_ref), i
        assert np.array_equal(ds.view(np.uint64), ds_ref.view(np.uint64)), i
        total += n
        deep += int((ds_ref[ds_ref > 0] < 0.5 * 3.0857e16).sum())   # segments shorter than half a parsec: cells of level >= 17
    assert total > 3000 and deep > 200


def test_octree_of_twenty_levels_and_the_limit(tmp_path):
    """the deepest tree the engine's packed indices hold (PMC_MAX_LEVEL = 20: 2^20 finest cells per axis, lower-wall indices up to 2^20 - 1 in
    20 bits, size exponents up to 20 through the escape of the box code): cfg4deepest.ski with maxLevel="20" -- rays through the cusp against
    the oracle, bit for bit, and the photon loop; one level more must fail with a message that names the limit (TreePolicy allows 99)"""
    import shutil
    from skirt9_amd.engine import Engine
    from skirt9_amd.host import scene_head
    text = open(ski("cfg4deepest.ski")).read()
    assert 'maxLevel="18"' in text
    shutil.copy(ski("cfg4deepest_sph.txt"), tmp_path / "cfg4deepest_sph.txt")
    for level in (20, 21):
        path = tmp_path / f"deep{level}.ski"
        path.write_text(text.replace('maxLevel="18"', f'maxLevel="{level}"'))
        n = 5000
        sim = Simulation(str(path), num_packets=n).setup()
        g = scene_head(sim).grid
        assert np.ctypeslib.as_array(g.node_level, (g.num_nodes,)).max() == level
        if level == 21:
            with pytest.raises(RuntimeError, match="deeper than 20 levels"):
                Engine(sim.scene, 0)
            continue
        eng = _engine(sim)
        pc = 3.0857e16
        centre = np.array([1234.5, -567.25, 89.125]) * pc
        rng = np.random.default_rng(5)
        segments = short = 0
        for i in range(60):
            k = rng.normal(size=3)
            k /= np.linalg.norm(k)
            b = (rng.random(3) - 0.5) * 2 * 10 * pc * 2.0 ** (-0.4 * i)     # impact parameters from 10 pc down to 1e-6 pc
            r = centre + b - k * 300 * pc
            m_ref, ds_ref = O.trace_ray(sim, r, k)
            m_gpu, ds_gpu = eng.trace_ray(r, k)
            assert np.array_equal(m_ref, m_gpu), i
            assert np.array_equal(ds_ref.view(np.uint64), ds_gpu.view(np.uint64)), i
            segments += len(m_ref)
            short += int((ds_ref[ds_ref > 0] < 0.05 * pc).sum())
        assert segments > 3000 and short > 100     # (cells of a twentieth of a parsec: levels 19 and 20)
        eng.run_primary(0, n, 8)
        gpu = eng.download()
        ref, counters = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=8)
        assert abs(eng.counters()["cell_visits"] - counters.cell_visits) <= 1e-4 * counters.cell_visits
        _compare_frames(sim, gpu, ref, n)


def _pixel_shapes(sim):
    """(ny, nx) of every instrument of the ski file in order ((1, 1) for an SEDInstrument: FluxRecorder's one bin)"""
    import re
    shapes = []
    for kind, attrs in re.findall(r"<(FullInstrument|FrameInstrument|SEDInstrument)\b([^>]*)>", open(sim.path).read()):
        if kind == "SEDInstrument":
            shapes.append((1, 1))
            continue
        nx = re.search(r'numPixelsX="(\d+)"', attrs)
        ny = re.search(r'numPixelsY="(\d+)"', attrs)
        shapes.append((int(ny.group(1)) if ny else 250, int(nx.group(1)) if nx else 250))
    return shapes


def _box3(plane):
    """sums over the 3 x 3 neighbourhood of every pixel"""
    p = np.pad(plane, 1)
    return sum(p[1 + dj:1 + dj + plane.shape[0], 1 + di:1 + di + plane.shape[1]] for dj in (-1, 0, 1) for di in (-1, 0, 1))


def _compare_frames(sim, gpu, ref, n):
    """flux arrays: relative agreement of the totals 1e-9; per element 1e-6 relative + tiny absolute, with a
    small allowance (<= 0.1 % of non-zero elements) for packets that land in a neighbouring pixel because of
    last-bit differences in the device libm -- and such a contribution must reappear NEXT to where the oracle put it: around
    every element of a flux frame (and of the sum-of-w frame of the statistics) that differs, the sums over the 3 x 3
    neighbourhood agree to 1e-9 of the frame's largest element; statistics arrays likewise (the powers of a history's
    per-pixel sum are not additive over pixels: for them the count allowance alone)."""
    lay = sim.layout(0)
    assert gpu.shape == ref.shape
    tot = np.abs(ref).sum()
    assert abs(gpu.sum() - ref.sum()) <= 1e-9 * tot
    # element by element, with the absolute allowance scaled PER ARRAY (flux arrays, and every power of the statistics on its
    # own: the sums of w^4 are seventy orders of magnitude above the fluxes)
    shapes = _pixel_shapes(sim)
    bad = np.zeros(gpu.shape, dtype=bool)
    covered = 0
    for inst in range(len(shapes)):
        li = sim.layout(inst)
        blocks = []
        if li.sed_offset >= 0:
            blocks.append((li.sed_offset, li.num_components * li.num_lambda))
        if li.ifu_offset >= 0:
            blocks.append((li.ifu_offset, li.num_components * li.num_lambda * li.npix))
        for k in range(5):
            if li.wsed_offset >= 0:
                blocks.append((li.wsed_offset + k * li.num_lambda, li.num_lambda))
            if li.wifu_offset >= 0:
                blocks.append((li.wifu_offset + k * li.num_lambda * li.npix, li.num_lambda * li.npix))
        for at, count in blocks:
            a, b = gpu[at:at + count], ref[at:at + count]
            bad[at:at + count] = np.abs(a - b) > (1e-6 * np.abs(b) + 1e-12 * np.abs(b).max())
            covered += count
    assert covered == gpu.size, (covered, gpu.size)
    nz = max(1, np.count_nonzero(ref))
    assert bad.sum() <= max(4, 1e-3 * nz), f"{bad.sum()} of {nz} elements differ"
    if bad.any():
        for inst, (ny, nx) in enumerate(shapes):
            li = sim.layout(inst)
            if li.npix != nx * ny or li.npix == 1:
                continue
            planes = []
            if li.ifu_offset >= 0:
                planes += [li.ifu_offset + q * li.npix for q in range(li.num_components * li.num_lambda)]
            if li.wifu_offset >= 0:
                planes += [li.wifu_offset + (li.num_lambda + ell) * li.npix for ell in range(li.num_lambda)]  # k = 1: sum of w
            for at in planes:
                where = bad[at:at + li.npix].reshape(ny, nx)
                if not where.any():
                    continue
                a, b = gpu[at:at + li.npix].reshape(ny, nx), ref[at:at + li.npix].reshape(ny, nx)
                gap = np.abs(_box3(a) - _box3(b))[where]
                assert gap.max() <= 1e-9 * np.abs(b).max(), f"instrument {inst}: a contribution moved beyond the neighbouring pixel"
    # SED block exact to summation order
    if lay.sed_offset >= 0:
        nsed = lay.num_components * lay.num_lambda
        a = gpu[lay.sed_offset:lay.sed_offset + nsed]
        b = ref[lay.sed_offset:lay.sed_offset + nsed]
        assert np.allclose(a, b, rtol=1e-9, atol=0)
    if lay.wsed_offset >= 0:
        a = gpu[lay.wsed_offset:lay.wsed_offset + 5 * lay.num_lambda]
        b = ref[lay.wsed_offset:lay.wsed_offset + 5 * lay.num_lambda]
        assert np.allclose(a, b, rtol=1e-9, atol=0)
        # sum of w^0 over the wavelength bins = histories that reached the SED: all of them, unless the instrument sits in the
        # observer frame of a model at redshift z (cfg3z: packets whose lambda (1+z) falls outside the instrument's grid)
        assert a[:lay.num_lambda].sum() == b[:lay.num_lambda].sum() and a[:lay.num_lambda].sum() <= n
        if "cfg3z" not in sim.path:
            assert a[:lay.num_lambda].sum() == n


@pytest.mark.parametrize("name,n", [("cfg1.ski", 20000), ("cfg2small.ski", 20000), ("cfg3small.ski", 20000), ("cfg1nf.ski", 50000), ("cfg2nf.ski", 50000), ("cfg4small.ski", 20000), ("cfg4deepest.ski", 20000), ("cfg1file.ski", 20000), ("cfg3file.ski", 20000), ("cfg5small.ski", 20000), ("cfg1sed.ski", 20000), ("cfg3sed.ski", 20000), ("cfg3disk.ski", 20000), ("cfg3plum.ski", 20000), ("cfg3multi.ski", 20000), ("cfg3ten.ski", 20000), ("cfg3twelve.ski", 20000), ("cfg1mesh.ski", 20000), ("cfg1mesh2.ski", 20000), ("cfg1long.ski", 5000), ("cfg3flat.ski", 20000), ("cfg3off.ski", 20000), ("cfg3z.ski", 20000), ("cfg2deep.ski", 20000), ("cfg2deeper.ski", 20000), ("cfg2ea.ski", 20000), ("cfg1nfea.ski", 50000), ("cfg2mm.ski", 20000), ("cfg2mmea.ski", 20000), ("cfg1mmnf.ski", 50000), ("cfg3mm.ski", 20000), ("cfg1con.ski", 20000), ("cfg1netzer.ski", 20000), ("cfg1laser.ski", 20000), ("cfg2agn.ski", 20000), ("cfg1nomed.ski", 20000)])
def test_photon_loop_matches_oracle(name, n):
    sim = Simulation(ski(name), num_packets=n).setup()
    eng = _engine(sim)
    seed = 12345
    eng.run_primary(0, n, seed)
    gpu = eng.download()
    ref, counters = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=seed)
    c = eng.counters()
    assert c["histories"] == n
    assert c["stat_overflows"] == 0
    # identical histories => identical amounts of work (allow a handful of last-bit decision flips)
    assert abs(c["cell_visits"] - counters.cell_visits) <= 1e-4 * counters.cell_visits
    assert abs(c["scatterings"] - counters.scatterings) <= 1e-4 * counters.scatterings + 2
    _compare_frames(sim, gpu, ref, n)


@pytest.mark.parametrize("name,n", [("cfg1.ski", 20000), ("cfg3small.ski", 20000), ("cfg5small.ski", 20000), ("cfg2deep.ski", 20000), ("cfg2nf.ski", 50000),
                                    ("cfg3ten.ski", 20000)])
def test_explicit_absorption_matches_oracle(name, n, tmp_path):
    """PhotonPacketOptions::explicitAbsorption (MonteCarloSimulation.cpp:568-569, 729-733, 751-766; MediumSystem.cpp:905-932, 1075-1110): the
    scenes of the parity list with explicitAbsorption="true" -- Cartesian, octree (incl. the 21-bit index variant and the non-forced
    cycle), Voronoi, panchromatic (the absorption cross section travels in the slot) and a ten-source system -- HIP engine against
    the oracle, which the reference's own files pin on cfg2ea / cfg1nfea / cfg1rfea (test_oracle_golden.py)"""
    import shutil
    text = open(ski(name)).read()
    assert 'explicitAbsorption="false"' in text
    for f in os.listdir(os.path.dirname(ski(name))):
        if f.endswith(".txt"):
            shutil.copy(ski(f), tmp_path / f)   # (input files named in a ski file are read from its directory)
    path = tmp_path / name.replace(".ski", "ea.ski")
    path.write_text(text.replace('explicitAbsorption="false"', 'explicitAbsorption="true"'))
    sim = Simulation(str(path), num_packets=n).setup()
    eng = _engine(sim)
    eng.run_primary(0, n, 777)
    gpu = eng.download()
    ref, counters = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=777)
    plain, _ = O.run_primary(Simulation(ski(name), num_packets=n).setup(), 0, n, O.RNG_PHILOX, seed=777)
    assert abs(plain.sum() - ref.sum()) > 1e-6 * np.abs(ref).sum()      # (the cycle changes the weights: not the same numbers)
    c = eng.counters()
    assert c["histories"] == n and c["stat_overflows"] == 0
    assert abs(c["cell_visits"] - counters.cell_visits) <= 1e-4 * counters.cell_visits
    assert abs(c["scatterings"] - counters.scatterings) <= 1e-4 * counters.scatterings + 2
    _compare_frames(sim, gpu, ref, n)


@pytest.mark.parametrize("changes", [{'forceScattering="true"': 'forceScattering="false"'},
                                     {'pathLengthBias="0.5"': 'pathLengthBias="0"'},
                                     {'pathLengthBias="0.5"': 'pathLengthBias="1"', 'minWeightReduction="1e4"': 'minWeightReduction="1e2"'}])
def test_voronoi_propagation_kernel_options(changes, tmp_path):
    """voroPropKernel (the propagation walks of a Voronoi grid, two lanes per walk) in the cycles the plain scene does not reach: without forced
    scattering (one walk up to the drawn depth, escapes), and with the path-length bias at its ends (exponential deviates only -- the branch with
    a rejection round that continues the slot's random stream --, uniform deviates only) -- HIP engine against the oracle on the reduced
    configs[4] scene"""
    import shutil
    name, n = "cfg5small.ski", 30000
    text = open(ski(name)).read()
    for old, new in changes.items():
        assert old in text
        text = text.replace(old, new)
    for f in os.listdir(os.path.dirname(ski(name))):
        if f.endswith(".txt"):
            shutil.copy(ski(f), tmp_path / f)
    path = tmp_path / "cfg5variant.ski"
    path.write_text(text)
    sim = Simulation(str(path), num_packets=n).setup()
    eng = _engine(sim)
    eng.run_primary(0, n, 31337)
    gpu = eng.download()
    ref, counters = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=31337)
    c = eng.counters()
    assert c["histories"] == n and c["stat_overflows"] == 0
    assert abs(c["cell_visits"] - counters.cell_visits) <= 1e-4 * counters.cell_visits
    assert abs(c["scatterings"] - counters.scatterings) <= 1e-4 * counters.scatterings + 2
    _compare_frames(sim, gpu, ref, n)


SECOND_COMPONENT = """</GeometricMedium><GeometricMedium velocityMagnitude="0 km/s" magneticFieldStrength="0 uG">
     <geometry type="Geometry"><PlummerGeometry scaleLength="1500 pc"/></geometry>
     <materialMix type="MaterialMix"><MeanListDustMix wavelengths="0.1 micron, 1 micron" extinctionCoefficients="5000 m2/kg, 1200 m2/kg" albedos="0.8, 0.4" asymmetryParameters="0.1, 0.9"/></materialMix>
     <normalization type="MaterialNormalization"><MassMaterialNormalization mass="2e6 Msun"/></normalization>
    </GeometricMedium>"""


@pytest.mark.parametrize("name,n,ea", [("cfg5small.ski", 20000, False), ("cfg2deep.ski", 20000, False), ("cfg3ten.ski", 20000, True), ("cfg2nf.ski", 50000, True),
                                       ("cfg2nf.ski", 50000, False), ("cfg3rfea.ski", 20000, True), ("cfg5small.ski", 20000, True)])
def test_several_components_match_oracle(name, n, ea, tmp_path):
    """hasMultipleConstantSectionMedia branches (MediumSystem.cpp:874-887, 934-955, 1013-1037, 1112-1153, 1225-1242; albedo and weights
    :678-730; component pick :806-817; consolidated peel-off :734-767): a second component (Plummer sphere, another dust mix) added to
    scenes of the parity list -- Voronoi, the 21-bit octree kernels, ten sources (panchromatic: a wavelength bin per component in the
    slot), the non-forced cycle with and without explicit absorption, radiation field -- HIP engine against the oracle, which the
    reference's own files pin on cfg2mm / cfg2mmea / cfg1mmnf / cfg3mm / cfg1mmrf (test_oracle_golden.py)"""
    import shutil
    text = open(ski(name)).read()
    assert text.count("</GeometricMedium>") == 1
    text = text.replace("</GeometricMedium>", SECOND_COMPONENT)
    if ea: text = text.replace('explicitAbsorption="false"', 'explicitAbsorption="true"')
    for f in os.listdir(os.path.dirname(ski(name))):
        if f.endswith(".txt"):
            shutil.copy(ski(f), tmp_path / f)
    path = tmp_path / name.replace(".ski", "mm.ski")
    path.write_text(text)
    sim = Simulation(str(path), num_packets=n).setup()
    eng = _engine(sim)
    eng.run_primary(0, n, 4242)
    gpu = eng.download()
    ref, counters = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=4242)
    plain, _ = O.run_primary(Simulation(ski(name), num_packets=n).setup(), 0, n, O.RNG_PHILOX, seed=4242)
    assert abs(plain.sum() - ref.sum()) > 1e-6 * np.abs(ref).sum()
    c = eng.counters()
    assert c["histories"] == n and c["stat_overflows"] == 0
    assert abs(c["cell_visits"] - counters.cell_visits) <= 1e-4 * counters.cell_visits
    assert abs(c["scatterings"] - counters.scatterings) <= 1e-4 * counters.scatterings + 2
    _compare_frames(sim, gpu, ref, n)
    if sim.radiation_field_size > 0:
        gpu_rf = eng.download_radiation_field()
        _, ref_rf, _ = O.run_primary_rf(sim, 0, n, O.RNG_PHILOX, seed=4242)
        assert abs(gpu_rf.sum() - ref_rf.sum()) <= 1e-9 * ref_rf.sum()
        assert np.array_equal(gpu_rf > 0, ref_rf > 0)
        assert (np.abs(gpu_rf - ref_rf) > 1e-6 * np.abs(ref_rf) + 1e-13 * ref_rf.max()).sum() == 0


RF_ON = ('<RadiationFieldOptions storeRadiationField="true"><radiationFieldWLG type="DisjointWavelengthGrid"><LogWavelengthGrid minWavelength="0.15 micron" '
         'maxWavelength="8 micron" numWavelengths="6"/></radiationFieldWLG></RadiationFieldOptions>')


@pytest.mark.parametrize("rf,ea,mm", [(True, False, False), (False, True, False), (False, False, True), (True, True, False), (True, True, True)])
def test_voronoi_flavours_run_in_the_voronoi_kernels(rf, ea, mm, tmp_path):
    """Radiation field, explicit absorption and several medium components on a Voronoi grid (cfg5small with those options switched on): since
    round 6 their propagation walks run in voroPropKernel<RF, EA, MM> and, with several components, their peel-off walks in voroPeelKernel<true>
    (two lanes per walk on the tables of runs) instead of the generic one-lane-per-walk kernel (VoronoiMeshSnapshot.cpp:1058-1188 under
    MediumSystem.cpp:849-932, 1075-1110, 1192-1260; MonteCarloSimulation.cpp:638-662).  The HIP engine against the oracle on the same histories --
    frames, radiation field, counted work -- and against the generic kernel on the same tables (PMC_VORO_PLAIN_PROP_ONLY, PMC_VORO_NO_PEEL_KERNEL:
    the same cells visited, frames to summation order)."""
    import shutil
    from skirt9_amd.engine import Engine, clear_tuning
    name, n = "cfg5small.ski", 20000
    text = open(ski(name)).read()
    if mm:
        assert text.count("</GeometricMedium>") == 1
        text = text.replace("</GeometricMedium>", SECOND_COMPONENT)
    if ea:
        text = text.replace('explicitAbsorption="false"', 'explicitAbsorption="true"')
    if rf:
        assert '<RadiationFieldOptions storeRadiationField="false"/>' in text
        text = text.replace('<RadiationFieldOptions storeRadiationField="false"/>', RF_ON)
    for f in os.listdir(os.path.dirname(ski(name))):
        if f.endswith(".txt"):
            shutil.copy(ski(f), tmp_path / f)
    path = tmp_path / "cfg5flavour.ski"
    path.write_text(text)
    sim = Simulation(str(path), num_packets=n).setup()
    assert (sim.radiation_field_size > 0) == rf

    def run(engine):
        engine.run_primary(0, n, 90125)
        c = engine.counters()
        return engine.download(), (engine.download_radiation_field() if rf else None), c

    eng = _engine(sim)
    gpu, gpu_rf, c = run(eng)
    if rf:
        ref, ref_rf, counters = O.run_primary_rf(sim, 0, n, O.RNG_PHILOX, seed=90125)
    else:
        ref, counters = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=90125)
    assert c["histories"] == n and c["stat_overflows"] == 0
    assert abs(c["cell_visits"] - counters.cell_visits) <= 1e-4 * counters.cell_visits
    assert abs(c["scatterings"] - counters.scatterings) <= 1e-4 * counters.scatterings + 2
    _compare_frames(sim, gpu, ref, n)
    if rf:
        assert abs(gpu_rf.sum() - ref_rf.sum()) <= 1e-9 * ref_rf.sum()
        assert np.array_equal(gpu_rf > 0, ref_rf > 0)
        assert (np.abs(gpu_rf - ref_rf) > 1e-6 * np.abs(ref_rf) + 1e-13 * ref_rf.max()).sum() == 0
    # the generic kernel on the same tables
    set_tuning("PMC_VORO_PLAIN_PROP_ONLY", "1")
    set_tuning("PMC_VORO_NO_PEEL_KERNEL", "1")
    try:
        other = Engine(sim.scene, 0)
        frames, field, co = run(other)
        other.close()
    finally:
        clear_tuning()
    assert (co["histories"], co["cell_visits"], co["scatterings"], co["detector_updates"]) == \
        (c["histories"], c["cell_visits"], c["scatterings"], c["detector_updates"])
    assert np.allclose(frames, gpu, rtol=1e-10, atol=1e-13 * np.abs(gpu).max())
    if rf:
        assert np.allclose(field, gpu_rf, rtol=1e-9, atol=1e-13 * gpu_rf.max())


def test_deep_octree_uses_the_wide_kernels():
    """cfg2deep.ski reaches level 12: the octree walk kernels with 21-bit index fields (and, for want of LDS next to the
    98 KB coordinate table, the peel-off kernel with service rounds and the propagation kernel without pass-1 records)"""
    from skirt9_amd.host import scene_head
    sim = Simulation(ski("cfg2deep.ski")).setup()
    g = scene_head(sim).grid
    levels = np.ctypeslib.as_array(g.node_level, (g.num_nodes,))
    assert levels.max() == 12
    # cfg2deeper.ski reaches level 14: the coordinate table (3 x 16385 doubles) does not fit in LDS, the walk kernels read the walls
    # of a step from global memory (TreePolicy allows maxLevel up to 99; the engine's box codes hold levels up to 20)
    sim = Simulation(ski("cfg2deeper.ski")).setup()
    g = scene_head(sim).grid
    assert np.ctypeslib.as_array(g.node_level, (g.num_nodes,)).max() == 14
    # cfg4deepest.ski reaches level 18 (round 6: box codes of three table indices, five-bit size exponents; PMC_MAX_LEVEL = 20)
    sim = Simulation(ski("cfg4deepest.ski")).setup()
    g = scene_head(sim).grid
    assert np.ctypeslib.as_array(g.node_level, (g.num_nodes,)).max() == 18


def test_eight_observers_and_three_slot_groups():
    """cfg3eight.ski: eight instruments with eight different observers (eight peel-off walk kernels per generation, each with
    its own task cursor) at a packet count that gives three slot groups -- frames against the oracle on the same histories"""
    n = 250000
    sim = Simulation(ski("cfg3eight.ski"), num_packets=n).setup()
    eng = _engine(sim)
    eng.run_primary(0, n, 3)
    gpu = eng.download()
    ref, counters = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=3)
    c = eng.counters()
    assert c["histories"] == n and c["stat_overflows"] == 0
    assert abs(c["cell_visits"] - counters.cell_visits) <= 1e-4 * counters.cell_visits
    assert eng.last_timing()["generations"] > 20
    tot = np.abs(ref).sum()
    assert abs(gpu.sum() - ref.sum()) <= 1e-9 * tot
    scale = np.abs(ref).max()
    bad = np.abs(gpu - ref) > (1e-6 * np.abs(ref) + 1e-12 * scale)
    assert bad.sum() <= max(4, 1e-3 * np.count_nonzero(ref)), int(bad.sum())


def test_partition_independence():
    """histories are keyed by index: two launches over [0,n/2) and [n/2,n) give the same frames as one over [0,n)"""
    n = 8000
    sim = Simulation(ski("cfg2small.ski"), num_packets=n).setup()
    eng = _engine(sim)
    eng.run_primary(0, n, 7)
    whole = eng.download()
    eng.clear()
    eng.run_primary(0, n // 2, 7)
    eng.run_primary(n // 2, n - n // 2, 7)
    split = eng.download()
    assert np.allclose(whole, split, rtol=1e-10, atol=1e-14 * np.abs(whole).max())


@pytest.mark.parametrize("name", ["cfg2small.ski", "cfg1.ski", "cfg5small.ski", "cfg3small.ski", "cfg1rf.ski", "cfg2mm.ski", "cfg2ea.ski", "cfg1nf.ski"])
def test_results_do_not_depend_on_what_device_memory_held(name):
    """the engine zero-fills only the arrays whose zeros it relies on; everything else is written before it is read, also in the record
    a walk kernel loads for a slot that turns out to have no walk.  With every other array filled with 0xA5 bytes at allocation
    (PMC_POISON_ALLOCATIONS: neither the zeros of fresh device memory nor the plausible values a destroyed context leaves behind) the same
    histories do the same work and fill the same frames -- and a second context created in the memory of a destroyed one runs like the first."""
    from skirt9_amd.engine import Engine, set_tuning, clear_tuning
    n = 20000
    sim = Simulation(ski(name), num_packets=n).setup()

    def run(engine):
        engine.clear()
        engine.reset_counters()
        engine.run_primary(0, n, 4242)
        c = engine.counters()
        return engine.download(), (c["histories"], c["cell_visits"], c["scatterings"], c["detector_updates"])

    keep = _engine(sim)
    base, base_counts = run(keep)
    set_tuning("PMC_POISON_ALLOCATIONS")
    poisoned, counts = run(Engine(sim.scene, 0))
    clear_tuning()
    assert counts == base_counts
    assert np.allclose(poisoned, base, rtol=1e-10, atol=1e-13 * np.abs(base).max())
    # (slot reuse next to a live context: the second context's arrays land in what the first one's freed)
    first = Engine(sim.scene, 0)
    run(first)
    del first
    again, counts = run(Engine(sim.scene, 0))
    assert counts == base_counts
    assert np.allclose(again, base, rtol=1e-10, atol=1e-13 * np.abs(base).max())


def test_fits_output_from_gpu(tmp_path):
    """end to end: ski -> scene -> GPU -> calibrated FITS/SED files with the reference's names"""
    n = 20000
    sim = Simulation(ski("cfg1.ski"), num_packets=n).setup()
    eng = _engine(sim)
    eng.run_primary(0, n, sim.seed)
    sim.write(eng.download(), str(tmp_path))
    names = sorted(p.name for p in tmp_path.iterdir())
    assert "cfg1_i0_total.fits" in names and "cfg1_i0_sed.dat" in names and "cfg1_i0_stats4.fits" in names
    sed = np.loadtxt(tmp_path / "cfg1_i0_sed.dat")
    # reference value at 1e6 packets: total 1.5817e-06 Jy, transparent 2.9432e-06 Jy (tests/golden/cfg1_sed.dat);
    # at 2e4 packets the relative noise R of the total is about 4e-3 -> 5 sigma tolerance
    assert abs(sed[1] - 1.581714115e-06) < 5 * 4e-3 * 1.58e-06
    assert abs(sed[2] - 2.943198361e-06) < 1e-9 * 2.94e-06 + 5 * 4e-3 * 2.94e-06


@pytest.mark.parametrize("name,n", [("cfg1rf.ski", 20000), ("cfg3rf.ski", 20000), ("cfg1rfea.ski", 20000), ("cfg3rfea.ski", 20000), ("cfg1mmrf.ski", 20000)])
def test_radiation_field_matches_oracle(name, n):
    """storeRadiationField on the GPU (walk kernel flavour RF: L * lnmean(e^-tau0, e^-tau1) * ds per path segment, f64
    atomics into rf[m * nbins + ell]) against the oracle following the same Philox histories: totals to 1e-9, cells
    with a contribution to 1e-6 relative (summation order and libm last bits); the detector frames must not change"""
    sim = Simulation(ski(name), num_packets=n).setup()
    eng = _engine(sim)
    assert eng.radiation_field_size == sim.radiation_field_size > 0
    eng.run_primary(0, n, 5)
    gpu_frames = eng.download()
    gpu_rf = eng.download_radiation_field()
    ref_frames, ref_rf, _ = O.run_primary_rf(sim, 0, n, O.RNG_PHILOX, seed=5)
    assert abs(gpu_rf.sum() - ref_rf.sum()) <= 1e-9 * ref_rf.sum()
    assert np.array_equal(gpu_rf > 0, ref_rf > 0)
    bad = np.abs(gpu_rf - ref_rf) > 1e-6 * np.abs(ref_rf) + 1e-13 * ref_rf.max()
    assert bad.sum() == 0, int(bad.sum())
    assert abs(gpu_frames.sum() - ref_frames.sum()) <= 1e-9 * np.abs(ref_frames).sum()
    # accumulation over segments and reset
    eng.run_primary(n, n, 5)
    twice = eng.download_radiation_field()
    assert twice.sum() > 1.9 * gpu_rf.sum()
    eng.clear_radiation_field()
    assert eng.download_radiation_field().sum() == 0.


def test_radiation_field_log_overflow_falls_back_to_atomics(monkeypatch):
    """octree: the contributions of a generation go to a log of 128 entries per slot; with one entry per slot most waves find
    the log full and add their contributions atomically -- the table must not change (same tolerances as above)"""
    n = 20000
    set_tuning("PMC_RF_LOG_PER_SLOT", "1")
    sim = Simulation(ski("cfg3rf.ski"), num_packets=n).setup()
    eng = _engine(sim)
    eng.run_primary(0, n, 5)
    small = eng.download_radiation_field()
    set_tuning("PMC_RF_LOG_PER_SLOT", None)
    set_tuning("PMC_RF_ATOMICS", "1")
    eng2 = _engine(sim)
    eng2.run_primary(0, n, 5)
    atomics = eng2.download_radiation_field()
    _, ref_rf, _ = O.run_primary_rf(sim, 0, n, O.RNG_PHILOX, seed=5)
    for got in (small, atomics):
        assert abs(got.sum() - ref_rf.sum()) <= 1e-9 * ref_rf.sum()
        assert np.array_equal(got > 0, ref_rf > 0)
        assert (np.abs(got - ref_rf) > 1e-6 * np.abs(ref_rf) + 1e-13 * ref_rf.max()).sum() == 0


@pytest.mark.parametrize("name,n", [("cfg2small.ski", 60000), ("cfg3small.ski", 40000)])
def test_statistics_log_matches_atomics_and_oracle(name, n):
    """FluxRecorder::recordContributions (FluxRecorder.cpp:962-1014): the sums of w^k per bin from the statistics log of the slot groups
    (partitioned and summed in LDS: the default), from a log of ONE chunk per group (it fills up: most sums take the atomic path, the
    two mix) and from the atomic path alone (PMC_STAT_ATOMICS): each against the oracle's statistics arrays, over several segments"""
    sim = Simulation(ski(name), num_packets=n).setup()
    ref, _ = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=11)
    for switches in ({}, {"PMC_STAT_LOG_ENTRIES": "4096"}, {"PMC_STAT_ATOMICS": "1"}):
        for k, v in switches.items():
            set_tuning(k, v)
        eng = _engine(sim)
        half = n // 2
        eng.run_primary(0, half, 11)       # two segments: the log starts again at zero
        eng.run_primary(half, n - half, 11)
        gpu = eng.download()
        assert eng.counters()["stat_overflows"] == 0
        _compare_frames(sim, gpu, ref, n)
        for k in switches:
            set_tuning(k, None)
        eng.close()


def test_radiation_field_absent_unless_requested():
    sim = Simulation(ski("cfg1.ski"), num_packets=1000).setup()
    eng = _engine(sim)
    assert eng.radiation_field_size == 0 and eng.radiation_field_device_ptr == 0
    with pytest.raises(RuntimeError, match="does not store"):
        eng.download_radiation_field()


def _read_fits(path):
    """primary image of a FITS file written by FITSInOut::write / the host layer: float32, big endian"""
    raw = open(path, "rb").read()
    cards = {}
    pos = 0
    while True:
        card = raw[pos:pos + 80].decode("ascii")
        pos += 80
        if card.startswith("END"):
            break
        if "=" in card[:10]:
            cards[card[:8].strip()] = card[10:].split("/")[0].strip()
    pos = (pos + 2879) // 2880 * 2880
    shape = [int(cards[f"NAXIS{i}"]) for i in range(int(cards["NAXIS"]), 0, -1)]
    count = int(np.prod(shape))
    return np.frombuffer(raw[pos:pos + 4 * count], dtype=">f4").astype(np.float64).reshape(shape)


def test_fits_cube_within_noise_of_the_reference(tmp_path):
    """north_star: 'FITS output within 1 sigma of the CPU reference at equal packet count'.  Config 1 with 10^6
    packets on the GPU (Philox streams) against the files the UNMODIFIED reference wrote with its own generator
    (tests/golden/cfg1_i0_*.fits): per pixel the difference of the total surface brightness is compared with the
    Monte Carlo noise of both runs, sigma^2 = sigma_ref^2 + sigma_gpu^2 with sigma/F = R = sqrt(S2/S1^2 - 1/N) from each
    run's own statistics cubes (sum of w and of w^2 per pixel).  Stated tolerances: reduced chi^2 over the well sampled
    pixels within [0.85, 1.2] (1 expected), no pixel beyond 5.5 sigma, the integrated flux within 1 sigma x 3."""
    n = 1000000
    sim = Simulation(ski("cfg1.ski"), num_packets=n).setup()
    eng = _engine(sim)
    eng.run_primary(0, n, 20260929)
    sim.write(eng.download(), str(tmp_path))

    def cube(where, name):
        return _read_fits(os.path.join(where, f"cfg1_i0_{name}.fits")).reshape(-1)

    from conftest import golden
    gold = os.path.dirname(golden("x"))
    f_ref, f_gpu = cube(gold, "total"), cube(str(tmp_path), "total")
    stats = {}
    for tag, where in (("ref", gold), ("gpu", str(tmp_path))):
        s0, s1, s2 = cube(where, "stats0"), cube(where, "stats1"), cube(where, "stats2")
        with np.errstate(divide="ignore", invalid="ignore"):
            r2 = np.where(s1 > 0, s2 / s1 ** 2 - 1.0 / n, np.inf)
        stats[tag] = (s0, np.sqrt(np.maximum(r2, 0)))
    good = (stats["ref"][0] >= 30) & (stats["gpu"][0] >= 30)       # at least 30 contributing packets in both runs
    assert good.sum() > 1000
    with np.errstate(invalid="ignore"):
        sigma = np.sqrt((stats["ref"][1] * f_ref) ** 2 + (stats["gpu"][1] * f_gpu) ** 2)
    z = (f_gpu - f_ref)[good] / sigma[good]
    chi2 = float(np.mean(z ** 2))
    assert 0.85 <= chi2 <= 1.2, chi2
    assert np.abs(z).max() < 5.5, float(np.abs(z).max())
    # integrated flux: noise of the sum from the per-pixel variances
    total_sigma = np.sqrt(np.sum(sigma[np.isfinite(sigma)] ** 2))
    assert abs(f_gpu.sum() - f_ref.sum()) <= 3 * total_sigma
