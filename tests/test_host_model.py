"""Host model layer (libskirthost.so): ski parsing, units, grids, densities, tables, FITS writer -- against values
dumped from the reference (tests/golden/*_cells.npz) and known answers."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, golden, ski
from skirt9_amd.host import Simulation, lib


from skirt9_amd.host import Grid, Medium, SceneHead, scene_head  # noqa: E402,F401  (ctypes mirrors of include/pmc.h)


@pytest.mark.parametrize("name,cells,nodes", [("cfg1", 32768, 0), ("cfg1mesh", 31 * 24 * 20, 0), ("cfg1mesh2", 30 * 25 * 16, 0), ("cfg2small", 17592, 20105), ("cfg2deep", 2318, 2649), ("cfg2deeper", 4992, 5705), ("cfg4small", 7274, 8313), ("cfg4deepest", 1793, 2049)])
def test_cell_densities_bit_exact(name, cells, nodes):
    sim = Simulation(ski(name + ".ski")).setup()
    head = scene_head(sim)
    assert head.abi_version == 9
    assert head.grid.num_cells == cells
    if nodes:
        assert head.grid.kind == 2 and head.grid.num_nodes == nodes
    gold = np.load(golden(name + "_cells.npz"))
    dens = np.ctypeslib.as_array(head.medium.number_density, shape=(cells,))
    assert np.array_equal(dens.view(np.uint64), gold["density"].view(np.uint64))
    # dust cross sections at 0.55 micron as the reference's MeanListDustMix returns them
    lam = np.ctypeslib.as_array(head.medium.lambda_border, shape=(head.medium.num_lambda,))
    idx = max(0, np.searchsorted(lam, 0.55e-6, side="right") - 1)
    ext = np.ctypeslib.as_array(head.medium.sigma_ext, shape=(head.medium.num_lambda,))[idx]
    sca = np.ctypeslib.as_array(head.medium.sigma_sca, shape=(head.medium.num_lambda,))[idx]
    g = np.ctypeslib.as_array(head.medium.asymmpar, shape=(head.medium.num_lambda,))[idx]
    assert [ext, sca, g] == list(gold["mix"][1:4])


def test_octree_volumes_and_structure():
    sim = Simulation(ski("cfg2small.ski")).setup()
    g = scene_head(sim).grid
    n = g.num_nodes
    box = np.ctypeslib.as_array(g.node_box, shape=(n, 6))
    cell = np.ctypeslib.as_array(g.node_cell, shape=(n,))
    first = np.ctypeslib.as_array(g.node_first_child, shape=(n,))
    leaf = first < 0
    vol = (box[:, 3] - box[:, 0]) * (box[:, 4] - box[:, 1]) * (box[:, 5] - box[:, 2])
    gold = np.load(golden("cfg2small_cells.npz"))
    order = np.argsort(cell[leaf])
    assert np.array_equal(vol[leaf][order].view(np.uint64), gold["volume"].view(np.uint64))
    # leaves tile the domain
    assert np.isclose(vol[leaf].sum(), vol[0], rtol=1e-12)
    # neighbour lists: symmetric and leaf-only for leaves
    start = np.ctypeslib.as_array(g.nbr_start, shape=(6 * n + 1,))
    nbr = np.ctypeslib.as_array(g.nbr_list, shape=(start[-1],))
    comp = [1, 0, 3, 2, 5, 4]
    ids = np.nonzero(leaf)[0][:2000]
    for i in ids:
        for w in range(6):
            for q in nbr[start[6 * i + w]:start[6 * i + w + 1]]:
                assert first[q] < 0
                back = nbr[start[6 * q + comp[w]]:start[6 * q + comp[w] + 1]]
                assert i in back


def test_setup_draw_count():
    """one parent-thread stream: 100 density samples x 3 deviates per cell for a Cartesian grid"""
    sim = Simulation(ski("cfg1.ski")).setup()
    assert sim.setup_draws == 32768 * 100 * 3


def test_units_and_defaults(tmp_path):
    text = open(ski("cfg1.ski")).read()
    # attribute omitted -> reference default; other units of the same quantity
    text2 = text.replace('minX="-1 pc"', 'minX="-3.08567758e16 m"').replace(' roll="0 deg"', "")
    p = tmp_path / "variant.ski"
    p.write_text(text2)
    a = Simulation(ski("cfg1.ski")).setup()
    b = Simulation(str(p)).setup()
    ga, gb = scene_head(a).grid, scene_head(b).grid
    assert ga.xmin == gb.xmin == -3.08567758e16
    assert ga.eps == gb.eps


@pytest.mark.parametrize("old,new,message", [
    ('<LinMesh numBins="32"/></meshX>', '<FileMesh filename="mesh.txt"/></meshX>', "FileMesh"),
    ("MeanListDustMix", "DraineLiDustMix", "DraineLiDustMix"),
    ('simulationMode="OligoExtinctionOnly"', 'simulationMode="DustEmission"', "simulationMode"),
    ('maxX="1 pc" minY', 'maxX="1 furlong" minY', "unit"),
    ("</MonteCarloSimulation>", "", "XML"),
])
def test_unsupported_or_broken_ski_fails_loudly(tmp_path, old, new, message):
    text = open(ski("cfg1.ski")).read()
    assert old in text
    p = tmp_path / "bad.ski"
    p.write_text(text.replace(old, new, 1))
    with pytest.raises(RuntimeError) as err:
        Simulation(str(p)).setup()
    assert message.lower() in str(err.value).lower()


def test_fits_layout(tmp_path):
    sim = Simulation(ski("cfg1.ski"), num_packets=10).setup()
    frames = np.zeros(sim.frame_size)
    lay = sim.layout(0)
    assert (lay.num_components, lay.npix, lay.num_lambda) == (3, 4096, 1)
    frames[lay.ifu_offset + 4096 + 5] = 1.0  # PrimaryDirect, pixel 5
    sim.write(frames, str(tmp_path))
    raw = open(tmp_path / "cfg1_i0_primarydirect.fits", "rb").read()
    assert len(raw) == 25920 and len(raw) % 2880 == 0
    cards = [raw[i:i + 80].decode() for i in range(0, 2880, 80)]
    assert cards[0] == "SIMPLE  =                    T / file does conform to FITS standard".ljust(80)
    assert cards[1].startswith("BITPIX  =                  -32")
    assert cards[16].startswith("CDELT1  =      8.057218994E-03")
    assert "BUNIT   = 'MJy/sr  '" in cards[13]
    data = np.frombuffer(raw[2880:2880 + 4 * 4096], dtype=">f4")
    assert np.count_nonzero(data) == 1 and data[5] > 0
    total = np.frombuffer(open(tmp_path / "cfg1_i0_total.fits", "rb").read()[2880:2880 + 4 * 4096], dtype=">f4")
    assert total[5] == data[5]


# ---------------------------------------------------------------- smoothed-particle import (config 4)

def _particle_ski(tmp_path, body, extra=""):
    """cfg4small.ski with its particle file replaced by `body` and extra ParticleMedium attributes"""
    text = open(ski("cfg4small.ski")).read().replace('filename="cfg4small_sph.txt"', 'filename="p.txt" ' + extra)
    text = text.replace('maxLevel="7"', 'maxLevel="3"')
    (tmp_path / "p.txt").write_text(body)
    path = tmp_path / "p.ski"
    path.write_text(text)
    return str(path)


def _densities(sim):
    head = scene_head(sim)
    return np.ctypeslib.as_array(head.medium.number_density, shape=(head.grid.num_cells,)).copy()


def test_particle_file_units_and_defaults(tmp_path):
    """the unit header of a column text file is honoured (TextInFile.cpp:16-47), and without a header the columns carry
    the default units of their role (Snapshot.cpp:62-85,130-134: pc, pc, pc, pc, Msun)"""
    rows = [(100.0, -200.0, 50.0, 3000.0, 2.0), (-4000.0, 1000.0, -100.0, 2500.0, 1.0), (0.0, 0.0, 0.0, 1000.0, 0.5)]
    plain = "".join("%g %g %g %g %g\n" % r for r in rows)
    kpc = ("# Column 1: position x (kpc)\n# column 2 : position y (kpc)\n#Column 3: position z (kpc)\n"
           "# a comment line\n# Column 4: size h (kpc)\n# Column 5: mass (kg)\n"
           + "".join("%.17g %.17g %.17g %.17g %.17g\n" % (r[0] / 1e3, r[1] / 1e3, r[2] / 1e3, r[3] / 1e3, r[4] * 1.9891e30)
                     for r in rows))
    d0 = _densities(Simulation(_particle_ski(tmp_path, plain)).setup())
    d1 = _densities(Simulation(_particle_ski(tmp_path, kpc)).setup())
    assert d0.max() > 0
    assert np.allclose(d0, d1, rtol=1e-12, atol=0)


def test_particle_mass_policy(tmp_path):
    """massFraction scales, metallicity multiplies and the temperature cut-off drops particles
    (ImportedMedium.cpp:44-52, ParticleSnapshot.cpp:57-69,92-108); zero-mass particles are ignored"""
    base = "0 0 0 4000 1\n3000 0 0 4000 1\n"
    d = _densities(Simulation(_particle_ski(tmp_path, base)).setup())
    path = _particle_ski(tmp_path, base)
    text = open(path).read().replace('massFraction="1"', 'massFraction="0.5"')
    open(path, "w").write(text)
    half = _densities(Simulation(path).setup())
    assert d.max() > 0 and np.array_equal(half, 0.5 * d)
    text = open(ski("cfg4small.ski")).read().replace('filename="cfg4small_sph.txt"', 'filename="p.txt"')
    text = text.replace('maxLevel="7"', 'maxLevel="3"')
    (tmp_path / "p.txt").write_text("0 0 0 4000 1 0.5 100\n3000 0 0 4000 1 0.5 1e6\n0 0 0 4000 0 1 100\n")
    t2 = text.replace('importMetallicity="false"', 'importMetallicity="true"').replace(
        'importTemperature="false"', 'importTemperature="true"').replace('maxTemperature="0 K"', 'maxTemperature="1e4 K"')
    (tmp_path / "q.ski").write_text(t2)
    dq = _densities(Simulation(str(tmp_path / "q.ski")).setup())
    (tmp_path / "p.txt").write_text("0 0 0 4000 0.5\n")
    (tmp_path / "r.ski").write_text(text)
    dr = _densities(Simulation(str(tmp_path / "r.ski")).setup())
    assert dq.max() > 0 and np.allclose(dq, dr, rtol=1e-12, atol=0)


@pytest.mark.parametrize("body,message", [
    ("1 2 3 4\n", "missing"),
    ("1 2 x 4 5\n", "floating point"),
    ("# Column 1: position x (furlong)\n# Column 2: position y (pc)\n# Column 3: position z (pc)\n"
     "# Column 4: size h (pc)\n# Column 5: mass (Msun)\n1 2 3 4 5\n", "Invalid units"),
    ("# Column 2: position x (pc)\n1 2 3 4 5\n", "Incorrect column index"),
])
def test_particle_file_errors(tmp_path, body, message):
    with pytest.raises(Exception, match=message):
        Simulation(_particle_ski(tmp_path, body)).setup()


def test_missing_particle_file_fails_loudly(tmp_path):
    text = open(ski("cfg4small.ski")).read().replace('filename="cfg4small_sph.txt"', 'filename="nope.txt"')
    (tmp_path / "m.ski").write_text(text)
    with pytest.raises(Exception, match="Could not open"):
        Simulation(str(tmp_path / "m.ski")).setup()


def test_mean_file_dust_mix_equals_listed_values(tmp_path):
    """MeanFileDustMix (MeanFileDustMix.cpp:11-22): the same four columns in a text file -- here in descending wavelength
    order and cgs units -- give the dust tables that MeanListDustMix builds from the equivalent attribute lists"""
    sim_file = Simulation(ski("cfg1file.ski")).setup()
    text = open(ski("cfg1file.ski")).read().replace(
        '<MeanFileDustMix filename="cfg1file_dust.txt"/>',
        '<MeanListDustMix wavelengths="0.1 micron, 0.55 micron, 2.2 micron, 10 micron" '
        'extinctionCoefficients="9000 m2/kg, 3000 m2/kg, 600 m2/kg, 60 m2/kg" albedos="0.4, 0.6, 0.45, 0.1" '
        'asymmetryParameters="0.6, 0.5, 0.25, 0.05"/>')
    (tmp_path / "l.ski").write_text(text)
    sim_list = Simulation(str(tmp_path / "l.ski")).setup()
    a, b = scene_head(sim_file).medium, scene_head(sim_list).medium
    assert a.num_lambda == b.num_lambda
    for field in ("lambda_border", "sigma_ext", "sigma_sca", "asymmpar"):
        x = np.ctypeslib.as_array(getattr(a, field), shape=(a.num_lambda,))
        y = np.ctypeslib.as_array(getattr(b, field), shape=(b.num_lambda,))
        assert np.allclose(x, y, rtol=1e-14, atol=0), field


# ---------------------------------------------------------------- Voronoi tessellation (config 5)

def _voronoi_sim(tmp_path, sites):
    """cfg5small.ski with its random sites replaced by the given ones (policy File)"""
    text = open(ski("cfg5small.ski")).read().replace('policy="Uniform" numSites="1500"', 'policy="File" filename="sites.txt"')
    (tmp_path / "sites.txt").write_text("".join("%.17g %.17g %.17g\n" % tuple(s) for s in sites))
    (tmp_path / "v.ski").write_text(text)
    return Simulation(str(tmp_path / "v.ski")).setup()


def _voronoi_tables(sim):
    g = scene_head(sim).grid
    n = g.num_cells
    start = np.ctypeslib.as_array(g.vnbr_start, shape=(n + 1,)).copy()
    nbr = np.ctypeslib.as_array(g.vnbr_list, shape=(start[n],)).copy()
    site = np.ctypeslib.as_array(g.site, shape=(n, 3)).copy()
    return n, start, nbr, site


@pytest.mark.parametrize("kind", ["random", "lattice", "clustered"])
def test_voronoi_tessellation_is_consistent(tmp_path, kind):
    """the host layer's own Voronoi construction (skirt9_amd/host/voronoi.cpp): sites come out in the reference's order
    (sorted by x, VoronoiMeshSnapshot.cpp:509); neighbour relations are symmetric; every wall of the domain is touched;
    a cell's neighbours are exactly the sites that a dense set of its bisector mid-points confirms -- also for a regular
    lattice (maximally degenerate: eight cells meet in every vertex) and for strongly clustered sites"""
    rng = np.random.default_rng(5)
    pc = 3.08567758e16
    half = np.array([20000.0, 20000.0, 4000.0])
    if kind == "random":
        sites = (rng.random((800, 3)) - 0.5) * 2 * half * 0.999
    elif kind == "lattice":
        ax = [np.linspace(-h, h, 9)[:-1] + h / 8 for h in half]
        sites = np.array([[x, y, z] for x in ax[0] for y in ax[1] for z in ax[2]])
    else:
        sites = np.concatenate([rng.normal(0, 300.0, (600, 3)), (rng.random((200, 3)) - 0.5) * 2 * half * 0.999])
        sites = sites[np.all(np.abs(sites) < half * 0.999, axis=1)]
    sim = _voronoi_sim(tmp_path, sites)
    n, start, nbr, site = _voronoi_tables(sim)
    assert n == len(sites)
    assert np.all(np.diff(site[:, 0]) >= 0)                      # the reference's order
    assert np.allclose(np.sort(site[:, 0]), np.sort(sites[:, 0] * pc), rtol=1e-15)
    lists = [set(nbr[start[m]:start[m + 1]].tolist()) for m in range(n)]
    assert all(len(l) >= 4 for l in lists)
    asym = sum(1 for m in range(n) for j in lists[m] if j >= 0 and m not in lists[j])
    assert asym == 0
    assert set(range(-6, 0)) <= set(nbr.tolist())
    interior = [m for m in range(n) if min(lists[m]) >= 0]
    assert len(interior) > 50
    if kind == "lattice":
        # the six face neighbours of an interior lattice cell (at one lattice spacing along an axis) must be present
        spacing = 2 * half / 8 * pc
        for m in interior:
            found = 0
            for j in lists[m]:
                d = np.abs(site[j] - site[m])
                if np.sum(d > 1e-6 * spacing.max()) == 1 and np.any(np.isclose(d, spacing, rtol=1e-9)):
                    found += 1
            assert found == 6, (m, found)
    else:
        # independent construction: the Delaunay neighbours (Qhull) of an interior cell are its Voronoi neighbours
        from scipy.spatial import Delaunay
        tri = Delaunay(site / pc)
        indptr, indices = tri.vertex_neighbor_vertices
        for m in interior:
            qhull = set(indices[indptr[m]:indptr[m + 1]].tolist())
            assert lists[m] <= qhull, (m, lists[m] - qhull)
            # Qhull may list neighbours through faces of vanishing area (co-spherical sites): at most a few
            assert len(qhull - lists[m]) <= 2, (m, qhull - lists[m])


def test_voronoi_sites_from_the_dust_density():
    """VoronoiMeshSpatialGrid policy DustDensity (its default; VoronoiMeshSpatialGrid.cpp:22-40,73-85): the sites are drawn
    from the medium's geometry with the simulation's random stream (ExpDiskGeometry::generatePosition with Lambert W,
    one extra deviate per site for the choice of the medium) -- the cells then carry the reference's sampled densities
    (to 1e-13: the bounding boxes of the cells, from which the sample positions are drawn, agree to rounding)"""
    sim = Simulation(ski("cfg5dd.ski")).setup()
    gold = np.load(golden("cfg5dd_cells.npz"))
    head = scene_head(sim)
    assert head.grid.kind == 3 and head.grid.num_cells == len(gold["density"]) == 1500
    dens = np.ctypeslib.as_array(head.medium.number_density, shape=(1500,))
    assert np.allclose(dens, gold["density"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("name,cells", [("cfg5peak", 1500), ("cfg5imp", 3000)])
def test_voronoi_site_policies(name, cells):
    """VoronoiMeshSpatialGrid policies CentralPeak (VoronoiMeshSpatialGrid.cpp:67-82: sites from the simulation's random stream, 1/r
    towards the origin) and ImportedSites (:127-132: the positions of the imported medium's entities, ImportedMedium.cpp:268-278): the
    same sites as the reference -- its cells carry the reference's volumes and sampled densities (to rounding: own tessellation)"""
    sim = Simulation(ski(name + ".ski")).setup()
    gold = np.load(golden(name + "_cells.npz"))
    head = scene_head(sim)
    assert head.grid.kind == 3 and head.grid.num_cells == len(gold["density"]) == cells
    dens = np.ctypeslib.as_array(head.medium.number_density, shape=(cells,))
    assert np.allclose(dens, gold["density"], rtol=1e-12, atol=0)


def test_voronoi_sites_from_a_flattened_gaussian():
    """policy DustDensity with a SpheroidalGeometryDecorator of a GaussianGeometry as the dust distribution: the sites follow
    GaussianGeometry::randomRadius (a 401-point table of the cumulative mass through Random::cdfLinLin, GaussianGeometry.cpp:17-50) and
    SpheroidalGeometryDecorator::generatePosition, in the reference's order of random draws"""
    sim = Simulation(ski("cfg5ddgauss.ski")).setup()
    gold = np.load(golden("cfg5ddgauss_cells.npz"))
    head = scene_head(sim)
    assert head.grid.kind == 3 and head.grid.num_cells == len(gold["density"]) == 1500
    assert np.allclose(np.ctypeslib.as_array(head.medium.number_density, shape=(1500,)), gold["density"], rtol=1e-12, atol=0)


def test_voronoi_relaxed_sites():
    """VoronoiMeshSpatialGrid relaxSites="true" (VoronoiMeshSnapshot.cpp:550-601): one relaxation step -- every site moves to the centroid of
    its cell in the tessellation of the sites as drawn -- before the final tessellation.  The reference takes the centroid from Voro++; the
    host layer sums the same tetrahedra over the faces of its own polyhedron, so the relaxed sites agree to rounding, NOT bit for bit: the
    cells carry the reference's densities to 1e-12, and the reference's 48 dumped rays run through the same cells with segment lengths that
    agree to 1e-11 (measured: 2e-13) -- parity to rounding, stated as such."""
    import oracle_lib as O
    sim = Simulation(ski("cfg5relax.ski")).setup()
    gold = np.load(golden("cfg5relax_cells.npz"))
    head = scene_head(sim)
    assert head.grid.kind == 3 and head.grid.num_cells == len(gold["density"]) == 1500
    dens = np.ctypeslib.as_array(head.medium.number_density, shape=(1500,))
    assert np.allclose(dens, gold["density"], rtol=1e-12, atol=0)
    plain = scene_head(Simulation(ski("cfg5small.ski")).setup())
    assert not np.allclose(np.ctypeslib.as_array(plain.medium.number_density, shape=(1500,)), dens, rtol=1e-3)  # (the sites have moved)
    rays = [[float.fromhex(t) for t in line.split()] for line in open(golden("cfg5relax_rays.txt"))]
    ref = open(golden("cfg5relax_rays_ref.txt")).read().split("\n")
    pos = total = 0
    for i, ray in enumerate(rays):
        h = ref[pos].split()
        n = int(h[2])
        k = np.array([float.fromhex(v) for v in h[3:6]])
        m_ref = np.array([int(ref[pos + 1 + j].split()[0]) for j in range(n)], dtype=np.int32)
        ds_ref = np.array([float.fromhex(ref[pos + 1 + j].split()[1]) for j in range(n)])
        pos += 1 + n
        m, ds = O.trace_ray(sim, ray[:3], k)
        assert len(m) == n and np.array_equal(m, m_ref), i
        assert np.allclose(ds, ds_ref, rtol=1e-11, atol=0), i
        total += n
    assert total > 300


@pytest.mark.parametrize("name", ["cfg2shell", "cfg2torus", "cfg2ring", "cfg2gauss"])
def test_more_medium_geometries_bit_exact(name):
    """ShellGeometry, TorusGeometry, RingGeometry and GaussianGeometry as the dust distribution (density, column density for the optical
    depth normalisation): the octree built by DensityTreePolicy has the reference's cells, and every cell the
    reference's volume and sampled density, bit for bit"""
    sim = Simulation(ski(name + ".ski")).setup()
    gold = np.load(golden(name + "_cells.npz"))
    head = scene_head(sim)
    n = head.grid.num_cells
    assert head.grid.kind == 2 and n == len(gold["density"])
    dens = np.ctypeslib.as_array(head.medium.number_density, shape=(n,))
    assert np.count_nonzero(dens) > n // 10
    assert np.array_equal(dens.view(np.uint64), gold["density"].view(np.uint64))


def test_list_mesh_bit_exact():
    """ListMesh (TabulatedMesh.cpp:12-33: sorted unique points, zero inserted, scaled by the last point) along z of a
    Cartesian grid: the reference's cell volumes and sampled densities, bit for bit"""
    sim = Simulation(ski("cfg1list.ski")).setup()
    gold = np.load(golden("cfg1list_cells.npz"))
    head = scene_head(sim)
    n = head.grid.num_cells
    assert head.grid.kind == 1 and n == 30 * 25 * 10 == len(gold["density"])
    zv = np.ctypeslib.as_array(head.grid.zv, shape=(11,))
    vol = gold["volume"].reshape(30, 25, 10)
    assert np.allclose(np.diff(zv) / np.diff(zv)[0], vol[0, 0, :] / vol[0, 0, 0], rtol=1e-12)
    dens = np.ctypeslib.as_array(head.medium.number_density, shape=(n,))
    assert np.array_equal(dens.view(np.uint64), gold["density"].view(np.uint64))


# ---------------------------------------------------------------- tabulated source spectra

class Options(C.Structure):
    """pmc_options (include/pmc.h)"""
    _fields_ = [("force_scattering", C.c_int32), ("min_weight_reduction", C.c_double), ("min_scatt_events", C.c_int32),
                ("path_length_bias", C.c_double), ("explicit_absorption", C.c_int32)]


class SourceHead(C.Structure):
    """the leading members of pmc_source, which follows pmc_medium and pmc_options in pmc_scene (include/pmc.h)"""
    _fields_ = [("kind", C.c_int32), ("position", C.c_double * 3), ("reff", C.c_double), ("sersic_n", C.c_int32),
                ("sersic_s", C.POINTER(C.c_double)), ("sersic_M", C.POINTER(C.c_double)), ("box", C.c_double * 6),
                ("packet_luminosity", C.c_double), ("lambda_mode", C.c_int32), ("num_oligo", C.c_int32),
                ("oligo_lambda", C.POINTER(C.c_double)), ("oligo_weight", C.POINTER(C.c_double)), ("lambda_bias", C.c_double),
                ("num_sed", C.c_int32), ("sed_lambda", C.POINTER(C.c_double)), ("sed_p", C.POINTER(C.c_double)),
                ("sed_P", C.POINTER(C.c_double))]


class SceneWithSource(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("grid", Grid), ("medium", Medium), ("options", Options), ("src", SourceHead)]


def _sed_tables(sim):
    src = SceneWithSource.from_address(sim.scene).src
    n = src.num_sed
    return (np.ctypeslib.as_array(src.sed_lambda, shape=(n,)).copy(), np.ctypeslib.as_array(src.sed_p, shape=(n,)).copy(),
            np.ctypeslib.as_array(src.sed_P, shape=(n,)).copy())


def test_list_sed_equals_file_sed(tmp_path):
    """ListSED (ListSED.cpp:11-18) and FileSED (FileSED.cpp:11-18) with the same numbers give the same sampling tables
    (TabulatedSED.cpp:14-21: table cut to the source range with interpolated end points, log-log cumulative
    distribution); the table is normalised to 1 and ends at 1"""
    sim_file = Simulation(ski("cfg3sed.ski")).setup()
    text = open(ski("cfg3sed.ski")).read().replace(
        '<FileSED filename="cfg3sed_sed.txt"/>',
        '<ListSED unitStyle="wavelengthmonluminosity" wavelengths="0.09 micron, 0.15 micron, 0.3 micron, 0.55 micron, 1 micron, '
        '2.2 micron, 5 micron, 20 micron" specificLuminosities="0.1 W/micron, 1.5 W/micron, 9 W/micron, 19 W/micron, '
        '14 W/micron, 6 W/micron, 0.9 W/micron, 0.02 W/micron"/>')
    (tmp_path / "l.ski").write_text(text)
    sim_list = Simulation(str(tmp_path / "l.ski")).setup()
    a, b = _sed_tables(sim_file), _sed_tables(sim_list)
    assert len(a[0]) == len(b[0]) == 8      # 0.1 | 0.15 0.3 0.55 1 2.2 5 | 10 micron
    assert np.isclose(a[0][0], 0.1e-6, rtol=1e-14) and np.isclose(a[0][-1], 10e-6, rtol=1e-14)
    for x, y in zip(a, b):
        # (the list carries W/micron values, the file's specific column is not scaled: the normalised tables agree)
        assert np.allclose(x, y, rtol=1e-13, atol=0)
    assert a[2][0] == 0.0 and a[2][-1] == 1.0 and np.all(np.diff(a[2]) > 0)


def test_file_sed_rejects_frequency_style_units(tmp_path):
    text = open(ski("cfg3sed.ski")).read().replace('filename="cfg3sed_sed.txt"', 'filename="s.txt"')
    (tmp_path / "s.txt").write_text("# Column 1: wavelength (micron)\n# Column 2: specific luminosity (W/Hz)\n0.1 1\n10 1\n")
    (tmp_path / "s.ski").write_text(text)
    with pytest.raises(Exception, match="only per-wavelength units"):
        Simulation(str(tmp_path / "s.ski")).setup()
