"""Host model layer (libskirthost.so): ski parsing, units, grids, densities, tables, FITS writer -- against values
dumped from the reference (tests/golden/*_cells.npz) and known answers."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, golden, ski
from skirt9_amd.host import Simulation, lib


class Grid(C.Structure):
    _fields_ = [("kind", C.c_int32), ("xmin", C.c_double), ("ymin", C.c_double), ("zmin", C.c_double),
                ("xmax", C.c_double), ("ymax", C.c_double), ("zmax", C.c_double), ("eps", C.c_double),
                ("num_cells", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("xv", C.POINTER(C.c_double)), ("yv", C.POINTER(C.c_double)), ("zv", C.POINTER(C.c_double)),
                ("num_nodes", C.c_int32), ("node_box", C.POINTER(C.c_double)), ("node_level", C.POINTER(C.c_int32)),
                ("node_first_child", C.POINTER(C.c_int32)), ("node_cell", C.POINTER(C.c_int32)),
                ("nbr_start", C.POINTER(C.c_int32)), ("nbr_list", C.POINTER(C.c_int32))]


class Medium(C.Structure):
    _fields_ = [("number_density", C.POINTER(C.c_double)), ("num_lambda", C.c_int32),
                ("lambda_border", C.POINTER(C.c_double)), ("sigma_ext", C.POINTER(C.c_double)),
                ("sigma_sca", C.POINTER(C.c_double)), ("asymmpar", C.POINTER(C.c_double))]


class SceneHead(C.Structure):
    """leading members of pmc_scene (include/pmc.h)"""
    _fields_ = [("abi_version", C.c_int32), ("grid", Grid), ("medium", Medium)]


def scene_head(sim):
    return SceneHead.from_address(sim.scene)


@pytest.mark.parametrize("name,cells,nodes", [("cfg1", 32768, 0), ("cfg2small", 17592, 20105)])
def test_cell_densities_bit_exact(name, cells, nodes):
    sim = Simulation(ski(name + ".ski")).setup()
    head = scene_head(sim)
    assert head.abi_version == 2
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
    ('<LinMesh numBins="32"/></meshX>', '<PowMesh numBins="32" ratio="2"/></meshX>', "PowMesh"),
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
