"""GPU tests at BASELINE.json's full grid size (configs[1]: the 953 688-cell octree of tests/ski/cfg2.ski), where the
scalar oracle is too slow for a photon-loop comparison: size-independent properties of the detector arrays, plus the
bit-exact traversal check (which the oracle does finish in seconds at any grid size).

* traversal: (m, ds) of 300 fixed rays bit-exact against the oracle on the full octree;
* conservation: every history reaches the SED of the FullInstrument exactly once per emission, so the number of
  contributing histories (sum of w^0 over the wavelength bins) equals N, and the transparent component of the SED sums
  to N times the packet luminosity (to summation order, 1e-12);
* ordering: extinction only removes flux: PrimaryDirect <= Transparent in every pixel;
* linearity / partition independence: two half segments accumulate to the arrays of one whole segment (1e-10);
* the counted work per history is in the range the reference's probe gave (V about 480-580).
"""
import numpy as np
import pytest

import oracle_lib as O
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
