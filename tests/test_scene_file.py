"""A set-up scene saved to one file and loaded again (include/skirt_host.h skh_scene_save / skh_scene_load): the ranks of a
multi-GPU job load the scene that one of them set up.  The loaded scene must drive the photon loop to the same result as the
live one -- checked with the CPU oracle, bit for bit, on every grid kind and on a system of several sources."""
import numpy as np
import pytest

import oracle_lib as O
from conftest import ski
from skirt9_amd.host import SceneFile, Simulation, scene_head


@pytest.mark.parametrize("name", ["cfg1.ski", "cfg2small.ski", "cfg3multi.ski", "cfg5small.ski", "cfg3rf.ski", "cfg1sed.ski"])
def test_loaded_scene_equals_the_live_one(name, tmp_path):
    n = 300
    sim = Simulation(ski(name), num_packets=n).setup()
    path = str(tmp_path / "scene.bin")
    sim.save_scene(path)
    loaded = SceneFile(path)
    assert loaded.seed == sim.seed and loaded.num_packets == n and loaded.frame_size == sim.frame_size
    assert loaded.radiation_field_size == sim.radiation_field_size and loaded.setup_draws == sim.setup_draws
    a, b = sim.layout(0), loaded.layout(0)
    assert [getattr(a, f) for f, _ in a._fields_] == [getattr(b, f) for f, _ in b._fields_]
    ga, gb = scene_head(sim).grid, scene_head(loaded).grid
    assert ga.kind == gb.kind and ga.num_cells == gb.num_cells
    if sim.radiation_field_size:
        x, rfx, _ = O.run_primary_rf(sim, 0, n, O.RNG_PHILOX, seed=7)
        y, rfy, _ = O.run_primary_rf(loaded, 0, n, O.RNG_PHILOX, seed=7)
        assert np.array_equal(rfx, rfy)
    else:
        x, _ = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=7)
        y, _ = O.run_primary(loaded, 0, n, O.RNG_PHILOX, seed=7)
    assert x.sum() > 0 and np.array_equal(x, y)
    # the file is self-contained: the simulation it came from may be gone
    sim.close()
    z, _ = (O.run_primary_rf(loaded, 0, n, O.RNG_PHILOX, seed=7)[0::2] if loaded.radiation_field_size else O.run_primary(loaded, 0, n, O.RNG_PHILOX, seed=7))
    assert np.array_equal(x, z)


def test_a_file_that_is_not_a_scene_is_refused(tmp_path):
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"\0" * 4096)
    with pytest.raises(RuntimeError, match="not a scene file"):
        SceneFile(str(bad))
