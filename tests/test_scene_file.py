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


def test_a_damaged_scene_file_is_refused(tmp_path):
    """the loader trusts nothing in the file (skirt9_amd/host/scenefile.cpp): a truncated file, a flipped byte in the tables, and a
    planted count or offset with a matching checksum are all refused before an offset becomes a pointer"""
    import struct
    sim = Simulation(ski("cfg2small.ski"), num_packets=100).setup()
    path = tmp_path / "scene.bin"
    sim.save_scene(str(path))
    good = path.read_bytes()
    SceneFile(str(path))
    (tmp_path / "short.bin").write_bytes(good[:len(good) // 2])
    with pytest.raises(RuntimeError, match="not a scene file"):
        SceneFile(str(tmp_path / "short.bin"))
    flipped = bytearray(good)
    flipped[len(good) // 2] ^= 0x40
    (tmp_path / "flipped.bin").write_bytes(bytes(flipped))
    with pytest.raises(RuntimeError, match="checksum"):
        SceneFile(str(tmp_path / "flipped.bin"))
    # a writer that knows the checksum: offsets and counts are still held against the file size
    header = struct.Struct("<QiiQQqqiiQQQIIIIQ")
    fields = list(header.unpack_from(good))
    assert fields[11] == len(good)

    def resealed(data):
        body = bytes(data[header.size:])
        h = [(0xcbf29ce484222325 + k) & 0xFFFFFFFFFFFFFFFF for k in range(8)]
        words = np.frombuffer(body[:len(body) // 64 * 64], dtype="<u8").reshape(-1, 8)
        for row in words.tolist():
            for k in range(8):
                h[k] = ((h[k] ^ row[k]) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        r = 0xcbf29ce484222325
        for k in range(8):
            r = ((r ^ h[k]) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        for b in body[len(body) // 64 * 64:]:
            r = ((r ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        f = list(header.unpack_from(data))
        f[16] = r
        header.pack_into(data, 0, *f)
        return bytes(data)

    assert resealed(bytearray(good)) == good          # (the test's checksum is the library's)
    planted = bytearray(good)
    scene_at = fields[9]
    # the pointer members of the pmc_scene in the file are offsets: the first plausible one is sent to the last 8 bytes of the file
    words = np.frombuffer(bytes(planted[scene_at:scene_at + 512]), dtype="<u8")
    inside = [i for i, w in enumerate(words.tolist()) if header.size <= w < len(good) and w % 16 == 0]
    assert inside, "no offset word found in the scene structure"
    struct.pack_into("<Q", planted, scene_at + 8 * inside[0], len(good) - 8)
    (tmp_path / "planted.bin").write_bytes(resealed(planted))
    with pytest.raises(RuntimeError, match="outside the file"):
        SceneFile(str(tmp_path / "planted.bin"))
    sim.close()
