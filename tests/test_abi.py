"""The C-ABI library of the HIP engine: it must load on a machine without a GPU, export every symbol that
include/pmc.h declares, answer the pure host helpers, and FAIL LOUDLY (no CPU fallback) when asked to compute
without a device."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT, ski


@pytest.fixture(scope="module")
def libpmc():
    path = os.path.join(ROOT, "skirt9_amd", "lib", "libpmc.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "skirt9_amd/lib/libpmc.so"], cwd=ROOT)  # hipcc cross-compiles gfx950
    from skirt9_amd import engine
    return engine.lib()


def test_every_declared_symbol_is_exported(libpmc):
    """the drop-in boundary (include/pmc.h) and the tuning aids (include/pmc_tuning.h: a header of its own, nothing a caller needs)"""
    boundary = open(os.path.join(ROOT, "include", "pmc.h")).read()
    tuning = open(os.path.join(ROOT, "include", "pmc_tuning.h")).read()
    declared = sorted(set(re.findall(r"\b(pmc_[a-z_]+)\s*\(", boundary + tuning)))
    from skirt9_amd import engine
    assert sorted(engine.SYMBOLS) == declared
    for name in declared:
        assert hasattr(libpmc, name), name
    assert libpmc.pmc_abi_version() == 9
    # the boundary header declares no tuning entry
    for name in ("pmc_tuning_set", "pmc_set_launch", "pmc_debug_tables", "pmc_walk_work"):
        assert name not in boundary and name in tuning


def test_build_info_names_the_flags_that_decide_the_results(libpmc):
    """the binary says what it was built with; the flags bit-compatibility rests on must be among them (pmc_create also probes the
    contraction on the device and refuses a binary that fuses a * b + c: every -m gpu test passes through that probe)"""
    info = libpmc.pmc_build_info().decode()
    assert "-ffp-contract=off" in info and "-munsafe-fp-atomics" in info and "gfx950" in info and "ABI 9" in info


def test_the_library_reads_three_environment_settings_only():
    """experiment switches go through pmc_tuning_set (include/pmc_tuning.h), not through the environment of the process"""
    names = set()
    for f in os.listdir(os.path.join(ROOT, "skirt9_amd", "csrc")):
        names |= set(re.findall(r'getenv\("([A-Z_0-9]+)"\)', open(os.path.join(ROOT, "skirt9_amd", "csrc", f)).read()))
    assert names == {"PMC_NUM_SLOTS", "PMC_NUM_GROUPS", "PMC_STAT_POOL_BLOCKS"}


def test_tuning_switch_table(libpmc):
    from skirt9_amd import engine
    engine.set_tuning("PMC_NO_LIVE_LISTS", "1")
    libpmc.pmcTune.restype = C.c_char_p
    assert libpmc.pmcTune(b"PMC_NO_LIVE_LISTS") == b"1"
    engine.set_tuning("PMC_NO_LIVE_LISTS", None)
    assert libpmc.pmcTune(b"PMC_NO_LIVE_LISTS") is None
    engine.set_tuning("PMC_RF_LOG_PER_SLOT", 7)
    engine.clear_tuning()
    assert libpmc.pmcTune(b"PMC_RF_LOG_PER_SLOT") is None


def test_history_range_is_the_library_function(libpmc):
    """skirt9_amd.engine.history_range, the CLI driver and the bench share pmc_history_range (pure host code)"""
    from skirt9_amd.engine import history_range
    for n, world in ((10 ** 9, 8), (7, 3), (0, 4), (2 ** 63 + 12345, 8)):
        cuts = [history_range(n, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and sum(c for _, c in cuts) == n
        for r in range(world):
            assert cuts[r][0] == (r * n) // world and cuts[r][1] == ((r + 1) * n) // world - (r * n) // world


def test_host_library_exports(libpmc):
    header = open(os.path.join(ROOT, "include", "skirt_host.h")).read()
    from skirt9_amd import host
    L = host.lib()
    for name in sorted(set(re.findall(r"\b(skh_[a-z_]+)\s*\(", header))):
        assert hasattr(L, name), name


def test_frame_layout_helper_matches_host(libpmc):
    from skirt9_amd.host import FrameLayout, Simulation
    sim = Simulation(ski("cfg1.ski"), num_packets=10).setup()
    out = FrameLayout()
    total = libpmc.pmc_frame_layout_of(sim.scene, 0, C.byref(out))
    assert total == sim.frame_size == 3 * 1 + 3 * 4096 + 5 * 1 + 5 * 4096
    ref = sim.layout(0)
    for field, _ in FrameLayout._fields_:
        assert getattr(out, field) == getattr(ref, field)


def test_no_cpu_fallback(libpmc):
    """without a HIP device pmc_create must return an error and say why; nothing computes on the CPU"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from skirt9_amd.engine import Engine
    from skirt9_amd.host import Simulation
    sim = Simulation(ski("cfg1.ski"), num_packets=10).setup()
    with pytest.raises(RuntimeError) as err:
        Engine(sim.scene, 0)
    assert "no hip device" in str(err.value).lower() or "pmc error" in str(err.value).lower()


def test_product_does_not_reference_the_oracle():
    """the oracle is test infrastructure: nothing under skirt9_amd/ or include/ may import, link or open it"""
    for base in ("skirt9_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle_lib" not in text and "liboracle" not in text and "oracle/" not in text.replace("test oracle", ""), f
    mk = open(os.path.join(ROOT, "Makefile")).read()
    rule = mk[mk.index("skirt9_amd/lib/skirt_mi355x:"):mk.index("# ---- test oracle")]
    assert "oracle" not in rule
