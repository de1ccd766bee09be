"""ctypes binding of the CPU test oracle (oracle/_build/liboracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "oracle"], cwd=ROOT)
        L = C.CDLL(path)
        L.oracle_run_primary.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64,
                                         C.c_void_p, C.c_void_p]
        L.oracle_run_primary_rf.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_trace_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_void_p]
        _lib = L
    return _lib


RNG_MT19937 = 0   # the reference's single-thread stream, continued after setup
RNG_PHILOX = 1    # the engine's per-history streams


def run_primary(sim, first, count, rng_kind, seed=None, skip_draws=None, frames=None):
    """runs histories [first, first+count) of a set-up skirt9_amd.host.Simulation; returns (frames, counters)"""
    from skirt9_amd.host import CounterValues
    if frames is None:
        frames = np.zeros(sim.frame_size, dtype=np.float64)
    if seed is None:
        seed = sim.seed
    if skip_draws is None:
        skip_draws = sim.setup_draws if rng_kind == RNG_MT19937 else 0
    counters = CounterValues()
    rc = lib().oracle_run_primary(sim.scene, first, count, rng_kind, seed, skip_draws,
                                  frames.ctypes.data_as(C.c_void_p), C.byref(counters))
    assert rc == 0
    return frames, counters


def run_primary_rf(sim, first, count, rng_kind, seed=None, skip_draws=None):
    """like run_primary for a simulation that stores the radiation field; returns (frames, rf, counters)"""
    from skirt9_amd.host import CounterValues
    frames = np.zeros(sim.frame_size, dtype=np.float64)
    rf = np.zeros(sim.radiation_field_size, dtype=np.float64)
    if seed is None:
        seed = sim.seed
    if skip_draws is None:
        skip_draws = sim.setup_draws if rng_kind == RNG_MT19937 else 0
    counters = CounterValues()
    rc = lib().oracle_run_primary_rf(sim.scene, first, count, rng_kind, seed, skip_draws, frames.ctypes.data_as(C.c_void_p),
                                     rf.ctypes.data_as(C.c_void_p), C.byref(counters))
    assert rc == 0
    return frames, rf, counters


def trace_ray(sim, r, k, cap=4096):
    r = np.asarray(r, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    m = np.zeros(cap, dtype=np.int32)
    ds = np.zeros(cap, dtype=np.float64)
    n = C.c_int32(0)
    rc = lib().oracle_trace_ray(sim.scene, r.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p),
                                m.ctypes.data_as(C.c_void_p), ds.ctypes.data_as(C.c_void_p), cap, C.byref(n))
    assert rc == 0
    return m[:n.value].copy(), ds[:n.value].copy()
