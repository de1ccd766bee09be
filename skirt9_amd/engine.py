"""ctypes binding of the HIP engine (libpmc.so, include/pmc.h).

The engine replaces the parallel section of ``MonteCarloSimulation::runPrimaryEmission``
(SKIRT/core/MonteCarloSimulation.cpp:126-129: ``parallel->call(Npp, performLifeCycle)`` followed by
``instrumentSystem()->flush()``).  There is NO CPU fallback: importing works anywhere (so that the build can be
checked on a machine without a GPU) but creating an ``Engine`` without a HIP device raises ``RuntimeError``.
"""
import ctypes as C
import os

import numpy as np

from .host import CounterValues, FrameLayout

_LIBDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")

# every symbol include/pmc.h and include/pmc_tuning.h declare
SYMBOLS = ["pmc_abi_version", "pmc_build_info", "pmc_last_error", "pmc_frame_layout_of", "pmc_create", "pmc_destroy", "pmc_bind_frames",
           "pmc_clear_frames", "pmc_run_primary", "pmc_set_progress", "pmc_sync", "pmc_download", "pmc_frames_device", "pmc_frames_size",
           "pmc_last_kernel_ms", "pmc_counters", "pmc_reset_counters", "pmc_trace_ray", "pmc_set_launch",
           "pmc_set_num_slots", "pmc_last_timing", "pmc_last_walk_timing", "pmc_walk_work", "pmc_radiation_field_size", "pmc_radiation_field_device",
           "pmc_download_radiation_field", "pmc_clear_radiation_field", "pmc_bind_radiation_field", "pmc_sampler_create",
           "pmc_sampler_density", "pmc_sampler_destroy", "pmc_history_range", "pmc_comm_init_all", "pmc_comm_unique_id",
           "pmc_comm_init_rank", "pmc_comm_size", "pmc_comm_destroy", "pmc_reduce_frames", "pmc_allreduce_radiation_field",
           "pmc_debug_tables", "pmc_tuning_set", "pmc_tuning_clear"]

_lib = None


class DebugTables(C.Structure):
    """pmc_debug_table_values (include/pmc_tuning.h): device addresses for profiles/microbench/bridge.hip"""
    _fields_ = [("cell_table", C.c_void_p), ("cell_slots", C.c_int64), ("loose_base", C.c_int64), ("task_cell", C.c_void_p), ("num_slots", C.c_int64)]


class WalkWork(C.Structure):
    """pmc_walk_work_values (include/pmc_tuning.h)"""
    _fields_ = [(n, C.c_uint64) for n in ("peel_wave_steps", "peel_lane_steps", "peel_rounds", "prop_wave_steps",
                                          "prop_lane_steps", "prop_rounds")]


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("PMC_LIBRARY") or os.path.join(_LIBDIR, "libpmc.so")  # override: kernel A/B experiments
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: the HIP engine has not been built (run `make` or "
                               "__graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(path)
        L.pmc_abi_version.restype = C.c_int
        L.pmc_last_error.restype = C.c_char_p
        L.pmc_build_info.restype = C.c_char_p
        L.pmc_frame_layout_of.restype = C.c_int64
        L.pmc_frame_layout_of.argtypes = [C.c_void_p, C.c_int32, C.POINTER(FrameLayout)]
        L.pmc_create.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        L.pmc_destroy.argtypes = [C.c_void_p]
        L.pmc_bind_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.pmc_clear_frames.argtypes = [C.c_void_p]
        L.pmc_run_primary.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        L.pmc_sync.argtypes = [C.c_void_p]
        L.pmc_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.pmc_frames_device.restype = C.c_void_p
        L.pmc_frames_device.argtypes = [C.c_void_p]
        L.pmc_frames_size.restype = C.c_int64
        L.pmc_frames_size.argtypes = [C.c_void_p]
        L.pmc_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.pmc_counters.argtypes = [C.c_void_p, C.POINTER(CounterValues)]
        L.pmc_reset_counters.argtypes = [C.c_void_p]
        L.pmc_trace_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.pmc_set_launch.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.pmc_set_num_slots.argtypes = [C.c_void_p, C.c_int64]
        L.pmc_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.POINTER(C.c_int32)]
        if hasattr(L, "pmc_walk_work"):
            # (absent only from engines built from an older commit and loaded through PMC_LIBRARY for kernel A/B experiments:
            # tools/sweep.py copes without these two; every other prototype below is still set)
            L.pmc_last_walk_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
            L.pmc_walk_work.argtypes = [C.c_void_p, C.POINTER(WalkWork)]
        L.pmc_radiation_field_size.restype = C.c_int64
        L.pmc_radiation_field_size.argtypes = [C.c_void_p]
        L.pmc_radiation_field_device.restype = C.c_void_p
        L.pmc_radiation_field_device.argtypes = [C.c_void_p]
        L.pmc_download_radiation_field.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.pmc_clear_radiation_field.argtypes = [C.c_void_p]
        L.pmc_bind_radiation_field.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.pmc_history_range.restype = None
        L.pmc_history_range.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.pmc_comm_init_all.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]
        L.pmc_comm_unique_id.argtypes = [C.c_void_p]
        L.pmc_comm_init_rank.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
        L.pmc_comm_size.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.pmc_comm_destroy.restype = None
        L.pmc_comm_destroy.argtypes = [C.c_void_p]
        L.pmc_reduce_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.pmc_allreduce_radiation_field.argtypes = [C.c_void_p, C.c_void_p]
        if hasattr(L, "pmc_tuning_set"):
            L.pmc_tuning_set.argtypes = [C.c_char_p, C.c_char_p]
            L.pmc_tuning_clear.restype = None
        _lib = L
    return _lib


# settings the library itself reads from the environment (include/pmc.h); every other PMC_* name is a tuning switch
ENVIRONMENT_SETTINGS = ("PMC_NUM_SLOTS", "PMC_NUM_GROUPS", "PMC_STAT_POOL_BLOCKS")


def set_tuning(name, value="1"):
    """a tuning switch of the engine (include/pmc_tuning.h pmc_tuning_set): alternative code paths for cross-checks and A/B
    timing; value None removes it.  (Engines built before the switch table existed read the same names from the environment.)"""
    L = lib()
    if hasattr(L, "pmc_tuning_set"):
        _check(L.pmc_tuning_set(name.encode(), None if value is None else str(value).encode()))
    elif value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)


def clear_tuning():
    L = lib()
    if hasattr(L, "pmc_tuning_clear"):
        L.pmc_tuning_clear()


def tuning_from_environment():
    """tools only (bench.py, tools/sweep.py): hands the PMC_* variables of the environment that are not library settings to the
    engine as tuning switches -- the library itself does not read them"""
    for key, value in os.environ.items():
        if key.startswith("PMC_") and key not in ENVIRONMENT_SETTINGS and key != "PMC_LIBRARY":
            set_tuning(key, value)


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"pmc error {rc}: {lib().pmc_last_error().decode()}")


def history_range(num_packets, rank, num_ranks):
    """[first, first + count) of rank `rank` of `num_ranks` for a segment of num_packets histories (pmc_history_range:
    the one implementation the CLI driver, the bench and the tests share; pure host code)"""
    first, count = C.c_uint64(0), C.c_uint64(0)
    lib().pmc_history_range(int(num_packets), int(rank), int(num_ranks), C.byref(first), C.byref(count))
    return int(first.value), int(count.value)


class Communicator:
    """RCCL communicator(s) behind the C ABI: ``Communicator.all(devices)`` for one process that drives several
    devices (one handle per device), ``Communicator.rank(device, num_ranks, rank, unique_id)`` for one process per
    device (``Communicator.unique_id()`` on rank 0, handed to the others by the launcher)."""

    def __init__(self, handles):
        self.handles = handles

    @classmethod
    def all(cls, devices):
        n = len(devices)
        dev = (C.c_int32 * n)(*devices)
        out = (C.c_void_p * n)()
        _check(lib().pmc_comm_init_all(n, dev, out))
        return cls([C.c_void_p(h) for h in out])

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().pmc_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def rank(cls, device, num_ranks, rank, unique_id):
        out = C.c_void_p()
        _check(lib().pmc_comm_init_rank(device, num_ranks, rank, C.c_char_p(unique_id), C.byref(out)))
        return cls([out])

    def size(self, index=0):
        """(ranks of the communicator, rank of handle `index` in it) as RCCL reports them"""
        n, me = C.c_int32(0), C.c_int32(0)
        _check(lib().pmc_comm_size(self.handles[index], C.byref(n), C.byref(me)))
        return int(n.value), int(me.value)

    def close(self):
        for h in self.handles:
            lib().pmc_comm_destroy(h)
        self.handles = []


class Engine:
    """One engine context on one MI355X: device copies of a scene plus the detector arrays."""

    def __init__(self, scene_ptr, device=0):
        """scene_ptr: address of a pmc_scene (e.g. ``skirt9_amd.host.Simulation.scene``)"""
        self._h = C.c_void_p()
        _check(lib().pmc_create(scene_ptr, device, C.byref(self._h)))
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().pmc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def frame_size(self):
        return int(lib().pmc_frames_size(self._h))

    @property
    def frames_device_ptr(self):
        return int(lib().pmc_frames_device(self._h))

    def bind_frames(self, device_ptr, num_doubles):
        """accumulate into caller-owned device memory (e.g. ``tensor.data_ptr()`` of a zeroed float64 torch tensor)"""
        _check(lib().pmc_bind_frames(self._h, C.c_void_p(device_ptr), num_doubles))

    def set_launch(self, block=0, grid=0):
        _check(lib().pmc_set_launch(self._h, block, grid))

    def clear(self):
        _check(lib().pmc_clear_frames(self._h))

    def run_primary(self, first, count, seed):
        """asynchronous launch of histories [first, first+count); accumulates into the frames"""
        _check(lib().pmc_run_primary(self._h, first, count, seed))

    def set_progress(self, report, interval_seconds=3.0):
        """report(launched, count) is called from run_primary at most once per interval (MonteCarloSimulation::logProgress); None: off"""
        proto = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64)
        self._progress = proto(lambda user, launched, count: report(int(launched), int(count))) if report else None  # (kept alive)
        lib().pmc_set_progress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        _check(lib().pmc_set_progress(self._h, C.cast(self._progress, C.c_void_p) if self._progress else None, None, float(interval_seconds)))

    def sync(self):
        _check(lib().pmc_sync(self._h))

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _check(lib().pmc_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def set_num_slots(self, n):
        _check(lib().pmc_set_num_slots(self._h, int(n)))

    def last_timing(self):
        """HIP-event times of the last segment: dict(total_ms, walk_ms, transition_ms, generations)"""
        t, w, x, g = C.c_float(0), C.c_float(0), C.c_float(0), C.c_int32(0)
        _check(lib().pmc_last_timing(self._h, C.byref(t), C.byref(w), C.byref(x), C.byref(g)))
        return {"total_ms": float(t.value), "walk_ms": float(w.value), "transition_ms": float(x.value),
                "generations": int(g.value)}

    def last_walk_timing(self):
        """octree: HIP-event spans of the peel-off kernels and of the propagation kernel of the last segment (they overlap)"""
        a, b = C.c_float(0), C.c_float(0)
        _check(lib().pmc_last_walk_timing(self._h, C.byref(a), C.byref(b)))
        return {"peel_ms": float(a.value), "prop_ms": float(b.value)}

    def walk_work(self):
        """counted wave-steps, lane-steps and bookkeeping rounds of the octree walk kernels since create / reset"""
        w = WalkWork()
        _check(lib().pmc_walk_work(self._h, C.byref(w)))
        return {n: int(getattr(w, n)) for n, _ in w._fields_}

    def debug_tables(self):
        """device addresses of the octree walk's hot table and of the task records' first cells (tuning aid)"""
        t = DebugTables()
        lib().pmc_debug_tables.argtypes = [C.c_void_p, C.POINTER(DebugTables)]
        _check(lib().pmc_debug_tables(self._h, C.byref(t)))
        return t

    def reduce_frames(self, comm_handle, root=0):
        """end of a segment on several devices: ONE ncclReduce (f64, sum) of the detector arrays onto `root`
        (ProcessManager::sumToRoot behind FluxRecorder::flush); the other ranks' arrays are cleared"""
        _check(lib().pmc_reduce_frames(self._h, comm_handle, root))

    def allreduce_radiation_field(self, comm_handle):
        """ncclAllReduce of the radiation field table (MediumSystem::communicateRadiationField)"""
        _check(lib().pmc_allreduce_radiation_field(self._h, comm_handle))

    def download(self):
        out = np.empty(self.frame_size, dtype=np.float64)
        _check(lib().pmc_download(self._h, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    @property
    def radiation_field_size(self):
        return int(lib().pmc_radiation_field_size(self._h))

    @property
    def radiation_field_device_ptr(self):
        """device address of the radiation field table (for an RCCL all-reduce across ranks)"""
        return int(lib().pmc_radiation_field_device(self._h) or 0)

    def bind_radiation_field(self, device_ptr, num_doubles):
        """accumulate the radiation field into caller-owned device memory (a zeroed float64 torch tensor), so that
        ``Engine.allreduce_radiation_field`` / pmc_allreduce_radiation_field sums it over the ranks"""
        _check(lib().pmc_bind_radiation_field(self._h, C.c_void_p(device_ptr), num_doubles))

    def download_radiation_field(self):
        """rf[m * nbins + ell] accumulated by the segments run so far (MediumSystem::_rf1)"""
        out = np.empty(self.radiation_field_size, dtype=np.float64)
        _check(lib().pmc_download_radiation_field(self._h, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def clear_radiation_field(self):
        _check(lib().pmc_clear_radiation_field(self._h))

    def counters(self):
        c = CounterValues()
        _check(lib().pmc_counters(self._h, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        _check(lib().pmc_reset_counters(self._h))

    def trace_ray(self, r, k, cap=4096):
        r = np.ascontiguousarray(r, dtype=np.float64)
        k = np.ascontiguousarray(k, dtype=np.float64)
        m = np.zeros(cap, dtype=np.int32)
        ds = np.zeros(cap, dtype=np.float64)
        n = C.c_int32(0)
        _check(lib().pmc_trace_ray(self._h, r.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p),
                                   m.ctypes.data_as(C.c_void_p), ds.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return m[:n.value].copy(), ds[:n.value].copy()
