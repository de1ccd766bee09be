"""ctypes binding of the host model layer (libskirthost.so, include/skirt_host.h).

The host layer reads an unchanged SKIRT ``.ski`` file, runs the reference's setup for the supported classes and
exposes the flattened ``pmc_scene`` that the HIP engine (``skirt9_amd.engine``) consumes; after the photon loop it
calibrates the detector arrays and writes SKIRT's output files.  Mirrors the call sequence of
``SkirtCommandLineHandler::doSimulation`` (SKIRT/main/SkirtCommandLineHandler.cpp:295-372):
``load -> setup -> [engine] -> write``.
"""
import ctypes as C
import os

_LIBDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")


class FrameLayout(C.Structure):
    """pmc_frame_layout (include/pmc.h)"""
    _fields_ = [("num_components", C.c_int64), ("npix", C.c_int64), ("num_lambda", C.c_int64),
                ("sed_offset", C.c_int64), ("ifu_offset", C.c_int64), ("wsed_offset", C.c_int64),
                ("wifu_offset", C.c_int64), ("end_offset", C.c_int64)]


class CounterValues(C.Structure):
    """pmc_counter_values (include/pmc.h)"""
    _fields_ = [("histories", C.c_uint64), ("paths", C.c_uint64), ("cell_visits", C.c_uint64),
                ("detector_updates", C.c_uint64), ("scatterings", C.c_uint64), ("stat_overflows", C.c_uint64),
                ("rewalk_visits", C.c_uint64)]

    def as_dict(self):
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


_lib = None


# ---- ctypes mirrors of the leading members of pmc_scene (include/pmc.h): inspection of the tables a Simulation hands over

class Grid(C.Structure):
    _fields_ = [("kind", C.c_int32), ("xmin", C.c_double), ("ymin", C.c_double), ("zmin", C.c_double),
                ("xmax", C.c_double), ("ymax", C.c_double), ("zmax", C.c_double), ("eps", C.c_double),
                ("num_cells", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("xv", C.POINTER(C.c_double)), ("yv", C.POINTER(C.c_double)), ("zv", C.POINTER(C.c_double)),
                ("num_nodes", C.c_int32), ("node_box", C.POINTER(C.c_double)), ("node_level", C.POINTER(C.c_int32)),
                ("node_first_child", C.POINTER(C.c_int32)), ("node_cell", C.POINTER(C.c_int32)),
                ("nbr_start", C.POINTER(C.c_int32)), ("nbr_list", C.POINTER(C.c_int32)),
                ("site", C.POINTER(C.c_double)), ("vnbr_start", C.POINTER(C.c_int32)), ("vnbr_list", C.POINTER(C.c_int32)),
                ("vblock_n", C.c_int32), ("vblock_start", C.POINTER(C.c_int32)), ("vblock_list", C.POINTER(C.c_int32))]


class Medium(C.Structure):
    _fields_ = [("number_density", C.POINTER(C.c_double)), ("num_lambda", C.c_int32),
                ("lambda_border", C.POINTER(C.c_double)), ("sigma_ext", C.POINTER(C.c_double)),
                ("sigma_sca", C.POINTER(C.c_double)), ("asymmpar", C.POINTER(C.c_double)),
                ("sigma_abs", C.POINTER(C.c_double))]


class SceneHead(C.Structure):
    """leading members of pmc_scene (include/pmc.h)"""
    _fields_ = [("abi_version", C.c_int32), ("grid", Grid), ("medium", Medium)]


def scene_head(sim):
    """the leading members (grid, medium) of the pmc_scene of a set-up Simulation, for inspection"""
    return SceneHead.from_address(sim.scene)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_LIBDIR, "libskirthost.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make` (or __graft_entry__.build()) first")
        L = C.CDLL(path)
        L.skh_last_error.restype = C.c_char_p
        L.skh_load.restype = C.c_void_p
        L.skh_load.argtypes = [C.c_char_p]
        L.skh_free.argtypes = [C.c_void_p]
        L.skh_set_num_packets.argtypes = [C.c_void_p, C.c_uint64]
        L.skh_set_tree_topology_file.argtypes = [C.c_void_p, C.c_char_p]
        L.skh_setup.argtypes = [C.c_void_p]
        L.skh_scene.restype = C.c_void_p
        L.skh_scene.argtypes = [C.c_void_p]
        L.skh_num_packets.restype = C.c_uint64
        L.skh_num_packets.argtypes = [C.c_void_p]
        L.skh_seed.restype = C.c_int32
        L.skh_seed.argtypes = [C.c_void_p]
        L.skh_packet_luminosity.restype = C.c_double
        L.skh_packet_luminosity.argtypes = [C.c_void_p, C.c_int32]
        L.skh_setup_draws.restype = C.c_uint64
        L.skh_setup_draws.argtypes = [C.c_void_p]
        L.skh_frame_size.restype = C.c_int64
        L.skh_frame_size.argtypes = [C.c_void_p]
        L.skh_frame_layout.argtypes = [C.c_void_p, C.c_int32, C.POINTER(FrameLayout)]
        L.skh_write.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        L.skh_summary.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
        L.skh_set_particle_sampler.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.skh_radiation_field_size.restype = C.c_int64
        L.skh_radiation_field_size.argtypes = [C.c_void_p]
        L.skh_write_radiation_field.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        L.skh_scene_save.argtypes = [C.c_void_p, C.c_char_p]
        L.skh_scene_load.restype = C.c_void_p
        L.skh_scene_load.argtypes = [C.c_char_p]
        L.skh_scene_file_free.argtypes = [C.c_void_p]
        L.skh_scene_file_scene.restype = C.c_void_p
        L.skh_scene_file_scene.argtypes = [C.c_void_p]
        L.skh_scene_file_number.restype = C.c_int64
        L.skh_scene_file_number.argtypes = [C.c_void_p, C.c_int32]
        L.skh_scene_file_layout.argtypes = [C.c_void_p, C.c_int32, C.POINTER(FrameLayout)]
        _lib = L
    return _lib


class Simulation:
    """A MonteCarloSimulation constructed from a ski file."""

    def __init__(self, ski_path, num_packets=None, tree_topology=None):
        L = lib()
        self.path = str(ski_path)
        self._h = L.skh_load(os.fsencode(ski_path))
        if not self._h:
            raise RuntimeError(L.skh_last_error().decode())
        if num_packets is not None:
            L.skh_set_num_packets(self._h, int(num_packets))
        if tree_topology is not None:
            if L.skh_set_tree_topology_file(self._h, os.fsencode(tree_topology)) != 0:
                raise RuntimeError(L.skh_last_error().decode())
        self._setup = False

    def use_device_sampler(self, device=0):
        """before setup(): evaluate the densities of an imported particle medium on the MI355X (libpmc.so
        pmc_sampler_*): bit-identical to the host evaluation, and the setup of 10^6 particles takes seconds"""
        from . import engine
        E = engine.lib()
        ptr = lambda f: C.cast(f, C.c_void_p)  # noqa: E731
        if lib().skh_set_particle_sampler(self._h, ptr(E.pmc_sampler_create), ptr(E.pmc_sampler_density),
                                          ptr(E.pmc_sampler_destroy), ptr(E.pmc_last_error), device) != 0:
            raise RuntimeError(lib().skh_last_error().decode())
        return self

    def setup(self):
        if lib().skh_setup(self._h) != 0:
            raise RuntimeError(lib().skh_last_error().decode())
        self._setup = True
        return self

    def close(self):
        if self._h:
            lib().skh_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def scene(self):
        """address of the pmc_scene (valid while this object lives)"""
        assert self._setup, "call setup() first"
        return lib().skh_scene(self._h)

    @property
    def num_packets(self):
        return int(lib().skh_num_packets(self._h))

    @property
    def seed(self):
        return int(lib().skh_seed(self._h))

    def packet_luminosity(self, index=0):
        """luminosity carried by one launched packet at oligochromatic wavelength `index`"""
        value = float(lib().skh_packet_luminosity(self._h, index))
        if value < 0:
            raise ValueError("not an oligochromatic simulation, or wavelength index out of range")
        return value

    @property
    def setup_draws(self):
        return int(lib().skh_setup_draws(self._h))

    @property
    def frame_size(self):
        return int(lib().skh_frame_size(self._h))

    def layout(self, instrument=0):
        out = FrameLayout()
        if lib().skh_frame_layout(self._h, instrument, C.byref(out)) != 0:
            raise IndexError(instrument)
        return out

    @property
    def radiation_field_size(self):
        """doubles of the radiation field table rf[m * nbins + ell]; 0 if the simulation does not store it"""
        return int(lib().skh_radiation_field_size(self._h))

    def write_radiation_field(self, rf, outdir):
        """write the RadiationFieldProbe files (<prefix>_<probe>_J.dat) from the table the engine accumulated"""
        import numpy as np
        data = np.ascontiguousarray(rf, dtype=np.float64)
        assert data.size == self.radiation_field_size
        os.makedirs(outdir, exist_ok=True)
        if lib().skh_write_radiation_field(self._h, data.ctypes.data_as(C.c_void_p), os.fsencode(outdir)) != 0:
            raise RuntimeError(lib().skh_last_error().decode())

    def save_scene(self, path):
        """the set-up scene as one file (skh_scene_save): other processes of a multi-GPU job load it with SceneFile instead of
        repeating the setup"""
        assert self._setup, "call setup() first"
        if lib().skh_scene_save(self._h, os.fsencode(path)) != 0:
            raise RuntimeError(lib().skh_last_error().decode())

    def summary(self):
        buf = C.create_string_buffer(2048)
        lib().skh_summary(self._h, buf, len(buf))
        return buf.value.decode()

    def write(self, frames, outdir):
        """calibrate a COPY of the detector arrays (numpy float64) and write the output files into outdir"""
        import numpy as np
        data = np.ascontiguousarray(frames, dtype=np.float64).copy()
        assert data.size == self.frame_size
        os.makedirs(outdir, exist_ok=True)
        if lib().skh_write(self._h, data.ctypes.data_as(C.c_void_p), os.fsencode(outdir)) != 0:
            raise RuntimeError(lib().skh_last_error().decode())
        return data


class SceneFile:
    """A scene saved by ``Simulation.save_scene``: what the engine needs (``scene``, ``seed``, ``frame_size``, ``layout``,
    ``radiation_field_size``), without the model behind it -- the process that holds the Simulation writes the output."""

    def __init__(self, path):
        L = lib()
        self.path = str(path)
        self._f = L.skh_scene_load(os.fsencode(path))
        if not self._f:
            raise RuntimeError(L.skh_last_error().decode())

    def close(self):
        if getattr(self, "_f", None):
            lib().skh_scene_file_free(self._f)
            self._f = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def scene(self):
        return lib().skh_scene_file_scene(self._f)

    def _number(self, what):
        return int(lib().skh_scene_file_number(self._f, what))

    seed = property(lambda self: self._number(0))
    num_packets = property(lambda self: self._number(1))
    frame_size = property(lambda self: self._number(2))
    radiation_field_size = property(lambda self: self._number(3))
    setup_draws = property(lambda self: self._number(4))

    def layout(self, instrument=0):
        out = FrameLayout()
        if lib().skh_scene_file_layout(self._f, instrument, C.byref(out)) != 0:
            raise IndexError(instrument)
        return out

