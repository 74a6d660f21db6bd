"""ctypes binding of the C ABI in include/sph_hip.h (libsph_hip.so).

This is the reference-side binding a maintainer would write for Assets/SPH.cs, in Python:
`Engine` plays the part of UnityEngine.ComputeShader + ComputeBuffer (SPH.cs:366-443).
There is no CPU fallback: if the shared library is missing, importing the engine raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PARTICLE_FLOATS = 18
PARTICLE_BYTES = 72

# enum sph_kernel == "#pragma kernel" order (compute:9-17)
K_CLEAR, K_CLEAR_GRID, K_BUILD_GRID, K_DENSITY_PRESSURE, K_FORCE_XSPH, K_VV_HALF_STEP, \
    K_VV_FULL_STEP, K_BOUND_OBS, K_VELOCITY_DAMPING = range(9)
KERNEL_NAMES = ["CS_Clear", "CS_ClearGrid", "CS_BuildGrid", "CS_DensityPressure", "CS_ForceXSPH",
                "CS_VvHalfStep", "CS_VvFullStep", "CS_BoundObs", "CS_VelocityDamping"]

CTR_COLLISIONS, CTR_BIN_OVERFLOW, CTR_OUT_OF_GRID, CTR_MAX_PLANE_SHIFT, CTR_BAND_MISSES, CTR_NAN = range(6)
SLAB_RECORD_BYTES = 64
PLANE_COUNTS_MAX = 8192      # SPH_PLANE_COUNTS_MAX

# every symbol include/sph_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "sph_create", "sph_destroy", "sph_last_error", "sph_version", "sph_alloc_particles",
    "sph_upload_particles", "sph_upload_particles_in_place", "sph_download_particles", "sph_export_particles_device", "sph_resize_grid", "sph_set_params",
    "sph_set_delta_time", "sph_set_phase_table", "sph_set_phase_ids", "sph_apply_phase_table",
    "sph_dispatch", "sph_step_frame", "sph_synchronize", "sph_reset_counter", "sph_read_counter",
    "sph_download_grid", "sph_download_neighbors", "sph_download_hit_lists", "sph_get_timings", "sph_device_pointer",
    "sph_get_stream", "sph_slab_configure", "sph_slab_reserve", "sph_slab_upload", "sph_slab_download",
    "sph_slab_pack", "sph_slab_rebuild", "sph_slab_arm_pack", "sph_slab_pack_wait", "sph_slab_plane_counts",
    "sph_slab_plane_counts_noalloc", "sph_particle_count",
]


class SphParams(C.Structure):
    """struct sph_params: field-for-field mirror of cbuffer SimulationParams (compute:47-81)."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("particleCount", C.c_int32),
        ("deltaTime", C.c_float),
        ("particleMass", C.c_float),
        ("restDensity", C.c_float),
        ("soundSpeed", C.c_float),
        ("gamma", C.c_float),
        ("smoothingRadius", C.c_float),
        ("gravity", C.c_float),
        ("surfaceTensionCoefficient", C.c_float),
        ("surfaceTensionThreshold", C.c_float),
        ("xsphEpsilon", C.c_float),
        ("particleCollisionDamping", C.c_float),
        ("obstacleRepulsionStiffness", C.c_float),
        ("obstacleRadius", C.c_float),
        ("boundsCenter", C.c_float * 3),
        ("boundsSize", C.c_float * 3),
        ("spawnCenter", C.c_float * 3),
        ("spawnSize", C.c_float * 3),
        ("obstaclePos", C.c_float * 3),
        ("gridResolutionX", C.c_int32),
        ("gridResolutionY", C.c_int32),
        ("gridResolutionZ", C.c_int32),
        ("maxParticlesPerCell", C.c_int32),
        ("cellSize", C.c_float),
        ("minBound", C.c_float * 3),
        ("boundaryMatrix", C.c_float * 16),
        ("boundaryInvMatrix", C.c_float * 16),
        ("boundaryHalfExtents", C.c_float * 3),
    ]

    def __init__(self, **kw):
        super().__init__()
        self.struct_size = C.sizeof(SphParams)
        for k, v in kw.items():
            self.set(k, v)

    def set(self, name, value):
        cur = getattr(self, name)
        if isinstance(cur, C.Array):
            flat = np.asarray(value, dtype=np.float32).reshape(-1)
            if flat.size != len(cur):
                raise ValueError(f"{name}: expected {len(cur)} floats, got {flat.size}")
            for i, x in enumerate(flat):
                cur[i] = float(x)
        else:
            setattr(self, name, value)

    def copy(self):
        other = SphParams()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(SphParams))
        return other


class SphPhase(C.Structure):
    """struct sph_phase: one spawn box's static properties (SpawnBox.cs:10-23)."""
    _fields_ = [("restDensity", C.c_float), ("viscosity", C.c_float), ("mass", C.c_float),
                ("color", C.c_float * 4)]


class SphTimings(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("build_grid_ms", "half_step_ms", "density_ms", "force_ms",
                                         "full_step_ms", "bound_obs_ms", "damping_ms", "frame_ms")] + [("frames", C.c_int32)]


def phase_array(phases):
    """[(restDensity, viscosity, mass, (r,g,b,a)), ...] -> ctypes array of sph_phase."""
    arr = (SphPhase * len(phases))()
    for i, (rd, visc, mass, col) in enumerate(phases):
        arr[i].restDensity, arr[i].viscosity, arr[i].mass = float(rd), float(visc), float(mass)
        for k in range(4):
            arr[i].color[k] = float(col[k])
    return arr


class SphError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsph_hip error {code}: {msg}")
        self.code = code


_LIB = None


def library_path() -> str:
    return os.environ.get("SPH_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                         "libsph_hip.so")


def load_library() -> C.CDLL:
    """dlopen libsph_hip.so and declare prototypes.  Raises if it is not built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the engine has no CPU fallback)")
    lib = C.CDLL(path)
    vp, ip = C.c_void_p, C.c_int
    lib.sph_create.argtypes = [C.POINTER(vp), ip]
    lib.sph_destroy.argtypes = [vp]
    lib.sph_last_error.argtypes = [vp]
    lib.sph_last_error.restype = C.c_char_p
    lib.sph_version.restype = C.c_char_p
    lib.sph_alloc_particles.argtypes = [vp, ip]
    lib.sph_upload_particles.argtypes = [vp, vp, ip]
    lib.sph_upload_particles_in_place.argtypes = [vp, vp, ip]
    lib.sph_download_particles.argtypes = [vp, vp, ip]
    lib.sph_export_particles_device.argtypes = [vp, vp, ip]
    lib.sph_resize_grid.argtypes = [vp, ip, ip, ip, ip]
    lib.sph_set_params.argtypes = [vp, C.POINTER(SphParams)]
    lib.sph_set_delta_time.argtypes = [vp, C.c_float]
    lib.sph_set_phase_table.argtypes = [vp, C.POINTER(SphPhase), ip]
    lib.sph_set_phase_ids.argtypes = [vp, vp, ip]
    lib.sph_apply_phase_table.argtypes = [vp]
    lib.sph_dispatch.argtypes = [vp, ip, ip]
    lib.sph_step_frame.argtypes = [vp, ip, C.c_float]
    lib.sph_synchronize.argtypes = [vp]
    lib.sph_reset_counter.argtypes = [vp, ip]
    lib.sph_read_counter.argtypes = [vp, ip, C.POINTER(ip)]
    lib.sph_download_grid.argtypes = [vp, vp, vp, vp, vp]
    lib.sph_download_neighbors.argtypes = [vp, vp, vp, ip]
    lib.sph_download_hit_lists.argtypes = [vp, vp, vp, ip, C.POINTER(ip)]
    lib.sph_get_timings.argtypes = [vp, C.POINTER(SphTimings)]
    lib.sph_device_pointer.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64)]
    lib.sph_get_stream.argtypes = [vp, C.POINTER(vp)]
    lib.sph_slab_configure.argtypes = [vp, ip, ip, ip, ip]
    lib.sph_slab_reserve.argtypes = [vp, ip]
    lib.sph_slab_upload.argtypes = [vp, vp, ip]
    lib.sph_slab_download.argtypes = [vp, vp, ip, ip, C.POINTER(ip)]
    lib.sph_slab_pack.argtypes = [vp, ip, ip, vp, ip, vp, ip, C.POINTER(ip), C.POINTER(ip)]
    lib.sph_slab_rebuild.argtypes = [vp, ip, ip, vp, ip, vp, ip, C.POINTER(ip)]
    lib.sph_slab_arm_pack.argtypes = [vp, ip, ip, ip, vp, ip, vp, ip]
    lib.sph_slab_pack_wait.argtypes = [vp, C.POINTER(ip), C.POINTER(ip), C.POINTER(ip)]
    lib.sph_slab_plane_counts.argtypes = [vp, vp, ip]
    lib.sph_slab_plane_counts_noalloc.argtypes = [vp, vp, ip]
    lib.sph_particle_count.argtypes = [vp, C.POINTER(ip)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("sph_last_error", "sph_version"):
            fn.restype = ip
    _LIB = lib
    return lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One GPU's SPH engine (an sph_ctx).  Method names follow the C ABI."""

    def __init__(self, device_id: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.sph_create(C.byref(h), device_id)
        if rc != 0:
            raise SphError(rc, "sph_create failed (no gfx950 device visible?)")
        self.h = h
        self.n = 0
        self.cells = 0

    # -- plumbing
    def _ck(self, rc):
        if rc != 0:
            raise SphError(rc, self.lib.sph_last_error(self.h).decode())

    def destroy(self):
        if getattr(self, "h", None):
            self.lib.sph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # -- buffers
    def alloc_particles(self, n):
        self._ck(self.lib.sph_alloc_particles(self.h, n))
        self.n = n

    def upload_particles(self, aos: np.ndarray):
        aos = np.ascontiguousarray(aos, dtype=np.float32).reshape(-1, PARTICLE_FLOATS)
        self._ck(self.lib.sph_upload_particles(self.h, _ptr(aos), aos.shape[0]))

    def upload_particles_in_place(self, aos: np.ndarray):
        """SetData in the middle of a frame: slot order and grid stay (sph_upload_particles_in_place)."""
        aos = np.ascontiguousarray(aos, dtype=np.float32).reshape(-1, PARTICLE_FLOATS)
        self._ck(self.lib.sph_upload_particles_in_place(self.h, _ptr(aos), aos.shape[0]))

    def download_particles(self) -> np.ndarray:
        out = np.empty((self.n, PARTICLE_FLOATS), dtype=np.float32)
        self._ck(self.lib.sph_download_particles(self.h, _ptr(out), self.n))
        return out

    def export_particles_device(self, dev_ptr: int, n=None):
        """Render feed: the 72-byte records (id order) into a DEVICE buffer of n * 72 bytes; asynchronous."""
        self._ck(self.lib.sph_export_particles_device(self.h, C.c_void_p(dev_ptr), self.n if n is None else n))

    def resize_grid(self, rx, ry, rz, cap):
        self._ck(self.lib.sph_resize_grid(self.h, rx, ry, rz, cap))
        self.cells = rx * ry * rz

    def set_params(self, p: SphParams):
        self._ck(self.lib.sph_set_params(self.h, C.byref(p)))

    def set_delta_time(self, dt):
        self._ck(self.lib.sph_set_delta_time(self.h, float(dt)))

    def set_phase_table(self, phases):
        arr = phases if isinstance(phases, C.Array) else phase_array(phases)
        self._ck(self.lib.sph_set_phase_table(self.h, arr, len(arr)))

    def set_phase_ids(self, ids: np.ndarray):
        ids = np.ascontiguousarray(ids, dtype=np.uint8)
        self._ck(self.lib.sph_set_phase_ids(self.h, _ptr(ids), ids.size))

    def apply_phase_table(self):
        self._ck(self.lib.sph_apply_phase_table(self.h))

    # -- dispatch
    def dispatch(self, kernel_id, count):
        self._ck(self.lib.sph_dispatch(self.h, kernel_id, count))

    def step_frame(self, sub_steps, time_step):
        self._ck(self.lib.sph_step_frame(self.h, sub_steps, float(time_step)))

    def synchronize(self):
        self._ck(self.lib.sph_synchronize(self.h))

    # -- counters
    def reset_counter(self, which=CTR_COLLISIONS):
        self._ck(self.lib.sph_reset_counter(self.h, which))

    def read_counter(self, which=CTR_COLLISIONS) -> int:
        v = C.c_int(0)
        self._ck(self.lib.sph_read_counter(self.h, which, C.byref(v)))
        return v.value

    # -- parity exports
    def download_grid(self):
        cs = np.empty(self.cells, dtype=np.int32)
        ce = np.empty(self.cells, dtype=np.int32)
        ids = np.empty(self.n, dtype=np.int32)
        cop = np.empty(self.n, dtype=np.uint32)
        self._ck(self.lib.sph_download_grid(self.h, _ptr(cs), _ptr(ce), _ptr(ids), _ptr(cop)))
        return cs, ce, ids, cop

    def download_neighbors(self, max_neighbors):
        counts = np.empty(self.n, dtype=np.int32)
        lists = np.full((self.n, max_neighbors), -1, dtype=np.int32)
        self._ck(self.lib.sph_download_neighbors(self.h, _ptr(counts), _ptr(lists), max_neighbors))
        return counts, lists

    def download_hit_lists(self, max_neighbors):
        """(mask_state, counts, lists): the force kernel's hit mask decoded into ids (sph_download_hit_lists).
        mask_state 1 = lists valid, 0 = no fresh mask, 2 = a bin overflowed (capped scan path, nothing recorded)."""
        counts = np.full(self.n, -3, dtype=np.int32)
        lists = np.full((self.n, max_neighbors), -1, dtype=np.int32)
        state = C.c_int(0)
        self._ck(self.lib.sph_download_hit_lists(self.h, _ptr(counts), _ptr(lists), max_neighbors, C.byref(state)))
        return state.value, counts, lists

    def get_timings(self) -> dict:
        t = SphTimings()
        self._ck(self.lib.sph_get_timings(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in SphTimings._fields_}

    def device_pointer(self, name: str):
        p, b = C.c_void_p(), C.c_uint64()
        self._ck(self.lib.sph_device_pointer(self.h, name.encode(), C.byref(p), C.byref(b)))
        return p.value, b.value

    # -- slab decomposition
    def slab_configure(self, rx_global, x_offset, own_lo, own_hi):
        self._ck(self.lib.sph_slab_configure(self.h, rx_global, x_offset, own_lo, own_hi))

    def slab_reserve(self, capacity):
        self._ck(self.lib.sph_slab_reserve(self.h, capacity))

    def slab_upload(self, records: np.ndarray):
        records = np.ascontiguousarray(records, dtype=SLAB_RECORD_DTYPE)
        self._ck(self.lib.sph_slab_upload(self.h, _ptr(records), records.shape[0]))
        self.n = records.shape[0]

    def slab_download(self, owned_only=True) -> np.ndarray:
        out = np.zeros(max(self.particle_count(), 1), dtype=SLAB_RECORD_DTYPE)
        n = C.c_int(0)
        self._ck(self.lib.sph_slab_download(self.h, _ptr(out), out.shape[0], int(owned_only), C.byref(n)))
        return out[:n.value]

    def slab_pack(self, left_below, right_from, dev_left, cap_left, dev_right, cap_right):
        nl, nr = C.c_int(0), C.c_int(0)
        self._ck(self.lib.sph_slab_pack(self.h, left_below, right_from, dev_left, cap_left, dev_right, cap_right,
                                        C.byref(nl), C.byref(nr)))
        return nl.value, nr.value

    def slab_rebuild(self, keep_lo, keep_hi, dev_left, n_left, dev_right, n_right) -> int:
        n = C.c_int(0)
        self._ck(self.lib.sph_slab_rebuild(self.h, keep_lo, keep_hi, dev_left, n_left, dev_right, n_right, C.byref(n)))
        self.n = n.value
        return n.value

    def slab_arm_pack(self, left_below, right_from, band_margin, dev_left, cap_left, dev_right, cap_right):
        self._ck(self.lib.sph_slab_arm_pack(self.h, left_below, right_from, band_margin, dev_left, cap_left, dev_right, cap_right))

    def slab_pack_wait(self):
        nl, nr, nb = C.c_int(0), C.c_int(0), C.c_int(0)
        self._ck(self.lib.sph_slab_pack_wait(self.h, C.byref(nl), C.byref(nr), C.byref(nb)))
        self.band_workgroups = nb.value
        return nl.value, nr.value

    def slab_plane_counts(self, rx_local) -> np.ndarray:
        out = np.zeros(rx_local, dtype=np.int32)
        self._ck(self.lib.sph_slab_plane_counts(self.h, _ptr(out), rx_local))
        return out

    def slab_plane_counts_noalloc(self, rx_local, out=None) -> np.ndarray:
        """The variant that allocates nothing on the device (static scratch, at most PLANE_COUNTS_MAX planes).  `out`:
        an int32 array to fill (host), or an int = device pointer of rx_local int32."""
        if isinstance(out, int):
            self._ck(self.lib.sph_slab_plane_counts_noalloc(self.h, C.c_void_p(out), rx_local))
            return None
        out = np.zeros(rx_local, dtype=np.int32) if out is None else out
        self._ck(self.lib.sph_slab_plane_counts_noalloc(self.h, _ptr(out), rx_local))
        return out

    def particle_count(self) -> int:
        n = C.c_int(0)
        self._ck(self.lib.sph_particle_count(self.h, C.byref(n)))
        return n.value

    def get_stream(self) -> int:
        s = C.c_void_p()
        self._ck(self.lib.sph_get_stream(self.h, C.byref(s)))
        return s.value or 0


# numpy view of struct sph_slab_record (64 B)
SLAB_RECORD_DTYPE = np.dtype([("pos", np.float32, 3), ("mass", np.float32), ("vel", np.float32, 3),
                              ("viscosity", np.float32), ("acc", np.float32, 3), ("restDensity", np.float32),
                              ("id", np.uint32), ("density", np.float32), ("pressure", np.float32),
                              ("phase", np.uint32)])
assert SLAB_RECORD_DTYPE.itemsize == SLAB_RECORD_BYTES


def records_from_aos(aos: np.ndarray, ids: np.ndarray, phase=0) -> np.ndarray:
    """72-byte AoS rows (+ global ids, + phase = spawn box index, scalar or per row) -> slab records."""
    aos = np.asarray(aos, dtype=np.float32).reshape(-1, PARTICLE_FLOATS)
    r = np.zeros(aos.shape[0], dtype=SLAB_RECORD_DTYPE)
    r["pos"], r["vel"], r["acc"] = aos[:, 0:3], aos[:, 3:6], aos[:, 6:9]
    r["density"], r["pressure"], r["restDensity"] = aos[:, 9], aos[:, 10], aos[:, 11]
    r["viscosity"], r["mass"] = aos[:, 12], aos[:, 13]
    r["id"] = ids
    r["phase"] = phase
    return r


def aos_from_records(r: np.ndarray) -> np.ndarray:
    """slab records -> 72-byte AoS rows (colour zero), in the records' order."""
    aos = np.zeros((r.shape[0], PARTICLE_FLOATS), dtype=np.float32)
    aos[:, 0:3], aos[:, 3:6], aos[:, 6:9] = r["pos"], r["vel"], r["acc"]
    aos[:, 9], aos[:, 10], aos[:, 11] = r["density"], r["pressure"], r["restDensity"]
    aos[:, 12], aos[:, 13] = r["viscosity"], r["mass"]
    return aos
