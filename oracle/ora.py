"""ctypes wrapper of the CPU oracle (oracle/liboracle.so, oracle/liboracle_omp.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  It exposes the same method names as sph_amd.abi.Engine so that one host
driver (sph_amd.host.SPH) can run either, which is how lock-step parity tests are written.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from sph_amd.abi import PARTICLE_FLOATS, SphParams, SphPhase, phase_array, CTR_COLLISIONS

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    """Compile both oracle variants with the committed Makefile."""
    import subprocess
    if force:
        subprocess.run(["make", "-C", _HERE, "clean"], check=True, capture_output=True)
    subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)


def _load(omp: bool) -> C.CDLL:
    key = "omp" if omp else "serial"
    if key in _LIBS:
        return _LIBS[key]
    path = os.path.join(_HERE, "liboracle_omp.so" if omp else "liboracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    vp, ip = C.c_void_p, C.c_int
    lib.ora_create.restype = vp
    lib.ora_destroy.argtypes = [vp]
    lib.ora_last_error.argtypes = [vp]
    lib.ora_last_error.restype = C.c_char_p
    lib.ora_alloc_particles.argtypes = [vp, ip]
    lib.ora_upload_particles.argtypes = [vp, vp, ip]
    lib.ora_download_particles.argtypes = [vp, vp, ip]
    lib.ora_resize_grid.argtypes = [vp, ip, ip, ip, ip]
    lib.ora_set_params.argtypes = [vp, C.POINTER(SphParams)]
    lib.ora_set_delta_time.argtypes = [vp, C.c_float]
    lib.ora_set_xsph_inplace.argtypes = [vp, ip]
    lib.ora_update_static.argtypes = [vp, C.POINTER(SphPhase), ip, vp, ip]
    lib.ora_dispatch.argtypes = [vp, ip]
    lib.ora_step_frame.argtypes = [vp, ip, C.c_float]
    lib.ora_reset_counter.argtypes = [vp, ip]
    lib.ora_read_counter.argtypes = [vp, ip, C.POINTER(ip)]
    lib.ora_download_grid.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.ora_download_neighbors.argtypes = [vp, vp, vp, ip]
    lib.ora_get_constants.argtypes = [vp, vp]
    _LIBS[key] = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleError(RuntimeError):
    pass


class Oracle:
    """CPU oracle with the Engine's method names."""

    def __init__(self, omp: bool = False):
        self.lib = _load(omp)
        self.h = C.c_void_p(self.lib.ora_create())
        self.n = 0
        self.cells = 0
        self._table = None
        self._ids = None
        self._static_dirty = False

    def _ck(self, rc):
        if rc != 0:
            raise OracleError(self.lib.ora_last_error(self.h).decode())

    def threads(self) -> int:
        return self.lib.ora_threads()

    def destroy(self):
        if getattr(self, "h", None):
            self.lib.ora_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def alloc_particles(self, n):
        self._ck(self.lib.ora_alloc_particles(self.h, n))
        self.n = n

    def upload_particles(self, aos):
        aos = np.ascontiguousarray(aos, dtype=np.float32).reshape(-1, PARTICLE_FLOATS)
        self._ck(self.lib.ora_upload_particles(self.h, _ptr(aos), aos.shape[0]))
        self._static_dirty = True

    def download_particles(self):
        out = np.empty((self.n, PARTICLE_FLOATS), dtype=np.float32)
        self._ck(self.lib.ora_download_particles(self.h, _ptr(out), self.n))
        return out

    def resize_grid(self, rx, ry, rz, cap):
        self._ck(self.lib.ora_resize_grid(self.h, rx, ry, rz, cap))
        self.cells = rx * ry * rz

    def set_params(self, p):
        self._ck(self.lib.ora_set_params(self.h, C.byref(p)))

    def set_delta_time(self, dt):
        self._ck(self.lib.ora_set_delta_time(self.h, float(dt)))

    def set_xsph_inplace(self, on):
        self._ck(self.lib.ora_set_xsph_inplace(self.h, int(on)))

    # SPH.UpdateParticleStaticProperties (SPH.cs:284-298): the oracle really rewrites the
    # four static fields of every particle from its spawn box.
    def set_phase_table(self, phases):
        self._table = phases if isinstance(phases, C.Array) else phase_array(phases)
        self._static_dirty = True

    def set_phase_ids(self, ids):
        self._ids = np.ascontiguousarray(ids, dtype=np.uint8)
        self._static_dirty = True

    def apply_phase_table(self):
        if self._table is None or self._ids is None:
            return
        self._ck(self.lib.ora_update_static(self.h, self._table, len(self._table), _ptr(self._ids), self._ids.size))
        self._static_dirty = False

    def dispatch(self, kernel_id, count=None):
        self._ck(self.lib.ora_dispatch(self.h, kernel_id))

    def step_frame(self, sub_steps, time_step):
        self.apply_phase_table()                      # SPH.cs:198
        self._ck(self.lib.ora_step_frame(self.h, sub_steps, float(time_step)))

    def synchronize(self):
        pass

    def reset_counter(self, which=CTR_COLLISIONS):
        self._ck(self.lib.ora_reset_counter(self.h, which))

    def read_counter(self, which=CTR_COLLISIONS):
        v = C.c_int(0)
        self._ck(self.lib.ora_read_counter(self.h, which, C.byref(v)))
        return v.value

    def download_grid(self, with_counts=False):
        cs = np.empty(self.cells, dtype=np.int32)
        ce = np.empty(self.cells, dtype=np.int32)
        ids = np.full(self.n, -1, dtype=np.int32)
        cop = np.empty(self.n, dtype=np.uint32)
        gc = np.empty(self.cells, dtype=np.int32)
        self._ck(self.lib.ora_download_grid(self.h, _ptr(cs), _ptr(ce), _ptr(ids), _ptr(cop), _ptr(gc)))
        if with_counts:
            return cs, ce, ids, cop, gc
        return cs, ce, ids, cop

    def download_neighbors(self, max_neighbors):
        counts = np.empty(self.n, dtype=np.int32)
        lists = np.full((self.n, max_neighbors), -1, dtype=np.int32)
        self._ck(self.lib.ora_download_neighbors(self.h, _ptr(counts), _ptr(lists), max_neighbors))
        return counts, lists

    def constants(self):
        out = np.zeros(6, dtype=np.float32)
        self._ck(self.lib.ora_get_constants(self.h, _ptr(out)))
        return dict(zip(["h2", "poly6", "spikyGradFactor", "viscLapFactor", "lapColFactor", "B1"], out.tolist()))
