"""CPU stand-in for sph_amd.slab.DevicePort, built on the oracle (TEST INFRASTRUCTURE).

Lets the slab driver (partitioning, message classification, halo width, migration) run with
world_size 2 on the gloo backend without a GPU.  The oracle keeps hashing on the GLOBAL grid, so
the port only has to keep the local particle list: owned + ghosts, uploaded in ascending global
id so that the oracle's bin order equals the engine's (cell, id) order."""
import numpy as np
import torch

from sph_amd import abi

REC = abi.SLAB_RECORD_BYTES


class _Shim:
    """The few Engine attributes SlabEngine touches directly."""

    def __init__(self, port):
        self._p = port
        self.n = 0
        self.cells = 0

    def __getattr__(self, name):
        return getattr(self._p.ora, name)

    def step_frame(self, sub, ts):
        self._p.ora.step_frame(sub, ts)
        self._p.after_frame()

    def dispatch(self, kernel, count):
        self._p.ora.dispatch(kernel)
        if kernel == abi.K_BUILD_GRID:
            self._p.after_build()

    def set_phase_table(self, phases):
        self._p.table = [tuple(float(v) for v in row[0:3]) for row in phases]      # (restDensity, viscosity, mass) per phase

    def apply_phase_table(self):
        self._p.apply_table()

    def read_counter(self, which=0):
        if which == abi.CTR_MAX_PLANE_SHIFT:
            return self._p.max_shift
        if which == abi.CTR_BAND_MISSES:
            return 0
        return self._p.ora.read_counter(which)


class OraclePort:
    def __init__(self, oracle):
        self.ora = oracle
        self.e = _Shim(self)
        self.device = torch.device("cpu")
        self.ids = np.zeros(0, np.uint32)          # global id of local particle k
        self.phase = np.zeros(0, np.uint32)        # its phase (spawn box): travels in the records
        self.table = None                          # the phase table in force (SlabEngine.set_phase_table)
        self.owned_mask = None                     # owned at the last grid build (None = all)
        self.carry_owned = None                    # after a rebuild: ids that were owned before it (received records are not)
        self.max_shift = 0

    def configure(self, rx_global, x_offset, rx_local, ry, rz, cap, own_lo, own_hi, capacity):
        self.rx, self.own = rx_global, (own_lo, own_hi)
        self.ora.resize_grid(rx_global, ry, rz, cap)
        self.e.cells = rx_global * ry * rz

    def reconfigure(self, rx_global, x_offset, rx_local, ry, rz, cap, own_lo, own_hi):
        self.own = (own_lo, own_hi)                # owned_mask stays that of the last build until the next one

    def owned_plane_histogram(self, rx_global, x_offset, rx_local, own_lo, own_hi):
        valid = self.ids != 0xFFFFFFFF
        owned = valid if self.owned_mask is None else (self.owned_mask & valid)
        return np.bincount(self._bin_plane[owned], minlength=rx_global).astype(np.int64)

    def _set_local(self, recs):
        recs = recs[np.argsort(recs["id"], kind="stable")]
        self.ids = recs["id"].copy()
        self.phase = recs["phase"].astype(np.uint32)
        n = recs.shape[0]
        self.ora.alloc_particles(max(n, 1))
        aos = abi.aos_from_records(recs) if n else np.zeros((1, 18), np.float32)
        if n == 0:
            aos[0, 0:3] = 1e9                      # a parked dummy; never owned
            aos[0, 9] = aos[0, 11] = 1.0
            self.ids = np.array([0xFFFFFFFF], np.uint32)
            self.phase = np.zeros(1, np.uint32)
        self._table_into(aos)                      # the table in force is authoritative (engine: store_record)
        self.ora.upload_particles(aos)
        self.ora._static_dirty = False
        self.owned_mask = None
        self.e.n = max(n, 1)
        return n

    def _table_into(self, aos):
        if self.table is None:
            return
        rows = np.asarray(self.table, dtype=np.float32)
        known = (self.phase < rows.shape[0]) & (self.ids != 0xFFFFFFFF)
        aos[known, 11:14] = rows[self.phase[known]]

    def apply_table(self):
        """A live edit of a spawn box: every local particle takes its phase's row (the grid and bins stay as they are)."""
        if self.table is None or self.ids.size == 0:
            return
        aos = self.ora.download_particles()
        self._table_into(aos)
        self.ora.upload_particles(aos)
        self.ora._static_dirty = False

    def upload_owned(self, records):
        self._set_local(np.array(records, copy=True))
        self.carry_owned = None

    def _planes(self, x, p):
        c = np.floor(((x - np.float32(p.minBound[0])) / np.float32(p.cellSize)).astype(np.float32)).astype(np.int64)
        return np.clip(c, 0, self.rx - 1)

    def set_params_ref(self, p):
        self.params = p

    def _current(self):
        aos = self.ora.download_particles()
        recs = abi.records_from_aos(aos, self.ids, self.phase)
        valid = self.ids != 0xFFFFFFFF
        if self.owned_mask is not None:
            owned = self.owned_mask & valid
        elif self.carry_owned is not None:         # rebuilt, not stepped yet: what the engine's slab_download reports
            owned = valid & np.isin(self.ids, self.carry_owned)
        else:
            owned = valid
        return recs, owned, self._planes(aos[:, 0], self.params)

    def pack(self, left_below, right_from):
        recs, owned, plane = self._current()
        to_l = recs[owned & (plane < left_below)]
        to_r = recs[owned & (plane >= right_from)]
        return (torch.from_numpy(to_l.view(np.uint8).reshape(-1, REC).copy()),
                torch.from_numpy(to_r.view(np.uint8).reshape(-1, REC).copy()))

    def arm_pack(self, left_below, right_from, band_margin):
        self._armed = (left_below, right_from)

    def pack_wait(self):
        return self.pack(*self._armed)              # the oracle ran the whole frame: nothing to overlap with

    def rebuild(self, keep_lo, keep_hi, from_left, from_right):
        recs, owned, plane = self._current()
        parts = [recs[owned & (plane >= keep_lo) & (plane < keep_hi)]]
        carry = recs["id"][owned].copy()
        for t in (from_left, from_right):
            if t is not None and t.shape[0]:
                r = np.frombuffer(t.contiguous().numpy().tobytes(), dtype=abi.SLAB_RECORD_DTYPE)
                pl = self._planes(r["pos"][:, 0], self.params)
                parts.append(r[(pl >= keep_lo) & (pl < keep_hi)])
        n = self._set_local(np.concatenate(parts))
        self.carry_owned = carry
        return n

    def after_build(self):
        aos = self.ora.download_particles()
        self._bin_plane = self._planes(aos[:, 0], self.params)
        self.owned_mask = (self._bin_plane >= self.own[0]) & (self._bin_plane < self.own[1])

    def after_frame(self):
        # the oracle ran the whole frame: ownership is by FRAME-START plane = the bins' plane
        cop = self.ora.download_grid()[3].astype(np.int64)
        self._bin_plane = cop % self.rx
        self.owned_mask = (self._bin_plane >= self.own[0]) & (self._bin_plane < self.own[1])
        # what the engine's density kernel measures (SPH_CTR_MAX_PLANE_SHIFT), from the frame's final positions:
        # planes moved against the binned plane, owned particles
        aos = self.ora.download_particles()
        now = self._planes(aos[:, 0], self.params)
        own = self.owned_mask & (self.ids != 0xFFFFFFFF)
        self.max_shift = int(np.abs(now[own] - self._bin_plane[own]).max(initial=0))

    def download_owned(self):
        recs, owned, _ = self._current()
        return recs[owned]
