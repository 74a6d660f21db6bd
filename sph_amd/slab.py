"""Multi-GPU slab decomposition of the SPH frame step.

The reference is single-GPU; this is new work with no reference counterpart.  One process per
GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on MI355X; "gloo" for tests).

Decomposition
  * the GLOBAL grid (RX x RY x RZ cells) is cut into G slabs of whole cell planes along x
    (gravity stratifies y, so x keeps slabs balanced); rank r owns planes [x0_r, x1_r);
  * a rank's local particle set = the particles it owns + ghost copies of `ghost_planes` planes on
    each side; its local grid is planes [x0 - ghost, x1 + ghost) of the global grid; cell
    coordinates are still computed against the GLOBAL origin and shifted by an integer plane offset,
    so the hash is bit-identical to a single-GPU run;
  * particle ids are global, and the (cell, id) order inside the engine makes every neighbour
    sum visit the same particles in the same order as a single-GPU run: results are bit-identical
    for owned particles, whatever G is.

Per frame (before the grid build of SPH.Update, SPH.cs:201-204) each rank sends ONE message to each
slab neighbour: its owned particles whose current plane lies within `ghost_planes` of that face or
beyond it.  The receiver keeps what falls into its local plane range: records in its own planes
are immigrants (now owned), the others are ghosts.  The sender keeps its emigrants as ghosts.  Last
frame's ghosts are dropped.  The frame then runs with NO further communication: ghosts are advanced
redundantly (half-step and density are recomputed locally for them from identical inputs, hence
bit-identical), which needs ghost_planes >= 2 * subSteps * (1 + max plane shift of a particle against
its frame-start plane): every sub-step a band of 2 * (1 + shift) planes at the outer edge of the ghost
region stops being right (density there misses neighbours, then forces next to it use that density), and
the owned planes must stay clear of it -- wider ghosts instead of a ghost refresh per sub-step.  The engine
measures the shift every frame (SPH_CTR_MAX_PLANE_SHIFT, owned particles) and the driver counts violations.  Point-to-point only: each GPU talks to at most two peers over their direct xGMI
links; there is no collective on the data path.

Load balance (SlabEngine(rebalance_every=K)): every K frames the ranks add up their per-plane particle counts
(read from the cell table, `sph_slab_plane_counts`; one small all-reduce of RX ints -- the only collective, and
not on the data path), cut the planes into equal shares and move their faces by at most `max_cut_shift` planes.
Ownership stays that of the last grid build until the next one, so the frame's ordinary face messages carry
whatever changes hands; no extra exchange.  Results stay bit-identical to a one-GPU run.

A moving boundary cube (grid origin / cell size / resolution change, SPH.cs:176-185) re-defines the planes:
the driver then re-partitions from scratch (`_repartition`: download, all-to-all by new owner, upload; slow path).

Limits of this round: migration to adjacent slabs only between re-partitions (particles conserved is
asserted by `count_global`).
"""
from __future__ import annotations

import os
import sys
import time
from typing import Optional

import numpy as np

from . import abi

REC = abi.SLAB_RECORD_BYTES


def slab_ranges(rx_global: int, world: int):
    """Equal split of the x planes: [(x0, x1)] per rank."""
    cuts = [(rx_global * r) // world for r in range(world + 1)]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def balanced_cuts(hist, old_cuts, min_width: int, max_shift: int):
    """Cut points (len world+1) that split `hist` (particles per global plane) into equal shares, each face at most
    `max_shift` planes from its old place and every slab at least `min_width` planes wide.  Needs old slabs that
    are themselves at least `min_width` wide (then the two constraints never contradict each other)."""
    hist = np.asarray(hist, dtype=np.int64)
    rx, world = hist.shape[0], len(old_cuts) - 1
    cum = np.concatenate([[0], np.cumsum(hist)])
    total = int(cum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        c = int(np.searchsorted(cum, target, side="left"))
        if c > 0 and abs(cum[c - 1] - target) <= abs(cum[min(c, rx)] - target):
            c -= 1
        lo = max(old_cuts[r] - max_shift, cuts[-1] + min_width)
        hi = min(old_cuts[r] + max_shift, rx - (world - r) * min_width)
        cuts.append(min(max(c, lo), hi))
    cuts.append(rx)
    return cuts


# ------------------------------------------------------------------------------------------------
# ports: how the driver talks to one rank's engine.  Messages are torch uint8 tensors [n, 64].
# ------------------------------------------------------------------------------------------------
class DevicePort:
    """libsph_hip engine; message buffers are torch tensors on the engine's GPU."""

    def __init__(self, engine: abi.Engine, device_index: int, msg_capacity: int):
        import torch
        self.torch = torch
        self.e = engine
        self.device = torch.device("cuda", device_index)
        self.cap = int(msg_capacity)
        # one spare row in front of the records: the header of the face message (NeighbourComm.exchange sends
        # header + records as ONE tensor, straight out of these buffers)
        self.send_bases = [torch.empty((self.cap + 1, REC), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.send_l, self.send_r = self.send_bases[0][1:], self.send_bases[1][1:]

    def configure(self, rx_global, x_offset, rx_local, ry, rz, cap, own_lo, own_hi, capacity):
        self.e.slab_configure(rx_global, x_offset, own_lo, own_hi)
        self.e.slab_reserve(capacity)
        self.e.resize_grid(rx_local, ry, rz, cap)

    def reconfigure(self, rx_global, x_offset, rx_local, ry, rz, cap, own_lo, own_hi):
        """Move the slab's cut between two frames (before that frame's pack)."""
        self.e.slab_configure(rx_global, x_offset, own_lo, own_hi)
        self.e.resize_grid(rx_local, ry, rz, cap)

    def owned_plane_histogram(self, rx_global, x_offset, rx_local, own_lo, own_hi):
        """Particles per GLOBAL plane in this rank's owned planes, as binned by the last grid build."""
        local = self.e.slab_plane_counts(rx_local)
        hist = np.zeros(rx_global, dtype=np.int64)
        hist[own_lo:own_hi] = local[own_lo - x_offset:own_hi - x_offset]
        return hist

    def upload_owned(self, records):
        self.e.slab_upload(records)

    def pack(self, left_below, right_from):
        nl, nr = self.e.slab_pack(left_below, right_from, self.send_l.data_ptr(), self.cap, self.send_r.data_ptr(), self.cap)
        return self.send_l[:nl], self.send_r[:nr]

    def arm_pack(self, left_below, right_from, band_margin):
        """The next step_frame packs its own face messages, early (sph_slab_arm_pack)."""
        self.e.slab_arm_pack(left_below, right_from, band_margin, self.send_l.data_ptr(), self.cap, self.send_r.data_ptr(), self.cap)

    def pack_wait(self):
        nl, nr = self.e.slab_pack_wait()
        return self.send_l[:nl], self.send_r[:nr]

    def rebuild(self, keep_lo, keep_hi, from_left, from_right):
        pl = from_left.data_ptr() if from_left is not None and from_left.shape[0] else None
        pr = from_right.data_ptr() if from_right is not None and from_right.shape[0] else None
        # (the received records are complete: NeighbourComm.exchange returned after it had read their headers on the
        # host, behind the transfer in stream order; the engine's own stream may still be running the frame whose
        # messages these are -- that is the overlap)
        nl = 0 if pl is None else from_left.shape[0]
        nr = 0 if pr is None else from_right.shape[0]
        return self.e.slab_rebuild(keep_lo, keep_hi, pl, nl, pr, nr)

    def download_owned(self):
        return self.e.slab_download(owned_only=True)

    def local_count(self):
        return self.e.particle_count()


# ------------------------------------------------------------------------------------------------
# communication: neighbour-to-neighbour send/recv of record tensors
# ------------------------------------------------------------------------------------------------
HDR_COUNT, HDR_NEED, HDR_FATAL = 0, 1, 2   # int64 fields of a message's header row (64 bytes = 8 fields)
FIRST_BOUND = 4096                   # records both ends of a face expect in the first message (before any was seen)


def next_bound(n: int, cap: int = 0) -> int:
    """Records both ends of a face reserve for the NEXT message in a direction, from the size of the last one (which
    both know): a sixteenth of headroom -- face populations change by a few per cent per frame -- so that the message
    is one transfer of a size the receiver can post without having been told it."""
    b = int(n) + (int(n) >> 4) + 256
    return min(b, cap) if cap else b


class NeighbourComm:
    """Neighbour-to-neighbour exchange of record tensors, ONE message per face and direction per frame: a header row
    (record count, ghost width the sender asks for) followed by the records, padded to the size both ends predicted
    from the previous frame's message (next_bound) -- send and receive sizes must match and the receiver cannot know
    the count before it has the message.  A message that outgrew its bound is completed by a second transfer of exactly
    the missing records (both ends see the count in the header), which happens in the first frame and whenever a face
    population jumps by more than a sixteenth.  Per exchange the host waits once: for the incoming headers (behind the
    transfer in stream order, so the records are complete as well and the send buffers are free again)."""

    def __init__(self, rank: int, world: int, device, backend: Optional[str] = None, msg_capacity: int = 0):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.device = rank, world, device
        self.backend = backend or dist.get_backend()
        self.stage_cpu = self.backend == "gloo"
        self.peer_need = (0, 0)          # what the left / right neighbour said it needs, as of the last exchange
        self.cap = int(msg_capacity)     # records per message at most (0: unlimited); must be the same on every rank
        self.bound_send = [FIRST_BOUND, FIRST_BOUND]
        self.bound_recv = [FIRST_BOUND, FIRST_BOUND]
        self.rbuf = [None, None]
        self.exchanges = self.rounds = self.host_waits = 0
        self.padded_records = self.sent_records = 0
        self.peer_fatal = (False, False)  # the left / right neighbour announced in its last header that it is stopping
        self.error = None                 # a failure this exchange detected itself (the caller makes the stop collective)

    def _dev(self):
        return self.torch.device("cpu") if self.stage_cpu else self.device

    def _xfer(self, sends, recvs):
        ops = []
        for t, peer in sends:
            ops.append(self.dist.P2POp(self.dist.isend, t, peer))
        for t, peer in recvs:
            ops.append(self.dist.P2POp(self.dist.irecv, t, peer))
        if ops:
            self.rounds += 1
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()

    def _recv_rows(self, side, rows):
        """Receive buffer of this face, at least `rows` rows (header included); contents are kept when it grows."""
        torch = self.torch
        cur = self.rbuf[side]
        if cur is None or cur.shape[0] < rows:
            want = max(rows, (self.cap + 1) if self.cap else rows + (rows >> 2))
            new = torch.empty((want, REC), dtype=torch.uint8, device=self._dev())
            if cur is not None:
                new[:cur.shape[0]] = cur
            self.rbuf[side] = cur = new
        return cur

    def empty(self):
        """A message of no records."""
        return self.torch.empty((0, REC), dtype=self.torch.uint8, device=self._dev())

    def _message(self, side, recs, base, bound, need, fatal=False):
        """The tensor to send: header row + the first min(n, bound) records + padding up to `bound` rows.  `base`: a
        buffer whose rows 1.. already hold the records (DevicePort.send_bases), used in place."""
        torch = self.torch
        n = int(recs.shape[0])
        dev = self._dev()
        hdr = torch.zeros(REC // 8, dtype=torch.int64)
        hdr[HDR_COUNT], hdr[HDR_NEED], hdr[HDR_FATAL] = n, int(need), 1 if fatal else 0
        in_place = (base is not None and base.device == dev and base.shape[0] >= bound + 1 and
                    (n == 0 or recs.data_ptr() == base[1:].data_ptr()))
        if in_place:
            msg = base[:bound + 1]
        else:
            msg = torch.empty((bound + 1, REC), dtype=torch.uint8, device=dev)
            k = min(n, bound)
            if k:
                msg[1:1 + k] = recs[:k].to(dev)
        msg[0:1].view(torch.int64).copy_(hdr.view(1, -1))
        return msg

    def exchange(self, to_left, to_right, need=0, bases=None, fatal=False):
        """Send to_left to rank-1 and to_right to rank+1; returns (from_left, from_right).  `need` (an int) rides in
        the header to both neighbours; theirs end up in self.peer_need.  `bases`: the two buffers the records sit in
        behind a spare header row (DevicePort.send_bases), or None.  `fatal`: this rank is about to stop with an error
        and says so in the header (the neighbours' announcements end up in self.peer_fatal).  A message beyond the
        capacity is still transferred completely -- both ends stay in step -- and reported in self.error."""
        torch = self.torch
        peers = [self.rank - 1 if self.rank > 0 else None, self.rank + 1 if self.rank < self.world - 1 else None]
        if peers[0] is None and peers[1] is None:
            return None, None
        self.exchanges += 1
        dev = self._dev()
        out = [to_left, to_right]
        sends, recvs, sent_first = [], [], [0, 0]
        for k in (0, 1):
            if peers[k] is None:
                continue
            bound = self.bound_send[k]
            msg = self._message(k, out[k], None if bases is None else bases[k], bound, need, fatal)
            sent_first[k] = min(int(out[k].shape[0]), bound)
            self.sent_records += sent_first[k]
            self.padded_records += bound - sent_first[k]
            sends.append((msg, peers[k]))
            recvs.append((self._recv_rows(k, self.bound_recv[k] + 1)[:self.bound_recv[k] + 1], peers[k]))
        self._xfer(sends, recvs)
        # the one host wait: the incoming headers (behind the transfer in stream order)
        hdrs = torch.stack([self.rbuf[k][0] if peers[k] is not None else torch.zeros(REC, dtype=torch.uint8, device=dev)
                            for k in (0, 1)]).cpu().view(torch.int64).reshape(2, -1)
        self.host_waits += 1
        n_from = [int(hdrs[k, HDR_COUNT]) if peers[k] is not None else 0 for k in (0, 1)]
        self.peer_need = tuple(int(hdrs[k, HDR_NEED]) if peers[k] is not None else 0 for k in (0, 1))
        self.peer_fatal = tuple(peers[k] is not None and int(hdrs[k, HDR_FATAL]) != 0 for k in (0, 1))
        # beyond the capacity: the sender and the receiver both see it (the count is in the header), both finish the
        # transfer as announced and report it; the caller stops the run on every rank (SlabEngine._stop_together)
        n_out_max = max(int(out[k].shape[0]) for k in (0, 1) if peers[k] is not None)
        if self.cap and max(max(n_from), n_out_max) > self.cap:
            self.error = (f"face message of {max(max(n_from), n_out_max)} records exceeds the message capacity {self.cap} "
                          f"(rank {self.rank})")
        # a message that outgrew its bound: exactly the missing records, in a second transfer
        sends, recvs = [], []
        for k in (0, 1):
            if peers[k] is None:
                continue
            n_out = int(out[k].shape[0])
            if n_out > sent_first[k]:
                sends.append((out[k][sent_first[k]:].to(dev).contiguous(), peers[k]))
                self.sent_records += n_out - sent_first[k]
            if n_from[k] > self.bound_recv[k]:
                buf = self._recv_rows(k, n_from[k] + 1)
                recvs.append((buf[self.bound_recv[k] + 1:n_from[k] + 1], peers[k]))
        if sends or recvs:
            self._xfer(sends, recvs)
            if not self.stage_cpu:
                torch.cuda.current_stream(self.device).synchronize()
                self.host_waits += 1
        res = []
        for k in (0, 1):
            if peers[k] is None:
                res.append(None)
                continue
            self.bound_send[k] = next_bound(int(out[k].shape[0]), self.cap)
            self.bound_recv[k] = next_bound(n_from[k], self.cap)
            got = self.rbuf[k][1:1 + n_from[k]]
            if self.stage_cpu and self.device is not None and torch.device(self.device).type != "cpu":
                got = got.to(self.device)
            res.append(got)
        return res[0], res[1]

    def max_code(self, code: int) -> int:
        """The largest `code` of any rank, on every rank (one tiny all-reduce; not on the data path)."""
        t = self.torch.tensor([int(code)], dtype=self.torch.int64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        self.host_waits += 1                          # the code is read on the host
        return int(t.item())

    def any_rank(self, flag: bool) -> bool:
        """True on every rank if `flag` is set on any."""
        return self.max_code(1 if flag else 0) != 0

    def allreduce_sum(self, value: int) -> int:
        t = self.torch.tensor([value], dtype=self.torch.int64, device=self._dev())
        self.dist.all_reduce(t)
        return int(t.item())

    def all_to_all_records(self, parts):
        """parts[r]: records (SLAB_RECORD_DTYPE) for rank r; returns the records every rank had for this one.  Sizes
        travel in one all-gather, the payload point to point (re-partition after a grid change: rare, not tuned)."""
        torch, dist = self.torch, self.dist
        dev = self._dev()
        counts = torch.tensor([int(x.shape[0]) for x in parts], dtype=torch.int64, device=dev)
        table = [torch.zeros(self.world, dtype=torch.int64, device=dev) for _ in range(self.world)]
        dist.all_gather(table, counts)
        incoming = [int(table[r][self.rank]) for r in range(self.world)]
        sends, recvs, bufs = [], [], {}
        for r in range(self.world):
            if r == self.rank:
                continue
            if parts[r].shape[0]:
                t = torch.from_numpy(np.ascontiguousarray(parts[r]).view(np.uint8).reshape(-1, REC).copy()).to(dev)
                sends.append((t, r))
            if incoming[r]:
                bufs[r] = torch.empty((incoming[r], REC), dtype=torch.uint8, device=dev)
                recvs.append((bufs[r], r))
        self._xfer(sends, recvs)
        if not self.stage_cpu and (sends or recvs):
            torch.cuda.current_stream(self.device).synchronize()
        got = [np.ascontiguousarray(parts[self.rank])]
        for r in sorted(bufs):
            got.append(np.frombuffer(bufs[r].cpu().numpy().tobytes(), dtype=abi.SLAB_RECORD_DTYPE))
        return np.concatenate(got) if got else np.zeros(0, dtype=abi.SLAB_RECORD_DTYPE)

    def allreduce_sum_array(self, a: np.ndarray) -> np.ndarray:
        """Element-wise sum over the ranks of a small int64 array (the per-plane histogram of a re-cut)."""
        t = self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(self._dev())
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def allreduce_max(self, value: float) -> float:
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        self.dist.barrier()


# ------------------------------------------------------------------------------------------------
# the engine-shaped object the host driver (sph_amd.host.SPH) talks to
# ------------------------------------------------------------------------------------------------
class SlabEngine:
    """Stands where sph_amd.abi.Engine stands for host.SPH, for one rank of a slab-decomposed run."""

    def __init__(self, port, comm: NeighbourComm, rank: int, world: int, ghost_planes: int = 8,
                 capacity_factor: float = 1.6, rebalance_every: int = 0, max_cut_shift: Optional[int] = None,
                 rebalance_threshold: float = 1.05, overlap: bool = False, adaptive_ghost: bool = False,
                 min_ghost: int = 4, shift_window: int = 8):
        self.port, self.comm, self.rank, self.world = port, comm, rank, world
        self.ghost = int(ghost_planes)                        # the widest the ghost region ever gets (capacity, slab widths)
        # Adaptive ghost width: the halo condition is ghost >= 2 subSteps (1 + max plane shift), and the shift is what
        # the blow-up of the first frames makes it (5 planes at the spawn density of configs 4 / 5, 2 afterwards).  Each
        # FACE keeps its own width: both of its ranks tell each other, in the count message of the frame's exchange,
        # the width they need (from the shifts their density kernels measured over the last `shift_window` frames, + 2
        # planes of margin); the face takes the larger.  Narrower first on the keep side (what is in flight was packed
        # wider), wider first on the send side.  No collective, no extra message.
        self.adaptive = bool(adaptive_ghost)
        self.min_ghost = int(min_ghost)
        self.wk = [self.ghost, self.ghost]                    # planes of the left / right neighbour's territory kept as ghosts
        self.ws = [self.ghost, self.ghost]                    # planes of OWN territory sent to the left / right neighbour
        self._shifts = []
        self._shift_window = int(shift_window)
        self.width_changes = 0
        self.capacity_factor = capacity_factor
        self.rebalance_every = int(rebalance_every)           # 0: static, equal plane counts
        self.max_cut_shift = self.ghost if max_cut_shift is None else int(max_cut_shift)
        self.rebalance_threshold = float(rebalance_threshold)  # re-cut when max share / mean share exceeds it
        self.cuts = None
        self.recuts = 0
        self.imbalance = 1.0
        # overlap: a frame packs its own face messages between the two launches of its last force kernel, and the
        # exchange for the NEXT frame runs while the rest of the frame is still on the GPU (sph_slab_arm_pack)
        self.overlap = bool(overlap)
        # overlap + moving cuts: a frame packs its messages itself, so the cuts of the NEXT build have to be known when
        # the frame starts: a re-cut is planned at the end of one frame (from that frame's grid), used for the send
        # range of the next frame's in-frame pack and applied at that frame's end (step_frame below)
        self._cuts_next = None
        # A particle that crosses a WHOLE neighbouring slab in one frame belongs to a rank the face messages never reach.
        # far_guard: every frame the ranks agree (one tiny all-reduce of a flag) whether any of them measured a plane
        # shift that large; if so the frame's owned particles are routed to their owners all-to-all (SURVEY 8e's
        # fallback, _reroute) instead of being lost.
        self.far_guard = True
        self._fatal = None               # why this rank has to stop (the stop itself is collective: _agree / _stop_together)
        self.reroutes = 0
        self.band_misses = 0
        self.band_fraction = 0.0       # share of the last force kernel that ran before the pack
        self._primed = False
        self.e = port.e
        self.n_global = 0
        self.n = 0
        self.cells = 0
        self.grid = None
        self.own = None
        self._rows = []
        self._min_x = None
        self._cell = None
        self.halo_violations = 0
        self.sub_steps = 1            # of the last frame (sets the halo width the frame needed)
        self.frames = 0
        self.exchange_s = 0.0
        self.n_built = 0

    # ---- particle creation: the host feeds spawned rows chunk by chunk (host.SPH.CreateParticles)
    grid_first = True

    def alloc_particles(self, n):
        self.n_global = int(n)

    def preview_grid(self, min_x, cell_size):
        self._min_x, self._cell = np.float32(min_x), np.float32(cell_size)

    def _planes(self, x):
        rx = self.grid[0]
        plane = np.floor(((np.asarray(x, dtype=np.float32) - self._min_x) / self._cell).astype(np.float32)).astype(np.int64)
        return np.clip(plane, 0, rx - 1)

    def box_may_matter(self, lo_x, hi_x):
        lo, hi = self._planes([lo_x, hi_x])
        return hi >= self.own[0] - 1 and lo <= self.own[1]

    def add_rows(self, first_id, rows, phase=0):
        """rows: (k, 18) float32 AoS rows of global ids first_id .. first_id+k-1, all of spawn box
        `phase`; only those in the owned planes are kept (ghosts arrive with the first exchange)."""
        plane = self._planes(rows[:, 0])
        m = (plane >= self.own[0]) & (plane < self.own[1])
        ids = (int(first_id) + np.flatnonzero(m)).astype(np.uint32)
        self._rows.append(abi.records_from_aos(rows[m], ids, phase))

    def upload_particles(self, aos):
        if self._rows is None:
            raise RuntimeError("the particles are distributed already: replace the state with resync(aos, phase_of_id)")
        self.add_rows(0, np.asarray(aos, dtype=np.float32).reshape(-1, abi.PARTICLE_FLOATS))

    def set_phase_table(self, phases):
        # The table is authoritative for the static properties (restDensity, viscosity, mass), like the spawn boxes are
        # in the reference (SPH.cs:284-298): a live edit of a box reaches the local slots with apply_phase_table and the
        # records that arrive later when they are stored.  Every rank gets the same table from its own host.SPH.
        self._phases = phases
        if self._rows is None:
            self.e.set_phase_table(phases)

    def set_phase_ids(self, ids):
        pass                                          # a slab keeps the phase per slot: it travels in the records

    def apply_phase_table(self):
        if self._rows is None:
            self.e.apply_phase_table()

    # ---- grid: global resolution in, local slab out
    def resize_grid(self, rx, ry, rz, cap):
        if self.grid is not None:
            if self.grid != (rx, ry, rz, cap):
                self._new_grid = (rx, ry, rz, cap)     # the boundary cube moved: re-partition at the next set_params
            return
        self._set_grid(rx, ry, rz, cap)

    def _set_grid(self, rx, ry, rz, cap):
        self.grid = (rx, ry, rz, cap)
        self.cells = rx * ry * rz
        ranges = slab_ranges(rx, self.world)
        self.cuts = [r[0] for r in ranges] + [rx]
        x0, x1 = ranges[self.rank]
        narrowest = min(b - a for a, b in ranges)
        if narrowest < 2 * self.ghost and self.world > 1:
            raise ValueError(f"slab of {narrowest} planes is narrower than 2 x {self.ghost} ghost planes")
        if self.rebalance_every and self.world > 1 and narrowest < self.min_width:
            raise ValueError(f"moving cuts need slabs of at least 2 x {self.ghost} ghost planes + {self.max_cut_shift} "
                             f"planes of cut shift; the narrowest has {narrowest}")
        self.own = (x0, x1)
        self.wk = [self.ghost, self.ghost]
        self.ws = [self.ghost, self.ghost]
        self._shifts = []
        self.keep = self._keep_range()

    def _keep_range(self):
        return (max(self.own[0] - self.wk[0], 0), min(self.own[1] + self.wk[1], self.grid[0]))

    @property
    def band_margin(self):
        """Planes a particle may move per frame while the halo condition holds (half the ghost width)."""
        return max(min(self.ws) // 2, 1)

    # ---- adaptive ghost width
    def _needed(self):
        """Ghost width this rank asks of its faces: 2 subSteps (1 + largest shift of the last frames) + 2 planes of margin
        (a width takes a frame or two to grow); the full width until a window of frames has been seen."""
        if not self.adaptive or len(self._shifts) < self._shift_window:
            return self.ghost
        need = 2 * self.sub_steps * (1 + max(self._shifts[-self._shift_window:])) + 2
        return int(min(max(need, self.min_ghost), self.ghost))

    def _note_shift(self, shift, widths=None):
        """`shift`: the largest plane shift the density kernel measured in the frame that ran with ghost widths `widths`."""
        widths = self.wk if widths is None else widths
        self._shifts.append(int(shift))
        del self._shifts[:-self._shift_window]
        faces = [widths[k] for k in (0, 1) if (self.rank > 0 if k == 0 else self.rank < self.world - 1)]
        if faces and 2 * self.sub_steps * (1 + shift) > min(faces):
            self.halo_violations += 1
            self._shifts = []                    # back to the full ghost width at once, until a new window has been seen

    def _far_danger(self, shift):
        """Could an owned particle of this rank have crossed a whole neighbouring slab (and so belong to a rank the face
        messages do not reach)?  Necessary: a plane shift of at least that slab's width."""
        if self.world < 3 or self.cuts is None:
            return False                         # with two slabs every other rank IS the face neighbour
        widths = [self.cuts[r + 1] - self.cuts[r] for r in (self.rank - 1, self.rank + 1) if 0 <= r < self.world]
        return bool(widths) and shift >= min(widths)

    # ---- stopping together
    # A failure on one rank (a face message or a slab beyond its capacity, a HIP error) must end the run on EVERY rank:
    # a rank that simply raised would leave its neighbours waiting in their next send/recv until the process group's
    # timeout.  The failing rank keeps `_fatal` and stays in step: with more than two ranks the per-frame code
    # all-reduce (the one that also carries the far-jump flag) tells everybody, with two the header of the face
    # message does; all ranks raise at the same point of the same frame.
    def _agree(self, far: bool) -> bool:
        """More than two ranks: one all-reduce per frame.  Returns whether some rank saw a far jump; raises on every
        rank if some rank has failed."""
        if self.world < 3:
            return False
        code = self.comm.max_code(2 if self._fatal else (1 if (far and self.far_guard) else 0))
        if code >= 2:
            raise RuntimeError(self._fatal or "another rank of the slab run stopped with an error (see its message)")
        return code == 1

    def _stop_together(self):
        """After a face exchange: raise on both ranks of a face if either announced or detected a failure."""
        if self.comm.error and not self._fatal:
            self._fatal = self.comm.error
        # (more than two ranks: a rank that left here would be missed in the next frame's all-reduce, which is where
        # everybody stops -- _agree)
        if self.world < 3 and (self._fatal or any(self.comm.peer_fatal)):
            raise RuntimeError(self._fatal or "the neighbouring rank of the slab run stopped with an error (see its message)")

    def _guarded(self, fn, *args):
        """fn(*args), or two empty messages and `_fatal` set if it failed (the stop is made collective later)."""
        try:
            return fn(*args)
        except (abi.SphError, RuntimeError, MemoryError) as exc:
            self._fatal = self._fatal or f"rank {self.rank}: {exc}"
            return self.comm.empty(), self.comm.empty()

    def _reroute(self):
        """SURVEY 8e's fallback for particles that jumped further than the neighbouring slab: every rank takes its owned
        particles off the device and sends each to the rank that owns its CURRENT plane (all-to-all by searchsorted on
        the cuts, as after a grid change); each rank starts again from its owned particles, the frame's ordinary
        exchange supplies the ghosts.  Conserves every particle; collective."""
        recs = self.port.download_owned()
        self._redistribute(recs)
        self.reroutes += 1

    def _adapt(self, in_flight: bool):
        """New face widths from this rank's need and the neighbours' (comm.peer_need, same frame): both ranks of a face
        compute the same numbers.  in_flight: the records about to be appended were packed with the OLD send widths, so a
        face can only be kept as wide as that (growing takes effect on the send side first)."""
        if not self.adaptive or self.world == 1:
            return
        own = self._needed()
        changed = False
        for k in (0, 1):
            if (k == 0 and self.rank == 0) or (k == 1 and self.rank == self.world - 1):
                continue
            target = int(min(max(own, self.comm.peer_need[k], self.min_ghost), self.ghost))
            keep = min(target, self.ws[k]) if in_flight else target
            changed |= keep != self.wk[k]
            self.wk[k], self.ws[k] = keep, target
        if changed:
            rx, ry, rz, cap = self.grid
            self.keep = self._keep_range()
            self.port.reconfigure(rx, self.keep[0], self.keep[1] - self.keep[0], ry, rz, cap, self.own[0], self.own[1])
            self.width_changes += 1

    @property
    def min_width(self):
        # A new face plus the ghost width must stay inside the two slabs next to the old face, also for particles
        # that moved since the last build (at most ghost/2 - 1 planes, the halo condition): width >= 2 ghost + shift.
        return 2 * self.ghost + self.max_cut_shift

    def _plan_recut(self, base_cuts):
        """New cuts by particle count, at most max_cut_shift planes from `base_cuts`, or None if the shares are balanced
        enough.  Collective (every rank calls it in the same frame); needs the grid of the last frame."""
        rx = self.grid[0]
        hist = (np.zeros(rx, dtype=np.int64) if self._fatal else     # a failed rank stays in step until everybody stops
                self.port.owned_plane_histogram(rx, self.keep[0], self.keep[1] - self.keep[0], self.own[0], self.own[1]))
        hist = self.comm.allreduce_sum_array(hist)
        cum = np.concatenate([[0], np.cumsum(hist)])
        shares = np.diff(cum[np.asarray(base_cuts)])
        mean = max(float(cum[-1]) / self.world, 1.0)
        self.imbalance = float(shares.max()) / mean
        if self.imbalance <= self.rebalance_threshold:
            return None
        cuts = balanced_cuts(hist, base_cuts, self.min_width, self.max_cut_shift)
        return None if cuts == list(base_cuts) else cuts

    def _apply_cuts(self, cuts):
        rx, ry, rz, cap = self.grid
        self.cuts = list(cuts)
        x0, x1 = cuts[self.rank], cuts[self.rank + 1]
        self.own = (x0, x1)
        self.keep = self._keep_range()
        self.port.reconfigure(rx, self.keep[0], self.keep[1] - self.keep[0], ry, rz, cap, x0, x1)
        self.recuts += 1

    def rebalance(self):
        """Re-cut the slabs by particle count and apply it at once (classic order: must run before the frame's pack,
        while the grid of the last frame is still valid).  Collective."""
        cuts = self._plan_recut(self.cuts)
        if cuts is None:
            return False
        self._apply_cuts(cuts)
        return True

    def set_params(self, p):
        self._params = p
        if self._rows is not None:
            self._distribute()
        elif (getattr(self, "_new_grid", None) is not None or np.float32(p.minBound[0]) != self._min_x
              or np.float32(p.cellSize) != self._cell):
            self._repartition(p)
        self.e.set_params(p)

    def _repartition(self, p):
        """The boundary cube moved (SPH.cs:176-185 -> 303-346): origin, cell size or resolution of the grid changed,
        which re-defines every cell plane.  Slow path, rare: every rank takes its own particles off the
        device, the records go to the ranks that own their planes under the NEW grid, and each rank starts again from
        its owned particles alone -- the frame's ordinary exchange then supplies the ghosts, as after the initial
        distribution.  Collective."""
        recs = self.port.download_owned()
        grid = getattr(self, "_new_grid", None) or self.grid
        self._new_grid = None
        self._min_x, self._cell = np.float32(p.minBound[0]), np.float32(p.cellSize)
        self._set_grid(*grid)
        self._redistribute(recs)
        self.repartitions = getattr(self, "repartitions", 0) + 1

    def _redistribute(self, recs):
        """`recs` (this rank's owned particles) to the ranks that own their planes under the current grid and cuts."""
        rx, ry, rz, cap = self.grid
        plane = self._planes(recs["pos"][:, 0])
        dest = np.searchsorted(np.asarray(self.cuts[1:-1]), plane, side="right")
        mine = self.comm.all_to_all_records([recs[dest == r] for r in range(self.world)])
        over = mine.shape[0] > self._capacity
        if (self.world > 1 and self.comm.max_code(2 if over else 0) >= 2) or over:       # every rank stops, not just this one
            raise RuntimeError(f"re-partition: {mine.shape[0]} particles exceed this slab's capacity of {self._capacity}" if over
                               else "re-partition: another rank's share exceeds its slab's capacity (see its message)")
        x0, x1 = self.own
        self.port.configure(rx, self.keep[0], self.keep[1] - self.keep[0], ry, rz, cap, x0, x1, self._capacity)
        self.port.upload_owned(mine)
        self.n = mine.shape[0]
        self.e.n = self.n
        self._primed = False          # overlap mode: the next frame gets its ghosts the classic way
        self._cuts_next = None

    def _distribute(self):
        rx, ry, rz, cap = self.grid
        x0, x1 = self.own
        owned = np.concatenate(self._rows) if self._rows else np.zeros(0, dtype=abi.SLAB_RECORD_DTYPE)
        self._rows = None
        share = max(self.n_global // self.world, owned.shape[0], 1024)
        # room for the ghosts of both faces and, between a rebuild and its grid build, for the records just received
        # on top of last frame's ghosts: 4 ghost widths on top of the owned planes (narrow slabs: config 4 on 8 GPUs)
        width = max(x1 - x0, 1)
        factor = max(self.capacity_factor, 1.25 * (1.0 + 4.0 * self.ghost / width)) if self.world > 1 else self.capacity_factor
        capacity = int(share * factor) + 4096
        self._capacity = capacity
        self.port.configure(rx, self.keep[0], self.keep[1] - self.keep[0], ry, rz, cap, x0, x1, capacity)
        if getattr(self, "_phases", None) and hasattr(self.e, "set_phase_table"):
            self.e.set_phase_table(self._phases)       # before the upload: records are checked against it
        self.port.upload_owned(owned)
        self.n = owned.shape[0]

    def resync(self, aos, phase_of_id):
        """Lock-step parity against a single-domain run (tests): replace the distributed state by `aos`, the 72-byte
        records of ALL particles in global id order (phase_of_id: their spawn box).  Every rank keeps the rows of its
        owned planes under the current cuts; the ghosts come with the next frame's exchange, as after the initial
        distribution.  No communication."""
        aos = np.asarray(aos, dtype=np.float32).reshape(-1, abi.PARTICLE_FLOATS)
        plane = self._planes(aos[:, 0])
        m = (plane >= self.own[0]) & (plane < self.own[1])
        ids = np.flatnonzero(m).astype(np.uint32)
        if ids.shape[0] > self._capacity:
            raise RuntimeError(f"resync: {ids.shape[0]} particles exceed this slab's capacity of {self._capacity}")
        self.port.upload_owned(abi.records_from_aos(aos[m], ids, np.asarray(phase_of_id)[m]))
        self.n = int(ids.shape[0])
        self.e.n = self.n
        self._primed = False
        self._cuts_next = None
        if self.adaptive and (self.wk != [self.ghost] * 2 or self.ws != [self.ghost] * 2):
            # the measured shifts describe the state that was replaced: back to the full ghost width
            rx, ry, rz, cap = self.grid
            self.wk, self.ws, self._shifts = [self.ghost] * 2, [self.ghost] * 2, []
            self.keep = self._keep_range()
            self.port.reconfigure(rx, self.keep[0], self.keep[1] - self.keep[0], ry, rz, cap, self.own[0], self.own[1])
        self._shifts = []

    def set_delta_time(self, dt):
        self.e.set_delta_time(dt)

    # ---- the per-frame exchange
    def exchange(self):
        t0 = time.perf_counter()
        if (self.rebalance_every and self.world > 1 and self.frames > 0 and self.frames % self.rebalance_every == 0
                and not self.overlap):
            self.rebalance()
        # classic order: nothing is in flight, so the widths agreed on from LAST frame's needs apply to this frame's
        # pack, keep range and grid at once
        ran_with = list(self.wk)
        self._adapt(in_flight=False)
        left_below, right_from = self._send_range()
        to_l, to_r = self._guarded(self.port.pack, left_below, right_from)
        if self.frames > 0 and self.world > 1:
            shift = 0 if self._fatal else self.e.read_counter(abi.CTR_MAX_PLANE_SHIFT)    # came back with pack's counts: no extra sync
            self._note_shift(shift, ran_with)
            if self._agree(self._far_danger(shift)):
                self._reroute()                     # owned particles only from here; pack again for the ghosts
                to_l, to_r = self._guarded(self.port.pack, left_below, right_from)
        from_l, from_r = self.comm.exchange(to_l, to_r, need=self._needed(), bases=getattr(self.port, "send_bases", None),
                                            fatal=bool(self._fatal))
        self._stop_together()
        if self._fatal:                             # more than two ranks: everybody learns of it in the next frame's all-reduce
            return
        self.n = self.port.rebuild(self.keep[0], self.keep[1], from_l, from_r)
        self.e.n = self.n
        self.exchange_s += time.perf_counter() - t0

    def _refresh_count(self):
        """The rebuild only appends; the grid build that follows drops what does not belong to the slab,
        so the local count is known after it."""
        if hasattr(self.port, "local_count"):
            self.n = self.port.local_count()
            self.e.n = self.n
        self.n_built = self.n          # owned + ghosts as of the last grid build

    def _send_range(self, cuts=None):
        x0, x1 = self.own if cuts is None else (cuts[self.rank], cuts[self.rank + 1])
        return (x0 + self.ws[0] if self.rank > 0 else 0,
                x1 - self.ws[1] if self.rank < self.world - 1 else self.grid[0])

    def step_frame(self, sub_steps, time_step):
        self.sub_steps = int(sub_steps)
        if self._fatal and self.world > 2:
            self._agree(False)                      # raises, on every rank
        if not (self.overlap and self.world > 1):
            self.exchange()
            if self._fatal:
                return
            self.e.step_frame(sub_steps, time_step)
            self._refresh_count()
            self.frames += 1
            return
        # overlapped: frame f's messages are packed inside frame f and travel while its second half runs
        if not self._primed:
            self.exchange()                         # the very first frame needs its ghosts the classic way
            if self._fatal:
                return
            self._primed = True
        left_below, right_from = self._send_range(self._cuts_next)      # what the NEXT build's owners and ghosts need

        def frame_and_pack():
            self.port.arm_pack(left_below, right_from, self.band_margin)
            self.e.step_frame(sub_steps, time_step)     # returns once everything is enqueued (one sync: survivor count)
            self._refresh_count()
            if self.frames > 0:
                self.band_misses += self.e.read_counter(abi.CTR_BAND_MISSES)    # cached by the build: no sync
            return self.port.pack_wait()            # the band part of the frame is done; the GPU carries on

        t0 = time.perf_counter()
        to_l, to_r = self._guarded(frame_and_pack)
        self.band_fraction = getattr(self.e, "band_workgroups", 0) / max((self.n + 255) // 256, 1)
        shift = 0 if self._fatal else self.e.read_counter(abi.CTR_MAX_PLANE_SHIFT)
        self._note_shift(shift)
        if self._agree(self._far_danger(shift)):
            # some particle may belong to a rank the face messages do not reach: let the frame finish, then route every
            # owned particle to its owner; the next frame starts the classic way (ghosts from its own exchange)
            self.e.synchronize()
            if self._cuts_next is not None:
                self._apply_cuts(self._cuts_next)
            self._reroute()
            self.exchange_s += time.perf_counter() - t0
            self.frames += 1
            return
        planned = None
        if self.rebalance_every and self.frames > 0 and self.frames % self.rebalance_every == 0:
            # from THIS frame's grid (still valid: the rebuild below invalidates it), relative to the cuts about to apply
            planned = self._plan_recut(self._cuts_next or self.cuts)
        from_l, from_r = self.comm.exchange(to_l, to_r, need=self._needed(), bases=getattr(self.port, "send_bases", None),
                                            fatal=bool(self._fatal))
        self._stop_together()
        if self._fatal:
            return
        if self._cuts_next is not None:          # the records just received were packed for these cuts
            self._apply_cuts(self._cuts_next)
        self._cuts_next = planned
        self._adapt(in_flight=True)              # the records just received were packed with the widths of this frame
        self.n = self.port.rebuild(self.keep[0], self.keep[1], from_l, from_r)   # enqueued behind the frame
        self.e.n = self.n
        self.exchange_s += time.perf_counter() - t0
        self.frames += 1

    def dispatch(self, kernel_id, count):
        if kernel_id == abi.K_CLEAR_GRID:
            self.exchange()
            self.e.dispatch(kernel_id, self.e.cells)
        else:
            self.e.dispatch(kernel_id, self.n)
            if kernel_id == abi.K_BUILD_GRID:
                self._refresh_count()
        if kernel_id == abi.K_CLEAR_GRID:
            self._half_steps = 0
        elif kernel_id == abi.K_VV_HALF_STEP:
            self._half_steps = getattr(self, "_half_steps", 0) + 1
        if kernel_id == abi.K_VELOCITY_DAMPING:
            self.sub_steps = max(getattr(self, "_half_steps", 1), 1)
            self.frames += 1

    def synchronize(self):
        self.e.synchronize()

    def reset_counter(self, which=abi.CTR_COLLISIONS):
        self.e.reset_counter(which)

    def read_counter(self, which=abi.CTR_COLLISIONS):
        return self.e.read_counter(which)

    def read_counter_global(self, which=abi.CTR_COLLISIONS):
        return self.comm.allreduce_sum(self.e.read_counter(which))

    def get_timings(self):
        return self.e.get_timings()

    def owned_records(self):
        return self.port.download_owned()

    def download_particles(self, root=None):
        """GetParticleBuffer of a slab run (SPH.cs:105-110): the 72-byte records of ALL particles in global id order,
        owned particles gathered from every rank -- on every rank (root None) or on `root` only (the others get None).
        Colours come from the phase table (a slab does not carry them).  Collective."""
        recs = self.owned_records()
        if self.world > 1:
            parts = [recs if (root is None or r == root) else recs[:0] for r in range(self.world)]
            recs = self.comm.all_to_all_records(parts)
            if root is not None and self.rank != root:
                return None
        if recs.shape[0] != self.n_global or not np.array_equal(np.sort(recs["id"]), np.arange(self.n_global, dtype=np.uint32)):
            raise RuntimeError(f"gathered {recs.shape[0]} records for {self.n_global} particles: the slabs do not partition the ids")
        recs = recs[np.argsort(recs["id"], kind="stable")]
        aos = abi.aos_from_records(recs)
        self.gathered_phase = recs["phase"].astype(np.uint8)       # by id: what a checkpoint of this state must hold
        rows = getattr(self, "_phases", None)
        if rows:
            colours = np.asarray([np.asarray(r[3], dtype=np.float32).reshape(4) for r in rows], dtype=np.float32)
            known = recs["phase"] < len(rows)
            aos[known, 14:18] = colours[recs["phase"][known]]
        return aos

    def count_global(self):
        return self.comm.allreduce_sum(int(self.owned_records().shape[0]))

    def destroy(self):
        self.e.destroy()


# ------------------------------------------------------------------------------------------------
# bench.py --gpus N (N > 1): BASELINE config 5, weak scaling, 8,388,608 particles per GPU
# ------------------------------------------------------------------------------------------------
def bench_main(args, rank, local_rank, world):
    import json
    import torch
    import torch.distributed as dist
    from . import scenes

    # Rehearsal on a one-GPU box: SPH_BENCH_BACKEND=gloo SPH_BENCH_SHARE_GPU=1 runs all ranks on
    # device 0 with host-staged messages.  The real run is one rank per GPU over RCCL.
    backend = os.environ.get("SPH_BENCH_BACKEND", "nccl")
    share = os.environ.get("SPH_BENCH_SHARE_GPU")
    rehearsal = None
    if share is None and torch.cuda.device_count() < world:
        # fewer GPUs than ranks (a one-GPU box): a functional rehearsal, said so in the JSON line -- RCCL refuses
        # two ranks on one device, so the messages are host-staged over gloo
        share, backend = "1", "gloo"
    if share == "1":
        rehearsal = f"{world} ranks share {torch.cuda.device_count()} GPU(s), host-staged messages ({backend}): functional rehearsal, not a scaling measurement"
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        # a collective or send/recv that never completes must end the run (the parent then reports the failure) rather
        # than hold the node until somebody else's limit
        import datetime
        limit = datetime.timedelta(seconds=int(os.environ.get("SPH_BENCH_COMM_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, timeout=limit,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=limit)
    # default: BASELINE config 5 (weak scaling, 8,388,608 particles per GPU); --workload config4: BASELINE config 4
    # (strong scaling, 16,777,216 particles in all, --particles overrides the TOTAL)
    strong = getattr(args, "workload", "") == "config4"
    if strong:
        n_all = args.particles or 16777216
        per_gpu = n_all // world
    else:
        per_gpu = args.particles or 8388608
    eng = abi.Engine(local_rank)
    msg_cap = max(per_gpu // (2 if strong else 4), 65536)      # narrow slabs: the face band is a large part of a slab
    port = DevicePort(eng, local_rank, msg_capacity=msg_cap)
    comm = NeighbourComm(rank, world, torch.device("cuda", local_rank), backend=backend, msg_capacity=msg_cap)
    # config 5 starts balanced and stays so (G equal blocks side by side); SPH_BENCH_REBALANCE=K re-cuts the
    # slabs by particle count every K frames anyway (0 = static cuts)
    # SPH_BENCH_OVERLAP=0: classic order (pack after the frame, exchange before the next one) instead of the
    # in-frame pack whose exchange runs under the second half of the frame
    rebalance = int(os.environ.get("SPH_BENCH_REBALANCE", "0"))
    overlap = os.environ.get("SPH_BENCH_OVERLAP", "1") != "0"
    # 12 ghost planes: at the spawn density of configs 4/5 (7 x rest density) the first frames throw particles up to
    # 5 cell planes per frame (scripts/max_shift.py), and the halo condition is ghost >= 2 (1 + shift); 2-3 planes later
    ghost = int(os.environ.get("SPH_BENCH_GHOST", "12"))
    # SPH_BENCH_ADAPTIVE_GHOST=0: keep the full ghost width for the whole run (A/B)
    adaptive = os.environ.get("SPH_BENCH_ADAPTIVE_GHOST", "1") != "0"
    se = SlabEngine(port, comm, rank, world, ghost_planes=ghost, rebalance_every=rebalance, overlap=overlap,
                    adaptive_ghost=adaptive)
    sc = scenes.config4(se, n=n_all) if strong else scenes.config5(se, gpus=world, per_gpu=per_gpu)
    sc.sph.use_step_frame = True
    sc.start()
    for _ in range(args.warmup):
        sc.frame()
    eng.synchronize()
    try:
        eng.get_timings()
    except abi.SphError:
        pass
    se.exchange_s = 0.0
    warm_violations = se.halo_violations
    se.halo_violations = 0
    waits0, rounds0, exch0 = comm.host_waits, comm.rounds, comm.exchanges
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sc.frame()
    eng.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    tm = eng.get_timings()
    n_total = sc.sph.particleCount       # conservation is asserted by tests/test_slab_*.py
    viol = comm.allreduce_sum(se.halo_violations)
    warm_viol = comm.allreduce_sum(warm_violations)
    reroutes = se.reroutes
    misses = comm.allreduce_sum(se.band_misses)
    overflow = comm.allreduce_sum(eng.read_counter(abi.CTR_BIN_OVERFLOW))
    force_ms = comm.allreduce_max(tm["force_ms"])
    # after the timed region: every id owned exactly once over all ranks (count and id sum), and the ghost share
    n_owned_here, conserved = -1, None
    if os.environ.get("SPH_BENCH_GHOST_RATIO", "1") != "0":
        owned_ids = se.owned_records()["id"].astype(np.int64)
        n_owned_here = int(owned_ids.shape[0])
        totals = comm.allreduce_sum_array(np.array([n_owned_here, int(owned_ids.sum())], dtype=np.int64))
        conserved = bool(totals[0] == n_total and totals[1] == n_total * (n_total - 1) // 2)
    ghost_ratio = comm.allreduce_max((se.n_built - n_owned_here) / max(n_owned_here, 1)) if n_owned_here >= 0 else None
    faces = [se.wk[k] for k in (0, 1) if (rank > 0 if k == 0 else rank < world - 1)] or [ghost]
    widths_min = -comm.allreduce_max(-float(min(faces)))
    widths_max = comm.allreduce_max(float(max(faces)))
    # The same per-GPU load on ONE GPU, no slabs (every rank runs it on its own GPU, concurrently): what
    # "weak scaling" compares with.  bench.py --gpus 1 runs config 3 (1 M particles), a different workload.
    solo = None
    if os.environ.get("SPH_BENCH_SOLO", "1") != "0" and not strong:
        # a failure here must not cost the slab result, nor leave the other ranks waiting in the reduction
        solo_dt = float("inf")
        try:
            eng.destroy()
            eng = abi.Engine(local_rank)
            s1 = scenes.config5(eng, gpus=1, per_gpu=per_gpu)
            s1.sph.use_step_frame = True
            s1.start()
            for _ in range(args.warmup):
                s1.frame()
            eng.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                s1.frame()
            eng.synchronize()
            solo_dt = time.perf_counter() - t1
        except Exception as exc:                     # noqa: BLE001 - reported, not fatal
            print(f"[bench] rank {rank}: one-GPU reference run failed: {exc}", file=sys.stderr)
        solo_dt = comm.allreduce_max(solo_dt)
        if solo_dt != float("inf"):
            solo = per_gpu * sc.sph.subSteps * args.steps / solo_dt
    # A timed frame whose halo was too narrow is not the single-domain computation any more, and ids that are not owned
    # exactly once mean lost or duplicated particles: either makes the number invalid (the line says why, value is null,
    # the exit code is non-zero).
    invalid = None
    if viol > 0:
        invalid = f"{viol} halo violations in the timed frames (ghost region narrower than 2 subSteps (1 + plane shift))"
    if conserved is False:
        invalid = "particle ids are not owned exactly once over the ranks"
    if rank == 0:
        sub = sc.sph.subSteps
        n_local = se.n
        achieved = 64.0 * n_local / (force_ms * 1e-3) / 1e9
        # counters are per GPU: the committed one-GPU PMC pass of the same per-GPU load (config 5, G = 1) at this
        # warm-up, if there is one; the CPU baseline is timed on the one-GPU leg only (bench.py --gpus 1) and quoted
        from . import benchutil as _bench
        run = None if strong else _bench.counter_run("config5", per_gpu, args.warmup)
        cpu = None
        try:
            cpu = dict(json.load(open(os.path.join(_bench.ROOT, "profiles", "cpu_baseline_1gpu.json"))))
            cpu["sample"] = "NOT re-timed in this run: " + cpu.get("sample", "") + " (one-GPU leg, profiles/cpu_baseline_1gpu.json)"
        except Exception:
            cpu = None
        print(json.dumps({
            "metric": "particle-updates/sec", "value": None if invalid else n_total * sub * args.steps / dt,
            "invalid_reason": invalid, "unit": "particle-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"config4: {n_total} particles in all (strong scaling), " if strong else
                                    f"config5: {per_gpu} particles per GPU, {n_total} total, ") +
                                   f"two-phase, x-slabs with "
                                   f"{ghost} ghost planes, one neighbour message per face per frame ({backend} send/recv)",
                       "particles": n_total, "sub_steps": sub, "parallelism": f"slab{world}",
                       "world_size": dist.get_world_size(), "backend": dist.get_backend(), "rehearsal": rehearsal,
                       "self_launched": os.environ.get("SPH_BENCH_SELF_LAUNCHED") == "1",
                       "halo_violations": viol, "halo_violations_warmup": warm_viol, "far_jump_reroutes": reroutes,
                       "bin_overflow_cells": overflow, "every_id_owned_once": conserved,
                       # blocking host waits of an overlapped frame on rank 0: the pack event, the incoming message
                       # headers (one per exchange; + one when a message outgrew its predicted size), the survivor count
                       "host_waits_per_frame_rank0": round(2 + (comm.host_waits - waits0) / max(args.steps, 1), 3),
                       "transfer_rounds_per_exchange_rank0": round((comm.rounds - rounds0) / max(comm.exchanges - exch0, 1), 3),
                       "adaptive_ghost": se.adaptive, "ghost_planes_at_end": [int(widths_min), int(widths_max)],
                       "ghost_to_owned_particles_worst_rank": ghost_ratio,
                       "rebalance_every": se.rebalance_every, "recuts": se.recuts,
                       "exchange_overlapped": se.overlap, "band_fraction_rank0": round(se.band_fraction, 4),
                       "band_misses": misses,
                       "exchange_ms_per_step_rank0": se.exchange_s / args.steps * 1e3,
                       "one_gpu_same_load": None if solo is None else {
                           "value": solo, "workload": f"config5, {per_gpu} particles, one GPU, no slabs (slowest of the "
                                                      f"{world} GPUs, run concurrently)",
                           "weak_scaling_efficiency": n_total * sub * args.steps / dt / (world * solo)}},
            "roofline": {"bound": "hbm", "measured_limiter": _bench.MEASURED_LIMITER,
                         "kernel": "k_force_xsph_hits (CS_ForceXSPH), slowest rank", "achieved": achieved,
                         "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": _bench.measured_traffic(run, "k_force_xsph_hits"),
                         "traffic_from": None if run is None else f"one GPU, same per-GPU load, warmup {run['warmup']} ({run.get('source', 'profiles/')})",
                         "alg_bytes_per_particle": 64, "particles_rank0": n_local, "avg_launch_ms": force_ms},
            "cpu_baseline": cpu,
            "stage_ms_rank0": {k: round(v, 4) for k, v in tm.items() if k.endswith("_ms")},
        }))
    eng.destroy()
    dist.destroy_process_group()
    if invalid:
        raise SystemExit(3)
