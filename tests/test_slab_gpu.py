"""Slab decomposition on the GPU: two ranks sharing the one MI355X of the test box (gloo backend,
messages staged through the host) must reproduce a single-engine run BIT FOR BIT, through the
C ABI's sph_slab_* entry points."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sph_amd import abi, slab
from tests.test_slab_gloo import _free_port, edited_tank, moving_tank, small_tank

pytestmark = pytest.mark.gpu


def long_tank(engine, n=0, sub_steps=1):
    """640 x 4 x 1 units (2560 x 16 x 4 cells) with two thin bars of fluid 600 units long: even after the bars have
    spread over a few dozen cell rows, a rank's part of a row is several workgroups long, so the workgroups near a
    slab face are a small part of all."""
    from sph_amd import scenes
    from sph_amd.host import SPH, Scene, SpawnBox, Time, Transform
    cube = Transform(position=(0, 2, 0), localScale=(640, 4, 1))
    boxes = [SpawnBox(48000, (600.0, 0.5, 0.45), 20, 0.1, 0.2, position=(0, 2.75, 0.0)),
             SpawnBox(24000, (600.0, 0.25, 0.45), 10, 0.1, 0.2, position=(0, 1.125, 0.0))]
    f = dict(scenes.SCENE_COMMON)
    f.update(gridResolution=4, subSteps=sub_steps)
    sph = SPH(engine, **f, surfaceTensionCoefficient=0.01, xsphEpsilon=0.2, boundaryCube=cube, spawnBoxes=boxes)
    return Scene(sph, Time(), [])


def _worker(rank, world, port, frames, out_dir, n, ghost, drop_phases, rebalance_every=0, sub_steps=1, overlap=False,
            tank="small", adaptive=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = abi.Engine(0)
    dport = slab.DevicePort(eng, 0, msg_capacity=n)
    comm = slab.NeighbourComm(rank, world, torch.device("cuda", 0), backend="gloo")
    se = slab.SlabEngine(dport, comm, rank, world, ghost_planes=ghost, rebalance_every=rebalance_every,
                         max_cut_shift=4, rebalance_threshold=1.0, overlap=overlap, adaptive_ghost=adaptive)
    if drop_phases:
        # records arrive with spawn-box indices the phase table does not know (200, 201 against a table of two rows):
        # they keep the static properties they carry, and the engine has to notice (PhaseCheck) and keep gathering
        # mass and viscosity per particle instead of looking them up by phase
        add = se.add_rows
        se.add_rows = lambda first_id, rows, phase=0: add(first_id, rows, 200 + phase)
    if tank in ("drift", "rock"):
        sc = moving_tank(se, tank, n=n)
    elif tank == "edit":
        sc = edited_tank(se, n=n)
    else:
        sc = (long_tank if tank == "long" else small_tank)(se, n=n, sub_steps=sub_steps)
    sc.sph.use_step_frame = True
    sc.start()
    counts = []
    for _ in range(frames):
        sc.frame()
        counts.append(se.count_global())
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), se.owned_records())
    np.save(os.path.join(out_dir, f"meta{rank}.npy"), np.array(counts + [se.halo_violations + se.band_misses]))
    np.save(os.path.join(out_dir, f"cuts{rank}.npy"), np.array(se.cuts + [se.recuts]))
    np.save(os.path.join(out_dir, f"band{rank}.npy"), np.array([se.band_fraction]))
    owned = se.owned_records().shape[0]
    np.save(os.path.join(out_dir, f"widths{rank}.npy"), np.array(se.wk + se.ws + [se.width_changes, se.n_built, owned]))
    eng.destroy()
    dist.destroy_process_group()


# (6, 4): six slabs of 8 planes with 4 ghost planes; the outer slabs hold NO particle at first
# (an empty slab must step like any other) and fill up as the fluid spreads
@pytest.mark.parametrize("world,ghost,drop_phases", [(1, 8, False), (2, 8, False), (3, 8, False), (6, 4, False),
                                                     (2, 8, True)])
def test_slabs_equal_single_engine_bitwise(tmp_path, world, ghost, drop_phases):
    frames, n = 8, 3000
    eng = abi.Engine(0)
    sc = small_tank(eng, n=n)
    sc.sph.use_step_frame = True
    sc.run(frames)
    want = eng.download_particles()
    eng.destroy()
    n = want.shape[0]

    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), n, ghost, drop_phases), nprocs=world, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    meta = np.load(tmp_path / "meta0.npy")
    assert meta[:-1].tolist() == [n] * frames, "particles not conserved across slabs"
    assert meta[-1] == 0, "halo too narrow for the particle speeds (halo_violations)"
    assert np.array_equal(np.sort(recs["id"]), np.arange(n, dtype=np.uint32))
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), f"{world} slabs differ from one engine"


@pytest.mark.parametrize("world,overlap", [(2, False), (3, False), (2, True), (3, True)])
def test_moving_cuts_equal_single_engine_bitwise(tmp_path, world, overlap):
    """Load-balanced cuts (sph_slab_plane_counts + sph_slab_configure between frames): the faces move every 2 frames,
    ownership changes hands through the ordinary face messages, the result stays that of one engine bit for bit."""
    frames, n = 9, 3000
    eng = abi.Engine(0)
    sc = small_tank(eng, n=n)
    sc.sph.use_step_frame = True
    sc.run(frames)
    want = eng.download_particles()
    eng.destroy()
    n = want.shape[0]
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), n, 4, False, 2, 1, overlap), nprocs=world, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    meta = np.load(tmp_path / "meta0.npy")
    cuts = [np.load(tmp_path / f"cuts{r}.npy").tolist() for r in range(world)]
    assert all(c == cuts[0] for c in cuts) and cuts[0][-1] >= 1, f"cuts {cuts}"
    assert meta[:-1].tolist() == [n] * frames and meta[-1] == 0
    assert np.array_equal(np.sort(recs["id"]), np.arange(n, dtype=np.uint32))
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), f"{world} re-cut slabs differ from one engine"


def test_two_sub_steps_two_slabs_bitwise(tmp_path):
    """subSteps = 2, no exchange between the sub-steps: 8 ghost planes >= 2 x 2 x (1 + shift) keep the owned planes
    clear of the ghost band that goes stale; halo_violations (measured, owned particles) must stay 0."""
    frames, n = 5, 3000
    eng = abi.Engine(0)
    sc = small_tank(eng, n=n, sub_steps=2)
    sc.sph.use_step_frame = True
    sc.run(frames)
    want = eng.download_particles()
    eng.destroy()
    n = want.shape[0]
    mp.spawn(_worker, args=(2, _free_port(), frames, str(tmp_path), n, 8, False, 0, 2), nprocs=2, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(2)])
    meta = np.load(tmp_path / "meta0.npy")
    assert meta[:-1].tolist() == [n] * frames and meta[-1] == 0, meta
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), "two sub-steps: 2 slabs differ from one engine"


@pytest.mark.parametrize("world,ghost,sub_steps,drop_phases", [(2, 8, 1, False), (3, 8, 1, False), (6, 4, 1, False),
                                                                (2, 8, 2, False), (2, 8, 1, True)])
def test_overlapped_exchange_bitwise(tmp_path, world, ghost, sub_steps, drop_phases):
    """sph_slab_arm_pack: every frame packs its own face messages between the two launches of its last force kernel
    (band workgroups first) and the exchange runs while the rest of the frame is still on the GPU.  Same results as
    one engine, bit for bit; no particle missed by the band (SPH_CTR_BAND_MISSES)."""
    frames, n = 8, 3000
    eng = abi.Engine(0)
    sc = small_tank(eng, n=n, sub_steps=sub_steps)
    sc.sph.use_step_frame = True
    sc.run(frames)
    want = eng.download_particles()
    eng.destroy()
    n = want.shape[0]
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), n, ghost, drop_phases, 0, sub_steps, True),
             nprocs=world, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    meta = np.load(tmp_path / "meta0.npy")
    assert meta[:-1].tolist() == [n] * frames, "particles not conserved across slabs"
    assert meta[-1] == 0, "halo violations or particles missed by the band"
    assert np.array_equal(np.sort(recs["id"]), np.arange(n, dtype=np.uint32))
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), f"{world} overlapped slabs differ from one engine"


def test_overlapped_exchange_really_splits_the_force_kernel(tmp_path):
    """A tank with long, dense cell rows: only the workgroups near the slab faces run before the pack (band_fraction
    well below 1), the rest after it -- and the result is still one engine's, bit for bit."""
    frames, world = 6, 2
    eng = abi.Engine(0)
    sc = long_tank(eng)
    sc.sph.use_step_frame = True
    sc.run(frames)
    want = eng.download_particles()
    assert eng.read_counter(abi.CTR_BIN_OVERFLOW) == 0
    eng.destroy()
    n = want.shape[0]
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), n, 12, False, 0, 1, True, "long"), nprocs=world, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    meta = np.load(tmp_path / "meta0.npy")
    band = [float(np.load(tmp_path / f"band{r}.npy")[0]) for r in range(world)]
    assert all(0.0 < b < 0.6 for b in band), f"band fractions {band}: the force kernel was not really split"
    assert meta[:-1].tolist() == [n] * frames and meta[-1] == 0, meta
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), "split force kernel: slabs differ from one engine"


@pytest.mark.parametrize("motion,overlap", [("drift", False), ("rock", False), ("rock", True)])
def test_moving_boundary_cube_repartitions_bitwise(tmp_path, motion, overlap):
    """The boundary cube moves every frame (SPH.cs:176-185): the grid changes under the slabs and the driver
    re-partitions; two ranks still reproduce one engine bit for bit (also when the exchange is overlapped: a
    re-partition drops the records that were waiting for the next grid build)."""
    frames, n, world = 6, 3000, 2
    eng = abi.Engine(0)
    sc = moving_tank(eng, motion, n=n)
    sc.sph.use_step_frame = True
    sc.run(frames)
    want = eng.download_particles()
    eng.destroy()
    n = want.shape[0]
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), n, 8, False, 0, 1, overlap, motion), nprocs=world, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    meta = np.load(tmp_path / "meta0.npy")
    assert meta[:-1].tolist() == [n] * frames, "particles not conserved across slabs"
    assert np.array_equal(np.sort(recs["id"]), np.arange(n, dtype=np.uint32))
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), f"moving cube ({motion}): 2 slabs differ from one engine"


def test_long_run_overlapped_slabs_bitwise(tmp_path):
    """120 frames of the long tank on 4 ranks with the overlapped exchange: spawn-density blow-up, fall, impact on the
    floor, settling against the walls -- conserved, no halo violation, no particle missed by the band, and the same
    bits as one engine at the end (scripts/soak_slab.py runs the same check for longer)."""
    frames, world = 120, 4
    eng = abi.Engine(0)
    sc = long_tank(eng)
    sc.sph.use_step_frame = True
    sc.run(frames)
    want = eng.download_particles()
    assert np.isfinite(want).all() and eng.read_counter(abi.CTR_BIN_OVERFLOW) == 0
    eng.destroy()
    n = want.shape[0]
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), n, 8, False, 0, 1, True, "long"), nprocs=world, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    meta = np.load(tmp_path / "meta0.npy")
    assert meta[:-1].tolist() == [n] * frames and meta[-1] == 0, meta[-8:]
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), "120 frames: 4 overlapped slabs differ from one engine"


@pytest.mark.parametrize("world,overlap", [(2, False), (4, True)])
def test_adaptive_ghost_width_bitwise(tmp_path, world, overlap):
    """Ghost widths that follow the measured plane shifts (per face, agreed through the count message): 12 planes
    through the blow-up of the first frames, then what 2 (1 + shift) + 2 asks for.  Same bits as one engine, no halo
    violation, no particle missed by the band, and clearly fewer ghost particles at the end than with the full width."""
    frames = 40
    eng = abi.Engine(0)
    sc = long_tank(eng)
    sc.sph.use_step_frame = True
    sc.run(frames)
    want = eng.download_particles()
    eng.destroy()
    n = want.shape[0]
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), n, 12, False, 0, 1, overlap, "long", True), nprocs=world,
             join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    meta = np.load(tmp_path / "meta0.npy")
    assert meta[:-1].tolist() == [n] * frames and meta[-1] == 0, meta[-8:]
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), "adaptive ghost width: slabs differ from one engine"
    w = [np.load(tmp_path / f"widths{r}.npy").tolist() for r in range(world)]     # wkL wkR wsL wsR changes n_built owned
    for r in range(world - 1):
        assert w[r][1] == w[r + 1][0] and w[r][3] == w[r + 1][2], f"face {r}|{r + 1}: the ranks disagree: {w}"
        assert w[r][1] < 12, f"face {r}|{r + 1} never narrowed: {w}"
    print("ghost / owned particles per rank:", [round((x[5] - x[6]) / max(x[6], 1), 3) for x in w], "widths", [x[:2] for x in w])


def test_plane_counts_equal_download():
    """sph_slab_plane_counts (from the cell table) against the planes of the downloaded records."""
    dist.init_process_group("gloo", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{_free_port()}")
    try:
        eng = abi.Engine(0)
        se = slab.SlabEngine(slab.DevicePort(eng, 0, msg_capacity=4096),
                             slab.NeighbourComm(0, 1, torch.device("cuda", 0), backend="gloo"), 0, 1)
        sc = small_tank(se, n=3000)
        sc.sph.use_step_frame = True
        sc.start()
        sc.run(3)
        rx = se.grid[0]
        counts = eng.slab_plane_counts(rx)
        cs, ce = np.empty(eng.cells, dtype=np.int32), np.empty(eng.cells, dtype=np.int32)
        eng._ck(eng.lib.sph_download_grid(eng.h, cs.ctypes.data, ce.ctypes.data, None, None))
        assert counts.sum() == eng.particle_count()
        per_cell = (ce - cs).astype(np.int64)          # reference cell order: x fastest (compute:127)
        assert np.array_equal(counts, per_cell.reshape(-1, rx).sum(axis=0))
        with pytest.raises(abi.SphError):
            eng.slab_plane_counts(rx + 1)
        eng.destroy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap", [(2, False), (3, True)])
def test_live_spawn_box_edit_bitwise(tmp_path, world, overlap):
    """A live edit of a spawn box (viscosity of phase 0 at frame 2; rest density and mass of phase 1 at frame 3:
    tests/test_slab_gloo._BoxEdit) reaches every slab like it reaches one engine (SPH.cs:284-298): local slots through
    sph_apply_phase_table, records that were packed before the edit when they are stored.  Mass differs between the
    phases afterwards, so the run also leaves the uniform-mass paths mid-way.  Bit-identical, static fields included."""
    frames, n = 8, 3000
    eng = abi.Engine(0)
    sc = edited_tank(eng, n=n)
    sc.sph.use_step_frame = True
    sc.run(frames)
    want = eng.download_particles()
    eng.destroy()
    n = want.shape[0]
    assert set(np.unique(want[:, 13]).tolist()) == {np.float32(0.2), np.float32(0.25)}
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), n, 8, False, 0, 1, overlap, "edit"), nprocs=world, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    meta = np.load(tmp_path / "meta0.npy")
    assert meta[:-1].tolist() == [n] * frames and meta[-1] == 0, meta
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), f"live edit: {world} slabs differ from one engine"


# ---------------------------------------------------------------------------------------------- checkpoints
def _ckpt_worker(rank, world, port, out_dir, mode, n, overlap):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sph_amd import statefile
    eng = abi.Engine(0)
    se = slab.SlabEngine(slab.DevicePort(eng, 0, msg_capacity=n), slab.NeighbourComm(rank, world, torch.device("cuda", 0), backend="gloo"),
                         rank, world, ghost_planes=8, overlap=overlap, adaptive_ghost=True)
    sc = small_tank(se, n=n)
    sc.sph.use_step_frame = True
    sc.start()
    if mode == "save":
        sc.run(4)
        meta = statefile.save_state(sc, os.path.join(out_dir, "slab"))
        assert (meta is not None) == (rank == 0)
    else:
        sc.run(2)
        statefile.load_state(sc, os.path.join(out_dir, "single"))
    sc.run(4)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), se.owned_records())
    np.save(os.path.join(out_dir, f"meta{rank}.npy"), np.array([se.halo_violations + se.band_misses]))
    eng.destroy()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap", [(2, True), (3, False)])
def test_checkpoints_move_between_one_gpu_and_slabs(tmp_path, world, overlap):
    """A raw-state file (sph_amd.statefile) written by one engine is continued bit for bit by 2 / 3 slabs, and the file
    the slabs write (rank 0, from the gathered owned particles) is byte for byte the one-engine file of that frame."""
    import json
    from sph_amd import statefile
    n = 3000
    eng = abi.Engine(0)
    sc = small_tank(eng, n=n)
    sc.sph.use_step_frame = True
    sc.run(4)
    single = statefile.save_state(sc, str(tmp_path / "single"))
    sc.run(4)
    want = eng.download_particles()
    eng.destroy()
    for mode in ("save", "load"):
        mp.spawn(_ckpt_worker, args=(world, _free_port(), str(tmp_path), mode, want.shape[0], overlap), nprocs=world, join=True)
        recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
        assert sum(int(np.load(tmp_path / f"meta{r}.npy")[0]) for r in range(world)) == 0
        got = abi.aos_from_records(recs[np.argsort(recs["id"])])
        for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
            assert np.array_equal(got[:, sl], want[:, sl]), f"{mode}: {world} slabs do not continue the one-engine run"
        if mode == "save":
            for ext in (".bin", ".phase"):
                assert open(tmp_path / ("slab" + ext), "rb").read() == open(tmp_path / ("single" + ext), "rb").read(), ext
            assert json.load(open(tmp_path / "slab.json")) == json.loads(json.dumps(single))


def test_one_message_protocol_on_device_buffers_in_place():
    """NeighbourComm.exchange with DEVICE buffers and no host staging (the RCCL configuration), two ranks played by two
    threads of this process whose transfers are copies on the GPU: the message is the send buffer itself (header row +
    records, no copy), sizes agree on both ends without being announced, a message that outgrew its predicted size is
    completed by exactly one more transfer, and what arrives is what was packed."""
    import threading
    dev = torch.device("cuda", 0)
    cap, frames = 6000, 9
    sizes = {0: [100, 110, 4500, 4600, 0, 30, 31, 5999, 6000], 1: [7, 3000, 3100, 3150, 3151, 0, 0, 5, 6000]}
    box, bar, errors = {}, threading.Barrier(2), []

    class LoopComm(slab.NeighbourComm):
        def _xfer(self, sends, recvs):
            self.rounds += 1
            for t, peer in sends:
                box[(self.rank, peer)] = t
            bar.wait()
            for t, peer in recvs:
                src = box[(peer, self.rank)]
                assert tuple(src.shape) == tuple(t.shape), (src.shape, t.shape)
                assert src.device.type == "cuda" and t.device.type == "cuda"
                t.copy_(src)
            torch.cuda.synchronize()
            bar.wait()

    def run(rank):
        try:
            torch.cuda.set_device(0)
            comm = LoopComm(rank, 2, dev, backend="nccl", msg_capacity=cap)
            bases = [torch.zeros((cap + 1, slab.REC), dtype=torch.uint8, device=dev) for _ in range(2)]
            side = 1 if rank == 0 else 0                       # rank 0 talks to its right neighbour, rank 1 to its left
            g = torch.Generator(device="cpu").manual_seed(100 + rank)
            for f in range(frames):
                n = sizes[rank][f]
                payload = torch.randint(0, 256, (n, slab.REC), dtype=torch.uint8, generator=g)
                bases[side][1:1 + n] = payload.to(dev)
                out = [bases[0][1:1], bases[1][1:1]]
                out[side] = bases[side][1:1 + n]
                rounds0 = comm.rounds
                got = comm.exchange(out[0], out[1], need=10 * rank + f, bases=bases)[side]
                # what the peer packed this frame (same generator sequence, replayed here)
                peer = 1 - rank
                gp = torch.Generator(device="cpu").manual_seed(100 + peer)
                for ff in range(f + 1):
                    want = torch.randint(0, 256, (sizes[peer][ff], slab.REC), dtype=torch.uint8, generator=gp)
                assert got.device.type == "cuda" and got.shape[0] == sizes[peer][f]
                assert torch.equal(got.cpu(), want), f"rank {rank} frame {f}: payload differs"
                assert comm.peer_need[side] == 10 * peer + f
                last = {0: slab.FIRST_BOUND, 1: slab.FIRST_BOUND} if f == 0 else \
                    {rank: slab.next_bound(sizes[rank][f - 1], cap), peer: slab.next_bound(sizes[peer][f - 1], cap)}
                overflow = n > last[rank] or sizes[peer][f] > last[peer]
                assert comm.rounds - rounds0 == (2 if overflow else 1), f"rank {rank} frame {f}: {comm.rounds - rounds0} rounds"
        except BaseException as exc:   # noqa: BLE001 - reported in the main thread
            errors.append(exc)
            bar.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
