"""world_size-2 slab decomposition on CPU (gloo): the slab driver of sph_amd/slab.py, run on
oracle-backed ports, must reproduce a single-domain oracle run bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sph_amd import abi, scenes, slab
from sph_amd.host import SPH, Scene, SpawnBox, Time, Transform


def small_tank(engine, n=1500, sub_steps=1):
    """A tank 12 x 3 x 2 units (48 x 12 x 8 cells of 0.25); fluid straddling the slab cuts of a
    2-way (x = 0) and a 3-way (x = -2, +2) decomposition, moving across them."""
    cube = Transform(position=(0, 1.5, 0), localScale=(12, 3, 2))
    # ~1.5 x rest number density (the reference's own scenes spawn at 8 x and throw a few particles
    # 4 cells per frame, which needs more ghost planes than this little tank has room for)
    na, nb = n * 2 // 3, n // 3
    sa = min((na / 150.0 / 4.0) ** 0.5, 1.9)
    sb = min((nb / 75.0 / 2.0) ** 0.5, 1.9)
    boxes = [SpawnBox(na, (4.0, sa, sa), 20, 0.1, 0.2, position=(-0.5, 1.0, 0.0), initialVelocity=(1.5, 0, 0)),
             SpawnBox(nb, (2.0, sb, sb), 10, 0.05, 0.2, position=(1.5, 1.9, 0.0), initialVelocity=(-2.0, 0, 0))]
    f = dict(scenes.SCENE_COMMON)
    f.update(gridResolution=8, subSteps=sub_steps)          # min axis 2 / 8 -> cell 0.25
    sph = SPH(engine, **f, surfaceTensionCoefficient=0.1, xsphEpsilon=0.3, boundaryCube=cube, spawnBoxes=boxes)
    return Scene(sph, Time(), [])


class _CubeMotion:
    """Moves the boundary cube every frame (the reference only does this by hand in the editor; SPH.cs:176-185 reacts):
    'drift' translates it along x (grid origin changes, resolution stays), 'rock' rotates it about z (the AABB and
    with it the resolution change)."""

    def __init__(self, cube, time, kind):
        self.cube, self.time, self.kind = cube, time, kind

    def Update(self):
        from sph_amd.host import quat_from_axis_angle
        f = self.time.frame
        if self.kind == "drift":
            self.cube.position = np.array([0.004 * f, 1.5, 0.0], dtype=np.float32)
        else:
            self.cube.rotation = quat_from_axis_angle((0, 0, 1), 0.6 * np.sin(0.7 * f))


def moving_tank(engine, kind, n=1500):
    sc = small_tank(engine, n=n)
    sc.movers.append(_CubeMotion(sc.sph.boundaryCube, sc.time, kind))
    return sc


class _BoxEdit:
    """A live Inspector edit of the spawn boxes (SpawnBox.cs:11-17), as SPH.cs:284-298 picks it up on the next Update:
    from frame 2 on phase 0 is more viscous, from frame 3 on phase 1 is denser and heavier."""

    def __init__(self, sph, time):
        self.sph, self.time = sph, time

    def Update(self):
        if self.time.frame == 2:
            self.sph.spawnBoxes[0].viscosity = 0.25
        if self.time.frame == 3:
            self.sph.spawnBoxes[1].restDensity = 12.0
            self.sph.spawnBoxes[1].particleMass = 0.25


def edited_tank(engine, n=1500):
    sc = small_tank(engine, n=n)
    sc.movers.append(_BoxEdit(sc.sph, sc.time))
    return sc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def flying_tank(engine, n=1500):
    """small_tank plus a handful of particles shot along x at 26 cell planes per frame: from the first slab of a 4-way
    decomposition (12 planes each) they land in the third -- a rank the face messages never reach -- and keep crossing
    slabs (they bounce off the end walls)."""
    sc = small_tank(engine, n=n)
    sc.sph.spawnBoxes.append(SpawnBox(6, (0.4, 0.4, 0.4), 20, 0.1, 0.2, position=(-5.4, 2.4, 0.0), initialVelocity=(325.0, 0, 0)))
    return sc


def _worker(rank, world, port, frames, out_dir, step_frame, ghost=8, rebalance_every=0, sub_steps=1, overlap=False,
            motion=None, adaptive=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.ora import Oracle
    from tests.slab_oracle_port import OraclePort
    oport = OraclePort(Oracle())
    comm = slab.NeighbourComm(rank, world, torch.device("cpu"), backend="gloo")
    se = slab.SlabEngine(oport, comm, rank, world, ghost_planes=ghost, rebalance_every=rebalance_every,
                         max_cut_shift=4, rebalance_threshold=1.0, overlap=overlap, adaptive_ghost=adaptive, shift_window=3)
    orig = se.set_params

    def set_params(p):
        oport.set_params_ref(p)
        orig(p)
    se.set_params = set_params
    sc = (edited_tank(se) if motion == "edit" else flying_tank(se) if motion == "fly" else moving_tank(se, motion) if motion
          else small_tank(se, sub_steps=sub_steps))
    sc.sph.use_step_frame = step_frame
    sc.start()
    counts = []
    for _ in range(frames):
        sc.frame()
        counts.append(se.count_global())
    recs = se.owned_records()
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), recs)
    np.save(os.path.join(out_dir, f"counts{rank}.npy"), np.array(counts))
    np.save(os.path.join(out_dir, f"cuts{rank}.npy"), np.array(se.cuts + [se.recuts]))
    np.save(os.path.join(out_dir, f"repart{rank}.npy"), np.array([getattr(se, "repartitions", 0)]))
    np.save(os.path.join(out_dir, f"widths{rank}.npy"), np.array(se.wk + se.ws + [se.width_changes, se.halo_violations]))
    np.save(os.path.join(out_dir, f"comm{rank}.npy"), np.array([comm.exchanges, comm.rounds, comm.host_waits, se.reroutes,
                                                                 comm.sent_records, comm.padded_records]))
    dist.destroy_process_group()


@pytest.mark.parametrize("step_frame", [True, False])
def test_two_slabs_equal_single_domain(tmp_path, oracle_built, step_frame):
    frames = 6
    from oracle.ora import Oracle
    ref = Oracle()
    sc = small_tank(ref)
    sc.run(frames)
    want = ref.download_particles()
    n = want.shape[0]

    mp.spawn(_worker, args=(2, _free_port(), frames, str(tmp_path), step_frame), nprocs=2, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(2)])
    counts = np.load(tmp_path / "counts0.npy")
    assert counts.tolist() == [n] * frames                     # particles conserved every frame
    assert recs.shape[0] == n and np.array_equal(np.sort(recs["id"]), np.arange(n, dtype=np.uint32))
    recs = recs[np.argsort(recs["id"])]
    got = abi.aos_from_records(recs)
    per_rank = [np.load(tmp_path / f"rank{r}.npy").shape[0] for r in range(2)]
    assert min(per_rank) > 0.1 * n                              # both slabs really hold fluid
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), "slab run differs from the single-domain run"


@pytest.mark.parametrize("step_frame", [True, False])
def test_two_sub_steps_two_slabs_equal_single_domain(tmp_path, oracle_built, step_frame):
    """subSteps = 2 with no ghost refresh between the sub-steps: the ghost region is wide enough (8 planes >=
    2 x 2 x (1 + shift 1)) for the owned planes to stay clear of the band that goes stale."""
    frames = 4
    from oracle.ora import Oracle
    ref = Oracle()
    sc = small_tank(ref, sub_steps=2)
    sc.run(frames)
    want = ref.download_particles()
    n = want.shape[0]
    mp.spawn(_worker, args=(2, _free_port(), frames, str(tmp_path), step_frame, 8, 0, 2), nprocs=2, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(2)])
    assert np.load(tmp_path / "counts0.npy").tolist() == [n] * frames
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), "two sub-steps: slab run differs from the single-domain run"


def test_overlapped_driver_equals_single_domain(tmp_path, oracle_built):
    """The overlap protocol of the driver (first frame classic, then arm -> step -> pack_wait -> exchange -> rebuild at
    the END of every frame) on oracle ports: same particles, same results."""
    frames = 6
    from oracle.ora import Oracle
    ref = Oracle()
    sc = small_tank(ref)
    sc.run(frames)
    want = ref.download_particles()
    n = want.shape[0]
    mp.spawn(_worker, args=(3, _free_port(), frames, str(tmp_path), True, 4, 0, 1, True), nprocs=3, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(3)])
    assert np.load(tmp_path / "counts0.npy").tolist() == [n] * frames
    assert np.array_equal(np.sort(recs["id"]), np.arange(n, dtype=np.uint32))
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), "overlapped slab run differs from the single-domain run"


@pytest.mark.parametrize("world,overlap,step_frame", [(2, False, True), (2, False, False), (3, True, True)])
def test_adaptive_ghost_width_equals_single_domain(tmp_path, oracle_built, world, overlap, step_frame):
    """Per-face ghost widths that follow the measured plane shifts (need rides in the count message, no collective):
    the faces start 8 planes wide, narrow to what 2 (1 + shift) + 2 asks for after a window of frames, both ranks of a
    face agree, and the result is still the single-domain run bit for bit."""
    frames = 10
    from oracle.ora import Oracle
    ref = Oracle()
    sc = small_tank(ref)
    sc.run(frames)
    want = ref.download_particles()
    n = want.shape[0]
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), step_frame, 8, 0, 1, overlap, None, True),
             nprocs=world, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    assert np.load(tmp_path / "counts0.npy").tolist() == [n] * frames
    assert np.array_equal(np.sort(recs["id"]), np.arange(n, dtype=np.uint32))
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), "adaptive ghost width: slab run differs from the single-domain run"
    w = [np.load(tmp_path / f"widths{r}.npy").tolist() for r in range(world)]       # wkL wkR wsL wsR changes violations
    assert all(x[5] == 0 for x in w), f"halo violations {w}"
    assert sum(x[4] for x in w) >= world, f"the widths never changed: {w}"
    for r in range(world - 1):                                                       # the two ranks of a face agree
        assert w[r][1] == w[r + 1][0] and w[r][3] == w[r + 1][2], f"face {r}|{r + 1}: {w[r]} vs {w[r + 1]}"
        assert 4 <= w[r][1] < 8, f"face {r}|{r + 1} did not narrow: {w[r]}"


@pytest.mark.parametrize("motion", ["drift", "rock"])
def test_moving_boundary_cube_repartitions_and_equals_single_domain(tmp_path, oracle_built, motion):
    """The cube moves every frame: the grid (origin, or origin + cell size + resolution) changes under the slabs, the
    driver re-partitions (download, all-to-all by new owner, upload) and the result is still the single-domain run."""
    frames = 5
    from oracle.ora import Oracle
    ref = Oracle()
    sc = moving_tank(ref, motion)
    sc.run(frames)
    want = ref.download_particles()
    n = want.shape[0]
    mp.spawn(_worker, args=(2, _free_port(), frames, str(tmp_path), True, 8, 0, 1, False, motion), nprocs=2, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(2)])
    assert int(np.load(tmp_path / "repart0.npy")[0]) >= 2, "the cube moved but the slabs were never re-partitioned"
    assert np.load(tmp_path / "counts0.npy").tolist() == [n] * frames
    assert np.array_equal(np.sort(recs["id"]), np.arange(n, dtype=np.uint32))
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), f"moving cube ({motion}): slab run differs from the single-domain run"


@pytest.mark.parametrize("world", [2, 3])
def test_moving_cuts_with_overlapped_exchange_equal_single_domain(tmp_path, oracle_built, world):
    """Moving cuts AND the overlapped exchange: a frame packs its own messages, so a re-cut is planned one frame ahead
    (from the grid of the frame that just ran), steers the send range of the next frame's in-frame pack and is applied
    at that frame's end.  Still the single-domain run bit for bit."""
    frames = 9
    from oracle.ora import Oracle
    ref = Oracle()
    sc = small_tank(ref)
    sc.run(frames)
    want = ref.download_particles()
    n = want.shape[0]
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), True, 4, 2, 1, True), nprocs=world, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    cuts = [np.load(tmp_path / f"cuts{r}.npy").tolist() for r in range(world)]
    assert all(c == cuts[0] for c in cuts), "ranks disagree about the cuts"
    assert cuts[0][-1] >= 1, "the cuts never moved: the test does not exercise a re-cut"
    assert np.load(tmp_path / "counts0.npy").tolist() == [n] * frames
    assert np.array_equal(np.sort(recs["id"]), np.arange(n, dtype=np.uint32))
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), "re-cut + overlapped slab run differs from the single-domain run"


@pytest.mark.parametrize("world", [2, 3])
def test_moving_cuts_equal_single_domain(tmp_path, oracle_built, world):
    """Load-balanced cuts: every 2 frames the ranks re-cut the planes by particle count (one all-reduce of the
    per-plane histogram) -- the faces move, particles change owner through the ordinary face messages, and the
    result is still the single-domain run bit for bit."""
    frames = 7
    from oracle.ora import Oracle
    ref = Oracle()
    sc = small_tank(ref)
    sc.run(frames)
    want = ref.download_particles()
    n = want.shape[0]
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), True, 4, 2), nprocs=world, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    cuts = [np.load(tmp_path / f"cuts{r}.npy").tolist() for r in range(world)]
    assert all(c == cuts[0] for c in cuts), "ranks disagree about the cuts"
    assert cuts[0][-1] >= 1, "the cuts never moved: the test does not exercise a re-cut"
    assert cuts[0][:-1] != [r[0] for r in slab.slab_ranges(48, world)] + [48]
    assert np.load(tmp_path / "counts0.npy").tolist() == [n] * frames
    assert np.array_equal(np.sort(recs["id"]), np.arange(n, dtype=np.uint32))
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), "re-cut slab run differs from the single-domain run"


def test_balanced_cuts_respect_width_and_shift():
    rng = np.random.default_rng(1)
    for _ in range(300):
        world, g, d = int(rng.integers(2, 9)), int(rng.integers(1, 5)), int(rng.integers(1, 6))
        mw = 2 * g + d
        rx = int(world * mw + rng.integers(0, 200))
        cuts = [(rx * r) // world for r in range(world + 1)]
        for _ in range(5):
            hist = rng.integers(0, 50, rx) * (rng.random(rx) < 0.3)
            new = slab.balanced_cuts(hist, cuts, mw, d)
            assert new[0] == 0 and new[-1] == rx
            assert all(b - a >= mw for a, b in zip(new, new[1:])) and all(abs(a - b) <= d for a, b in zip(new, cuts))
            cuts = new
    # unconstrained, the shares come out equal to within one plane's population
    hist = np.full(400, 10)
    assert slab.balanced_cuts(hist, [0, 100, 200, 300, 400], 1, 400) == [0, 100, 200, 300, 400]
    hist[:200] = 30
    cuts = slab.balanced_cuts(hist, [0, 100, 200, 300, 400], 1, 400)
    shares = np.diff(np.concatenate([[0], np.cumsum(hist)])[cuts])
    assert shares.max() - shares.min() <= 30


def test_slab_ranges_cover_the_grid():
    for rx in (16, 320, 2560, 7):
        for w in (1, 2, 3, 8):
            r = slab.slab_ranges(rx, w)
            assert r[0][0] == 0 and r[-1][1] == rx
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))


@pytest.mark.parametrize("world,overlap", [(2, False), (3, True)])
def test_live_spawn_box_edit_reaches_every_slab(tmp_path, oracle_built, world, overlap):
    """SPH.cs:284-298 rewrites restDensity / viscosity / mass of every particle from its spawn box each frame, so an
    Inspector edit takes effect on the next Update.  In slab mode the edit travels as the phase table: local slots take
    the new row at once, records packed BEFORE the edit (overlapped exchange: in flight) take it when they are stored.
    Bit-identical to the single-domain run with the same edits, static fields included."""
    frames = 7
    from oracle.ora import Oracle
    ref = Oracle()
    sc = edited_tank(ref)
    sc.run(frames)
    want = ref.download_particles()
    n = want.shape[0]
    assert set(np.unique(want[:, 12]).tolist()) == {np.float32(0.25), np.float32(0.05)}      # the edit reached the single domain
    assert set(np.unique(want[:, 13]).tolist()) == {np.float32(0.2), np.float32(0.25)}
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), True, 8, 0, 1, overlap, "edit"), nprocs=world, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    assert np.load(tmp_path / "counts0.npy").tolist() == [n] * frames
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), "live edit: slab run differs from the single-domain run"


# ---------------------------------------------------------------------------------------------- checkpoints
def _ckpt_worker(rank, world, port, out_dir, mode, overlap):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.ora import Oracle
    from sph_amd import statefile
    from tests.slab_oracle_port import OraclePort
    oport = OraclePort(Oracle())
    se = slab.SlabEngine(oport, slab.NeighbourComm(rank, world, torch.device("cpu"), backend="gloo"), rank, world,
                         ghost_planes=8, overlap=overlap)
    orig = se.set_params

    def set_params(p):
        oport.set_params_ref(p)
        orig(p)
    se.set_params = set_params
    sc = small_tank(se)
    sc.sph.use_step_frame = True
    sc.start()
    if mode == "save":
        sc.run(4)
        meta = statefile.save_state(sc, os.path.join(out_dir, "slab"))
        assert (meta is not None) == (rank == 0)
    else:
        sc.run(2)                                    # some other state first: the load must replace all of it
        statefile.load_state(sc, os.path.join(out_dir, "single"))
    sc.run(3)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), se.owned_records())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap", [(2, False), (3, True)])
def test_checkpoints_move_between_one_domain_and_slabs(tmp_path, oracle_built, world, overlap):
    """sph_amd.statefile on a slab run: rank 0 writes, from the gathered owned particles, byte for byte the file the
    single-domain run writes at that frame; and a slab run that loads the single-domain file continues it bit for bit
    (every rank keeps the particles of its owned planes, ghosts come with the next exchange)."""
    from oracle.ora import Oracle
    from sph_amd import statefile
    ref = Oracle()
    sc = small_tank(ref)
    sc.run(4)
    single = statefile.save_state(sc, str(tmp_path / "single"))
    sc.run(3)
    want = ref.download_particles()
    for mode in ("save", "load"):
        mp.spawn(_ckpt_worker, args=(world, _free_port(), str(tmp_path), mode, overlap), nprocs=world, join=True)
        recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
        got = abi.aos_from_records(recs[np.argsort(recs["id"])])
        for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
            assert np.array_equal(got[:, sl], want[:, sl]), f"{mode}: the slab run does not continue the single-domain run"
        if mode == "save":
            for ext in (".bin", ".phase"):
                assert open(tmp_path / ("slab" + ext), "rb").read() == open(tmp_path / ("single" + ext), "rb").read(), ext
            import json
            slabmeta = json.load(open(tmp_path / "slab.json"))
            assert slabmeta == json.loads(json.dumps(single))


# ---------------------------------------------------------------------------------------------- the exchange protocol
def test_face_exchange_is_one_round_per_frame(tmp_path, oracle_built):
    """One message per face and direction per frame (header row + records, padded to the size both ends predicted from
    the last message): after the first frames the exchange is ONE batch of sends and receives, and one host wait."""
    frames = 12
    mp.spawn(_worker, args=(3, _free_port(), frames, str(tmp_path), True, 4, 0, 1, True), nprocs=3, join=True)
    for r in range(3):
        exchanges, rounds, waits, reroutes, sent, padded = np.load(tmp_path / f"comm{r}.npy").tolist()
        assert exchanges == frames + 1 and reroutes == 0               # overlapped: the first frame also exchanges the classic way
        assert rounds <= exchanges + 3, f"rank {r}: {rounds} transfer rounds for {exchanges} exchanges"
        assert waits == exchanges + frames                             # gloo: the header read, and with three ranks the far-jump flag of every frame
        assert padded <= 0.35 * sent + 4 * 4096 * 2, f"rank {r}: {padded} padding records for {sent} sent"


def test_next_bound_is_what_both_ends_compute():
    assert slab.next_bound(0) == 256 and slab.next_bound(1600) == 1600 + 100 + 256
    assert slab.next_bound(10 ** 6, cap=1000) == 1000


@pytest.mark.parametrize("overlap", [False, True])
def test_particles_that_jump_two_slabs_are_rerouted_and_conserved(tmp_path, oracle_built, overlap):
    """SURVEY 8e's all-to-all fallback: a few particles cross MORE than a whole neighbouring slab in one frame (26
    planes; the slabs are 12 wide), i.e. they belong to a rank the face messages never reach.  The ranks agree on it
    (one flag all-reduce per frame), route every owned particle to the owner of its current plane and go on: every id
    is owned exactly once after every frame.  (Their own sums in the frames they fly are not the single-domain run's --
    no ghost region is 26 planes wide -- so this is a conservation test, not a bit-identity test.)"""
    frames, world = 8, 4
    from oracle.ora import Oracle
    ref = Oracle()
    sc = flying_tank(ref)
    sc.start()
    n = ref.n
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), True, 4, 0, 1, overlap, "fly"), nprocs=world, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    assert np.load(tmp_path / "counts0.npy").tolist() == [n] * frames, "particles were lost or duplicated"
    assert np.array_equal(np.sort(recs["id"]), np.arange(n, dtype=np.uint32))
    reroutes = [int(np.load(tmp_path / f"comm{r}.npy")[3]) for r in range(world)]
    assert len(set(reroutes)) == 1 and reroutes[0] >= 1, f"the fallback never ran (or not on every rank): {reroutes}"
    assert np.isfinite(abi.aos_from_records(recs)).all()
    # the bulk of the fluid (everything that never flew) still follows the single-domain run closely
    sc.run(frames)
    want = ref.download_particles()
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    slow = np.arange(n) < n - 6
    assert np.abs(got[slow, 0:3] - want[slow, 0:3]).max() < 0.05


@pytest.mark.parametrize("world,overlap", [(2, False), (3, False), (3, True)])
def test_adaptive_ghost_width_with_moving_cuts_equals_single_domain(tmp_path, oracle_built, world, overlap):
    """Both at once (round 2 refused the combination): the faces move every 2 frames by particle count AND every face
    follows its own measured ghost width.  Still the single-domain run bit for bit, ranks agree on cuts and widths."""
    frames = 12
    from oracle.ora import Oracle
    ref = Oracle()
    sc = small_tank(ref)
    sc.run(frames)
    want = ref.download_particles()
    n = want.shape[0]
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path), True, 4, 2, 1, overlap, None, True), nprocs=world, join=True)
    recs = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    cuts = [np.load(tmp_path / f"cuts{r}.npy").tolist() for r in range(world)]
    assert all(c == cuts[0] for c in cuts), "ranks disagree about the cuts"
    assert cuts[0][-1] >= 1, "the cuts never moved"
    assert np.load(tmp_path / "counts0.npy").tolist() == [n] * frames
    assert np.array_equal(np.sort(recs["id"]), np.arange(n, dtype=np.uint32))
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)):
        assert np.array_equal(got[:, sl], want[:, sl]), "adaptive widths + moving cuts: slab run differs from the single-domain run"
    w = [np.load(tmp_path / f"widths{r}.npy").tolist() for r in range(world)]
    assert all(x[5] == 0 for x in w), f"halo violations {w}"
    for r in range(world - 1):
        assert w[r][1] == w[r + 1][0] and w[r][3] == w[r + 1][2], f"face {r}|{r + 1}: {w[r]} vs {w[r + 1]}"



def _failing_worker(rank, world, port, out_dir, overlap, mode, fail_rank, fail_frame):
    """A run in which ONE rank fails (mode "pack": its engine refuses the pack of frame `fail_frame`; mode "cap": the face
    messages outgrow the message capacity).  Every rank writes the frame it stopped in and why."""
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    from oracle.ora import Oracle
    from tests.slab_oracle_port import OraclePort
    oport = OraclePort(Oracle())
    comm = slab.NeighbourComm(rank, world, torch.device("cpu"), backend="gloo", msg_capacity=12 if mode == "cap" else 0)
    se = slab.SlabEngine(oport, comm, rank, world, ghost_planes=4 if world == 3 else 8, overlap=overlap)
    orig = se.set_params

    def set_params(p):
        oport.set_params_ref(p)
        orig(p)
    se.set_params = set_params
    state = {"frame": 0}
    if mode == "pack" and rank == fail_rank:
        for name in ("pack", "pack_wait"):
            good = getattr(oport, name)

            def failing(*a, _good=good):
                if state["frame"] >= fail_frame:
                    raise abi.SphError(-4, "slab message buffer too small (injected)")
                return _good(*a)
            setattr(oport, name, failing)
    sc = small_tank(se)
    sc.sph.use_step_frame = True
    stopped = "ran to the end"
    try:
        sc.start()
        for f in range(8):
            state["frame"] = f
            sc.frame()
    except RuntimeError as exc:
        stopped = f"{state['frame']}|{exc}"
    with open(os.path.join(out_dir, f"stopped{rank}.txt"), "w") as fh:
        fh.write(stopped)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap,mode", [(2, False, "pack"), (2, True, "pack"), (3, False, "pack"), (3, True, "pack"),
                                                (2, True, "cap"), (3, False, "cap")])
def test_a_failure_on_one_rank_stops_every_rank_in_the_same_frame(tmp_path, oracle_built, world, overlap, mode):
    """A rank that raised on its own would leave its neighbours waiting in their next send/recv until the process
    group's timeout.  The failing rank stays in step instead and says so -- in the header of its face
    messages (two ranks) or in the frame's code all-reduce (more) -- and all ranks raise together."""
    fail_rank, fail_frame = world - 1, 3
    mp.spawn(_failing_worker, args=(world, _free_port(), str(tmp_path), overlap, mode, fail_rank, fail_frame), nprocs=world, join=True)
    got = [open(tmp_path / f"stopped{r}.txt").read() for r in range(world)]
    assert all("|" in g for g in got), got                       # nobody ran to the end, nobody hung
    frames = {int(g.split("|")[0]) for g in got}
    assert len(frames) == 1, got                                  # ... and all in the same frame
    if mode == "pack":
        assert frames == {fail_frame}
        assert "injected" in got[fail_rank]                       # the failing rank reports its own error
    else:
        assert any("exceeds the message capacity" in g for g in got)
