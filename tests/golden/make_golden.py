#!/usr/bin/env python3
"""Generates tests/golden/*.npz: one-frame input/expected-output vectors for the SPH step.

Provenance: the reference (CPearson109/SPH) ships no tests, fixtures or CPU solver, and its
HLSL/C# cannot be compiled here, so these vectors are produced by oracle/sph_oracle.cpp and
cross-checked (asserted below) against the independent numpy restatement oracle/sph_ref.py.
They pin the oracle against regressions and give the GPU engine fixed targets; they are NOT
outputs of the reference itself ("parity unpinned by the reference").

    python tests/golden/make_golden.py        # rewrites the .npz files next to this script
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.ora import Oracle            # noqa: E402
from oracle.sph_ref import Ref           # noqa: E402
from sph_amd import abi, scenes          # noqa: E402
from sph_amd.host import SPH, Scene, SpawnBox, SphereCollider, Time, Transform  # noqa: E402


def params_bytes(p):
    return np.frombuffer(bytes(p), dtype=np.uint8).copy()


def csr(lists):
    counts = np.array([len(x) for x in lists], dtype=np.int32)
    flat = np.concatenate(lists).astype(np.int32) if counts.sum() else np.zeros(0, np.int32)
    return counts, flat


def capture(name, scene_builder, warm_frames, check_tol=2e-5):
    """Runs `warm_frames` frames on the oracle, then records frame warm_frames+1 as the vector."""
    ora = Oracle()
    sc = scene_builder(ora)
    sc.start()
    for _ in range(warm_frames):
        sc.frame()
    state0 = ora.download_particles()
    sc.frame()
    prm = sc.sph.build_params()
    sub, ts = sc.sph.subSteps, sc.sph.timeStep
    state1 = ora.download_particles()
    cs, ce, ids, cop, counts = ora.download_grid(with_counts=True)
    # neighbour lists as CS_DensityPressure sees them at the END of the frame (current positions,
    # this frame's bins) -- exactly what sph_download_neighbors returns after the same frame
    ncnt, nlst = ora.download_neighbors(512)
    assert ncnt.max(initial=0) <= 512
    lists = [nlst[i, :ncnt[i]] for i in range(ora.n)]
    # ---- cross-check against the numpy restatement
    ref = Ref(prm.copy(), state0)
    ref.step_frame(sub, ts)
    assert np.array_equal(ref.cell_of_particle, cop), name
    assert np.array_equal(ref.grid_counts, counts), name
    for fld, sl in (("pos", slice(0, 3)), ("vel", slice(3, 6)), ("acc", slice(6, 9)), ("rho", 9), ("p", 10)):
        a, b = ref.P[:, sl].astype(np.float64), state1[:, sl].astype(np.float64)
        scale = max(np.abs(b).max(), 1e-30)
        err = np.abs(a - b).max() / scale
        assert err < check_tol, f"{name}: numpy restatement disagrees on {fld}: {err:.3e}"
    rl = ref.neighbor_lists()
    assert all(np.array_equal(a, b) for a, b in zip(rl, lists)), f"{name}: neighbour lists differ"
    nc, nf = csr(lists)
    out = os.path.join(HERE, name + ".npz")
    np.savez_compressed(out, params=params_bytes(prm), sub_steps=np.int32(sub), time_step=np.float32(ts),
                        state_in=state0, state_out=state1, cell_of_particle=cop, sorted_ids=ids,
                        grid_counts_nonzero_cells=np.flatnonzero(counts).astype(np.int32),
                        grid_counts_nonzero=counts[np.flatnonzero(counts)].astype(np.int32),
                        neighbor_counts=nc, neighbor_flat=nf,
                        counters=np.array([ora.read_counter(k) for k in range(3)], dtype=np.int32))
    print(f"{name}: n={ora.n} cells={ora.cells} neighbours/particle={nc.mean():.1f} "
          f"collisions={ora.read_counter(0)} overflow={ora.read_counter(1)} oog={ora.read_counter(2)} -> {os.path.getsize(out)//1024} KiB")
    ora.destroy()


def two_particles(e):
    """Two-particle KAT: r = 0.2 < h along x, AABB branch, resting on nothing."""
    boxes = [SpawnBox(2, (0.0, 0.0, 0.0), 20, 0.1, 0.2, position=(0.0, 5.0, 0.0))]
    f = dict(scenes.SCENE_COMMON)
    f.update(gridResolution=40)
    sph = SPH(e, **f, surfaceTensionCoefficient=0.1, xsphEpsilon=0.3, spawnBoxes=boxes)
    sc = Scene(sph, Time(), [])
    orig = sph.CreateParticles

    def create():
        orig()
        P = e.download_particles()
        P[0, 0:3] = (0.0, 5.0, 0.0)
        P[1, 0:3] = (0.2, 5.0, 0.0)
        e.upload_particles(P)
    sph.CreateParticles = create
    return sc


def lattice27(e):
    """3 x 3 x 3 lattice with spacing 0.25 (= cell size): every particle in its own cell."""
    boxes = [SpawnBox(27, (0.0, 0.0, 0.0), 20, 0.1, 0.2, position=(0.0, 5.0, 0.0))]
    f = dict(scenes.SCENE_COMMON)
    f.update(gridResolution=40)
    sph = SPH(e, **f, surfaceTensionCoefficient=0.1, xsphEpsilon=0.3, spawnBoxes=boxes)
    sc = Scene(sph, Time(), [])
    orig = sph.CreateParticles

    def create():
        orig()
        P = e.download_particles()
        g = np.stack(np.meshgrid(np.arange(3), np.arange(3), np.arange(3), indexing="ij"), -1).reshape(-1, 3)
        P[:, 0:3] = (np.array([0.1, 5.1, 0.1], np.float32) + g.astype(np.float32) * np.float32(0.25))
        e.upload_particles(P)
    sph.CreateParticles = create
    return sc


def overflow_scene(e):
    """Bin-cap overflow: 600 particles in a 0.5^3 box with maxParticlesPerCell = 8 (rule O2)."""
    boxes = [SpawnBox(600, (0.5, 0.5, 0.5), 20, 0.1, 0.2, position=(0.0, 5.0, 0.0))]
    f = dict(scenes.SCENE_COMMON)
    f.update(gridResolution=40, maxParticlesPerCell=8)
    sph = SPH(e, **f, surfaceTensionCoefficient=0.1, xsphEpsilon=0.3, spawnBoxes=boxes)
    return Scene(sph, Time(), [])


def escaped_scene(e):
    """Particles spawned partly OUTSIDE the cube: clamped hash (compute:126), unclamped search
    cell (rule O7), OBB push-back with a corner normal, obstacle sphere in the fluid."""
    cube = Transform(position=(0, 2, 0), localScale=(4, 4, 4))
    obstacle = SphereCollider(radius=0.5, transform=Transform(position=(0.5, 1.0, 0.0), localScale=(2, 2, 2)))
    boxes = [SpawnBox(1500, (5.0, 3.0, 5.0), 20, 0.1, 0.2, position=(0.0, 1.0, 0.0)),
             SpawnBox(500, (2.0, 2.0, 2.0), 10, 0.05, 0.2, position=(1.5, 3.5, 1.5), initialVelocity=(3, 2, 1))]
    f = dict(scenes.SCENE_COMMON)
    f.update(gridResolution=16)
    sph = SPH(e, **f, surfaceTensionCoefficient=0.1, xsphEpsilon=0.3, boundaryCube=cube, obstacleCollider=obstacle,
              spawnBoxes=boxes)
    return Scene(sph, Time(), [])


def rotated_cube_scene(e):
    from sph_amd.host import quat_from_axis_angle
    cube = Transform(position=(0, 3, 0), rotation=quat_from_axis_angle((0, 0, 1), 20.0), localScale=(4, 6, 4))
    boxes = [SpawnBox(2048, (3.0, 3.0, 3.0), 20, 0.1, 0.2, position=(0.0, 3.0, 0.0), initialVelocity=(4, 0, 0))]
    f = dict(scenes.SCENE_COMMON)
    f.update(gridResolution=20)
    sph = SPH(e, **f, surfaceTensionCoefficient=0.1, xsphEpsilon=0.3, boundaryCube=cube, spawnBoxes=boxes)
    return Scene(sph, Time(), [])


CASES = [
    ("kat_two_particles", two_particles, 0),
    ("lattice27", lattice27, 0),
    ("config1_aabb_1024_f3", lambda e: scenes.config1(e, n=1024), 3),
    ("config1_aabb_4096_f6", lambda e: scenes.config1(e, n=4096), 6),
    ("scene2_obb_two_phase_2250_f4", lambda e: scenes.unity_scene(e, which=2, scale_count=0.01), 4),
    ("scene4_obstacle_two_phase_2250_f5", lambda e: scenes.unity_scene(e, which=4, scale_count=0.01), 5),
    ("bin_overflow_600_f1", overflow_scene, 1),
    ("escaped_obb_sphere_2000_f2", escaped_scene, 2),
    ("rotated_cube_2048_f3", rotated_cube_scene, 3),
    # BASELINE config 3 in small: the cube rocks (the frame runs with LAST frame's grid against THIS frame's matrices,
    # SPH.cs:176-185) while the sphere orbits through the fluid on the floor (compute:467-483)
    ("config3_rocking_sphere_3000_f26", lambda e: scenes.config3(e, n=3000, scale=0.25), 26),
]

if __name__ == "__main__":
    only = set(sys.argv[1:])                  # optional: names of the cases to (re)write
    for name, builder, warm in CASES:
        if not only or name in only:
            capture(name, builder, warm)
