"""Small versions of what the reference's README pictures show (readme_files/MuliPhase1.PNG, box.jpg, Splash.jpg,
Ball.jpg), with the reference's own Inspector values (scenes.SCENE_COMMON: h 0.3, cell 0.25, mass 0.2, rest densities
20 / 10, c 10, gamma 4, dt 0.02) in tanks a few units wide, and the statistics that say whether a run ended the way the
picture does.  The pictures are the only OUTPUTS of the path the reference holds; tests/test_readme_pictures_gpu.py runs
these scenes on the engine and holds the end states to them (light phase on top of the dense one with a level
interface; a level free surface in a tilted cube; a dropped block that ends as a level layer over the whole floor; no
particle inside the moving sphere, which ploughs a hole into a shallow layer)."""
from __future__ import annotations

import numpy as np

from .host import (SPH, CircularMotion, Scene, SpawnBox, SphereCollider, Time, Transform, quat_from_axis_angle)
from .scenes import SCENE_BLUE, SCENE_COMMON, SCENE_YELLOW

FRAMES = dict(two_phase_tank=2500, tilted_cube=1500, splash=1500, ball=700)


def _sph(engine, cube, boxes, obstacle=None, sigma=0.01, eps=0.2, seed=12345):
    f = dict(SCENE_COMMON)
    f.update(gridResolution=int(round(min(cube.localScale) / 0.25)))          # cell size 0.25 like the scenes (SPH.cs:329)
    return SPH(engine, **f, surfaceTensionCoefficient=sigma, xsphEpsilon=eps, boundaryCube=cube, obstacleCollider=obstacle,
               spawnBoxes=boxes, seed=seed)


def _ke_per_particle(P):
    return float((0.5 * P[:, 13] * (P[:, 3:6].astype(np.float64) ** 2).sum(1)).mean())


# ---------------------------------------------------------------------------------- MuliPhase1.PNG
def two_phase_tank(engine, n=6000) -> Scene:
    """Scene 2 / 4 in small: the two phases spawn side by side in z (rest densities 20 and 10, 2:1 by count = equal rest
    volumes) in a 4 x 12 x 2 tank."""
    cube = Transform(position=(0, 6, 0), localScale=(4, 12, 2))
    a = n - n // 3
    boxes = [SpawnBox(a, (3.4, 6.0, 0.7), 20, 0.1, 0.2, particleColor=SCENE_BLUE[4], position=(0, 3.4, 0.5)),
             SpawnBox(n - a, (3.4, 6.0, 0.7), 10, 0.1, 0.2, particleColor=SCENE_YELLOW, position=(0, 3.4, -0.5))]
    return Scene(_sph(engine, cube, boxes), Time(), [])


def two_phase_tank_stats(sc, P):
    dense, light = P[P[:, 11] == 20.0], P[P[:, 11] == 10.0]
    order = np.argsort(P[:, 1])
    top = P[order[-light.shape[0]:]]                       # the highest particles, as many as there are light ones
    # the interface along x: top of the dense phase in 0.5-wide columns
    k = np.floor(dense[:, 0] / 0.5).astype(np.int64)
    tops = np.array([np.percentile(dense[k == b, 1], 95) for b in np.unique(k) if (k == b).sum() >= 25])
    return dict(interface_std=float(tops.std()) if tops.size else float("nan"), y_dense=float(dense[:, 1].mean()),
                y_light=float(light[:, 1].mean()),
                light_share_of_top=float((top[:, 11] == 10.0).mean()), ke=_ke_per_particle(P),
                y_max=float(P[:, 1].max()), dense_top_p95=float(np.percentile(dense[:, 1], 95)),
                light_bottom_p05=float(np.percentile(light[:, 1], 5)))


# ---------------------------------------------------------------------------------- box.jpg
TILT_DEG = 25.0


def tilted_cube(engine, n=5000) -> Scene:
    cube = Transform(position=(0, 4, 0), rotation=quat_from_axis_angle((0, 0, 1), TILT_DEG), localScale=(6, 8, 2))
    boxes = [SpawnBox(n, (2.4, 2.4, 1.6), 20, 0.1, 0.2, particleColor=SCENE_BLUE[1], position=(0, 3.5, 0))]
    return Scene(_sph(engine, cube, boxes, sigma=0.1, eps=0.3), Time(), [])


def surface_profile(P, axis=0, bin_width=0.4, min_count=25):
    """(bin centres, surface height) along `axis`: the 95th percentile of y in every bin that holds enough particles."""
    x = P[:, axis].astype(np.float64)
    k = np.floor(x / bin_width).astype(np.int64)
    xs, ys = [], []
    for b in np.unique(k):
        m = k == b
        if m.sum() >= min_count:
            xs.append((b + 0.5) * bin_width)
            ys.append(np.percentile(P[m, 1], 95))
    return np.asarray(xs), np.asarray(ys)


def tilted_cube_stats(sc, P):
    xs, ys = surface_profile(P)
    # the wedge's thin end follows the floor: keep the bins whose column is at least 0.6 deep
    floor = np.array([np.percentile(P[np.floor(P[:, 0] / 0.4) == np.floor(x / 0.4), 1], 2) for x in xs])
    deep = ys - floor >= 0.6
    slope = float(np.polyfit(xs[deep], ys[deep], 1)[0]) if deep.sum() >= 3 else float("nan")
    M = np.asarray(sc.sph.boundaryCube.worldToLocalMatrix, dtype=np.float64)
    q = P[:, 0:3].astype(np.float64) @ M[:3, :3].T + M[:3, 3]           # cube-local coordinates: inside = |q| <= 0.5
    return dict(surface_slope=slope, floor_slope=float(np.tan(np.radians(TILT_DEG))), bins=int(deep.sum()),
                max_local=float(np.abs(q).max()), ke=_ke_per_particle(P), x_mean=float(P[:, 0].mean()),
                lowest=P[int(np.argmin(P[:, 1])), 0:2].astype(np.float64))


# ---------------------------------------------------------------------------------- Splash.jpg
def splash(engine, n=6000) -> Scene:
    cube = Transform(position=(0, 5, 0), localScale=(5, 10, 5))
    boxes = [SpawnBox(n, (3.0, 4.0, 3.0), 20, 0.1, 0.2, particleColor=SCENE_BLUE[1], position=(0, 6.0, 0))]
    return Scene(_sph(engine, cube, boxes, sigma=0.1, eps=0.3), Time(), [])


def splash_stats(sc, P):
    kx, kz = np.floor((P[:, 0] + 2.5) / 0.5).astype(int), np.floor((P[:, 2] + 2.5) / 0.5).astype(int)
    ok = (kx >= 0) & (kx < 10) & (kz >= 0) & (kz < 10)
    top = np.full((10, 10), np.nan)
    for i in range(10):
        for j in range(10):
            m = ok & (kx == i) & (kz == j)
            if m.sum() >= 5:
                top[i, j] = np.percentile(P[m, 1], 95)
    return dict(ke=_ke_per_particle(P), y_max=float(P[:, 1].max()), covered=float(np.isfinite(top).mean()),
                surface_mean=float(np.nanmean(top)), surface_std=float(np.nanstd(top)), y_mean=float(P[:, 1].mean()))


# ---------------------------------------------------------------------------------- Ball.jpg
def ball(engine, n=8000) -> Scene:
    time = Time()
    cube = Transform(position=(0, 4, 0), localScale=(8, 8, 8))
    boxes = [SpawnBox(n, (7.4, 1.4, 7.4), 20, 0.1, 0.2, particleColor=SCENE_BLUE[3], position=(0, 0.8, 0))]
    obstacle = SphereCollider(radius=0.5, transform=Transform(position=(2.0, 0.9, 0.0), localScale=(3, 3, 3)))
    movers = [CircularMotion(obstacle.transform, time, center=(0, 0.9, 0), radius=2.0, speed=1.0)]
    return Scene(_sph(engine, cube, boxes, obstacle=obstacle), time, movers)


def ball_stats(sc, P):
    c = np.asarray(sc.sph._obstaclePos, dtype=np.float64)
    r = float(sc.sph._obstacleRadius)
    d = np.linalg.norm(P[:, 0:3].astype(np.float64) - c[None, :], axis=1)
    shell = (d < r * 1.15)
    # the layer far from the sphere, for the depth the hole is measured against
    far = np.linalg.norm(P[:, [0, 2]] - c[None, [0, 2]], axis=1) > 2.0 * r
    return dict(min_dist_over_radius=float(d.min() / r), in_shell=int(shell.sum()), layer_top=float(np.percentile(P[far, 1], 95)),
                sphere_bottom=float(c[1] - r), ke=_ke_per_particle(P), y_max=float(P[:, 1].max()))
