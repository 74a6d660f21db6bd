"""Scene definitions: the reference's four Unity scenes (values read from
Assets/Scenes/Scene{1..4}.unity, tabulated in SURVEY.md section 5.1) and the five BASELINE
configurations.  Each builder returns a host.Scene driving `engine`.
"""
from __future__ import annotations

from .host import (SPH, CircularMotion, RockingMotion, Scene, SpawnBox, SphereCollider, Time, Transform)

WATER = (0.2, 0.5, 1.0, 0.8)                   # the BASELINE configurations (build-defined scenes)
OIL = (1.0, 0.8, 0.2, 0.8)
# SpawnBox.particleColor as serialized in the reference's scenes (Scene1.unity:543, Scene2.unity:499/594,
# Scene3.unity:543, Scene4.unity:499/594; tests/golden/unity_scene_values.json holds every serialized value)
SCENE_BLUE = {1: (0.0039215684, 0.18861637, 1.0, 1.0), 2: (0.0039215684, 0.30632228, 1.0, 1.0),
              3: (0.0039215684, 0.21572512, 1.0, 1.0), 4: (0.0039215684, 0.23613732, 1.0, 1.0)}
SCENE_YELLOW = (1.0, 0.94675124, 0.0, 1.0)

# Inspector values shared by Scene1-4 (Scene1.unity:604-622 etc.)
SCENE_COMMON = dict(soundSpeed=10.0, gamma=4.0, smoothingRadius=0.3, gravity=-9.81, timeStep=0.02, subSteps=1,
                    surfaceTensionThreshold=0.01, obstacleRepulsionStiffness=100.0, particleCollisionDamping=1.0,
                    maxParticlesPerCell=500, gridResolution=80, boundsCenter=(0, 5, 0), boundsSize=(10, 10, 10))


def _split21(n):
    """A:B = 2:1 by count: 262,144 -> 174,763 + 87,381."""
    b = n // 3
    return n - b, b


def _parked_obstacle():
    # SphereCollider radius 0.5, scale 8 -> radius 4, parked outside the cube (Scene1.unity:676, 740-741)
    return SphereCollider(radius=0.5, transform=Transform(position=(-6.08, 4.0, 41.53), localScale=(8, 8, 8)))


def unity_scene(engine, which: int, scale_count: float = 1.0, seed: int = 12345) -> Scene:
    """Scene1..Scene4 of the reference.  scale_count < 1 (tests) shrinks the particle counts and
    the spawn boxes together, so the spawn number density stays the scene's own."""
    ls = float(scale_count) ** (1.0 / 3.0)

    def cnt(c):
        return max(1, int(round(c * scale_count)))

    def sz(x, y, z):
        return (x * ls, y * ls, z * ls)
    time = Time()
    blue = SCENE_BLUE.get(which, WATER)
    cube = Transform(position=(0, 15, 0), localScale=(20, 30, 20))                # Scene1.unity:933-935
    obstacle = _parked_obstacle()
    movers = []
    if which in (1, 2):
        sigma, eps = 0.1, 0.3
    else:
        sigma, eps = 0.01, 0.2
    if which == 1:
        boxes = [SpawnBox(cnt(200000), sz(5, 10, 5), 20, 0.1, 0.2, particleColor=blue, position=(0, 20, 0))]
    elif which == 2:
        boxes = [SpawnBox(cnt(150000), sz(5, 10, 5), 20, 0.01, 0.2, particleColor=blue, position=(0, 20, 5)),
                 SpawnBox(cnt(75000), sz(5, 10, 5), 10, 0.1, 0.2, particleColor=SCENE_YELLOW, position=(0, 20, -5))]
    elif which == 3:
        boxes = [SpawnBox(cnt(200000), sz(10, 10, 10), 20, 0.1, 0.2, particleColor=blue, position=(0, 15, 0))]
        movers.append(CircularMotion(obstacle.transform, time, center=(0, 0, 0), radius=5.0, speed=2.0))
    elif which == 4:
        boxes = [SpawnBox(cnt(150000), sz(5, 10, 5), 20, 0.1, 0.2, particleColor=blue, position=(0, 15, 5)),
                 SpawnBox(cnt(75000), sz(5, 10, 5), 10, 0.1, 0.2, particleColor=SCENE_YELLOW, position=(0, 15, -5))]
        movers.append(CircularMotion(obstacle.transform, time, center=(0, 0, 0), radius=5.0, speed=1.0))
    else:
        raise ValueError("scene 1..4")
    sph = SPH(engine, **SCENE_COMMON, surfaceTensionCoefficient=sigma, xsphEpsilon=eps, boundaryCube=cube,
              obstacleCollider=obstacle, spawnBoxes=boxes, seed=seed)
    return Scene(sph, time, movers)


def config1(engine, n=16384, seed=12345) -> Scene:
    """Config 1: single-phase dam break, AABB-fallback branch, 40^3 cells, no obstacle."""
    boxes = [SpawnBox(n, (4, 8.2, 5), 20, 0.1, 0.2, particleColor=WATER, position=(-3, 4.1, 0))]
    f = dict(SCENE_COMMON)
    f.update(gridResolution=40)
    sph = SPH(engine, **f, surfaceTensionCoefficient=0.1, xsphEpsilon=0.3, boundaryCube=None, obstacleCollider=None,
              spawnBoxes=boxes, seed=seed)
    return Scene(sph, Time(), [])


def config2(engine, n=262144, seed=12345) -> Scene:
    """Config 2: two-phase tank, Scene2/4 geometry, surface tension on, obstacle parked."""
    a, b = _split21(n)
    boxes = [SpawnBox(a, (5, 10, 5), 20, 0.1, 0.2, particleColor=WATER, position=(0, 20, 5)),
             SpawnBox(b, (5, 10, 5), 10, 0.1, 0.2, particleColor=OIL, position=(0, 20, -5))]
    cube = Transform(position=(0, 15, 0), localScale=(20, 30, 20))
    sph = SPH(engine, **SCENE_COMMON, surfaceTensionCoefficient=0.1, xsphEpsilon=0.3, boundaryCube=cube,
              obstacleCollider=_parked_obstacle(), spawnBoxes=boxes, seed=seed)
    return Scene(sph, Time(), [])


def config3(engine, n=1048576, seed=12345, scale=1.6, rocking=True) -> Scene:
    """Config 3: two-phase + orbiting sphere + rocking cube (Scene4 geometry x1.6)."""
    a, b = _split21(n)
    s = scale
    time = Time()
    boxes = [SpawnBox(a, (5 * s, 10 * s, 5 * s), 20, 0.1, 0.2, particleColor=WATER, position=(0, 15 * s, 5 * s)),
             SpawnBox(b, (5 * s, 10 * s, 5 * s), 10, 0.1, 0.2, particleColor=OIL, position=(0, 15 * s, -5 * s))]
    cube = Transform(position=(0, 15 * s, 0), localScale=(20 * s, 30 * s, 20 * s))
    obstacle = SphereCollider(radius=0.5, transform=Transform(position=(5 * s, 0, 0), localScale=(8 * s, 8 * s, 8 * s)))
    movers = [CircularMotion(obstacle.transform, time, center=(0, 0, 0), radius=5.0 * s, speed=1.0)]
    if rocking:
        movers.append(RockingMotion(cube, time, amplitude_deg=10.0, rate=0.5))
    f = dict(SCENE_COMMON)
    f.update(gridResolution=int(round(80 * s)))
    sph = SPH(engine, **f, surfaceTensionCoefficient=0.01, xsphEpsilon=0.2, boundaryCube=cube,
              obstacleCollider=obstacle, spawnBoxes=boxes, seed=seed)
    return Scene(sph, time, movers)


def config4(engine, n=16777216, seed=12345, cube=(80.0, 120.0, 80.0), box=(78.0, 10.0, 20.0), box_z=12.0) -> Scene:
    """Config 4: 16M two-phase tank (strong scaling); single-GPU form.  `cube`, `box`, `box_z` shrink the geometry
    for tests (CONFIG4_SMALL: the same layout -- two boxes spanning the x width side by side in z, static cube, no
    obstacle, cell 0.25 -- at the same spawn densities, ~700 and ~350 particles per unit^3)."""
    a, b = _split21(n)
    yc = 0.5 * cube[1]
    boxes = [SpawnBox(a, box, 20, 0.1, 0.2, particleColor=WATER, position=(0, yc, box_z)),
             SpawnBox(b, box, 10, 0.1, 0.2, particleColor=OIL, position=(0, yc, -box_z))]
    tank = Transform(position=(0, yc, 0), localScale=cube)
    f = dict(SCENE_COMMON)
    f.update(gridResolution=int(round(min(cube) / 0.25)))                  # cell size 0.25 (SPH.cs:329); 320 at full size
    sph = SPH(engine, **f, surfaceTensionCoefficient=0.01, xsphEpsilon=0.2, boundaryCube=tank, obstacleCollider=None,
              spawnBoxes=boxes, seed=seed)
    return Scene(sph, Time(), [])


# test-size geometry of configs 4 and 5 (tests/test_configs_gpu.py): long enough in x for 4 slabs of >= 2 x 12 ghost
# planes, spawn densities of the full-size configurations (their blow-up is what sets the ghost width)
CONFIG4_SMALL = dict(n=48000, cube=(26.0, 12.0, 8.0), box=(24.0, 1.2, 1.6), box_z=1.6)
CONFIG5_SMALL = dict(per_gpu=12800, block=(8.0, 8.0), height=12.0, box=(7.6, 1.0, 1.6))


def config5(engine, gpus=1, per_gpu=8388608, seed=12345, block=(80.0, 64.0), height=96.0, box=None) -> Scene:
    """Config 5: weak scaling, 8,388,608 particles per GPU, G blocks of 80 x 96 x 64 along x.
    `block` = (x, z) extent of one GPU's block; other shapes of the same volume and spawn density are
    used by locality experiments only.  `height`, `box` (spawn box size): test-size geometry, CONFIG5_SMALL."""
    G = gpus
    bx, bz = block
    a, b = _split21(per_gpu)
    boxes = []
    if box is None:
        box = (bx - 4.0, 8.0, 76.0 * 13.0 / (bx - 4.0))  # spawn box footprint of constant area
    yc = 0.5 * height
    for g in range(G):
        xc = -0.5 * bx * G + 0.5 * bx + bx * g
        boxes.append(SpawnBox(a, box, 20, 0.1, 0.2, particleColor=WATER, position=(xc, yc, 0.25 * bz)))
        boxes.append(SpawnBox(b, box, 10, 0.1, 0.2, particleColor=OIL, position=(xc, yc, -0.25 * bz)))
    cube = Transform(position=(0, yc, 0), localScale=(bx * G, height, bz))
    f = dict(SCENE_COMMON)
    f.update(gridResolution=int(round(min(bx * G, height, bz) / 0.25)))     # cell size 0.25 (SPH.cs:329)
    sph = SPH(engine, **f, surfaceTensionCoefficient=0.01, xsphEpsilon=0.2, boundaryCube=cube, obstacleCollider=None,
              spawnBoxes=boxes, seed=seed)
    return Scene(sph, Time(), [])


def from_values(engine, values: dict, scale_count: float = 1.0, seed: int = 12345) -> Scene:
    """A scene from a description in the schema of tests/golden/unity_scene_values.json (one entry of its "scenes":
    what a Unity scene file serializes for the path): {"sph": {Inspector fields of SPH.cs:5-49}, "boundaryCube":
    {position, rotation, localScale} | null, "obstacle": {radius, transform, mover: {center, radius, speed, enabled} |
    null} | null, "spawnBoxes": [{particleCount, spawnSize, restDensity, viscosity, particleMass, initialVelocity,
    particleColor, position}]}.  scale_count as in unity_scene."""
    ls = float(scale_count) ** (1.0 / 3.0)
    time = Time()
    fields = {k: v for k, v in values["sph"].items() if k != "totalCells"}           # totalCells is computed (SPH.cs:79)
    cube = None
    if values.get("boundaryCube"):
        c = values["boundaryCube"]
        cube = Transform(position=c["position"], rotation=c.get("rotation", (0, 0, 0, 1)), localScale=c["localScale"])
    obstacle, movers = None, []
    if values.get("obstacle"):
        o = values["obstacle"]
        t = o["transform"]
        obstacle = SphereCollider(radius=o["radius"], transform=Transform(position=t["position"],
                                  rotation=t.get("rotation", (0, 0, 0, 1)), localScale=t["localScale"]))
        m = o.get("mover")
        if m and m.get("enabled", 1):
            movers.append(CircularMotion(obstacle.transform, time, center=m["center"], radius=m["radius"], speed=m["speed"]))
    boxes = []
    for b in values["spawnBoxes"]:
        if not (b.get("enabled", 1) and b.get("active", 1)):
            continue
        boxes.append(SpawnBox(max(1, int(round(b["particleCount"] * scale_count))), tuple(x * ls for x in b["spawnSize"]),
                              b["restDensity"], b["viscosity"], b["particleMass"],
                              initialVelocity=b.get("initialVelocity", (0, 0, 0)),
                              particleColor=b.get("particleColor", (1, 1, 1, 1)), position=b["position"]))
    sph = SPH(engine, **fields, boundaryCube=cube, obstacleCollider=obstacle, spawnBoxes=boxes, seed=seed)
    return Scene(sph, time, movers)


def load_scene(engine, path: str, name: str = None, **kw) -> Scene:
    """from_values on a JSON file: either one scene description, or {"scenes": {name: description}} (then `name`
    picks one)."""
    import json
    with open(path) as fh:
        doc = json.load(fh)
    if "scenes" in doc:
        doc = doc["scenes"][name]
    return from_values(engine, doc, **kw)
