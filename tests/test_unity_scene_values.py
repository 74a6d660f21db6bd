"""The scene builders against the reference's own data: tests/golden/unity_scene_values.json holds every value
Assets/Scenes/Scene{1..4}.unity serialize for the path (extracted as text by tests/golden/make_unity_scene_values.py).
The reference ships no other concrete input of the path; this is the part of it that CAN pin something.  CPU only."""
import json
import os

import numpy as np
import pytest

from sph_amd import scenes
from sph_amd.host import CircularMotion
from tests.test_host import RecordingEngine

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "unity_scene_values.json")
VALUES = json.load(open(FIXTURE))["scenes"]


def f32(x):
    return np.asarray(x, dtype=np.float32)


@pytest.mark.parametrize("which", [1, 2, 3, 4])
def test_unity_scene_equals_the_serialized_scene(which):
    want = VALUES[f"Scene{which}"]
    sc = scenes.unity_scene(RecordingEngine(), which)
    s = sc.sph
    for key, val in want["sph"].items():
        if key == "totalCells":                      # computed at Start (SPH.cs:79); serialized as 0
            continue
        got = getattr(s, key)
        assert np.array_equal(f32(got), f32(val)), f"Scene{which}.unity SPH.{key}: {val}, builder has {got}"
    cube = want["boundaryCube"]
    assert (s.boundaryCube is None) == (cube is None)
    if cube:
        assert np.array_equal(f32(s.boundaryCube.position), f32(cube["position"]))
        assert np.array_equal(f32(s.boundaryCube.rotation), f32(cube["rotation"]))
        assert np.array_equal(f32(s.boundaryCube.localScale), f32(cube["localScale"]))
    ob = want["obstacle"]
    assert (s.obstacleCollider is None) == (ob is None)
    movers = [m for m in sc.movers if isinstance(m, CircularMotion)]
    if ob:
        assert ob["center"] == [0, 0, 0], "SPH.UpdateObstacle ignores SphereCollider.center (SPH.cs:350-354)"
        assert float(s.obstacleCollider.radius) == ob["radius"]
        assert np.array_equal(f32(s.obstacleCollider.transform.position), f32(ob["transform"]["position"]))
        assert np.array_equal(f32(s.obstacleCollider.transform.localScale), f32(ob["transform"]["localScale"]))
        if ob["mover"] and ob["mover"]["enabled"]:
            assert len(movers) == 1 and movers[0].transform is s.obstacleCollider.transform
            assert np.array_equal(f32(movers[0].center), f32(ob["mover"]["center"]))
            assert (float(movers[0].radius), float(movers[0].speed)) == (ob["mover"]["radius"], ob["mover"]["speed"])
        else:
            assert not movers
    boxes = [b for b in want["spawnBoxes"] if b["enabled"] and b["active"]]
    assert len(s.spawnBoxes) == len(boxes)
    for got, b in zip(s.spawnBoxes, boxes):          # order = order of SPH.spawnBoxes: it numbers the phases (SPH.cs:247-281)
        assert got.particleCount == b["particleCount"]
        for key in ("spawnSize", "initialVelocity", "particleColor"):
            assert np.array_equal(f32(getattr(got, key)), f32(b[key])), f"Scene{which} spawn box {key}"
        for key in ("restDensity", "viscosity", "particleMass"):
            assert f32(getattr(got, key)) == f32(b[key]), f"Scene{which} spawn box {key}"
        assert np.array_equal(f32(got.SpawnCenter), f32(b["position"]))


@pytest.mark.parametrize("which", [2, 3])
def test_scene_loaded_from_json_spawns_the_same_particles(which):
    """scenes.load_scene on the fixture = the coded builder: same parameter block, same 72-byte records."""
    from oracle.ora import Oracle
    a, b = Oracle(), Oracle()
    sa = scenes.unity_scene(a, which, scale_count=0.005)
    sb = scenes.load_scene(b, FIXTURE, name=f"Scene{which}", scale_count=0.005)
    sa.start()
    sb.start()
    assert bytes(sa.sph.build_params()) == bytes(sb.sph.build_params())
    assert np.array_equal(a.download_particles(), b.download_particles())
    for _ in range(2):
        sa.frame()
        sb.frame()
    assert np.array_equal(a.download_particles(), b.download_particles())
    assert len(sa.movers) == len(sb.movers)
    a.destroy()
    b.destroy()


def test_fixture_is_what_the_extraction_script_writes():
    """Here (not on the GPU box) the reference is present: the committed fixture must be the script's output."""
    ref = "/root/reference/Assets/Scenes/Scene1.unity"
    if not os.path.exists(ref):
        pytest.skip("reference not present")
    from tests.golden import make_unity_scene_values as mk
    for k in (1, 2, 3, 4):
        got = json.loads(json.dumps(mk.extract(f"/root/reference/Assets/Scenes/Scene{k}.unity")))
        assert got == VALUES[f"Scene{k}"]
