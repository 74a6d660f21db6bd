"""Raw-state files (sph_amd/statefile.py): format, resume and viewer exports, on the oracle-backed host (CPU)."""
import json
import os

import numpy as np
import pytest

from sph_amd import abi, scenes, statefile


def _resume_case(make_engine, tmp_path, builder, frames_a, frames_b, **kw):
    """Run frames_a + frames_b frames in one go; and frames_a, save, fresh engine, load, frames_b.  Bitwise equal."""
    e1 = make_engine()
    s1 = builder(e1, **kw)
    s1.start()
    s1.run(frames_a)
    prefix = str(tmp_path / "state")
    meta = statefile.save_state(s1, prefix)
    s1.run(frames_b)
    want = s1.sph.GetParticleBuffer()
    e1.destroy()
    e2 = make_engine()
    s2 = builder(e2, **kw)
    statefile.load_state(s2, prefix)
    assert s2.time.frame == frames_a == meta["frame"]
    s2.run(frames_b)
    got = s2.sph.GetParticleBuffer()
    e2.destroy()
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), "resumed run differs from the uninterrupted one"
    return prefix, meta


def test_file_is_the_72_byte_buffer(tmp_path, oracle_built):
    from oracle.ora import Oracle
    ora = Oracle()
    sc = scenes.unity_scene(ora, which=2, scale_count=0.004)
    sc.start()
    sc.run(2)
    prefix = str(tmp_path / "s")
    meta = statefile.save_state(sc, prefix)
    n = sc.sph.particleCount
    assert os.path.getsize(prefix + ".bin") == n * 72 and os.path.getsize(prefix + ".phase") == n
    raw = np.fromfile(prefix + ".bin", dtype="<f4").reshape(n, 18)
    assert np.array_equal(raw.view(np.uint32), ora.download_particles().view(np.uint32))
    assert meta["record_bytes"] == abi.PARTICLE_BYTES == 72 and json.load(open(prefix + ".json"))["frame"] == 2
    m2, aos, phase = statefile.read_state(prefix)
    assert np.array_equal(phase, sc.sph.particleSpawnBoxIndices) and aos.shape == (n, 18)
    # a damaged file is refused
    with open(prefix + ".bin", "r+b") as f:
        f.seek(100); f.write(b"\xff\xff")
    with pytest.raises(ValueError):
        statefile.read_state(prefix)
    ora.destroy()


def test_resume_is_bitwise_oracle_host(tmp_path, oracle_built):
    from oracle.ora import Oracle
    # moving sphere + rocking cube: frame number, movers and the lagged grid are all part of the state
    _resume_case(Oracle, tmp_path, scenes.config3, 3, 3, n=3000, scale=0.25)


def test_load_refuses_another_scene(tmp_path, oracle_built):
    from oracle.ora import Oracle
    ora = Oracle()
    sc = scenes.config1(ora, n=512)
    sc.start()
    prefix = str(tmp_path / "c1")
    statefile.save_state(sc, prefix)
    other = scenes.config1(Oracle(), n=256)
    with pytest.raises(ValueError):
        statefile.load_state(other, prefix)
    other2 = scenes.unity_scene(Oracle(), which=1, scale_count=512 / 200000.0)
    with pytest.raises(ValueError):
        statefile.load_state(other2, prefix)


def test_ply_round_trip_and_splat(tmp_path, oracle_built):
    from oracle.ora import Oracle
    ora = Oracle()
    sc = scenes.unity_scene(ora, which=2, scale_count=0.004)
    sc.start()
    sc.run(1)
    aos = ora.download_particles()
    path = str(tmp_path / "p.ply")
    statefile.write_ply(path, aos)
    v = statefile.read_ply_vertices(path)
    assert v.shape[0] == aos.shape[0]
    assert np.array_equal(v["x"], aos[:, 0]) and np.array_equal(v["z"], aos[:, 2]) and np.array_equal(v["density"], aos[:, 9])
    # Scene2's own colours (Scene2.unity:499, 594): r = 0.0039215684 (1/255) and 1
    assert set(np.unique(v["red"])) == {int(scenes.SCENE_BLUE[2][0] * 255 + 0.5), 255} == {1, 255}
    img = statefile.splat_png(str(tmp_path / "front.png"), aos, lo=(-10, 0, -10), hi=(10, 30, 10), width=200)
    assert img.shape == (300, 200, 3)
    lit = (img != np.array([16, 16, 24], dtype=np.uint8)).any(axis=2)
    assert 0 < lit.sum() <= aos.shape[0]
    ys, xs = np.nonzero(lit)
    # the two spawn boxes sit at y 15..25 of a 0..30 cube: rows 50..150 of 300
    assert ys.min() >= 45 and ys.max() <= 155
    data = open(str(tmp_path / "front.png"), "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and data[-8:-4] == b"IEND"
