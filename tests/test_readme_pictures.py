"""The same outcomes on the ORACLE (CPU), 250 frames: what tests/test_readme_pictures_gpu.py asks of the engine is what
the restated reference algorithm does -- the light phase is on top and the tilted cube's surface level by then."""
import numpy as np

from sph_amd import readme_scenes as rs


def _run(name, frames):
    from oracle.ora import Oracle
    ora = Oracle(omp=True)
    sc = getattr(rs, name)(ora)
    sc.start()
    for _ in range(frames):
        sc.frame()
    P = ora.download_particles()
    st = getattr(rs, name + "_stats")(sc, P)
    ora.destroy()
    assert np.isfinite(P).all()
    return st


def test_oracle_two_phase_tank_light_over_dense(oracle_built):
    st = _run("two_phase_tank", 250)
    assert st["light_share_of_top"] >= 0.95 and st["y_light"] - st["y_dense"] >= 0.7, st


def test_oracle_tilted_cube_level_surface(oracle_built):
    st = _run("tilted_cube", 250)
    assert abs(st["surface_slope"]) <= 0.06 < st["floor_slope"] and st["max_local"] <= 0.5 + 1e-4, st
