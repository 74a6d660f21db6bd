"""The outcomes the reference's README pictures show, on the engine in free run.  -m gpu.

readme_files/*.{PNG,jpg} are the only OUTPUTS of the path the reference holds: four pictures of end states.  The scenes
of sph_amd/readme_scenes.py repeat them in small with the reference's own Inspector values, and the end states are held
to what the pictures show (bounds: several times the spread seen over 2,500 frames on MI355X, scripts/readme_probe.py):

  MuliPhase1.PNG  the light phase (rest density 10, yellow) lies on the dense one (20, blue), level interface; the
                  rendered front view (statefile.splat_png) has the picture's arrangement of the two colour families
                  (tests/golden/readme_pictures.json: pixel statistics of the picture, make_readme_picture_values.py)
  box.jpg         in a boundary cube tilted by 25 degrees the free surface is level in WORLD space and the fluid sits in
                  the cube's lowest edge, inside the cube
  Splash.jpg      a block dropped into a box ends as a level layer over the whole floor
  Ball.jpg        the moving sphere ploughs through a shallow layer: no particle is ever inside it
"""
import json
import os

import numpy as np
import pytest

from sph_amd import abi, statefile
from sph_amd import readme_scenes as rs
from tests.golden.make_readme_picture_values import describe

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
PICTURES = json.load(open(os.path.join(HERE, "golden", "readme_pictures.json")))


def _run(name, frames, every=None, check=None):
    eng = abi.Engine(0)
    sc = getattr(rs, name)(eng)
    sc.sph.use_step_frame = True
    sc.start()
    for f in range(1, frames + 1):
        sc.frame()
        if every and f % every == 0:
            check(sc, eng.download_particles(), f)
    P = eng.download_particles()
    assert np.isfinite(P).all() and eng.read_counter(abi.CTR_NAN) == 0
    st = getattr(rs, name + "_stats")(sc, P)
    eng.destroy()
    return sc, P, st


def test_two_phase_tank_settles_light_over_dense_like_the_picture(tmp_path):
    sc, P, st = _run("two_phase_tank", 1500)
    assert st["light_share_of_top"] >= 0.97, st          # 0.994-0.999 from frame 500 on
    assert st["y_light"] - st["y_dense"] >= 0.8, st      # 1.0-1.2
    assert st["dense_top_p95"] <= st["light_bottom_p05"] + 0.05, st     # the layers do not interleave
    assert st["interface_std"] <= 0.12, st               # level interface
    assert st["ke"] <= 0.25, st                          # settled (residual jitter 0.04-0.08)
    # the picture: same two colour families, same arrangement
    img = statefile.splat_png(str(tmp_path / "two_phase.png"), P, lo=(-2, 0, -1), hi=(2, 3, 1), width=400)
    mine, ref = describe(img), PICTURES["MuliPhase1.PNG"]
    assert mine["yellow"]["pixels"] > 500 and mine["blue"]["pixels"] > 500
    assert (mine["yellow"]["mean_row"] < mine["blue"]["mean_row"]) == (ref["yellow"]["mean_row"] < ref["blue"]["mean_row"]) is True
    assert mine["yellow_share_of_upper_half"] >= ref["yellow_share_of_upper_half"] - 0.1, (mine, ref)
    assert mine["blue_share_of_lower_half"] >= ref["blue_share_of_lower_half"] - 0.1, (mine, ref)


def test_tilted_cube_free_surface_is_level_in_world_space():
    sc, P, st = _run("tilted_cube", 800)
    assert st["bins"] >= 6, st
    assert abs(st["surface_slope"]) <= 0.05 < 0.4 < st["floor_slope"], st          # measured +-0.01; the floor slopes 0.466
    assert st["max_local"] <= 0.5 + 1e-4, st                                        # inside the cube (compute:349-419)
    # the cube's lowest edge: local corner (-3, -4) of the 6 x 8 cross-section, rotated by 25 degrees about z at (0, 4)
    t = np.radians(rs.TILT_DEG)
    corner = np.array([-3 * np.cos(t) + 4 * np.sin(t), 4 - 3 * np.sin(t) - 4 * np.cos(t)])
    assert np.linalg.norm(st["lowest"] - corner) <= 0.3, (st["lowest"], corner)
    assert st["ke"] <= 0.2, st


def test_dropped_block_ends_as_a_level_layer_over_the_floor():
    sc, P, st = _run("splash", 800)
    assert st["covered"] == 1.0, st
    assert st["surface_std"] <= 0.1 and st["y_max"] <= 1.0, st      # 0.034 / 0.62
    assert st["ke"] <= 0.06, st


def test_moving_sphere_keeps_every_particle_outside():
    seen = []

    def check(sc, P, f):
        s = rs.ball_stats(sc, P)
        assert s["min_dist_over_radius"] >= 1.0 - 1e-5, (f, s)      # compute:467-483 projects onto the surface
        seen.append(s["in_shell"])

    sc, P, st = _run("ball", 600, every=25, check=check)
    assert min(seen) > 50, "the sphere never touched the layer"
    assert st["sphere_bottom"] < 0.0 < st["layer_top"] < 1.0, st   # it reaches through the whole layer: a hole, not a dent
