"""Calibration runs for tests/test_readme_pictures_gpu.py: the four outcomes the reference's README pictures show
(readme_files/MuliPhase1.PNG, box.jpg, Splash.jpg, Ball.jpg), free-running on the engine; prints the statistics the
tests bound.  GPU.  python scripts/readme_probe.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from sph_amd import abi, statefile, readme_scenes as rs          # noqa: E402

# front views written next to the statistics (gpurun_out/readme_<scene>.png): x-y window of every tank
VIEW = dict(two_phase_tank=((-2, 0, -1), (2, 3, 1)), tilted_cube=((-4.5, -1, -1), (4.5, 5, 1)), splash=((-2.5, 0, -2.5), (2.5, 2.5, 2.5)),
            ball=((-4, 0, -4), (4, 4, 4)))


def main():
    for name in ("two_phase_tank", "tilted_cube", "splash", "ball"):
        eng = abi.Engine(0)
        sc = getattr(rs, name)(eng)
        sc.sph.use_step_frame = True
        sc.start()
        t0 = time.perf_counter()
        for f in range(1, rs.FRAMES[name] + 1):
            sc.frame()
            if f % 250 == 0 or f == rs.FRAMES[name]:
                P = eng.download_particles()
                print(name, f, {k: (np.round(v, 3).tolist() if isinstance(v, np.ndarray) else round(float(v), 4))
                                for k, v in getattr(rs, name + "_stats")(sc, P).items()}, flush=True)
        print(name, "wall %.1f s" % (time.perf_counter() - t0), "nan", eng.read_counter(abi.CTR_NAN), flush=True)
        lo, hi = VIEW[name]
        axes = (0, 2) if name == "ball" else (0, 1)          # the ball's hole shows from above
        statefile.splat_png(f"gpurun_out/readme_{name}.png", eng.download_particles(), lo=lo, hi=hi, axes=axes, width=480)
        eng.destroy()


if __name__ == "__main__":
    main()
