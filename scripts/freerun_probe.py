import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from oracle.ora import Oracle
from sph_amd import abi, scenes
from tests.test_freerun_gpu import aggregates
def run(builder, frames, marks):
    eng, ora = abi.Engine(0), Oracle(omp=True)
    se, so = builder(eng), builder(ora)
    se.sph.use_step_frame = True
    se.start(); so.start()
    for f in range(1, frames+1):
        se.frame(); so.frame()
        if f in marks:
            P, Q = eng.download_particles(), ora.download_particles()
            for ph, name in ((20.0,'A'),(10.0,'B')):
                a = aggregates(P[P[:,11]==ph]); b = aggregates(Q[Q[:,11]==ph])
                print(f, name, {k: (np.round(a[k],4), np.round(b[k],4)) for k in a})
    eng.destroy()
run(lambda e: scenes.unity_scene(e, which=4, scale_count=0.05), 600, (10, 100, 300, 600))
