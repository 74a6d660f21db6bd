import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sph_amd import abi, slab
from tests.test_slab_gloo import small_tank
class NoComm:
    rank=0; world=1
    def exchange(self, a, b):
        e=torch.empty((0,64),dtype=torch.uint8,device='cuda'); return e,e
    def allreduce_sum(self,v): return v
eng = abi.Engine(0)
port = slab.DevicePort(eng, 0, 4096)
se = slab.SlabEngine(port, NoComm(), 0, 1, ghost_planes=8)
sc = small_tank(se, n=3000); sc.sph.use_step_frame=True; sc.start()
prev = None
for f in range(8):
    before = se.owned_records() if f>0 else None
    sc.frame()
    r = se.owned_records(); r = r[np.argsort(r['id'])]
    sh = eng.read_counter(abi.CTR_MAX_PLANE_SHIFT)
    if prev is not None:
        dx = np.abs(r['pos'][:,0]-prev['pos'][:,0]).max()/0.25
    else: dx = 0
    print(f, 'ctr shift', sh, 'true max |dx|/cell', round(float(dx),3), 'max|v|', float(np.abs(r['vel']).max()))
    prev = r
