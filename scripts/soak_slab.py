"""Long-run check of the slab path on one GPU (ranks share it, gloo): F frames of the long tank of
tests/test_slab_gpu.py on W ranks with the overlapped exchange, against one engine, bit for bit.
usage: python scripts/soak_slab.py [frames] [world] [ghost]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch.multiprocessing as mp
from sph_amd import abi
from tests.test_slab_gloo import _free_port
from tests.test_slab_gpu import _worker, long_tank

def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ghost = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    eng = abi.Engine(0)
    sc = long_tank(eng)
    sc.sph.use_step_frame = True
    sc.run(frames)
    want = eng.download_particles()
    print("one engine:", frames, "frames,", want.shape[0], "particles, finite", bool(np.isfinite(want).all()),
          "overflow", eng.read_counter(abi.CTR_BIN_OVERFLOW), "out of grid", eng.read_counter(abi.CTR_OUT_OF_GRID), flush=True)
    eng.destroy()
    n = want.shape[0]
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, _free_port(), frames, tmp, n, ghost, False, 0, 1, True, "long"), nprocs=world, join=True)
        recs = np.concatenate([np.load(os.path.join(tmp, f"rank{r}.npy")) for r in range(world)])
        meta = np.load(os.path.join(tmp, "meta0.npy"))
        band = [float(np.load(os.path.join(tmp, f"band{r}.npy"))[0]) for r in range(world)]
    got = abi.aos_from_records(recs[np.argsort(recs["id"])])
    same = all(np.array_equal(got[:, sl], want[:, sl]) for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 14)))
    print("slabs:", world, "ranks, conserved", meta[:-1].tolist() == [n] * frames, "violations+misses", int(meta[-1]),
          "band fractions", [round(b, 3) for b in band], "bitwise equal", same, flush=True)
    sys.exit(0 if same and meta[-1] == 0 else 1)


if __name__ == "__main__":
    main()
