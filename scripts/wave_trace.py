"""Per-wave timeline of the last density / force launch (debug build of the library, -DSPH_WG_TRACE): how long the
waves run, how many are resident over the launch, how much of the launch is tail.
  SPH_HIP_LIB=$PWD/sph_amd/libsph_hip_trace.so python scripts/wave_trace.py [frames=25] [n=1048576]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import abi, scenes

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 25
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
e = abi.Engine(0)
sc = scenes.config3(e, n=n)
sc.sph.use_step_frame = True
sc.start()
for _ in range(frames):
    sc.frame()
e.synchronize()
lib = abi.load_library()
waves = (n + 63) // 64
for k, name in ((0, "density"), (1, "force")):
    buf = np.zeros((waves, 3), dtype=np.uint64)
    rc = lib.sph_debug_wave_trace(k, buf.ctypes.data_as(C.c_void_p), waves)
    assert rc == 0, rc
    t0, t1 = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64)
    ok = t1 > 0
    t0, t1, hw = t0[ok], t1[ok], buf[ok, 2]
    start = t0.min()
    t0, t1 = (t0 - start) / 100.0, (t1 - start) / 100.0          # us (100 MHz)
    span = t1.max()
    dur = t1 - t0
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
    hwid = hw.astype(np.int64) & 0xFFFFFFFF
    cu = (hwid >> 8) & 0xF
    se = (hwid >> 13) & 0x7
    sh = (hwid >> 12) & 0x1
    simd = (hwid >> 4) & 0x3
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"== {name}: {ok.sum()} waves, launch span {span:.1f} us; wave duration mean {dur.mean():.1f} p50 {np.median(dur):.1f} "
          f"p90 {np.percentile(dur, 90):.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f} us")
    print(f"   sum of wave durations / span = {dur.sum() / span:.0f} resident waves on average; distinct CUs seen {np.unique(cuid).size}, "
          f"XCCs {np.unique(xcc).size}")
    edges = np.linspace(0, span, 21)
    res = [(np.minimum(t1, b) - np.maximum(t0, a)).clip(min=0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
    print("   resident waves per 5 % of the span:", " ".join(f"{r:.0f}" for r in res))
    last_start = t0.max()
    print(f"   last wave starts at {last_start:.1f} us ({last_start / span * 100:.0f} % of the span); "
          f"time with < 50 % of the peak residency: {sum(1 for r in res if r < 0.5 * max(res)) * 5} % of the span")
    # per CU: busy until
    ends = np.array([t1[cuid == c].max() for c in np.unique(cuid)])
    print(f"   per-CU last wave end: min {ends.min():.1f} p10 {np.percentile(ends, 10):.1f} p50 {np.median(ends):.1f} max {ends.max():.1f} us")
    # duration by dispatch order (quartiles of the wave index)
    idx = np.nonzero(ok)[0]
    q = np.array_split(np.arange(idx.size), 8)
    print("   mean duration by eighth of the grid (dispatch order):", " ".join(f"{dur[i].mean():.1f}" for i in q))
    print("   mean start by eighth of the grid:", " ".join(f"{t0[i].mean():.1f}" for i in q))
e.destroy()
