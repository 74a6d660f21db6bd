"""Raw-state files and viewer exports around the 72-byte particle record.

The reference has no checkpoint or export code: its only "wire format" is the particle ComputeBuffer
itself, 18 fp32 per particle in id order (SPH.cs:54-65, 248; ParticleRender.shader:54-67 reads the same
struct).  A scene reload restarts from a fresh random spawn (SceneManager.cs).  This module keeps that
record as the file format, so a Unity build could `SetData` the file unchanged:

  <prefix>.bin    particleCount x 72 bytes, id order, little-endian fp32 -- exactly the buffer contents
  <prefix>.phase  particleCount x uint8, particleSpawnBoxIndices (SPH.cs:75, 276)
  <prefix>.json   what else SPH.Update's next frame depends on: frame number (Time.time drives the
                  movers, rule O8), the lagged grid (SPH.cs:176-185 only regrids past a dead zone, so
                  the grid is state, not a function of the cube), the Inspector fields and spawn boxes

save_state / load_state resume a run bit for bit (tests/test_statefile_gpu.py): the engine's results
depend on the (cell, id) order only, never on the slot order the previous frames left behind.  The same
file serves slab-decomposed runs (sph_amd.slab): rank 0 writes it from the gathered owned particles, every
rank loads it and keeps its owned planes; slab runs are bit-identical to one-GPU runs, so a file moves
between any numbers of GPUs (tests/test_slab_gpu.py, tests/test_slab_gloo.py: test_checkpoints_*).

write_ply / splat_png are for visual regression against the reference's screenshots (readme_files/):
a binary PLY point cloud any viewer opens, and a small orthographic point splat with no dependencies.
"""
from __future__ import annotations

import json
import struct
import zlib

import numpy as np

from . import abi

FORMAT = "sph-raw-state"
VERSION = 1

_FIELDS = ("soundSpeed", "gamma", "smoothingRadius", "gravity", "timeStep", "subSteps", "surfaceTensionCoefficient",
           "surfaceTensionThreshold", "xsphEpsilon", "obstacleRepulsionStiffness", "particleCollisionDamping",
           "maxParticlesPerCell", "gridResolution", "seed")


def _l(v):
    return [float(x) for x in np.asarray(v).reshape(-1)]


def _phase_by_id(sph) -> np.ndarray:
    """particleSpawnBoxIndices (SPH.cs:75, 276); a slab run does not materialise it: ids are dealt box by box."""
    if sph.particleSpawnBoxIndices is not None:
        return np.ascontiguousarray(sph.particleSpawnBoxIndices, dtype=np.uint8)
    counts = [sb.particleCount for sb in sph.spawnBoxes]
    if sum(counts) == sph.particleCount and len(counts) <= 256:
        return np.repeat(np.arange(len(counts), dtype=np.uint8), counts)
    # (all-zero phases would be accepted by load_state and give every particle phase 0's mass, viscosity and rest density)
    raise ValueError(f"cannot derive the phase of every particle: {len(counts)} spawn boxes hold {sum(counts)} particles, "
                     f"the run has {sph.particleCount}")


def save_state(scene, prefix: str) -> dict:
    """Write <prefix>.bin/.phase/.json for a started host.Scene.  Synchronises.  On a slab-decomposed run
    (sph_amd.slab.SlabEngine) the call is collective: the owned particles of all ranks are gathered on rank 0, which
    writes the SAME file a one-GPU run writes at that frame (slab runs are bit-identical to it), so a run can be resumed
    on any number of GPUs; the other ranks return None."""
    sph = scene.sph
    e = sph.engine
    if hasattr(e, "resync"):                       # slab run
        aos = e.download_particles(root=0)
        if aos is None:
            return None
    else:
        aos = sph.GetParticleBuffer()
    aos = np.ascontiguousarray(aos, dtype="<f4")
    if aos.shape != (sph.particleCount, abi.PARTICLE_FLOATS):
        raise ValueError(f"particle buffer has shape {aos.shape}")
    aos.tofile(prefix + ".bin")
    # a slab run writes the phases its records actually carry (gathered with them); otherwise particleSpawnBoxIndices
    carried = getattr(e, "gathered_phase", None) if hasattr(e, "resync") else None
    phase = _phase_by_id(sph) if carried is None else np.ascontiguousarray(carried, dtype=np.uint8)
    if phase.shape[0] != sph.particleCount or (phase.size and int(phase.max()) >= max(len(sph.spawnBoxes), 1)):
        raise ValueError("phase ids do not match the particles / spawn boxes")
    np.ascontiguousarray(phase, dtype=np.uint8).tofile(prefix + ".phase")
    meta = {
        "format": FORMAT, "version": VERSION, "record_bytes": abi.PARTICLE_BYTES, "particleCount": int(sph.particleCount),
        "frame": int(scene.time.frame), "frame_period": float(scene.time.frame_period),
        "fields": {k: (int(getattr(sph, k)) if isinstance(getattr(sph, k), (int, np.integer)) else float(getattr(sph, k)))
                   for k in _FIELDS},
        "boundsCenter": _l(sph.boundsCenter), "boundsSize": _l(sph.boundsSize),
        "grid": {"min": _l(sph.gridMin), "cellSize": float(sph.cellSize),
                 "resolution": [int(sph.gridResolutionX), int(sph.gridResolutionY), int(sph.gridResolutionZ)]},
        "boundaryCube": None if sph.boundaryCube is None else {
            "position": _l(sph.boundaryCube.position), "rotation": _l(sph.boundaryCube.rotation),
            "localScale": _l(sph.boundaryCube.localScale),
            "lastPosition": _l(sph.lastBoundaryCubePosition), "lastRotation": _l(sph.lastBoundaryCubeRotation)},
        "spawnBoxes": [{"particleCount": int(sb.particleCount), "spawnSize": _l(sb.spawnSize), "restDensity": sb.restDensity,
                        "viscosity": sb.viscosity, "particleMass": sb.particleMass, "particleColor": _l(sb.particleColor),
                        "position": _l(sb.SpawnCenter)} for sb in sph.spawnBoxes],
        "crc32": {"bin": zlib.crc32(aos.tobytes()) & 0xFFFFFFFF},
    }
    with open(prefix + ".json", "w") as f:
        json.dump(meta, f, indent=1)
    return meta


def read_state(prefix: str):
    """(meta, aos[n,18] float32, phase[n] uint8) of a state file; checks sizes and the checksum."""
    with open(prefix + ".json") as f:
        meta = json.load(f)
    if meta.get("format") != FORMAT or meta.get("version") != VERSION:
        raise ValueError(f"{prefix}.json is not a {FORMAT} v{VERSION} file")
    n = int(meta["particleCount"])
    aos = np.fromfile(prefix + ".bin", dtype="<f4")
    if aos.size != n * abi.PARTICLE_FLOATS:
        raise ValueError(f"{prefix}.bin holds {aos.size * 4} bytes, expected {n} records of {abi.PARTICLE_BYTES}")
    if zlib.crc32(aos.tobytes()) & 0xFFFFFFFF != meta["crc32"]["bin"]:
        raise ValueError(f"{prefix}.bin does not match the checksum in {prefix}.json")
    phase = np.fromfile(prefix + ".phase", dtype=np.uint8)
    if phase.size != n:
        raise ValueError(f"{prefix}.phase holds {phase.size} entries, expected {n}")
    return meta, aos.reshape(n, abi.PARTICLE_FLOATS).astype(np.float32), phase


def load_state(scene, prefix: str) -> dict:
    """Put a started host.Scene of the SAME scene definition back into the saved state: particles, phase ids, frame
    number, movers, and the lagged grid.  The next scene.frame() continues the saved run bit for bit."""
    meta, aos, phase = read_state(prefix)
    sph = scene.sph
    if not scene.started:
        scene.start()
    if sph.particleCount != meta["particleCount"]:
        raise ValueError(f"scene has {sph.particleCount} particles, the file {meta['particleCount']}")
    if len(sph.spawnBoxes) != len(meta["spawnBoxes"]) or int(phase.max(initial=0)) >= max(len(sph.spawnBoxes), 1):
        raise ValueError("spawn boxes of the scene and of the file differ")
    for k, v in meta["fields"].items():
        if getattr(sph, k) != v:
            raise ValueError(f"scene field {k} = {getattr(sph, k)!r}, file has {v!r}")
    e = sph.engine
    scene.time.frame = int(meta["frame"])
    cube = meta["boundaryCube"]
    if (cube is None) != (sph.boundaryCube is None):
        raise ValueError("boundary cube of the scene and of the file differ")
    g = meta["grid"]
    sph.gridMin = np.asarray(g["min"], dtype=np.float32)
    sph.cellSize = np.float32(g["cellSize"])
    sph.gridResolutionX, sph.gridResolutionY, sph.gridResolutionZ = (int(v) for v in g["resolution"])
    sph.totalCells = sph.gridResolutionX * sph.gridResolutionY * sph.gridResolutionZ
    e.resize_grid(sph.gridResolutionX, sph.gridResolutionY, sph.gridResolutionZ, sph.maxParticlesPerCell)
    if cube is not None:
        sph.boundaryCube.position = np.asarray(cube["position"], dtype=np.float32)
        sph.boundaryCube.rotation = np.asarray(cube["rotation"], dtype=np.float32)
        sph.lastBoundaryCubePosition = np.asarray(cube["lastPosition"], dtype=np.float32)
        sph.lastBoundaryCubeRotation = np.asarray(cube["lastRotation"], dtype=np.float32)
    if hasattr(e, "resync"):
        # slab run: every rank reads the file; a grid that differs from the one the run started with re-partitions the
        # slabs (SetComputeParams -> SlabEngine.set_params), then every rank keeps the particles of its owned planes
        sph.SetComputeParams()
        e.set_phase_table(sph._phase_rows())
        e.resync(aos, phase)
        return meta
    e.upload_particles(aos)
    sph.particleSpawnBoxIndices = phase.copy()
    e.set_phase_table(sph._phase_rows())
    e.set_phase_ids(phase)
    sph.SetComputeParams()
    return meta


# ------------------------------------------------------------------------------------------------
# viewer exports
# ------------------------------------------------------------------------------------------------
def write_ply(path: str, aos: np.ndarray, with_fields: bool = True) -> None:
    """Binary little-endian PLY point cloud: x y z, rgba from the record's colour (what the billboard
    shader tints with, ParticleRender.shader:100-114), optionally density / pressure / speed."""
    aos = np.asarray(aos, dtype=np.float32).reshape(-1, abi.PARTICLE_FLOATS)
    n = aos.shape[0]
    cols = [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1"), ("alpha", "u1")]
    if with_fields:
        cols += [("density", "<f4"), ("pressure", "<f4"), ("speed", "<f4")]
    v = np.zeros(n, dtype=np.dtype(cols))
    v["x"], v["y"], v["z"] = aos[:, 0], aos[:, 1], aos[:, 2]
    rgba = np.clip(np.nan_to_num(aos[:, 14:18]) * 255.0 + 0.5, 0, 255).astype(np.uint8)
    v["red"], v["green"], v["blue"], v["alpha"] = rgba[:, 0], rgba[:, 1], rgba[:, 2], rgba[:, 3]
    if with_fields:
        v["density"], v["pressure"] = aos[:, 9], aos[:, 10]
        v["speed"] = np.sqrt((aos[:, 3:6].astype(np.float64) ** 2).sum(axis=1)).astype(np.float32)
    names = {"<f4": "float", "u1": "uchar"}
    header = "ply\nformat binary_little_endian 1.0\ncomment sph_amd particle state\n" + f"element vertex {n}\n"
    header += "".join(f"property {names[t]} {k}\n" for k, t in cols) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(v.tobytes())


def read_ply_vertices(path: str) -> np.ndarray:
    """Reads back what write_ply wrote (round-trip check)."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    lines = data[:end].decode("ascii").splitlines()
    n = next(int(ln.split()[2]) for ln in lines if ln.startswith("element vertex"))
    types = {"float": "<f4", "uchar": "u1"}
    cols = [(ln.split()[2], types[ln.split()[1]]) for ln in lines if ln.startswith("property")]
    return np.frombuffer(data, dtype=np.dtype(cols), count=n, offset=end)


def _png(path: str, rgb: np.ndarray) -> None:
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, payload):
        c = struct.pack(">I", len(payload)) + tag + payload
        return c + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def splat_png(path: str, aos: np.ndarray, lo, hi, axes=(0, 1), width: int = 512, background=(16, 16, 24)) -> np.ndarray:
    """Orthographic point splat of the particles onto the plane of `axes` (default: front view x-y), depth-sorted
    along the third axis (nearest last), one pixel per particle, tinted with the particle colour.  lo/hi: world-space
    box shown (e.g. the boundary cube's AABB).  Returns the image (h, w, 3) uint8 it wrote."""
    aos = np.asarray(aos, dtype=np.float32).reshape(-1, abi.PARTICLE_FLOATS)
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    a, b = axes
    c = 3 - a - b
    height = max(1, int(round(width * (hi[b] - lo[b]) / (hi[a] - lo[a]))))
    img = np.empty((height, width, 3), dtype=np.uint8)
    img[:] = np.asarray(background, dtype=np.uint8)
    ok = np.isfinite(aos[:, 0:3]).all(axis=1)
    p = aos[ok]
    px = np.floor((p[:, a] - lo[a]) / (hi[a] - lo[a]) * width).astype(np.int64)
    py = height - 1 - np.floor((p[:, b] - lo[b]) / (hi[b] - lo[b]) * height).astype(np.int64)
    inside = (px >= 0) & (px < width) & (py >= 0) & (py < height)
    p, px, py = p[inside], px[inside], py[inside]
    order = np.argsort(-p[:, c], kind="stable")           # far first, near overwrites
    rgb = np.clip(np.nan_to_num(p[:, 14:17]) * 255.0 + 0.5, 0, 255).astype(np.uint8)
    img[py[order], px[order]] = rgb[order]
    _png(path, img)
    return img


def main(argv=None):
    """python -m sph_amd.statefile <prefix> [--ply out.ply] [--png out.png --axes xy|xz|zy --width W]"""
    import argparse
    ap = argparse.ArgumentParser(description="export a raw-state file for viewers")
    ap.add_argument("prefix")
    ap.add_argument("--ply")
    ap.add_argument("--png")
    ap.add_argument("--axes", default="xy", choices=["xy", "xz", "zy"])
    ap.add_argument("--width", type=int, default=768)
    a = ap.parse_args(argv)
    meta, aos, _ = read_state(a.prefix)
    if a.ply:
        write_ply(a.ply, aos)
    if a.png:
        g = meta["grid"]
        lo = np.asarray(g["min"], dtype=np.float64)
        hi = lo + np.asarray(g["resolution"], dtype=np.float64) * g["cellSize"]
        axes = tuple("xyz".index(ch) for ch in a.axes)
        splat_png(a.png, aos, lo, hi, axes=axes, width=a.width)
    print(json.dumps({"particles": meta["particleCount"], "frame": meta["frame"], "ply": a.ply, "png": a.png}))


if __name__ == "__main__":
    main()
