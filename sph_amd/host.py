"""Headless host driver: the Python mirror of the reference's Unity scripts.

  SPH            <- Assets/SPH.cs           (same public fields, Start/Update/OnDestroy, same
                                              per-frame dispatch order, SPH.cs:173-223)
  SpawnBox       <- Assets/SpawnBox.cs      (SpawnBox.cs:7-25)
  CircularMotion <- Assets/SphereMover.cs   (SphereMover.cs:12-21)
  Transform / SphereCollider / Time         the slivers of UnityEngine those scripts touch

Where SPH.cs calls UnityEngine.ComputeShader / ComputeBuffer, this calls an *engine* object
with the libsph_hip method names (sph_amd.abi.Engine in production).  All host arithmetic that
reaches the kernels (grid sizing, matrices) is done in fp32 like Unity's Vector3/Matrix4x4.

Build-defined pieces (the reference has no counterpart that can run headless):
  * particle spawn RNG (ORACLE rule O9): a counter-based generator replaces UnityEngine.Random;
  * Time: fixed frame period 1/60 s, frame 0 at t = 0 (ORACLE rule O8);
  * RockingMotion: the boundary rotation script of BASELINE config 3 (the reference has none).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import abi

f32 = np.float32


def _v3(x) -> np.ndarray:
    return np.asarray(x, dtype=np.float32).reshape(3).copy()


# --------------------------------------------------------------------------------------------
# UnityEngine slivers
# --------------------------------------------------------------------------------------------
class Time:
    """Time.time with a fixed frame period (rule O8)."""
    frame_period = 1.0 / 60.0

    def __init__(self):
        self.frame = 0

    @property
    def time(self) -> float:
        return float(f32(self.frame) * f32(self.frame_period))

    def advance(self):
        self.frame += 1


def quat_from_axis_angle(axis, degrees) -> np.ndarray:
    """Quaternion.AngleAxis(degrees, axis) as (x, y, z, w)."""
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    half = float(degrees) * math.pi / 180.0 * 0.5      # same operation order as host/sph_host.hpp (AngleAxis)
    s = math.sin(half)
    return np.array([a[0] * s, a[1] * s, a[2] * s, math.cos(half)], dtype=np.float32)


def quat_angle(a, b) -> float:
    """Quaternion.Angle: degrees between two rotations; 0 when |dot| > 1 - 1e-6 (Unity's dead zone)."""
    d = abs(float(np.dot(np.asarray(a, np.float32), np.asarray(b, np.float32))))
    d = min(d, 1.0)
    if d > 1.0 - 1e-6:
        return 0.0
    return math.degrees(math.acos(d) * 2.0)


class Transform:
    """position / rotation / localScale of a root GameObject (no parenting)."""

    def __init__(self, position=(0, 0, 0), rotation=(0, 0, 0, 1), localScale=(1, 1, 1)):
        self.position = _v3(position)
        self.rotation = np.asarray(rotation, dtype=np.float32).reshape(4).copy()
        self.localScale = _v3(localScale)

    @property
    def lossyScale(self):
        return self.localScale

    def _state_key(self):
        return self.position.tobytes() + self.rotation.tobytes() + self.localScale.tobytes()

    @property
    def localToWorldMatrix(self) -> np.ndarray:
        """Matrix4x4.TRS(position, rotation, scale), fp32, row-major m[r, c] acting on columns.  Recomputed only when
        position / rotation / scale changed (the host asks for it ~10 times a frame: 8 corners + the parameter upload)."""
        key = self._state_key()
        if getattr(self, "_l2w_key", None) != key:
            self._l2w, self._l2w_key = self._trs(), key
        return self._l2w

    def _trs(self) -> np.ndarray:
        x, y, z, w = [f32(v) for v in self.rotation]
        two = f32(2)
        one = f32(1)
        R = np.array([
            [one - two * (y * y + z * z), two * (x * y - z * w), two * (x * z + y * w)],
            [two * (x * y + z * w), one - two * (x * x + z * z), two * (y * z - x * w)],
            [two * (x * z - y * w), two * (y * z + x * w), one - two * (x * x + y * y)],
        ], dtype=np.float32)
        M = np.zeros((4, 4), dtype=np.float32)
        M[:3, :3] = R * self.localScale[None, :]
        M[:3, 3] = self.position
        M[3, 3] = 1
        return M

    @property
    def worldToLocalMatrix(self) -> np.ndarray:
        key = self._state_key()
        if getattr(self, "_w2l_key", None) != key:
            self._w2l, self._w2l_key = self._affine_inverse(self.localToWorldMatrix), key
        return self._w2l

    @staticmethod
    def _affine_inverse(M) -> np.ndarray:
        """Inverse of the affine TRS matrix by the adjugate of its 3 x 3 part, in double, rounded once -- the same
        operations in the same order as host/sph_host.hpp (Transform::worldToLocalMatrix), so that the Python and the
        C++ host hand the kernels the same bits (a LAPACK inverse differs in the last place now and then, and one
        OBB collision later two runs of a chaotic tank have parted)."""
        a = [float(M[r, c]) for r in range(3) for c in range(3)]
        det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6])
        inv = [(a[4] * a[8] - a[5] * a[7]) / det, (a[2] * a[7] - a[1] * a[8]) / det, (a[1] * a[5] - a[2] * a[4]) / det,
               (a[5] * a[6] - a[3] * a[8]) / det, (a[0] * a[8] - a[2] * a[6]) / det, (a[2] * a[3] - a[0] * a[5]) / det,
               (a[3] * a[7] - a[4] * a[6]) / det, (a[1] * a[6] - a[0] * a[7]) / det, (a[0] * a[4] - a[1] * a[3]) / det]
        out = np.zeros((4, 4), dtype=np.float32)
        for r in range(3):
            t = 0.0
            for c in range(3):
                out[r, c] = inv[r * 3 + c]
                t -= inv[r * 3 + c] * float(M[c, 3])
            out[r, 3] = t
        out[3, 3] = 1
        return out

    def TransformPoint(self, p) -> np.ndarray:
        M = self.localToWorldMatrix
        p = _v3(p)
        out = np.empty(3, dtype=np.float32)
        for r in range(3):
            out[r] = f32(f32(f32(M[r, 0] * p[0]) + f32(M[r, 1] * p[1])) + f32(M[r, 2] * p[2])) + M[r, 3]
        return out


@dataclass
class SphereCollider:
    """UnityEngine.SphereCollider as SPH.UpdateObstacle reads it (SPH.cs:350-354)."""
    radius: float = 0.5
    transform: Transform = field(default_factory=Transform)


# --------------------------------------------------------------------------------------------
# Assets/SpawnBox.cs
# --------------------------------------------------------------------------------------------
class SpawnBox:
    def __init__(self, particleCount=1000, spawnSize=(2, 2, 2), restDensity=1000.0, viscosity=0.1,
                 particleMass=0.03, initialVelocity=(0, 0, 0), particleColor=(1, 1, 1, 1), position=(0, 0, 0)):
        self.particleCount = int(particleCount)            # SpawnBox.cs:7
        self.spawnSize = _v3(spawnSize)                    # SpawnBox.cs:8
        self.restDensity = float(restDensity)              # SpawnBox.cs:11
        self.viscosity = float(viscosity)                  # SpawnBox.cs:14
        self.particleMass = float(particleMass)            # SpawnBox.cs:17
        self.initialVelocity = _v3(initialVelocity)        # SpawnBox.cs:20
        self.particleColor = np.asarray(particleColor, dtype=np.float32).reshape(4).copy()  # SpawnBox.cs:23
        self.transform = Transform(position=position)

    @property
    def SpawnCenter(self):                                 # SpawnBox.cs:25
        return self.transform.position


# --------------------------------------------------------------------------------------------
# Assets/SphereMover.cs (class CircularMotion) and the build-defined boundary rocker
# --------------------------------------------------------------------------------------------
class CircularMotion:
    def __init__(self, transform: Transform, time: Time, center=(0, 0, 0), radius=5.0, speed=1.0):
        self.transform, self.time = transform, time
        self.center = _v3(center)
        self.radius = f32(radius)
        self.speed = f32(speed)

    def Update(self):                                      # SphereMover.cs:12-21
        angle = f32(f32(self.time.time) * self.speed)
        x = f32(self.center[0] + self.radius * f32(math.cos(float(angle))))
        z = f32(self.center[2] + self.radius * f32(math.sin(float(angle))))
        self.transform.position = np.array([x, self.center[1], z], dtype=np.float32)


class RockingMotion:
    """Build-defined (BASELINE config 3): rotate the boundary cube about z by
    theta(t) = amplitude_deg * sin(rate * t).  The reference only rotates the cube by hand in
    the editor; SPH.cs:176-185 reacts to it, and this drives that code path headless."""

    def __init__(self, transform: Transform, time: Time, amplitude_deg=10.0, rate=0.5, axis=(0, 0, 1)):
        self.transform, self.time = transform, time
        self.amplitude_deg, self.rate, self.axis = float(amplitude_deg), float(rate), axis

    def Update(self):
        theta = self.amplitude_deg * math.sin(self.rate * self.time.time)
        # Quaternion.AngleAxis takes a FLOAT angle: rounded here like in the C++ and C# hosts (the three must hand the
        # kernels the same matrices: one differing bit parts two runs at their first OBB collision)
        self.transform.rotation = quat_from_axis_angle(self.axis, float(f32(theta)))


# --------------------------------------------------------------------------------------------
# Spawn RNG (ORACLE rule O9)
# --------------------------------------------------------------------------------------------
def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def spawn_uniforms(seed: int, first_index: int, count: int) -> np.ndarray:
    """count x 3 fp32 uniforms in [0,1): uniform k of particle i is the top 24 bits of
    splitmix64(seed * 2^32 + 3*i + k) scaled by 2^-24 (counter-based, order independent)."""
    idx = (np.arange(first_index, first_index + count, dtype=np.uint64)[:, None] * np.uint64(3)
           + np.arange(3, dtype=np.uint64)[None, :])
    with np.errstate(over="ignore"):
        ctr = np.uint64(seed) * np.uint64(1 << 32) + idx
    bits = _splitmix64(ctr) >> np.uint64(40)
    return (bits.astype(np.float32) * f32(1.0 / (1 << 24))).astype(np.float32)


# --------------------------------------------------------------------------------------------
# Assets/SPH.cs
# --------------------------------------------------------------------------------------------
class SPH:
    THREAD_GROUP_SIZE = 256                                # SPH.cs:94
    boundaryUpdateThreshold = 0.001                        # SPH.cs:100

    def __init__(self, engine, **fields):
        self.engine = engine
        # Inspector fields with the code defaults of SPH.cs:5-49
        self.soundSpeed = 20.0
        self.gamma = 7.0
        self.smoothingRadius = 0.2
        self.gravity = -9.81
        self.timeStep = 0.003
        self.subSteps = 2
        self.surfaceTensionCoefficient = 0.03
        self.surfaceTensionThreshold = 0.01
        self.xsphEpsilon = 0.5
        self.boundsCenter = _v3((0, 0, 0))
        self.boundsSize = _v3((10, 10, 10))
        self.boundaryCube: Optional[Transform] = None
        self.obstacleCollider: Optional[SphereCollider] = None
        self.obstacleRepulsionStiffness = 5000.0
        self.particleCollisionDamping = 0.9999
        self.maxParticlesPerCell = 100
        self.gridResolution = 50
        self.spawnBoxes: List[SpawnBox] = []
        self.seed = 12345                                  # build-defined (rule O9)
        for k, v in fields.items():
            if not hasattr(self, k):
                raise AttributeError(f"SPH has no field '{k}'")
            setattr(self, k, _v3(v) if k in ("boundsCenter", "boundsSize") else v)
        self.particleCount = 0
        self.totalCells = 0                                # SPH.cs:79
        self.gridResolutionX = self.gridResolutionY = self.gridResolutionZ = 0
        self.gridMin = _v3((0, 0, 0))
        self.cellSize = f32(0)
        self._obstaclePos = _v3((0, 0, 0))
        self._obstacleRadius = f32(0)
        self._deltaTime = f32(0)
        self.particleSpawnBoxIndices = None
        self.use_step_frame = False                        # True: one sph_step_frame call per Update

    # SPH.cs:105-110
    @property
    def ParticleCount(self):
        return self.particleCount

    def GetParticleBuffer(self):
        """The 72-byte AoS records in id order (what ParticleRender.shader:54-67 consumes)."""
        return self.engine.download_particles()

    # ---------------------------------------------------------------- Start, SPH.cs:112-171
    def Start(self):
        self.particleCount = sum(sb.particleCount for sb in self.spawnBoxes)      # SPH.cs:124-128
        grid_first = getattr(self.engine, "grid_first", False)
        if not grid_first:
            self.CreateParticles()                                                # SPH.cs:134
        if self.boundaryCube is None:                                             # SPH.cs:137-148
            half = self.boundsSize * f32(0.5)
            self.gridMin = (self.boundsCenter - half).astype(np.float32)
            self._size_grid(self.boundsSize)
            self.engine.resize_grid(self.gridResolutionX, self.gridResolutionY, self.gridResolutionZ,
                                    self.maxParticlesPerCell)                     # SPH.cs:156-157
        else:
            self.UpdateGridParametersFromBoundary()                               # SPH.cs:151
            self.lastBoundaryCubePosition = self.boundaryCube.position.copy()
            self.lastBoundaryCubeRotation = self.boundaryCube.rotation.copy()
        if grid_first:
            # slab engines filter the spawned rows by plane as they arrive, so they must know the
            # grid first; the order of these two host steps does not change any result
            self.engine.preview_grid(float(self.gridMin[0]), float(self.cellSize))
            self.CreateParticles()
        self.engine.reset_counter(abi.CTR_COLLISIONS)                             # SPH.cs:158
        self.SetComputeParams()                                                   # SPH.cs:162

    # ---------------------------------------------------------------- Update, SPH.cs:173-223
    def Update(self):
        e = self.engine
        if self.boundaryCube is not None:                                         # SPH.cs:176-185
            moved = float(np.linalg.norm((self.boundaryCube.position - self.lastBoundaryCubePosition)
                                         .astype(np.float32)))
            if (moved > self.boundaryUpdateThreshold or
                    quat_angle(self.boundaryCube.rotation, self.lastBoundaryCubeRotation) > self.boundaryUpdateThreshold):
                self.UpdateGridParametersFromBoundary()
                self.lastBoundaryCubePosition = self.boundaryCube.position.copy()
                self.lastBoundaryCubeRotation = self.boundaryCube.rotation.copy()
        self.UpdateObstacle()                                                     # SPH.cs:187
        e.reset_counter(abi.CTR_COLLISIONS)                                       # SPH.cs:190
        self.SetComputeParams()                                                   # SPH.cs:193
        self.UpdateParticleStaticProperties()                                     # SPH.cs:198
        if self.use_step_frame:
            e.step_frame(self.subSteps, self.timeStep)                            # SPH.cs:201-222 in one call
            return
        self.DispatchCompute(abi.K_CLEAR_GRID, self.totalCells)                   # SPH.cs:201
        self.DispatchCompute(abi.K_BUILD_GRID, self.particleCount)                # SPH.cs:204
        dtSub = f32(f32(self.timeStep) / f32(self.subSteps))                      # SPH.cs:207
        self._deltaTime = dtSub
        e.set_delta_time(dtSub)                                                   # SPH.cs:208
        for _ in range(self.subSteps):                                            # SPH.cs:210-219
            self.DispatchCompute(abi.K_VV_HALF_STEP, self.particleCount)
            self.DispatchCompute(abi.K_CLEAR, self.particleCount)
            self.DispatchCompute(abi.K_DENSITY_PRESSURE, self.particleCount)
            self.DispatchCompute(abi.K_FORCE_XSPH, self.particleCount)
            self.DispatchCompute(abi.K_VV_FULL_STEP, self.particleCount)
            self.DispatchCompute(abi.K_BOUND_OBS, self.particleCount)
        self.DispatchCompute(abi.K_VELOCITY_DAMPING, self.particleCount)          # SPH.cs:222

    def OnDestroy(self):                                                          # SPH.cs:235-241
        self.engine.destroy()

    # ---------------------------------------------------------------- SPH.cs:246-282
    def _spawn_rows(self, bi, sb, first_index, cnt, offset):
        """Rows offset .. offset+cnt-1 of spawn box bi (global ids first_index+offset ..)."""
        rows = np.zeros((cnt, abi.PARTICLE_FLOATS), dtype=np.float32)
        halfSpawn = (sb.spawnSize * f32(0.5)).astype(np.float32)
        u = spawn_uniforms(self.seed, first_index + offset, cnt)
        # Random.Range(-half, half) -> lerp(-half, +half, u), fp32 (rule O9)
        r = ((-halfSpawn)[None, :] + u * (halfSpawn - (-halfSpawn))[None, :]).astype(np.float32)
        rows[:, 0:3] = (sb.SpawnCenter[None, :] + r).astype(np.float32)
        rows[:, 3:6] = sb.initialVelocity
        rows[:, 9] = sb.restDensity
        rows[:, 11] = sb.restDensity
        rows[:, 12] = sb.viscosity
        rows[:, 13] = sb.particleMass
        rows[:, 14:18] = sb.particleColor
        return rows

    def CreateParticles(self):
        n = self.particleCount
        e = self.engine
        if getattr(e, "grid_first", False):
            # slab run: every rank walks the same global id sequence but only materialises the
            # boxes that can touch its planes, in bounded chunks (counter-based RNG, rule O9)
            e.alloc_particles(n)
            e.set_phase_table(self._phase_rows())      # records are checked against it when they are uploaded
            index = 0
            for bi, sb in enumerate(self.spawnBoxes):
                half_x = float(sb.spawnSize[0]) * 0.5
                if e.box_may_matter(float(sb.SpawnCenter[0]) - half_x, float(sb.SpawnCenter[0]) + half_x):
                    for off in range(0, sb.particleCount, 1 << 20):
                        cnt = min(1 << 20, sb.particleCount - off)
                        e.add_rows(index + off, self._spawn_rows(bi, sb, index, cnt, off), phase=bi)
                index += sb.particleCount
            return
        init = np.zeros((n, abi.PARTICLE_FLOATS), dtype=np.float32)
        self.particleSpawnBoxIndices = np.zeros(n, dtype=np.uint8)
        index = 0
        for bi, sb in enumerate(self.spawnBoxes):
            cnt = sb.particleCount
            init[index:index + cnt] = self._spawn_rows(bi, sb, index, cnt, 0)
            self.particleSpawnBoxIndices[index:index + cnt] = bi                  # SPH.cs:276
            index += cnt
        e.alloc_particles(n)                                                      # SPH.cs:249
        e.upload_particles(init)                                                  # SPH.cs:280
        e.set_phase_table(self._phase_rows())
        e.set_phase_ids(self.particleSpawnBoxIndices)

    def _phase_rows(self):
        return [(sb.restDensity, sb.viscosity, sb.particleMass, sb.particleColor) for sb in self.spawnBoxes]

    # SPH.cs:284-298: the reference reads the whole buffer back, rewrites four fields on the CPU
    # and uploads it again, every frame.  Here the spawn boxes' current values go down as a
    # tiny table; the engine re-applies it on the device only when something changed.
    def UpdateParticleStaticProperties(self):
        self.engine.set_phase_table(self._phase_rows())
        self.engine.apply_phase_table()

    # ---------------------------------------------------------------- SPH.cs:303-346
    def _size_grid(self, gridSize):
        gridSize = gridSize.astype(np.float32)
        minSize = f32(min(gridSize[0], gridSize[1], gridSize[2]))
        self.cellSize = f32(minSize / f32(self.gridResolution))
        self.gridResolutionX = int(math.ceil(float(f32(gridSize[0] / self.cellSize))))
        self.gridResolutionY = int(math.ceil(float(f32(gridSize[1] / self.cellSize))))
        self.gridResolutionZ = int(math.ceil(float(f32(gridSize[2] / self.cellSize))))
        self.totalCells = self.gridResolutionX * self.gridResolutionY * self.gridResolutionZ

    def UpdateGridParametersFromBoundary(self):
        if self.boundaryCube is None:
            return
        h = 0.5
        local = [(-h, -h, -h), (h, -h, -h), (h, -h, h), (-h, -h, h), (-h, h, -h), (h, h, -h), (h, h, h), (-h, h, h)]
        corners = np.stack([self.boundaryCube.TransformPoint(c) for c in local])  # SPH.cs:309-317
        newMin = corners.min(axis=0).astype(np.float32)
        newMax = corners.max(axis=0).astype(np.float32)
        self.gridMin = newMin                                                     # SPH.cs:327
        self._size_grid((newMax - newMin).astype(np.float32))                     # SPH.cs:326-336
        self.engine.resize_grid(self.gridResolutionX, self.gridResolutionY, self.gridResolutionZ,
                                self.maxParticlesPerCell)                         # SPH.cs:339-345

    # ---------------------------------------------------------------- SPH.cs:348-361
    def UpdateObstacle(self):
        if isinstance(self.obstacleCollider, SphereCollider):
            self._obstacleRadius = f32(f32(self.obstacleCollider.radius) * self.obstacleCollider.transform.lossyScale[0])
            self._obstaclePos = self.obstacleCollider.transform.position.copy()
        else:
            self._obstaclePos = _v3((0, 0, 0))
            self._obstacleRadius = f32(0)

    # ---------------------------------------------------------------- SPH.cs:379-412
    def build_params(self) -> abi.SphParams:
        p = abi.SphParams()
        p.particleCount = self.particleCount
        p.deltaTime = float(self._deltaTime)
        p.soundSpeed = self.soundSpeed
        p.gamma = self.gamma
        p.smoothingRadius = self.smoothingRadius
        p.gravity = self.gravity
        p.surfaceTensionCoefficient = self.surfaceTensionCoefficient
        p.surfaceTensionThreshold = self.surfaceTensionThreshold
        p.xsphEpsilon = self.xsphEpsilon
        p.particleCollisionDamping = self.particleCollisionDamping
        p.obstacleRepulsionStiffness = self.obstacleRepulsionStiffness
        p.obstacleRadius = float(self._obstacleRadius)
        p.set("obstaclePos", self._obstaclePos)
        p.gridResolutionX, p.gridResolutionY, p.gridResolutionZ = (self.gridResolutionX, self.gridResolutionY,
                                                                    self.gridResolutionZ)
        p.maxParticlesPerCell = self.maxParticlesPerCell
        p.cellSize = float(self.cellSize)
        p.set("minBound", self.gridMin)
        if self.boundaryCube is not None:
            p.set("boundaryMatrix", self.boundaryCube.localToWorldMatrix)
            p.set("boundaryInvMatrix", self.boundaryCube.worldToLocalMatrix)
            p.set("boundaryHalfExtents", (0.5, 0.5, 0.5))                         # SPH.cs:405
        else:
            p.set("boundsCenter", self.boundsCenter)                              # SPH.cs:409-410
            p.set("boundsSize", self.boundsSize)
        return p

    def SetComputeParams(self):
        self.engine.set_params(self.build_params())

    # ---------------------------------------------------------------- SPH.cs:439-443
    def DispatchCompute(self, kernel, count):
        groups = int(math.ceil(count / float(self.THREAD_GROUP_SIZE)))
        assert groups >= 1
        self.engine.dispatch(kernel, count)


class Scene:
    """A headless 'scene': the SPH component plus the movers that run before it each frame."""

    def __init__(self, sph: SPH, time: Optional[Time] = None, movers=()):
        self.sph, self.time, self.movers = sph, time or Time(), list(movers)
        self.started = False

    def start(self):
        for m in self.movers:
            m.Update()            # frame 0 at t = 0 (rule O8)
        self.sph.Start()
        self.started = True

    def frame(self):
        if not self.started:
            self.start()
        for m in self.movers:
            m.Update()
        self.sph.Update()
        self.time.advance()

    def run(self, frames):
        for _ in range(frames):
            self.frame()
