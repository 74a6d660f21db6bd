// SphSimulation.cs -- the host side of the SPH frame (what Assets/SPH.cs does around its kernels) on libsph_hip,
// free of UnityEngine types so that the Unity component (Unity/SPH_Hip.cs) and the headless harness
// (Headless/Program.cs) run the SAME code.  Counterpart of sph_amd/host.py (the host this repository tests).
//
// NOT COMPILED HERE (no .NET SDK in the build image).  Reference lines are cited as "SPH.cs:n".
//
// Per frame (SPH.cs:173-223): regrid if the boundary cube moved past the dead zone -> obstacle -> zero the
// collision counter -> parameters -> static properties -> ClearGrid, BuildGrid, subSteps x {HalfStep, Clear,
// DensityPressure, ForceXSPH, FullStep, BoundObs}, VelocityDamping.
using System;
using System.Collections.Generic;

namespace SphHip
{
    public struct Vec3
    {
        public float x, y, z;
        public Vec3(float x, float y, float z) { this.x = x; this.y = y; this.z = z; }
        public static Vec3 operator -(Vec3 a, Vec3 b) { return new Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
        public float Length() { return (float)Math.Sqrt(x * x + y * y + z * z); }
    }

    /// position / rotation (x, y, z, w) / scale of a root object: the sliver of UnityEngine.Transform SPH.cs reads.
    public sealed class Pose
    {
        public Vec3 position = new Vec3(0, 0, 0);
        public float qx = 0, qy = 0, qz = 0, qw = 1;
        public Vec3 scale = new Vec3(1, 1, 1);

        /// Matrix4x4.TRS as 16 floats, row-major, acting on column vectors (what the engine expects).
        public float[] LocalToWorld()
        {
            float x = qx, y = qy, z = qz, w = qw;
            float[] r = {
                1f - 2f * (y * y + z * z), 2f * (x * y - z * w),      2f * (x * z + y * w),
                2f * (x * y + z * w),      1f - 2f * (x * x + z * z), 2f * (y * z - x * w),
                2f * (x * z - y * w),      2f * (y * z + x * w),      1f - 2f * (x * x + y * y) };
            float[] s = { scale.x, scale.y, scale.z };
            float[] t = { position.x, position.y, position.z };
            var m = new float[16];
            for (int i = 0; i < 3; i++)
            {
                for (int j = 0; j < 3; j++) m[i * 4 + j] = r[i * 3 + j] * s[j];
                m[i * 4 + 3] = t[i];
            }
            m[15] = 1f;
            return m;
        }

        /// Inverse of LocalToWorld (affine), computed in double and rounded once.
        public float[] WorldToLocal()
        {
            float[] m = LocalToWorld();
            double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], k = m[10];
            double det = a * (e * k - f * h) - b * (d * k - f * g) + c * (d * h - e * g);
            double[] inv = {
                (e * k - f * h) / det, (c * h - b * k) / det, (b * f - c * e) / det,
                (f * g - d * k) / det, (a * k - c * g) / det, (c * d - a * f) / det,
                (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det };
            var o = new float[16];
            for (int i = 0; i < 3; i++)
            {
                for (int j = 0; j < 3; j++) o[i * 4 + j] = (float)inv[i * 3 + j];
                o[i * 4 + 3] = (float)(-(inv[i * 3] * m[3] + inv[i * 3 + 1] * m[7] + inv[i * 3 + 2] * m[11]));
            }
            o[15] = 1f;
            return o;
        }

        public Vec3 TransformPoint(Vec3 p)
        {
            float[] m = LocalToWorld();
            return new Vec3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3],
                            m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
                            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
        }

        /// Quaternion.Angle in degrees, with Unity's dead zone (0 when |dot| > 1 - 1e-6).
        public static float AngleBetween(Pose a, float bx, float by, float bz, float bw)
        {
            double d = Math.Abs(a.qx * bx + a.qy * by + a.qz * bz + a.qw * bw);
            if (d > 1.0) d = 1.0;
            if (d > 1.0 - 1e-6) return 0f;
            return (float)(Math.Acos(d) * 2.0 * 180.0 / Math.PI);
        }
    }

    /// Assets/SpawnBox.cs:7-25.
    public sealed class SpawnBoxDesc
    {
        public int particleCount = 1000;
        public Vec3 spawnSize = new Vec3(2, 2, 2);
        public float restDensity = 1000f, viscosity = 0.1f, particleMass = 0.03f;
        public Vec3 initialVelocity = new Vec3(0, 0, 0);
        public float r = 1, g = 1, b = 1, a = 1;
        public Vec3 center = new Vec3(0, 0, 0);
    }

    public sealed class SphSimulation : IDisposable
    {
        // Inspector fields of SPH.cs:5-49, same names and code defaults
        public float soundSpeed = 20f, gamma = 7f, smoothingRadius = 0.2f, gravity = -9.81f;
        public float timeStep = 0.003f;
        public int subSteps = 2;
        public float surfaceTensionCoefficient = 0.03f, surfaceTensionThreshold = 0.01f, xsphEpsilon = 0.5f;
        public Vec3 boundsCenter = new Vec3(0, 0, 0), boundsSize = new Vec3(10, 10, 10);
        public Pose boundaryCube;                     // null: axis-aligned fallback (SPH.cs:137-148)
        public Pose obstacle;                         // null: no obstacle
        public float obstacleColliderRadius = 0.5f;   // SphereCollider.radius (scaled by lossyScale.x, SPH.cs:352)
        public float obstacleRepulsionStiffness = 5000f, particleCollisionDamping = 0.9999f;
        public int maxParticlesPerCell = 100, gridResolution = 50;
        public List<SpawnBoxDesc> spawnBoxes = new List<SpawnBoxDesc>();
        public ulong seed = 12345;                    // spawn generator (see CreateParticles)
        public bool fusedFrame = true;                // one sph_step_frame call instead of 2 + 6*subSteps + 1 dispatches

        public int ParticleCount { get; private set; }
        public int totalCells;
        public readonly Engine engine;

        const float BoundaryUpdateThreshold = 0.001f; // SPH.cs:100
        int rx, ry, rz;
        Vec3 gridMin;
        float cellSize;
        Vec3 lastCubePosition;
        float lastQx, lastQy, lastQz, lastQw = 1f;
        byte[] phaseIds;

        public SphSimulation(int deviceId = 0) { engine = new Engine(deviceId); }
        public void Dispose() { engine.Dispose(); }                               // SPH.cs:235-241

        // ------------------------------------------------------------------ SPH.cs:112-171
        public void Start()
        {
            ParticleCount = 0;
            foreach (var sb in spawnBoxes) ParticleCount += sb.particleCount;     // SPH.cs:124-128
            CreateParticles();                                                    // SPH.cs:134
            if (boundaryCube == null)
            {
                gridMin = new Vec3(boundsCenter.x - boundsSize.x * 0.5f, boundsCenter.y - boundsSize.y * 0.5f,
                                   boundsCenter.z - boundsSize.z * 0.5f);
                SizeGrid(boundsSize);
                engine.ResizeGrid(rx, ry, rz, maxParticlesPerCell);               // SPH.cs:156-157
            }
            else
            {
                RegridFromBoundary();                                             // SPH.cs:151
                RememberCube();
            }
            engine.ResetCounter(SphCounter.Collisions);                           // SPH.cs:158
            PushParams(0f);                                                       // SPH.cs:162
        }

        // ------------------------------------------------------------------ SPH.cs:173-223
        public void Update()
        {
            if (boundaryCube != null)
            {
                float moved = (boundaryCube.position - lastCubePosition).Length();
                float turned = Pose.AngleBetween(boundaryCube, lastQx, lastQy, lastQz, lastQw);
                if (moved > BoundaryUpdateThreshold || turned > BoundaryUpdateThreshold)   // SPH.cs:176-185
                {
                    RegridFromBoundary();
                    RememberCube();
                }
            }
            engine.ResetCounter(SphCounter.Collisions);                           // SPH.cs:190
            float dtSub = timeStep / subSteps;                                    // SPH.cs:207
            PushParams(dtSub);                                                    // SPH.cs:187, 193, 208
            RefreshStaticProperties();                                            // SPH.cs:198
            if (fusedFrame) { engine.StepFrame(subSteps, timeStep); return; }     // SPH.cs:201-222 in one call
            engine.Dispatch(SphKernel.ClearGrid, totalCells);
            engine.Dispatch(SphKernel.BuildGrid, ParticleCount);
            engine.SetDeltaTime(dtSub);
            for (int s = 0; s < subSteps; s++)
            {
                engine.Dispatch(SphKernel.VvHalfStep, ParticleCount);
                engine.Dispatch(SphKernel.Clear, ParticleCount);
                engine.Dispatch(SphKernel.DensityPressure, ParticleCount);
                engine.Dispatch(SphKernel.ForceXSPH, ParticleCount);
                engine.Dispatch(SphKernel.VvFullStep, ParticleCount);
                engine.Dispatch(SphKernel.BoundObs, ParticleCount);
            }
            engine.Dispatch(SphKernel.VelocityDamping, ParticleCount);
        }

        // ------------------------------------------------------------------ SPH.cs:246-282
        // UnityEngine.Random cannot be reproduced outside Unity; positions come from a counter-based generator
        // (splitmix64 of seed * 2^32 + 3 * id + axis, top 24 bits), the same one sph_amd/host.py and the CPU
        // oracle use, so a headless .NET run starts from the state the Python host starts from, bit for bit.
        static ulong SplitMix64(ulong x)
        {
            unchecked
            {
                x += 0x9E3779B97F4A7C15UL;
                ulong z = x;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9UL;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBUL;
                return z ^ (z >> 31);
            }
        }

        static float Uniform(ulong seed, ulong id, ulong axis)
        {
            unchecked
            {
                ulong bits = SplitMix64(seed * (1UL << 32) + id * 3UL + axis) >> 40;
                return (float)bits * (1.0f / 16777216.0f);
            }
        }

        static float Spread(float half, float u) { float lo = -half; float w = half - lo; float t = u * w; return lo + t; }

        void CreateParticles()
        {
            var init = new SphParticle[ParticleCount];
            phaseIds = new byte[ParticleCount];
            var table = new SphPhase[spawnBoxes.Count];
            int id = 0;
            for (int b = 0; b < spawnBoxes.Count; b++)
            {
                SpawnBoxDesc sb = spawnBoxes[b];
                float hx = sb.spawnSize.x * 0.5f, hy = sb.spawnSize.y * 0.5f, hz = sb.spawnSize.z * 0.5f;
                for (int k = 0; k < sb.particleCount; k++, id++)
                {
                    SphParticle p = new SphParticle();
                    p.px = sb.center.x + Spread(hx, Uniform(seed, (ulong)id, 0));
                    p.py = sb.center.y + Spread(hy, Uniform(seed, (ulong)id, 1));
                    p.pz = sb.center.z + Spread(hz, Uniform(seed, (ulong)id, 2));
                    p.vx = sb.initialVelocity.x; p.vy = sb.initialVelocity.y; p.vz = sb.initialVelocity.z;
                    p.density = sb.restDensity; p.restDensity = sb.restDensity;
                    p.viscosity = sb.viscosity; p.mass = sb.particleMass;
                    p.r = sb.r; p.g = sb.g; p.b = sb.b; p.a = sb.a;
                    init[id] = p;
                    phaseIds[id] = (byte)b;                                       // SPH.cs:276
                }
            }
            engine.AllocParticles(ParticleCount);                                 // SPH.cs:249
            engine.UploadParticles(init);                                         // SPH.cs:280
            engine.SetPhaseTable(PhaseTable());
            engine.SetPhaseIds(phaseIds);
        }

        unsafe SphPhase[] PhaseTable()
        {
            var t = new SphPhase[spawnBoxes.Count];
            for (int b = 0; b < t.Length; b++)
            {
                SpawnBoxDesc sb = spawnBoxes[b];
                SphPhase ph = new SphPhase();
                ph.restDensity = sb.restDensity; ph.viscosity = sb.viscosity; ph.mass = sb.particleMass;
                ph.color[0] = sb.r; ph.color[1] = sb.g; ph.color[2] = sb.b; ph.color[3] = sb.a;
                t[b] = ph;
            }
            return t;
        }

        // SPH.cs:284-298 reads the whole buffer back and rewrites four fields every frame; here the boxes' current
        // values go down as a small table and the engine re-applies it on the device only when something changed.
        void RefreshStaticProperties()
        {
            engine.SetPhaseTable(PhaseTable());
            engine.ApplyPhaseTable();
        }

        // ------------------------------------------------------------------ SPH.cs:303-346
        void SizeGrid(Vec3 size)
        {
            float smallest = Math.Min(size.x, Math.Min(size.y, size.z));
            cellSize = smallest / gridResolution;                                 // SPH.cs:329
            rx = (int)Math.Ceiling((double)(float)(size.x / cellSize));
            ry = (int)Math.Ceiling((double)(float)(size.y / cellSize));
            rz = (int)Math.Ceiling((double)(float)(size.z / cellSize));
            totalCells = rx * ry * rz;
        }

        void RegridFromBoundary()
        {
            float lox = float.MaxValue, loy = float.MaxValue, loz = float.MaxValue;
            float hix = float.MinValue, hiy = float.MinValue, hiz = float.MinValue;
            for (int c = 0; c < 8; c++)                                           // SPH.cs:309-317
            {
                Vec3 w = boundaryCube.TransformPoint(new Vec3((c & 1) != 0 ? 0.5f : -0.5f, (c & 2) != 0 ? 0.5f : -0.5f,
                                                              (c & 4) != 0 ? 0.5f : -0.5f));
                lox = Math.Min(lox, w.x); loy = Math.Min(loy, w.y); loz = Math.Min(loz, w.z);
                hix = Math.Max(hix, w.x); hiy = Math.Max(hiy, w.y); hiz = Math.Max(hiz, w.z);
            }
            gridMin = new Vec3(lox, loy, loz);                                    // SPH.cs:327
            SizeGrid(new Vec3(hix - lox, hiy - loy, hiz - loz));
            engine.ResizeGrid(rx, ry, rz, maxParticlesPerCell);                   // SPH.cs:339-345 (grow-only inside)
        }

        void RememberCube()
        {
            lastCubePosition = boundaryCube.position;
            lastQx = boundaryCube.qx; lastQy = boundaryCube.qy; lastQz = boundaryCube.qz; lastQw = boundaryCube.qw;
        }

        // ------------------------------------------------------------------ SPH.cs:348-361, 379-412
        unsafe void PushParams(float deltaTime)
        {
            SphParams p = new SphParams();
            p.particleCount = ParticleCount;
            p.deltaTime = deltaTime;
            p.soundSpeed = soundSpeed; p.gamma = gamma; p.smoothingRadius = smoothingRadius; p.gravity = gravity;
            p.surfaceTensionCoefficient = surfaceTensionCoefficient; p.surfaceTensionThreshold = surfaceTensionThreshold;
            p.xsphEpsilon = xsphEpsilon;
            p.particleCollisionDamping = particleCollisionDamping;
            p.obstacleRepulsionStiffness = obstacleRepulsionStiffness;
            if (obstacle != null)                                                 // SPH.cs:350-354
            {
                p.obstacleRadius = obstacleColliderRadius * obstacle.scale.x;
                p.obstaclePos[0] = obstacle.position.x; p.obstaclePos[1] = obstacle.position.y; p.obstaclePos[2] = obstacle.position.z;
            }
            p.gridResolutionX = rx; p.gridResolutionY = ry; p.gridResolutionZ = rz;
            p.maxParticlesPerCell = maxParticlesPerCell;
            p.cellSize = cellSize;
            p.minBound[0] = gridMin.x; p.minBound[1] = gridMin.y; p.minBound[2] = gridMin.z;
            if (boundaryCube != null)
            {
                float[] m = boundaryCube.LocalToWorld(), inv = boundaryCube.WorldToLocal();
                for (int i = 0; i < 16; i++) { p.boundaryMatrix[i] = m[i]; p.boundaryInvMatrix[i] = inv[i]; }
                p.boundaryHalfExtents[0] = 0.5f; p.boundaryHalfExtents[1] = 0.5f; p.boundaryHalfExtents[2] = 0.5f;   // SPH.cs:405
            }
            else
            {
                p.boundsCenter[0] = boundsCenter.x; p.boundsCenter[1] = boundsCenter.y; p.boundsCenter[2] = boundsCenter.z;
                p.boundsSize[0] = boundsSize.x; p.boundsSize[1] = boundsSize.y; p.boundsSize[2] = boundsSize.z;
            }
            engine.SetParams(ref p);
        }

        /// The 72-byte records in id order (what ParticleRender.shader:54-67 reads).  Synchronises.
        public void ReadParticles(SphParticle[] into) { engine.DownloadParticles(into); }
    }
}
