// SphNative.cs -- P/Invoke binding of libsph_hip (include/sph_hip.h), one extern per C entry point.
//
// NOT COMPILED IN THIS REPOSITORY'S BUILD IMAGE (no .NET SDK, no Unity): the same entry points, struct layouts
// and call order are exercised through ctypes by sph_amd/abi.py and through C++ by host/.  Layouts below are
// checked against the header by tests/test_abi.py::test_csharp_binding_matches_header (field order, counts,
// exported names).  Works under Unity (Mono/IL2CPP, "Allow unsafe code" on) and plain .NET 6+.
//
// Replaces, in Assets/SPH.cs of the reference: ComputeShader.FindKernel/SetInt/SetFloat/SetVector/SetMatrix/
// SetBuffer/Dispatch (SPH.cs:366-443) and ComputeBuffer ctor/SetData/GetData/Release (SPH.cs:156-158, 190,
// 237-240, 249, 280, 287, 297, 339-345).
using System;
using System.Runtime.InteropServices;

namespace SphHip
{
    /// struct sph_params == cbuffer SimulationParams (SPHCompute.compute:47-81), declaration order.
    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct SphParams
    {
        public uint structSize;                 // = sizeof(SphParams)
        public int particleCount;
        public float deltaTime;
        public float particleMass;              // declared by the reference, read by no kernel
        public float restDensity;               // declared by the reference, read by no kernel
        public float soundSpeed;
        public float gamma;
        public float smoothingRadius;
        public float gravity;
        public float surfaceTensionCoefficient;
        public float surfaceTensionThreshold;
        public float xsphEpsilon;
        public float particleCollisionDamping;
        public float obstacleRepulsionStiffness;
        public float obstacleRadius;
        public fixed float boundsCenter[3];
        public fixed float boundsSize[3];
        public fixed float spawnCenter[3];
        public fixed float spawnSize[3];
        public fixed float obstaclePos[3];
        public int gridResolutionX;
        public int gridResolutionY;
        public int gridResolutionZ;
        public int maxParticlesPerCell;
        public float cellSize;
        public fixed float minBound[3];
        public fixed float boundaryMatrix[16];      // row-major M[r*4+c], acts on column vectors: mul(M, v)
        public fixed float boundaryInvMatrix[16];
        public fixed float boundaryHalfExtents[3];
    }

    /// struct sph_phase: one spawn box's static particle properties (SpawnBox.cs:10-23).
    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct SphPhase
    {
        public float restDensity;
        public float viscosity;
        public float mass;
        public fixed float color[4];
    }

    /// struct sph_timings (milliseconds; see the header for the averaging rule).
    [StructLayout(LayoutKind.Sequential)]
    public struct SphTimings
    {
        public float buildGridMs, halfStepMs, densityMs, forceMs, fullStepMs, boundObsMs, dampingMs, frameMs;
        public int frames;
    }

    /// The reference's particle record (SPH.cs:54-65, compute:26-37): 18 floats, 72 bytes, blittable.
    [StructLayout(LayoutKind.Sequential)]
    public struct SphParticle
    {
        public float px, py, pz;
        public float vx, vy, vz;
        public float ax, ay, az;
        public float density, pressure, restDensity, viscosity, mass;
        public float r, g, b, a;
    }

    /// enum sph_kernel: "#pragma kernel" order of SPHCompute.compute:9-17 (what FindKernel returns, SPH.cs:366-377).
    public enum SphKernel
    {
        Clear = 0, ClearGrid = 1, BuildGrid = 2, DensityPressure = 3, ForceXSPH = 4,
        VvHalfStep = 5, VvFullStep = 6, BoundObs = 7, VelocityDamping = 8
    }

    /// enum sph_counter.
    public enum SphCounter { Collisions = 0, BinOverflow = 1, OutOfGrid = 2, MaxPlaneShift = 3, BandMisses = 4, NonFinite = 5 }

    public static class Native
    {
        const string Lib = "sph_hip";           // libsph_hip.so next to the player / on LD_LIBRARY_PATH

        [DllImport(Lib)] public static extern int sph_create(out IntPtr ctx, int deviceId);
        [DllImport(Lib)] public static extern int sph_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern IntPtr sph_last_error(IntPtr ctx);
        [DllImport(Lib)] public static extern IntPtr sph_version();
        [DllImport(Lib)] public static extern int sph_alloc_particles(IntPtr ctx, int n);
        [DllImport(Lib)] public static extern int sph_upload_particles(IntPtr ctx, [In] SphParticle[] aos72, int n);
        [DllImport(Lib)] public static extern int sph_upload_particles_in_place(IntPtr ctx, [In] SphParticle[] aos72, int n);
        [DllImport(Lib)] public static extern int sph_download_particles(IntPtr ctx, [Out] SphParticle[] aos72, int n);
        [DllImport(Lib)] public static extern int sph_export_particles_device(IntPtr ctx, IntPtr devAos72, int n);
        [DllImport(Lib)] public static extern int sph_resize_grid(IntPtr ctx, int rx, int ry, int rz, int maxPerCell);
        [DllImport(Lib)] public static extern int sph_set_params(IntPtr ctx, ref SphParams p);
        [DllImport(Lib)] public static extern int sph_set_delta_time(IntPtr ctx, float dt);
        [DllImport(Lib)] public static extern int sph_set_phase_table(IntPtr ctx, [In] SphPhase[] table, int nPhases);
        [DllImport(Lib)] public static extern int sph_set_phase_ids(IntPtr ctx, [In] byte[] ids, int n);
        [DllImport(Lib)] public static extern int sph_apply_phase_table(IntPtr ctx);
        [DllImport(Lib)] public static extern int sph_dispatch(IntPtr ctx, int kernelId, int count);
        [DllImport(Lib)] public static extern int sph_step_frame(IntPtr ctx, int subSteps, float timeStep);
        [DllImport(Lib)] public static extern int sph_synchronize(IntPtr ctx);
        [DllImport(Lib)] public static extern int sph_reset_counter(IntPtr ctx, int which);
        [DllImport(Lib)] public static extern int sph_read_counter(IntPtr ctx, int which, out int value);
        [DllImport(Lib)] public static extern int sph_download_grid(IntPtr ctx, [Out] int[] cellStart, [Out] int[] cellEnd,
                                                                    [Out] int[] sortedIds, [Out] uint[] cellOfParticle);
        [DllImport(Lib)] public static extern int sph_download_neighbors(IntPtr ctx, [Out] int[] counts, [Out] int[] lists, int maxNeighbors);
        [DllImport(Lib)] public static extern int sph_download_hit_lists(IntPtr ctx, [Out] int[] counts, [Out] int[] lists, int maxNeighbors, out int maskState);
        [DllImport(Lib)] public static extern int sph_get_timings(IntPtr ctx, out SphTimings t);
        [DllImport(Lib)] public static extern int sph_particle_count(IntPtr ctx, out int n);
    }

    /// Thrown for any negative return code; Message carries sph_last_error.
    public sealed class SphException : Exception
    {
        public readonly int Code;
        public SphException(int code, string message) : base("libsph_hip error " + code + ": " + message) { Code = code; }
    }

    /// One GPU's engine (an sph_ctx).  Not thread-safe: one owner thread at a time, like the header says.
    public sealed class Engine : IDisposable
    {
        IntPtr ctx;
        public int ParticleCount { get; private set; }
        public int TotalCells { get; private set; }

        public Engine(int deviceId = 0)
        {
            int rc = Native.sph_create(out ctx, deviceId);
            if (rc != 0) { ctx = IntPtr.Zero; throw new SphException(rc, "sph_create failed (no gfx950 device visible?)"); }
        }

        void Check(int rc)
        {
            if (rc == 0) return;
            string msg = Marshal.PtrToStringAnsi(Native.sph_last_error(ctx)) ?? "";
            throw new SphException(rc, msg);
        }

        public static string Version() { return Marshal.PtrToStringAnsi(Native.sph_version()) ?? ""; }

        public void AllocParticles(int n) { Check(Native.sph_alloc_particles(ctx, n)); ParticleCount = n; }
        public void UploadParticles(SphParticle[] p) { Check(Native.sph_upload_particles(ctx, p, p.Length)); }
        public void DownloadParticles(SphParticle[] p) { Check(Native.sph_download_particles(ctx, p, p.Length)); }
        public void ExportParticlesDevice(IntPtr deviceBuffer) { Check(Native.sph_export_particles_device(ctx, deviceBuffer, ParticleCount)); }
        public void ResizeGrid(int rx, int ry, int rz, int cap) { Check(Native.sph_resize_grid(ctx, rx, ry, rz, cap)); TotalCells = rx * ry * rz; }
        public unsafe void SetParams(ref SphParams p) { p.structSize = (uint)sizeof(SphParams); Check(Native.sph_set_params(ctx, ref p)); }
        public void SetDeltaTime(float dt) { Check(Native.sph_set_delta_time(ctx, dt)); }
        public void SetPhaseTable(SphPhase[] t) { Check(Native.sph_set_phase_table(ctx, t, t.Length)); }
        public void SetPhaseIds(byte[] ids) { Check(Native.sph_set_phase_ids(ctx, ids, ids.Length)); }
        public void ApplyPhaseTable() { Check(Native.sph_apply_phase_table(ctx)); }
        public void Dispatch(SphKernel k, int count) { Check(Native.sph_dispatch(ctx, (int)k, count)); }
        public void StepFrame(int subSteps, float timeStep) { Check(Native.sph_step_frame(ctx, subSteps, timeStep)); }
        public void Synchronize() { Check(Native.sph_synchronize(ctx)); }
        public void ResetCounter(SphCounter c) { Check(Native.sph_reset_counter(ctx, (int)c)); }
        public int ReadCounter(SphCounter c) { int v; Check(Native.sph_read_counter(ctx, (int)c, out v)); return v; }
        public SphTimings GetTimings() { SphTimings t; Check(Native.sph_get_timings(ctx, out t)); return t; }

        public void Dispose()
        {
            if (ctx != IntPtr.Zero) { Native.sph_destroy(ctx); ctx = IntPtr.Zero; }
            GC.SuppressFinalize(this);
        }
        ~Engine() { if (ctx != IntPtr.Zero) Native.sph_destroy(ctx); }
    }
}
