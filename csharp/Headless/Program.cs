// Headless .NET harness: the reference's four scenes (values of Assets/Scenes/Scene1-4.unity as tabulated in
// SURVEY.md 5.1 and sph_amd/scenes.py) on libsph_hip without Unity.  Same SphSimulation the Unity component runs.
//
//   dotnet run -- --scene 4 --frames 600 [--gpu 0] [--dump out.bin]
//
// NOT COMPILED HERE (no .NET SDK in the build image); host/sph_headless.cpp is the harness this repository builds
// and tests, with the same options and the same scenes.  Frame period 1/60 s for the movers (frame 0 at t = 0).
using System;
using System.Diagnostics;
using System.IO;
using System.Runtime.InteropServices;
using SphHip;

static class Program
{
    static SpawnBoxDesc Box(int count, Vec3 size, float rho0, float visc, Vec3 center, bool oil)
    {
        var b = new SpawnBoxDesc();
        b.particleCount = count; b.spawnSize = size; b.restDensity = rho0; b.viscosity = visc; b.particleMass = 0.2f;
        b.center = center;
        if (oil) { b.r = 1.0f; b.g = 0.8f; b.b = 0.2f; b.a = 0.8f; } else { b.r = 0.2f; b.g = 0.5f; b.b = 1.0f; b.a = 0.8f; }
        return b;
    }

    static SphSimulation BuildScene(int which, int gpu, out float orbitSpeed)
    {
        var s = new SphSimulation(gpu);
        s.soundSpeed = 10f; s.gamma = 4f; s.smoothingRadius = 0.3f; s.gravity = -9.81f; s.timeStep = 0.02f; s.subSteps = 1;
        s.surfaceTensionThreshold = 0.01f; s.obstacleRepulsionStiffness = 100f; s.particleCollisionDamping = 1f;
        s.maxParticlesPerCell = 500; s.gridResolution = 80;
        s.boundaryCube = new Pose(); s.boundaryCube.position = new Vec3(0, 15, 0); s.boundaryCube.scale = new Vec3(20, 30, 20);
        s.obstacle = new Pose(); s.obstacle.position = new Vec3(-6.08f, 4.0f, 41.53f); s.obstacle.scale = new Vec3(8, 8, 8);
        s.obstacleColliderRadius = 0.5f;
        bool strongTension = which <= 2;
        s.surfaceTensionCoefficient = strongTension ? 0.1f : 0.01f;
        s.xsphEpsilon = strongTension ? 0.3f : 0.2f;
        orbitSpeed = 0f;
        switch (which)
        {
            case 1: s.spawnBoxes.Add(Box(200000, new Vec3(5, 10, 5), 20f, 0.1f, new Vec3(0, 20, 0), false)); break;
            case 2:
                s.spawnBoxes.Add(Box(150000, new Vec3(5, 10, 5), 20f, 0.01f, new Vec3(0, 20, 5), false));
                s.spawnBoxes.Add(Box(75000, new Vec3(5, 10, 5), 10f, 0.1f, new Vec3(0, 20, -5), true)); break;
            case 3: s.spawnBoxes.Add(Box(200000, new Vec3(10, 10, 10), 20f, 0.1f, new Vec3(0, 15, 0), false)); orbitSpeed = 2f; break;
            case 4:
                s.spawnBoxes.Add(Box(150000, new Vec3(5, 10, 5), 20f, 0.1f, new Vec3(0, 15, 5), false));
                s.spawnBoxes.Add(Box(75000, new Vec3(5, 10, 5), 10f, 0.1f, new Vec3(0, 15, -5), true)); orbitSpeed = 1f; break;
            default: throw new ArgumentException("--scene 1..4");
        }
        return s;
    }

    // Assets/SphereMover.cs:12-21 (class CircularMotion): centre (0,0,0), radius 5
    static void Orbit(Pose p, int frame, float speed)
    {
        float t = (float)frame * (1.0f / 60.0f);
        float angle = t * speed;
        p.position = new Vec3(5f * (float)Math.Cos(angle), 0f, 5f * (float)Math.Sin(angle));
    }

    static int Main(string[] args)
    {
        int scene = 4, frames = 100, gpu = 0;
        string dump = null;
        for (int i = 0; i < args.Length; i++)
        {
            if (args[i] == "--scene") scene = int.Parse(args[++i]);
            else if (args[i] == "--frames") frames = int.Parse(args[++i]);
            else if (args[i] == "--gpu") gpu = int.Parse(args[++i]);
            else if (args[i] == "--dump") dump = args[++i];
            else { Console.Error.WriteLine("unknown option " + args[i]); return 2; }
        }
        try
        {
            float orbitSpeed;
            using (SphSimulation sim = BuildScene(scene, gpu, out orbitSpeed))
            {
                if (orbitSpeed != 0f) Orbit(sim.obstacle, 0, orbitSpeed);
                sim.Start();
                var clock = Stopwatch.StartNew();
                for (int f = 0; f < frames; f++)
                {
                    if (orbitSpeed != 0f) Orbit(sim.obstacle, f, orbitSpeed);
                    sim.Update();
                }
                sim.engine.Synchronize();
                double seconds = clock.Elapsed.TotalSeconds;
                Console.WriteLine("{0}: scene {1}, {2} particles, {3} frames, {4:F3} ms/frame, {5:E3} particle-updates/s, bin overflow {6}",
                                  Engine.Version(), scene, sim.ParticleCount, frames, seconds / frames * 1e3,
                                  (double)sim.ParticleCount * sim.subSteps * frames / seconds,
                                  sim.engine.ReadCounter(SphCounter.BinOverflow));
                if (dump != null)
                {
                    var p = new SphParticle[sim.ParticleCount];
                    sim.ReadParticles(p);
                    byte[] raw = new byte[p.Length * 72];
                    GCHandle h = GCHandle.Alloc(p, GCHandleType.Pinned);
                    try { Marshal.Copy(h.AddrOfPinnedObject(), raw, 0, raw.Length); } finally { h.Free(); }
                    File.WriteAllBytes(dump, raw);
                }
            }
            return 0;
        }
        catch (SphException e) { Console.Error.WriteLine(e.Message); return 1; }
    }
}
