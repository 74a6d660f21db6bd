// SPH_Hip.cs -- Unity component that stands where the reference's `SPH` component stands, with the fluid step
// running in libsph_hip instead of SPHCompute.compute.  Put SphNative.cs, SphSimulation.cs and this file under
// Assets/, libsph_hip.so under Assets/Plugins/x86_64/, and swap the component on the "SPH" object: the serialized
// field names are those of SPH.cs:5-49, so Inspector values typed into a scene carry over by name.
//
// NOT COMPILED HERE (no Unity in the build image).  The simulation logic lives in SphSimulation.cs, which the
// headless harness runs too; this file only adapts UnityEngine types and feeds the renderer.
//
// Rendering (SPH.cs:164-170, 225-233): the reference's material reads the simulation's own ComputeBuffer.  Unity
// cannot bind a HIP allocation, so the component keeps a 72-byte-stride ComputeBuffer for the material and
// refreshes it from sph_download_particles once per rendered frame (or every `renderEvery` frames).  A native
// rendering plugin that shares the buffer (Vulkan external memory) would call sph_export_particles_device instead.
using UnityEngine;
using SphHip;

public class SPH_Hip : MonoBehaviour
{
    [Header("Particle Settings")]
    public float soundSpeed = 20f;
    public float gamma = 7f;
    public float smoothingRadius = 0.2f;
    public float gravity = -9.81f;
    [Header("Time Settings")]
    public float timeStep = 0.003f;
    public int subSteps = 2;
    [Header("Surface Tension")]
    public float surfaceTensionCoefficient = 0.03f;
    public float surfaceTensionThreshold = 0.01f;
    [Header("XSPH Settings")]
    public float xsphEpsilon = 0.5f;
    [Header("Axis-Aligned Boundary (Fallback)")]
    public Vector3 boundsCenter = Vector3.zero;
    public Vector3 boundsSize = new Vector3(10f, 10f, 10f);
    [Header("Boundary Cube")]
    public Transform boundaryCube;
    [Header("Obstacle Settings")]
    public Collider obstacleCollider;
    public float obstacleRepulsionStiffness = 5000f;
    public float particleCollisionDamping = 0.9999f;
    [Header("Rendering Settings")]
    public Material fluidMaterial;
    [Header("Grid Settings")]
    public int maxParticlesPerCell = 100;
    [Header("Grid Resolution")]
    public int gridResolution = 50;
    [Header("Spawn Boxes")]
    public SpawnBox[] spawnBoxes;
    [Header("libsph_hip")]
    public int gpuIndex = 0;
    public int renderEvery = 1;
    public bool fusedFrame = true;

    SphSimulation sim;
    ComputeBuffer renderBuffer;
    SphParticle[] staging;
    Pose cubePose, obstaclePose;
    int frame;

    public int ParticleCount { get { return sim != null ? sim.ParticleCount : 0; } }
    public ComputeBuffer GetParticleBuffer() { return renderBuffer; }
    public int CollisionCount { get { return sim.engine.ReadCounter(SphCounter.Collisions); } }

    static Vec3 V(Vector3 v) { return new Vec3(v.x, v.y, v.z); }

    static void CopyPose(Transform t, Pose p)
    {
        p.position = V(t.position);
        Quaternion q = t.rotation;
        p.qx = q.x; p.qy = q.y; p.qz = q.z; p.qw = q.w;
        p.scale = V(t.lossyScale);
    }

    void PushInspector()
    {
        sim.soundSpeed = soundSpeed; sim.gamma = gamma; sim.smoothingRadius = smoothingRadius; sim.gravity = gravity;
        sim.timeStep = timeStep; sim.subSteps = subSteps;
        sim.surfaceTensionCoefficient = surfaceTensionCoefficient; sim.surfaceTensionThreshold = surfaceTensionThreshold;
        sim.xsphEpsilon = xsphEpsilon;
        sim.boundsCenter = V(boundsCenter); sim.boundsSize = V(boundsSize);
        sim.obstacleRepulsionStiffness = obstacleRepulsionStiffness; sim.particleCollisionDamping = particleCollisionDamping;
        sim.maxParticlesPerCell = maxParticlesPerCell; sim.gridResolution = gridResolution;
        sim.fusedFrame = fusedFrame;
        if (boundaryCube != null) { CopyPose(boundaryCube, cubePose); sim.boundaryCube = cubePose; } else sim.boundaryCube = null;
        SphereCollider sc = obstacleCollider as SphereCollider;       // SPH.cs:350: only spheres are obstacles
        if (sc != null) { CopyPose(sc.transform, obstaclePose); sim.obstacle = obstaclePose; sim.obstacleColliderRadius = sc.radius; }
        else sim.obstacle = null;
        // live edits of a spawn box (rest density, viscosity, mass, colour) reach the particles like SPH.cs:284-298
        for (int b = 0; b < spawnBoxes.Length; b++)
        {
            SpawnBox src = spawnBoxes[b];
            SpawnBoxDesc d = sim.spawnBoxes[b];
            d.restDensity = src.restDensity; d.viscosity = src.viscosity; d.particleMass = src.particleMass;
            d.r = src.particleColor.r; d.g = src.particleColor.g; d.b = src.particleColor.b; d.a = src.particleColor.a;
        }
    }

    void Start()
    {
        sim = new SphSimulation(gpuIndex);
        cubePose = new Pose(); obstaclePose = new Pose();
        foreach (SpawnBox src in spawnBoxes)
        {
            var d = new SpawnBoxDesc();
            d.particleCount = src.particleCount;
            d.spawnSize = V(src.spawnSize);
            d.initialVelocity = V(src.initialVelocity);
            d.center = V(src.SpawnCenter);
            sim.spawnBoxes.Add(d);
        }
        PushInspector();
        sim.Start();
        staging = new SphParticle[sim.ParticleCount];
        renderBuffer = new ComputeBuffer(sim.ParticleCount, 72);
        if (fluidMaterial != null) fluidMaterial.SetBuffer("_ParticleBuffer", renderBuffer);
        RefreshRenderBuffer();
    }

    void Update()
    {
        PushInspector();
        sim.Update();
        frame++;
        if (renderEvery > 0 && frame % renderEvery == 0) RefreshRenderBuffer();
    }

    void RefreshRenderBuffer()
    {
        sim.ReadParticles(staging);
        renderBuffer.SetData(staging);
    }

    void OnRenderObject()
    {
        if (fluidMaterial == null || renderBuffer == null) return;
        fluidMaterial.SetPass(0);
        Graphics.DrawProceduralNow(MeshTopology.Points, sim.ParticleCount);
    }

    void OnDestroy()
    {
        if (renderBuffer != null) renderBuffer.Release();
        if (sim != null) sim.Dispose();
    }
}
