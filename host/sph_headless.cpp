// sph_headless -- runs the reference's scenes / the BASELINE configurations without Unity, through
// libsph_hip's C ABI.  Usage:
//   sph_headless --scene 1..4 | --config 1..3  [--frames F] [--scale s] [--step-frame] [--dump out.bin]
// --scale shrinks particle counts and spawn boxes together (spawn density kept), like
// sph_amd.scenes.unity_scene(scale_count=...).  --dump writes the final 72-byte records in id order.
#include <chrono>
#include <cstdlib>
#include <memory>

#include "sph_host.hpp"

using namespace sphhost;

struct SceneObjects {
    Time time;
    Transform cube;
    SphereCollider obstacle;
    std::vector<std::unique_ptr<SpawnBox>> boxes;
    std::vector<CircularMotion> circ;
    std::vector<RockingMotion> rock;
};

static SpawnBox* addBox(SceneObjects& so, int count, Vector3 size, float rho0, float visc, float mass, Color col, Vector3 pos) {
    auto sb = std::make_unique<SpawnBox>();
    sb->particleCount = count; sb->spawnSize = size; sb->restDensity = rho0; sb->viscosity = visc;
    sb->particleMass = mass; sb->particleColor = col; sb->transform.position = pos;
    so.boxes.push_back(std::move(sb));
    return so.boxes.back().get();
}

static void common(SPH& s) {   // Inspector values shared by Scene1-4 (Scene1.unity:604-622)
    s.soundSpeed = 10.0f; s.gamma = 4.0f; s.smoothingRadius = 0.3f; s.gravity = -9.81f; s.timeStep = 0.02f;
    s.subSteps = 1; s.surfaceTensionThreshold = 0.01f; s.obstacleRepulsionStiffness = 100.0f;
    s.particleCollisionDamping = 1.0f; s.maxParticlesPerCell = 500; s.gridResolution = 80;
    s.boundsCenter = {0, 5, 0}; s.boundsSize = {10, 10, 10};
}

int main(int argc, char** argv) {
    int scene = 0, config = 0, frames = 100;
    double scale = 1.0;
    bool stepFrame = false;
    const char* dump = nullptr;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--scene") scene = atoi(next());
        else if (a == "--config") config = atoi(next());
        else if (a == "--frames") frames = atoi(next());
        else if (a == "--scale") scale = atof(next());
        else if (a == "--step-frame") stepFrame = true;
        else if (a == "--dump") dump = next();
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (!scene && !config) scene = 2;
    try {
        SceneObjects so;
        SPH sph(0);
        common(sph);
        sph.useStepFrame = stepFrame;
        const Color WATER{0.2f, 0.5f, 1.0f, 0.8f}, OIL{1.0f, 0.8f, 0.2f, 0.8f};      // the BASELINE configurations
        // SpawnBox.particleColor as serialized in the scenes (Scene1.unity:543, Scene2.unity:499/594, Scene3.unity:543,
        // Scene4.unity:499/594)
        const float blueG[5] = {0.5f, 0.18861637f, 0.30632228f, 0.21572512f, 0.23613732f};
        const Color BLUE{0.0039215684f, blueG[scene >= 1 && scene <= 4 ? scene : 0], 1.0f, 1.0f}, YELLOW{1.0f, 0.94675124f, 0.0f, 1.0f};
        const float ls = (float)std::pow(scale, 1.0 / 3.0);
        auto cnt = [&](int c) { int v = (int)std::lround(c * scale); return v < 1 ? 1 : v; };
        auto sz = [&](double x, double y, double z) { return Vector3{(float)(x * (double)std::pow(scale, 1.0 / 3.0)), (float)(y * (double)std::pow(scale, 1.0 / 3.0)), (float)(z * (double)std::pow(scale, 1.0 / 3.0))}; };
        (void)ls;
        if (scene) {
            so.cube.position = {0, 15, 0}; so.cube.localScale = {20, 30, 20};          // Scene1.unity:933-935
            so.obstacle.radius = 0.5f; so.obstacle.transform.position = {-6.08f, 4.0f, 41.53f};
            so.obstacle.transform.localScale = {8, 8, 8};                               // Scene1.unity:676, 740-741
            sph.boundaryCube = &so.cube; sph.obstacleCollider = &so.obstacle;
            if (scene <= 2) { sph.surfaceTensionCoefficient = 0.1f; sph.xsphEpsilon = 0.3f; }
            else { sph.surfaceTensionCoefficient = 0.01f; sph.xsphEpsilon = 0.2f; }
            if (scene == 1) addBox(so, cnt(200000), sz(5, 10, 5), 20, 0.1f, 0.2f, BLUE, {0, 20, 0});
            else if (scene == 2) { addBox(so, cnt(150000), sz(5, 10, 5), 20, 0.01f, 0.2f, BLUE, {0, 20, 5});
                                   addBox(so, cnt(75000), sz(5, 10, 5), 10, 0.1f, 0.2f, YELLOW, {0, 20, -5}); }
            else if (scene == 3) addBox(so, cnt(200000), sz(10, 10, 10), 20, 0.1f, 0.2f, BLUE, {0, 15, 0});
            else if (scene == 4) { addBox(so, cnt(150000), sz(5, 10, 5), 20, 0.1f, 0.2f, BLUE, {0, 15, 5});
                                   addBox(so, cnt(75000), sz(5, 10, 5), 10, 0.1f, 0.2f, YELLOW, {0, 15, -5}); }
            else { fprintf(stderr, "scene 1..4\n"); return 2; }
            if (scene >= 3) {
                CircularMotion m; m.transform = &so.obstacle.transform; m.time = &so.time; m.center = {0, 0, 0};
                m.radius = 5.0f; m.speed = scene == 3 ? 2.0f : 1.0f;                    // Scene3.unity:759-761 / Scene4 :811-813
                so.circ.push_back(m);
            }
        } else if (config == 1) {
            sph.gridResolution = 40; sph.surfaceTensionCoefficient = 0.1f; sph.xsphEpsilon = 0.3f;
            addBox(so, cnt(16384), {4, 8.2f, 5}, 20, 0.1f, 0.2f, WATER, {-3, 4.1f, 0});
        } else if (config == 2 || config == 3) {
            const int n = cnt(config == 2 ? 262144 : 1048576), b = n / 3, a = n - b;
            const float s = config == 2 ? 1.0f : 1.6f;
            so.cube.position = {0, 15 * s, 0}; so.cube.localScale = {20 * s, 30 * s, 20 * s};
            sph.boundaryCube = &so.cube; sph.obstacleCollider = &so.obstacle;
            sph.gridResolution = (int)std::lround(80 * s);
            so.obstacle.radius = 0.5f;
            if (config == 2) {
                sph.surfaceTensionCoefficient = 0.1f; sph.xsphEpsilon = 0.3f;
                so.obstacle.transform.position = {-6.08f, 4.0f, 41.53f}; so.obstacle.transform.localScale = {8, 8, 8};
                addBox(so, a, {5, 10, 5}, 20, 0.1f, 0.2f, WATER, {0, 20, 5});
                addBox(so, b, {5, 10, 5}, 10, 0.1f, 0.2f, OIL, {0, 20, -5});
            } else {
                sph.surfaceTensionCoefficient = 0.01f; sph.xsphEpsilon = 0.2f;
                so.obstacle.transform.position = {5 * s, 0, 0}; so.obstacle.transform.localScale = {8 * s, 8 * s, 8 * s};
                addBox(so, a, {5 * s, 10 * s, 5 * s}, 20, 0.1f, 0.2f, WATER, {0, 15 * s, 5 * s});
                addBox(so, b, {5 * s, 10 * s, 5 * s}, 10, 0.1f, 0.2f, OIL, {0, 15 * s, -5 * s});
                CircularMotion m; m.transform = &so.obstacle.transform; m.time = &so.time; m.center = {0, 0, 0};
                m.radius = 5.0f * s; m.speed = 1.0f;
                so.circ.push_back(m);
                RockingMotion r; r.transform = &so.cube; r.time = &so.time;
                so.rock.push_back(r);
            }
        } else { fprintf(stderr, "config 1..3\n"); return 2; }
        for (auto& b : so.boxes) sph.spawnBoxes.push_back(b.get());

        auto movers = [&]() { for (auto& m : so.circ) m.Update(); for (auto& m : so.rock) m.Update(); };
        movers();                       // frame 0 at t = 0 (rule O8)
        sph.Start();
        const auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; f++) {
            movers();
            sph.Update();
            so.time.advance();
        }
        sph_synchronize(sph.Context());
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const std::vector<float> P = sph.GetParticleBuffer();
        const int n = sph.ParticleCount();
        double cx = 0, cy = 0, cz = 0, ke = 0;
        for (int i = 0; i < n; i++) {
            const float* p = &P[(size_t)i * SPH_PARTICLE_FLOATS];
            cx += p[0]; cy += p[1]; cz += p[2];
            ke += 0.5 * p[13] * ((double)p[3] * p[3] + (double)p[4] * p[4] + (double)p[5] * p[5]);
        }
        printf("{\"particles\": %d, \"frames\": %d, \"grid\": [%d, %d, %d], \"seconds\": %.4f, \"particle_updates_per_s\": %.4g, "
               "\"centre_of_mass\": [%.6f, %.6f, %.6f], \"kinetic_energy\": %.6g, \"collisions_last_frame\": %d, "
               "\"bin_overflow_cells\": %d}\n",
               n, frames, sph.gridResolutionX, sph.gridResolutionY, sph.gridResolutionZ, sec,
               (double)n * sph.subSteps * frames / sec, cx / n, cy / n, cz / n, ke, sph.ReadCounter(SPH_CTR_COLLISIONS),
               sph.ReadCounter(SPH_CTR_BIN_OVERFLOW));
        if (dump) {
            FILE* f = fopen(dump, "wb");
            if (!f) { perror(dump); return 1; }
            fwrite(P.data(), sizeof(float), P.size(), f);
            fclose(f);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "sph_headless: %s\n", e.what());
        return 1;
    }
    return 0;
}
