// sph_host.hpp -- C++ headless host over the C ABI of libsph_hip (include/sph_hip.h).
//
// The reference's host layer is C# on Unity (Assets/SPH.cs, SpawnBox.cs, SphereMover.cs); this image
// has no .NET toolchain, so the compiled-language mirror of that layer is C++.  Class and member
// names follow the reference; every method cites the lines it restates.  It is the same logic as
// sph_amd/host.py (which the tests drive) and spawns bit-identical particles (rule O9 RNG).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/sph_hip.h"

namespace sphhost {

struct Vector3 {
    float x = 0, y = 0, z = 0;
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct Quaternion { float x = 0, y = 0, z = 0, w = 1; };
struct Color { float r = 1, g = 1, b = 1, a = 1; };

inline Quaternion AngleAxis(float degrees, Vector3 axis) {
    double n = std::sqrt((double)axis.x * axis.x + (double)axis.y * axis.y + (double)axis.z * axis.z);
    double half = (double)degrees * M_PI / 180.0 * 0.5, s = std::sin(half);
    return {(float)(axis.x / n * s), (float)(axis.y / n * s), (float)(axis.z / n * s), (float)std::cos(half)};
}

// Quaternion.Angle: 0 inside Unity's dead zone |dot| > 1 - 1e-6
inline float QuaternionAngle(const Quaternion& a, const Quaternion& b) {
    float d = std::fabs(a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w);
    if (d > 1.0f) d = 1.0f;
    if (d > 1.0f - 1e-6f) return 0.0f;
    return (float)(std::acos((double)d) * 2.0 * 180.0 / M_PI);
}

struct Matrix4x4 {
    float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   // row-major, acts on column vectors
};

struct Transform {
    Vector3 position;
    Quaternion rotation;
    Vector3 localScale{1, 1, 1};
    Vector3 lossyScale() const { return localScale; }

    Matrix4x4 localToWorldMatrix() const {            // Matrix4x4.TRS in fp32
        const float x = rotation.x, y = rotation.y, z = rotation.z, w = rotation.w;
        const float R[9] = {1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y - z * w), 2.0f * (x * z + y * w),
                            2.0f * (x * y + z * w), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z - x * w),
                            2.0f * (x * z - y * w), 2.0f * (y * z + x * w), 1.0f - 2.0f * (x * x + y * y)};
        Matrix4x4 M;
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) M.m[r * 4 + c] = R[r * 3 + c] * localScale[c];
            M.m[r * 4 + 3] = position[r];
        }
        M.m[12] = M.m[13] = M.m[14] = 0;
        M.m[15] = 1;
        return M;
    }

    Matrix4x4 worldToLocalMatrix() const {            // inverse of the affine TRS, in double, rounded once
        const Matrix4x4 M = localToWorldMatrix();
        double a[9], inv[9];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) a[r * 3 + c] = M.m[r * 4 + c];
        const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) +
                           a[2] * (a[3] * a[7] - a[4] * a[6]);
        inv[0] = (a[4] * a[8] - a[5] * a[7]) / det; inv[1] = (a[2] * a[7] - a[1] * a[8]) / det; inv[2] = (a[1] * a[5] - a[2] * a[4]) / det;
        inv[3] = (a[5] * a[6] - a[3] * a[8]) / det; inv[4] = (a[0] * a[8] - a[2] * a[6]) / det; inv[5] = (a[2] * a[3] - a[0] * a[5]) / det;
        inv[6] = (a[3] * a[7] - a[4] * a[6]) / det; inv[7] = (a[1] * a[6] - a[0] * a[7]) / det; inv[8] = (a[0] * a[4] - a[1] * a[3]) / det;
        Matrix4x4 I;
        for (int r = 0; r < 3; r++) {
            double t = 0;
            for (int c = 0; c < 3; c++) {
                I.m[r * 4 + c] = (float)inv[r * 3 + c];
                t -= inv[r * 3 + c] * (double)M.m[c * 4 + 3];
            }
            I.m[r * 4 + 3] = (float)t;
        }
        return I;
    }

    Vector3 TransformPoint(Vector3 p) const {
        const Matrix4x4 M = localToWorldMatrix();
        Vector3 o;
        for (int r = 0; r < 3; r++)
            o[r] = ((M.m[r * 4 + 0] * p.x + M.m[r * 4 + 1] * p.y) + M.m[r * 4 + 2] * p.z) + M.m[r * 4 + 3];
        return o;
    }
};

struct Time {                                         // rule O8: fixed 1/60 s frames, frame 0 at t = 0
    int frame = 0;
    float time() const { return (float)frame * (float)(1.0 / 60.0); }
    void advance() { frame++; }
};

struct SphereCollider {                               // what SPH.UpdateObstacle reads (SPH.cs:350-354)
    float radius = 0.5f;
    Transform transform;
};

struct SpawnBox {                                     // Assets/SpawnBox.cs:7-25
    int particleCount = 1000;
    Vector3 spawnSize{2, 2, 2};
    float restDensity = 1000.0f;
    float viscosity = 0.1f;
    float particleMass = 0.03f;
    Vector3 initialVelocity;
    Color particleColor;
    Transform transform;
    Vector3 SpawnCenter() const { return transform.position; }
};

struct CircularMotion {                               // Assets/SphereMover.cs:12-21
    Transform* transform = nullptr;
    const Time* time = nullptr;
    Vector3 center;
    float radius = 5.0f, speed = 1.0f;
    void Update() {
        const float angle = time->time() * speed;
        const float x = center.x + radius * (float)std::cos((double)angle);
        const float z = center.z + radius * (float)std::sin((double)angle);
        transform->position = {x, center.y, z};
    }
};

struct RockingMotion {                                // build-defined (BASELINE config 3): cube rotates about z
    Transform* transform = nullptr;
    const Time* time = nullptr;
    float amplitudeDeg = 10.0f, rate = 0.5f;
    void Update() {
        const double theta = (double)amplitudeDeg * std::sin((double)rate * (double)time->time());
        transform->rotation = AngleAxis((float)theta, {0, 0, 1});
    }
};

// counter-based spawn RNG (rule O9): top 24 bits of splitmix64(seed * 2^32 + 3 * i + k)
inline float spawn_uniform(uint64_t seed, uint64_t index, int k) {
    uint64_t x = seed * (1ull << 32) + index * 3ull + (uint64_t)k;
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

class SPH {                                           // Assets/SPH.cs
public:
    // Inspector fields with the code defaults of SPH.cs:5-49
    float soundSpeed = 20.0f, gamma = 7.0f, smoothingRadius = 0.2f, gravity = -9.81f;
    float timeStep = 0.003f;
    int subSteps = 2;
    float surfaceTensionCoefficient = 0.03f, surfaceTensionThreshold = 0.01f, xsphEpsilon = 0.5f;
    Vector3 boundsCenter{0, 0, 0}, boundsSize{10, 10, 10};
    Transform* boundaryCube = nullptr;
    SphereCollider* obstacleCollider = nullptr;
    float obstacleRepulsionStiffness = 5000.0f, particleCollisionDamping = 0.9999f;
    int maxParticlesPerCell = 100, gridResolution = 50;
    std::vector<SpawnBox*> spawnBoxes;
    uint64_t seed = 12345;                            // build-defined (rule O9)
    bool useStepFrame = false;                        // true: SPH.cs:201-222 as one sph_step_frame call
    int totalCells = 0;                               // SPH.cs:79

    explicit SPH(int device = 0) {
        if (sph_create(&ctx_, device) != SPH_OK) throw std::runtime_error("sph_create failed (no gfx950 device?)");
    }
    ~SPH() { OnDestroy(); }
    SPH(const SPH&) = delete;
    SPH& operator=(const SPH&) = delete;

    int ParticleCount() const { return particleCount_; }                 // SPH.cs:105
    std::vector<float> GetParticleBuffer() {                              // SPH.cs:107-110 (72-byte records)
        std::vector<float> out((size_t)particleCount_ * SPH_PARTICLE_FLOATS);
        ck(sph_download_particles(ctx_, out.data(), particleCount_));
        return out;
    }
    sph_ctx* Context() { return ctx_; }
    int gridResolutionX = 0, gridResolutionY = 0, gridResolutionZ = 0;
    float cellSize = 0;
    Vector3 gridMin;

    void Start() {                                                        // SPH.cs:112-171
        particleCount_ = 0;
        for (auto* sb : spawnBoxes) particleCount_ += sb->particleCount;  // SPH.cs:124-128
        CreateParticles();                                                // SPH.cs:134
        if (!boundaryCube) {                                              // SPH.cs:137-148
            for (int k = 0; k < 3; k++) gridMin[k] = boundsCenter[k] - boundsSize[k] * 0.5f;
            SizeGrid(boundsSize);
            ck(sph_resize_grid(ctx_, gridResolutionX, gridResolutionY, gridResolutionZ, maxParticlesPerCell));
        } else {
            UpdateGridParametersFromBoundary();                           // SPH.cs:151
            lastPos_ = boundaryCube->position;
            lastRot_ = boundaryCube->rotation;
        }
        ck(sph_reset_counter(ctx_, SPH_CTR_COLLISIONS));                  // SPH.cs:158
        SetComputeParams();                                               // SPH.cs:162
    }

    void Update() {                                                       // SPH.cs:173-223
        if (boundaryCube) {                                               // SPH.cs:176-185
            const float dx = boundaryCube->position.x - lastPos_.x, dy = boundaryCube->position.y - lastPos_.y,
                        dz = boundaryCube->position.z - lastPos_.z;
            const float moved = std::sqrt(dx * dx + dy * dy + dz * dz);
            if (moved > kBoundaryUpdateThreshold || QuaternionAngle(boundaryCube->rotation, lastRot_) > kBoundaryUpdateThreshold) {
                UpdateGridParametersFromBoundary();
                lastPos_ = boundaryCube->position;
                lastRot_ = boundaryCube->rotation;
            }
        }
        UpdateObstacle();                                                 // SPH.cs:187
        ck(sph_reset_counter(ctx_, SPH_CTR_COLLISIONS));                  // SPH.cs:190
        SetComputeParams();                                               // SPH.cs:193
        UpdateParticleStaticProperties();                                 // SPH.cs:198
        if (useStepFrame) {
            ck(sph_step_frame(ctx_, subSteps, timeStep));
            return;
        }
        DispatchCompute(SPH_K_CLEAR_GRID, totalCells);                    // SPH.cs:201
        DispatchCompute(SPH_K_BUILD_GRID, particleCount_);                // SPH.cs:204
        deltaTime_ = timeStep / (float)subSteps;                          // SPH.cs:207
        ck(sph_set_delta_time(ctx_, deltaTime_));                         // SPH.cs:208
        for (int i = 0; i < subSteps; i++) {                              // SPH.cs:210-219
            DispatchCompute(SPH_K_VV_HALF_STEP, particleCount_);
            DispatchCompute(SPH_K_CLEAR, particleCount_);
            DispatchCompute(SPH_K_DENSITY_PRESSURE, particleCount_);
            DispatchCompute(SPH_K_FORCE_XSPH, particleCount_);
            DispatchCompute(SPH_K_VV_FULL_STEP, particleCount_);
            DispatchCompute(SPH_K_BOUND_OBS, particleCount_);
        }
        DispatchCompute(SPH_K_VELOCITY_DAMPING, particleCount_);          // SPH.cs:222
    }

    void OnDestroy() {                                                    // SPH.cs:235-241
        if (ctx_) sph_destroy(ctx_);
        ctx_ = nullptr;
    }

    int ReadCounter(int which) {
        int v = 0;
        ck(sph_read_counter(ctx_, which, &v));
        return v;
    }

private:
    static constexpr float kBoundaryUpdateThreshold = 0.001f;             // SPH.cs:100
    sph_ctx* ctx_ = nullptr;
    int particleCount_ = 0;
    float deltaTime_ = 0;
    Vector3 obstaclePos_;
    float obstacleRadius_ = 0;
    Vector3 lastPos_;
    Quaternion lastRot_;
    std::vector<uint8_t> particleSpawnBoxIndices_;                        // SPH.cs:75

    void ck(int rc) {
        if (rc != SPH_OK) throw std::runtime_error(std::string("libsph_hip: ") + sph_last_error(ctx_));
    }

    void CreateParticles() {                                              // SPH.cs:246-282
        std::vector<float> init((size_t)particleCount_ * SPH_PARTICLE_FLOATS, 0.0f);
        particleSpawnBoxIndices_.assign((size_t)particleCount_, 0);
        uint64_t index = 0;
        for (size_t bi = 0; bi < spawnBoxes.size(); bi++) {
            const SpawnBox& sb = *spawnBoxes[bi];
            const Vector3 c = sb.SpawnCenter();
            for (int i = 0; i < sb.particleCount; i++, index++) {
                float* p = &init[index * SPH_PARTICLE_FLOATS];
                for (int k = 0; k < 3; k++) {                             // Random.Range(-half, half) -> lerp (rule O9)
                    const float half = sb.spawnSize[k] * 0.5f;
                    const float u = spawn_uniform(seed, index, k);
                    const float r = (-half) + u * (half - (-half));
                    p[k] = c[k] + r;
                    p[3 + k] = sb.initialVelocity[k];
                }
                p[9] = sb.restDensity; p[10] = 0.0f; p[11] = sb.restDensity;
                p[12] = sb.viscosity; p[13] = sb.particleMass;
                p[14] = sb.particleColor.r; p[15] = sb.particleColor.g; p[16] = sb.particleColor.b; p[17] = sb.particleColor.a;
                particleSpawnBoxIndices_[index] = (uint8_t)bi;            // SPH.cs:276
            }
        }
        ck(sph_alloc_particles(ctx_, particleCount_));                    // SPH.cs:249
        ck(sph_upload_particles(ctx_, init.data(), particleCount_));      // SPH.cs:280
        UploadPhaseTable();
        ck(sph_set_phase_ids(ctx_, particleSpawnBoxIndices_.data(), particleCount_));
    }

    void UploadPhaseTable() {
        std::vector<sph_phase> t(spawnBoxes.size());
        for (size_t i = 0; i < spawnBoxes.size(); i++) {
            const SpawnBox& sb = *spawnBoxes[i];
            t[i] = {sb.restDensity, sb.viscosity, sb.particleMass,
                    {sb.particleColor.r, sb.particleColor.g, sb.particleColor.b, sb.particleColor.a}};
        }
        ck(sph_set_phase_table(ctx_, t.data(), (int)t.size()));
    }

    // SPH.cs:284-298 without the GetData/SetData round trip: the spawn boxes' current values go down
    // as a table; the engine re-applies it on the device only when something changed.
    void UpdateParticleStaticProperties() {
        UploadPhaseTable();
        ck(sph_apply_phase_table(ctx_));
    }

    void SizeGrid(Vector3 gridSize) {                                     // SPH.cs:142-147 / 331-336
        const float minSize = std::fmin(gridSize.x, std::fmin(gridSize.y, gridSize.z));
        cellSize = minSize / (float)gridResolution;
        gridResolutionX = (int)std::ceil((double)(gridSize.x / cellSize));
        gridResolutionY = (int)std::ceil((double)(gridSize.y / cellSize));
        gridResolutionZ = (int)std::ceil((double)(gridSize.z / cellSize));
        totalCells = gridResolutionX * gridResolutionY * gridResolutionZ;
    }

    void UpdateGridParametersFromBoundary() {                             // SPH.cs:303-346
        if (!boundaryCube) return;
        const float h = 0.5f;
        const Vector3 local[8] = {{-h, -h, -h}, {h, -h, -h}, {h, -h, h}, {-h, -h, h}, {-h, h, -h}, {h, h, -h}, {h, h, h}, {-h, h, h}};
        Vector3 mn = boundaryCube->TransformPoint(local[0]), mx = mn;
        for (int i = 1; i < 8; i++) {
            const Vector3 c = boundaryCube->TransformPoint(local[i]);
            for (int k = 0; k < 3; k++) { mn[k] = std::fmin(mn[k], c[k]); mx[k] = std::fmax(mx[k], c[k]); }
        }
        gridMin = mn;
        SizeGrid({mx.x - mn.x, mx.y - mn.y, mx.z - mn.z});
        ck(sph_resize_grid(ctx_, gridResolutionX, gridResolutionY, gridResolutionZ, maxParticlesPerCell));   // SPH.cs:339-345
    }

    void UpdateObstacle() {                                               // SPH.cs:348-361
        if (obstacleCollider) {
            obstacleRadius_ = obstacleCollider->radius * obstacleCollider->transform.lossyScale().x;
            obstaclePos_ = obstacleCollider->transform.position;
        } else {
            obstaclePos_ = {0, 0, 0};
            obstacleRadius_ = 0.0f;
        }
    }

    void SetComputeParams() {                                             // SPH.cs:379-412
        sph_params p;
        std::memset(&p, 0, sizeof p);
        p.struct_size = sizeof p;
        p.particleCount = particleCount_;
        p.deltaTime = deltaTime_;
        p.soundSpeed = soundSpeed; p.gamma = gamma; p.smoothingRadius = smoothingRadius; p.gravity = gravity;
        p.surfaceTensionCoefficient = surfaceTensionCoefficient; p.surfaceTensionThreshold = surfaceTensionThreshold;
        p.xsphEpsilon = xsphEpsilon; p.particleCollisionDamping = particleCollisionDamping;
        p.obstacleRepulsionStiffness = obstacleRepulsionStiffness;
        p.obstacleRadius = obstacleRadius_;
        for (int k = 0; k < 3; k++) { p.obstaclePos[k] = obstaclePos_[k]; p.minBound[k] = gridMin[k]; }
        p.gridResolutionX = gridResolutionX; p.gridResolutionY = gridResolutionY; p.gridResolutionZ = gridResolutionZ;
        p.maxParticlesPerCell = maxParticlesPerCell;
        p.cellSize = cellSize;
        if (boundaryCube) {
            const Matrix4x4 M = boundaryCube->localToWorldMatrix(), I = boundaryCube->worldToLocalMatrix();
            std::memcpy(p.boundaryMatrix, M.m, sizeof M.m);
            std::memcpy(p.boundaryInvMatrix, I.m, sizeof I.m);
            p.boundaryHalfExtents[0] = p.boundaryHalfExtents[1] = p.boundaryHalfExtents[2] = 0.5f;   // SPH.cs:405
        } else {
            for (int k = 0; k < 3; k++) { p.boundsCenter[k] = boundsCenter[k]; p.boundsSize[k] = boundsSize[k]; }
        }
        ck(sph_set_params(ctx_, &p));
    }

    void DispatchCompute(int kernel, int count) {                         // SPH.cs:439-443
        ck(sph_dispatch(ctx_, kernel, count));
    }
};

}  // namespace sphhost
