/*
 * asan_check.cpp -- drives the CPU oracle (sph_oracle.cpp) under AddressSanitizer + UBSan.
 * TEST INFRASTRUCTURE ONLY ("Race detection / sanitizers": the oracle is built
 * with -fsanitize=address,undefined in CI).  `make -C oracle asan` builds and runs it; any report
 * aborts with a non-zero exit status.
 *
 * The states are chosen to walk every branch that indexes memory: bin-capacity overflow (compute:131),
 * particles outside the grid on every side (the clamp at compute:126 and the unclamped search cell at
 * compute:155-166), a NaN position, the OBB branch under a rotated, non-uniformly scaled cube
 * (compute:349-419), the AABB branch (compute:421-464), the sphere obstacle incl. the d <= eps branch
 * (compute:467-483), two sub-steps, and a grid that is resized between frames (SPH.cs:339-345).
 */
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/sph_hip.h"

struct ora_ctx;
extern "C" {
ora_ctx* ora_create(void);
void ora_destroy(ora_ctx*);
const char* ora_last_error(const ora_ctx*);
int ora_alloc_particles(ora_ctx*, int);
int ora_upload_particles(ora_ctx*, const void*, int);
int ora_download_particles(ora_ctx*, void*, int);
int ora_resize_grid(ora_ctx*, int, int, int, int);
int ora_set_params(ora_ctx*, const sph_params*);
int ora_update_static(ora_ctx*, const sph_phase*, int, const uint8_t*, int);
int ora_dispatch(ora_ctx*, int);
int ora_step_frame(ora_ctx*, int, float);
int ora_reset_counter(ora_ctx*, int);
int ora_read_counter(ora_ctx*, int, int*);
int ora_download_grid(ora_ctx*, int32_t*, int32_t*, int32_t*, uint32_t*, int32_t*);
int ora_download_neighbors(ora_ctx*, int32_t*, int32_t*, int);
}

namespace {

uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
float uni(uint64_t& s) { return (float)(splitmix(s) >> 40) * (1.0f / 16777216.0f); }

int check(ora_ctx* c, int rc, const char* what) {
    if (rc != 0) {
        fprintf(stderr, "asan_check: %s failed: %s\n", what, ora_last_error(c));
        return 1;
    }
    return 0;
}

sph_params base_params(int n) {
    sph_params p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.particleCount = n;
    p.soundSpeed = 10.f; p.gamma = 4.f; p.smoothingRadius = 0.3f; p.gravity = -9.81f;
    p.surfaceTensionCoefficient = 0.1f; p.surfaceTensionThreshold = 0.01f; p.xsphEpsilon = 0.3f;
    p.particleCollisionDamping = 1.0f;
    return p;
}

// one scenario: n particles uniform in [lo, hi) plus the hand-placed troublemakers
int run(const char* name, sph_params p, int rx, int ry, int rz, int cap, const float lo[3], const float hi[3], int n,
        int subSteps, int frames) {
    ora_ctx* c = ora_create();
    uint64_t seed = 12345;
    std::vector<float> aos((size_t)n * SPH_PARTICLE_FLOATS, 0.f);
    std::vector<uint8_t> ids((size_t)n);
    for (int i = 0; i < n; i++) {
        float* r = &aos[(size_t)i * SPH_PARTICLE_FLOATS];
        for (int k = 0; k < 3; k++) r[k] = lo[k] + uni(seed) * (hi[k] - lo[k]);
        r[9] = r[11] = 20.f; r[12] = 0.1f; r[13] = 0.2f;
        ids[i] = (uint8_t)(i & 1);
    }
    // troublemakers: far outside on each side, on the max face, NaN, and on the obstacle centre
    const float far_[6][3] = {{-1e9f, 0, 0}, {1e9f, 0, 0}, {0, -1e9f, 0}, {0, 1e9f, 0}, {0, 0, -3e38f}, {0, 0, 3e38f}};
    for (int k = 0; k < 6 && k < n; k++) memcpy(&aos[(size_t)k * SPH_PARTICLE_FLOATS], far_[k], 12);
    if (n > 8) {
        aos[6 * SPH_PARTICLE_FLOATS + 0] = NAN;
        memcpy(&aos[7 * SPH_PARTICLE_FLOATS], p.obstaclePos, 12);
        for (int k = 0; k < 3; k++) aos[8 * SPH_PARTICLE_FLOATS + k] = p.minBound[k] + p.cellSize * (float)(k == 0 ? rx : k == 1 ? ry : rz);
    }
    const sph_phase table[2] = {{20.f, 0.1f, 0.2f, {0.2f, 0.5f, 1.f, 0.8f}}, {10.f, 0.1f, 0.2f, {1.f, 0.8f, 0.2f, 0.8f}}};
    int bad = 0;
    bad |= check(c, ora_alloc_particles(c, n), "alloc");
    bad |= check(c, ora_upload_particles(c, aos.data(), n), "upload");
    p.gridResolutionX = rx; p.gridResolutionY = ry; p.gridResolutionZ = rz; p.maxParticlesPerCell = cap;
    std::vector<int32_t> cs, ce, sorted((size_t)n), counts((size_t)n), lists((size_t)n * 64), gc;
    std::vector<uint32_t> cop((size_t)n);
    for (int f = 0; f < frames && !bad; f++) {
        // the grid changes size between frames, like a rocking boundary cube (SPH.cs:339-345)
        const int gx = rx + (f & 1), gy = ry + (f & 1), gz = rz;
        p.gridResolutionX = gx; p.gridResolutionY = gy;
        bad |= check(c, ora_resize_grid(c, gx, gy, gz, cap), "resize_grid");
        bad |= check(c, ora_set_params(c, &p), "set_params");
        bad |= check(c, ora_update_static(c, table, 2, ids.data(), n), "update_static");
        for (int k = 0; k < 3; k++) ora_reset_counter(c, k);
        bad |= check(c, ora_step_frame(c, subSteps, 0.02f), "step_frame");
        const size_t cells = (size_t)gx * gy * gz;
        cs.assign(cells, 0); ce.assign(cells, 0); gc.assign(cells, 0);
        bad |= check(c, ora_download_grid(c, cs.data(), ce.data(), sorted.data(), cop.data(), gc.data()), "download_grid");
        bad |= check(c, ora_download_neighbors(c, counts.data(), lists.data(), 64), "download_neighbors");
    }
    bad |= check(c, ora_download_particles(c, aos.data(), n), "download");
    int ctr[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) ora_read_counter(c, k, &ctr[k]);
    printf("asan_check %-28s n=%d frames=%d  collisions=%d overflow_cells=%d out_of_grid=%d  %s\n", name, n, frames, ctr[0],
           ctr[1], ctr[2], bad ? "FAILED" : "ok");
    ora_destroy(c);
    return bad;
}

}  // namespace

int main() {
    int bad = 0;
    {   // AABB branch (boundaryHalfExtents = 0), dense enough to overflow a cap of 8, two sub-steps
        sph_params p = base_params(4000);
        p.boundsCenter[1] = 2.5f; p.boundsSize[0] = p.boundsSize[1] = p.boundsSize[2] = 5.f;
        p.cellSize = 0.25f; p.minBound[0] = -2.5f; p.minBound[1] = 0.f; p.minBound[2] = -2.5f;
        const float lo[3] = {-2.4f, 0.1f, -2.4f}, hi[3] = {-0.5f, 2.0f, 0.5f};
        bad |= run("aabb_overflow_substeps2", p, 20, 20, 20, 8, lo, hi, 4000, 2, 3);
    }
    {   // OBB branch: cube rotated 20 degrees about z, scale (4, 6, 4), centre (0, 3, 0); sphere obstacle inside
        sph_params p = base_params(3000);
        const float th = 20.0f * 3.14159265f / 180.0f, cs = cosf(th), sn = sinf(th);
        const float S[3] = {4.f, 6.f, 4.f}, T[3] = {0.f, 3.f, 0.f};
        const float R[3][3] = {{cs, -sn, 0}, {sn, cs, 0}, {0, 0, 1}};
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) p.boundaryMatrix[r * 4 + k] = R[r][k] * S[k];
            p.boundaryMatrix[r * 4 + 3] = T[r];
        }
        p.boundaryMatrix[15] = 1.f;
        for (int r = 0; r < 3; r++) {          // inverse = S^-1 R^T (x - T)
            float t = 0;
            for (int k = 0; k < 3; k++) {
                p.boundaryInvMatrix[r * 4 + k] = R[k][r] / S[r];
                t -= R[k][r] / S[r] * T[k];
            }
            p.boundaryInvMatrix[r * 4 + 3] = t;
        }
        p.boundaryInvMatrix[15] = 1.f;
        p.boundaryHalfExtents[0] = p.boundaryHalfExtents[1] = p.boundaryHalfExtents[2] = 0.5f;
        p.obstacleRadius = 1.2f; p.obstaclePos[0] = 0.5f; p.obstaclePos[1] = 1.0f; p.obstaclePos[2] = 0.f;
        p.cellSize = 0.25f; p.minBound[0] = -3.f; p.minBound[1] = -0.5f; p.minBound[2] = -2.f;
        const float lo[3] = {-2.8f, -0.4f, -1.9f}, hi[3] = {2.8f, 3.0f, 1.9f};
        bad |= run("obb_rotated_sphere", p, 24, 28, 16, 500, lo, hi, 3000, 1, 4);
    }
    return bad ? 1 : 0;
}
