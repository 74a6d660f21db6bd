// libsph_hip.hip -- the C ABI of include/sph_hip.h over the gfx950 kernels.
//
// Host side of the drop-in boundary: what UnityEngine.ComputeShader / ComputeBuffer do for
// Assets/SPH.cs (FindKernel / Set* / SetBuffer / Dispatch / SetData / GetData / Release) is done
// here over HIP.  No CPU fallback exists: without a gfx950 device sph_create fails.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "slab_kernels.h"

using namespace sph;

namespace {

constexpr int kNumEvents = 8;    // start / stop of the four timed spans of a frame
// Kernel times come from the dispatch packets themselves: hipExtLaunchKernelGGL attaches a start and a stop event to
// a launch, so timing adds no packet to the stream (a hipEventRecord between two kernels is a barrier packet of its own
// and cost ~5 us of stream time each: 6 % of a 0.44 ms frame with five of them).
enum { E_GRID0 = 0, E_GRID1, E_DENS0, E_DENS1, E_FORCE0, E_FORCE1, E_TAIL0, E_TAIL1 };
constexpr int kTimingRing = 32;  // frames whose events may be outstanding

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

// Scratch of sph_slab_plane_counts_noalloc: per-plane counters, one copy of the array per GPU the library runs on.
// In host code this NAME is only a handle for hipGetSymbolAddress; the device address is resolved once per device
// (planeScratchFor) and only that pointer is ever passed to kernels, memsets and copies.
static __device__ __align__(256) int scratch[SPH_PLANE_COUNTS_MAX];

struct sph_ctx {
    int device = 0;
    int* planeScratch = nullptr;      // this GPU's copy of `scratch` (device address)
    hipStream_t stream = nullptr;
    std::string err;

    int n = 0;            // particle count
    int capacity = 0;     // allocated particle capacity
    StateSet set[2] = {};
    int cur = 0;          // set[cur] holds acc/rhoP/rho0/origId; velCur picks velV, posCur picks posM
    int velCur = 0;
    int posCur = 0;       // (the force kernel with the fused tail writes the new positions into the other buffer)
    // SPH_HIP_LDS_DENSITY=1: CS_DensityPressure stages its candidates in LDS per workgroup (k_density_pressure_lds).
    // Off by default: correct (bit-identical, 99.8 % of the lane-steps served from LDS, 5x fewer vector loads) but
    // 1.7-1.9x SLOWER at 1 M particles -- 3 waves per SIMD (50 KB of LDS per workgroup) and two workgroup barriers per
    // step leave the waves waiting 56 % of their cycles (profiles/r3_lds_density/)
    bool ldsDensity = false;
    unsigned long long* ldsStats = nullptr;   // SPH_HIP_LDS_STATS=1: [0] lane-steps from LDS, [1] from memory, [2] chunks staged
    unsigned launchLds = 0;   // dynamic LDS of the next LAUNCH_T (occupancy experiments: densLdsPad / forceLdsPad)
    unsigned densLdsPad = 0, forceLdsPad = 0;   // SPH_HIP_DENSITY_LDS_KB / SPH_HIP_FORCE_LDS_KB
    bool fuseTailOk = true;   // SPH_HIP_FUSE_TAIL=0: sph_step_frame keeps the separate tail pass (A/B)
    bool tailFusedLast = false;   // the last CS_ForceXSPH launch applied the frame's tail itself
    float4* color = nullptr;       // by id
    uint8_t* phaseOfId = nullptr;  // by id
    bool havePhaseIds = false;
    sph_phase* phaseTableDev = nullptr;
    std::vector<sph_phase> phaseTable;
    bool staticDirty = false;

    // grid
    int rx = 0, ry = 0, rz = 0, cap = 0;
    long long cells = 0;              // rx * ry * rz, the reference's cell count
    long long tableCells = 0;         // entries of the tiled cell table (rows padded to whole tiles)
    long long cellCapacity = 0;
    int* cumBase = nullptr;           // allocation behind the COUNT array of the grid build
    int* cumT = nullptr;              // counts per cell during a build (16-byte aligned); all-zero between builds
    int* cumOutBase = nullptr;        // allocation behind the cumulative cell table
    int* cumOut = nullptr;            // T[c] = #particles with cell <= c (16-byte aligned, T[-1] = 0)
    int* cellCum = nullptr;           // = T - 1: cellCum[c] = #particles with cell < c; [cells + 1] entries
    uint8_t* tileTouched = nullptr;   // per 4096-cell tile of the count array: a particle was counted into it
    int* cumBlocks = nullptr;         // particles per 4096-cell tile of the table (k_cum_block_sum), the seeds of its scan
    uint32_t* keys[2] = {nullptr, nullptr};
    uint32_t* vals[2] = {nullptr, nullptr};
    uint32_t* sid = nullptr;
    uint32_t* sortedKeys = nullptr;   // alias into keys[] after the last build
    bool gridValid = false;
    uint32_t* hitMask = nullptr;      // kMaskWords words per slot, written by CS_DensityPressure
    float4* nbrA = nullptr;           // packed neighbour records (PackedNbr), written with the mask
    float4* nbrB = nullptr;
    bool packedValid = false;         // nbrA / nbrB match the mask (phase table of <= 4 rows in force)
    bool maskValid = false;           // hitMask matches the current positions and grid
    bool useHitMask = true;           // SPH_HIP_HITMASK=0 turns the hit path off
    // SPH_HIP_ROW_SLACK: how many cell rows a lane of the force kernel's hit walk may run ahead of its wave's slowest
    // lane (8 = ungated).  Measured at 1 M particles (scripts/gpu_row_slack*.sh), force kernel in the driver's window /
    // at the defaults: 0 -> 0.236 / 0.152 ms, 1 -> 0.190 / 0.127, 2 -> 0.181 / 0.121, 3 -> 0.174 / 0.116, 4 -> 0.172 / 0.114,
    // 8 -> 0.1745 / 0.116.  Walking the rows in step does make a gather touch fewer lines, but the lanes of a wave have
    // similar hit counts per row and stay roughly in step by themselves; a tight gate only adds idle lanes.
    int rowSlack = 4;
    int xcdRunOverride = 0;           // SPH_HIP_XCD_RUN=<n> pins the XCD run length (experiments)
    bool debug = false;               // SPH_HIP_DEBUG=1: report on stderr which force kernel runs
    int lastForceVariant = -1;

    // Two sets of SPH_CTR_COUNT ints: the per-build counters (bin overflow, out of grid, plane shift, band misses,
    // non-finite) alternate between them, every grid build starts on the set its predecessor zeroed on the way
    // (k_cum_block_sum), so no fill precedes a build.  The collision counter always lives in set 0.
    int* countersBase = nullptr;
    int* counters = nullptr;          // the current set (what the kernels and sph_read_counter use)
    // Collision counter (SPH_CTR_COLLISIONS), kept lazily: the boundary pass leaves one count per workgroup
    // in blockCollide; they are folded into the device counter only when somebody reads it (or before the
    // next pass overwrites them).  SPH.cs:190 resets the counter every frame and reads it never, so the
    // usual frame pays neither a fill nor a reduction launch.
    bool collisionsResetPending = false;   // the counter's stored value is void: the next fold overwrites it
    int collideFoldBlocks = 0;             // > 0: blockCollide[0 .. this) holds counts not yet folded
    int* blockCollide = nullptr;
    int* hostInts = nullptr;          // 16 pinned ints: landing zone of the small device -> host reads

    // slab decomposition (multi-GPU)
    bool slab = false;
    int rxGlobal = 0, xOffset = 0, ownLoG = 0, ownHiG = 0;   // global plane coordinates
    int* slabCounts = nullptr;        // 4 ints on the device (message / rebuild counters)
    int* blockInts = nullptr;         // one int per workgroup: collision counts of the boundary kernels; slab
                                      // mode: plane shifts of the density kernel (capacity / 256 + 1 ints)
    int* phaseMismatch = nullptr;     // device flag: some slab record's statics are not its phase's row (PhaseCheck)
    SlabFilter filter{};              // set by sph_slab_rebuild, consumed by the next grid build
    OwnedRef ownedRef{nullptr, 1, 0, 0};   // who owned which slot at the last grid build (survives a re-cut of the slab)
    // exchange overlapped with compute (sph_slab_arm_pack): the next sph_step_frame packs the face messages itself,
    // from the band workgroups, between the two launches of the last CS_ForceXSPH
    struct ArmedPack { bool armed = false; int leftBelow = 0, rightFrom = 0, margin = 0; void* devL = nullptr; int capL = 0;
                       void* devR = nullptr; int capR = 0; } armedPack;
    uint8_t* wgBand = nullptr;        // per workgroup of 256 slots: 1 = holds a particle near a slab face
    hipEvent_t packEvent = nullptr;   // recorded behind the in-frame pack
    bool packPending = false;         // counts of an in-frame pack not collected yet (sph_slab_pack_wait)
    int packCapL = 0, packCapR = 0;
    bool bandValid = false;           // wgBand / bandSendLo / bandSendHi describe the frame that just ran
    int bandSendLo = 0, bandSendHi = 0;
    int* planeCounts = nullptr; int planeCountsCap = 0;   // sph_slab_plane_counts
    int cachedShift = -1;             // SPH_CTR_MAX_PLANE_SHIFT as read along with sph_slab_pack's counts (-1 = not cached)
    float* pos3 = nullptr;            // x, y, z per slot (12 bytes): candidate positions of the density kernel's fast path
    bool pos3Valid = false;           // written by the half step that produced the current positions
    bool slabMismatchClean = false;   // slab mode: phaseMismatch was read as 0 at the last grid build (covers this frame's records)
    bool skipGhostForce = true;       // SPH_HIP_GHOST_FORCE=1 switches the skip off (A/B)
    bool uniformMassOk = true;        // SPH_HIP_UNIFORM_MASS=0 switches the 12-byte candidate path off (A/B)
    bool uniformPhaseOk = true;       // SPH_HIP_UNIFORM_PHASE=0: the force kernel always decodes mass / viscosity by phase (A/B)
    int cachedMisses = -1;            // SPH_CTR_BAND_MISSES as read along with the survivor count of a slab build
    bool slabPackedOff = false;       // the phase table changed after the records were uploaded
    SlabRecord* recStage = nullptr;   // device staging for slab upload / download
    size_t recStageCap = 0;
    float* aosStage = nullptr;        // device AoS staging (n * 72 B)
    void* pinned = nullptr;           // pinned host staging
    size_t pinnedBytes = 0;

    sph_params prm{};
    bool haveParams = false;
    DevParams dp{};

    // per-stage hipEvent timing without blocking the host: a ring of event sets, harvested
    // lazily (when a slot is reused, or in sph_get_timings)
    hipEvent_t ev[kTimingRing][kNumEvents] = {};
    bool slotPending[kTimingRing] = {};
    unsigned slotRecorded[kTimingRing] = {};   // bit k: ev[slot][k] was attached to a launch (an empty slab launches nothing)
    hipEvent_t* tev = nullptr;          // events of the frame being enqueued, null = this frame is not timed
    unsigned* tevMask = nullptr;
    bool tevSub = false;                // the sub-step being enqueued is the timed one (the last)
    int ringNext = 0;
    double stageSum[kNumEvents] = {};   // ms; [7] = whole frame
    int timingLevel = 1;                // 0 none, 1 every timingStride-th frame (default), 2 every frame
    // level 1: every this-many-th frame is timed (SPH_HIP_TIMING_STRIDE).  A launch that carries events costs ~70 us of
    // HOST time (hipExtLaunchKernel; five per timed frame), an ordinary one ~3 us: at 1 M particles (0.33-0.44 ms of
    // GPU time per frame) timing EVERY frame makes the frame loop host-bound (-4 % throughput, scripts/host_rate.py),
    // every second frame does not -- the host catches up on the untimed frames.
    int timingStride = 2;
    unsigned eventFlags = hipEventDisableSystemFence;   // SPH_HIP_EVENT_FENCE=1: default events (A/B)
    int framesSinceHarvest = 0;         // level 1 times every timingStride-th frame, counted from the last sph_get_timings
    int framesTimed = 0;
};

namespace {

// Device address of `scratch` on `device` (which must be current): resolved once per device, then cached.
constexpr int kMaxDevices = 64;
std::mutex g_scratchMutex;
std::mutex g_scratchUse[kMaxDevices];      // serialises the users of one GPU's `scratch` (sph_slab_plane_counts_noalloc)
int* g_scratchPtr[kMaxDevices] = {};

hipError_t planeScratchFor(int device, int** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if (!g_scratchPtr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratchPtr[device] = (int*)p;
    }
    *out = g_scratchPtr[device];
    return hipSuccess;
}

int fail(sph_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(c, e_ == hipErrorOutOfMemory ? SPH_E_NOMEM : SPH_E_HIP, "%s failed: %s", \
                        #expr, hipGetErrorString(e_));                                           \
    } while (0)

#define CHECK_LAUNCH(c) HIP_TRY(c, hipGetLastError())

// Launch on the ctx's stream; e0 / e1 >= 0: attach ev[e0] as the start and / or ev[e1] as the stop event of this launch
// when the frame is timed.
#define LAUNCH_T(c, e0, e1, kernel, grid, ...)                                                                        \
    do {                                                                                                              \
        hipEvent_t s_ = ((c)->tev && (e0) >= 0) ? (c)->tev[(e0) < 0 ? 0 : (e0)] : (hipEvent_t) nullptr;                \
        hipEvent_t t_ = ((c)->tev && (e1) >= 0) ? (c)->tev[(e1) < 0 ? 0 : (e1)] : (hipEvent_t) nullptr;                \
        if (s_ || t_) {                                                                                               \
            hipExtLaunchKernelGGL(kernel, grid, dim3(kBlock), (c)->launchLds, (c)->stream, s_, t_, 0, __VA_ARGS__);                \
            if (s_) *(c)->tevMask |= 1u << ((e0) < 0 ? 0 : (e0));                                                     \
            if (t_) *(c)->tevMask |= 1u << ((e1) < 0 ? 0 : (e1));                                                     \
        } else {                                                                                                      \
            hipLaunchKernelGGL(kernel, grid, dim3(kBlock), (c)->launchLds, (c)->stream, __VA_ARGS__);                  \
        }                                                                                                             \
    } while (0)

inline int blocksFor(long long n) { return (int)((n + kBlock - 1) / kBlock); }

template <class T>
int devAlloc(sph_ctx* c, T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(c, hipMalloc((void**)p, count * sizeof(T)));
    return SPH_OK;
}

template <class T>
void devFree(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

void freeParticles(sph_ctx* c) {
    for (int k = 0; k < 2; k++) {
        devFree(c->set[k].posM); devFree(c->set[k].velV); devFree(c->set[k].acc);
        devFree(c->set[k].rhoP); devFree(c->set[k].rho0); devFree(c->set[k].origId); devFree(c->set[k].phase);
        devFree(c->keys[k]); devFree(c->vals[k]);
    }
    devFree(c->color); devFree(c->phaseOfId); devFree(c->sid);
    devFree(c->aosStage); devFree(c->recStage); devFree(c->planeCounts); c->planeCountsCap = 0; devFree(c->wgBand); devFree(c->blockInts); devFree(c->blockCollide); devFree(c->hitMask); devFree(c->nbrA); devFree(c->nbrB); devFree(c->pos3);
    c->recStageCap = 0;
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->pinned = nullptr; c->pinnedBytes = 0;
    c->capacity = 0; c->n = 0; c->collideFoldBlocks = 0;
}

void freeGrid(sph_ctx* c) {
    devFree(c->cumBase); devFree(c->cumOutBase); devFree(c->cumBlocks); devFree(c->tileTouched);
    c->cumT = c->cumOut = c->cellCum = nullptr;
    c->cellCapacity = 0; c->cells = 0; c->tableCells = 0;
}

int ensurePinned(sph_ctx* c, size_t bytes) {
    if (bytes <= c->pinnedBytes) return SPH_OK;
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->pinned = nullptr; c->pinnedBytes = 0;
    HIP_TRY(c, hipHostMalloc(&c->pinned, bytes, hipHostMallocDefault));
    c->pinnedBytes = bytes;
    return SPH_OK;
}

// O4: constants derived once on the host in fp32 (compute:20-21, 150-151, 214-216, 262, 269).
void deriveParams(sph_ctx* c) {
    const sph_params& q = c->prm;
    DevParams& d = c->dp;
    const float PI = 3.14159265359f;
    d.n = c->n;
    d.dt = q.deltaTime;
    d.soundSpeed = q.soundSpeed; d.gamma = q.gamma; d.h = q.smoothingRadius; d.gravity = q.gravity;
    d.sigma = q.surfaceTensionCoefficient; d.stThreshold = q.surfaceTensionThreshold;
    d.xsphEps = q.xsphEpsilon; d.collisionDamping = q.particleCollisionDamping;
    d.obstacleRadius = q.obstacleRadius;
    for (int k = 0; k < 3; k++) {
        d.obstaclePos[k] = q.obstaclePos[k];
        d.boundsCenter[k] = q.boundsCenter[k];
        d.boundsSize[k] = q.boundsSize[k];
        d.minBound[k] = q.minBound[k];
    }
    d.rx = c->slab ? c->rx : q.gridResolutionX;
    d.ry = q.gridResolutionY; d.rz = q.gridResolutionZ;
    d.rxg = q.gridResolutionX;
    d.ryT = tiles_along(d.ry, kTileYLog);
    {   // XCD run length (see xcd_tile): 32 runs per XCD, between 16 and 128 workgroups.  Measured on the final
        // kernels (scripts/gpu_xcd.sh): config 3 (1 M particles) is flat from 16 to 48 and loses 2 % at 96; config 5
        // (8.4 M) is best at 64-128 (2.77 ms), 256 costs 1.5 %, 1024 6 %, 16 2 %.
        const int nwg = (c->n + kBlock - 1) / kBlock;
        int run = nwg / 256;
        if (run < 16) run = 16;
        if (run > 128) run = 128;
        d.xcdRun = c->xcdRunOverride > 0 ? c->xcdRunOverride : run;
    }
    d.xoff = c->slab ? c->xOffset : 0;
    d.ownLo = c->slab ? c->ownLoG - c->xOffset : 0;
    d.ownHi = c->slab ? c->ownHiG - c->xOffset : d.rx;
    d.cap = q.maxParticlesPerCell;
    d.cellSize = q.cellSize;
    for (int k = 0; k < 12; k++) { d.M[k] = q.boundaryMatrix[k]; d.Minv[k] = q.boundaryInvMatrix[k]; }
    const float* he = q.boundaryHalfExtents;
    d.obb = sqrtf((he[0] * he[0] + he[1] * he[1]) + he[2] * he[2]) > 0.001f ? 1 : 0;
    const float h = q.smoothingRadius;
    d.h2 = h * h;
    d.poly6 = 315.0f / (64.0f * PI * powf(h, 9.0f));
    d.spikyGradFactor = -45.0f / (PI * powf(h, 6.0f));
    d.viscLapFactor = 45.0f / (PI * powf(h, 6.0f));
    d.lapColFactor = 1890.0f / (64.0f * PI * powf(h, 9.0f));
    d.wgBand = nullptr; d.wgPhase = -1; d.ghostKeys = nullptr;
}

int checkReady(sph_ctx* c, bool needGrid) {
    if (!c) return SPH_E_INVALID;
    if (c->n <= 0 && !(c->slab && c->capacity > 0)) return fail(c, SPH_E_STATE, "no particles allocated");
    if (!c->haveParams) return fail(c, SPH_E_STATE, "sph_set_params not called");
    if (c->cells <= 0) return fail(c, SPH_E_STATE, "sph_resize_grid not called");
    deriveParams(c);   // picks up slab / grid changes made after sph_set_params
    if (c->dp.rx != c->rx || c->dp.ry != c->ry || c->dp.rz != c->rz || c->dp.cap != c->cap)
        return fail(c, SPH_E_STATE, "params grid (%d,%d,%d,cap %d) != allocated grid (%d,%d,%d,cap %d)", c->dp.rx,
                    c->dp.ry, c->dp.rz, c->dp.cap, c->rx, c->ry, c->rz, c->cap);
    if (c->slab) {
        if (c->dp.rxg != c->rxGlobal)
            return fail(c, SPH_E_STATE, "params gridResolutionX %d != slab global resolution %d", c->dp.rxg, c->rxGlobal);
        if (c->xOffset < 0 || c->xOffset + c->rx > c->rxGlobal || c->ownLoG < c->xOffset || c->ownHiG > c->xOffset + c->rx)
            return fail(c, SPH_E_STATE, "slab planes [%d,%d) own [%d,%d) do not fit the local grid of %d planes at %d",
                        c->xOffset, c->xOffset + c->rx, c->ownLoG, c->ownHiG, c->rx, c->xOffset);
    } else if (c->prm.particleCount != c->n) {
        return fail(c, SPH_E_STATE, "params particleCount %d != allocated %d", c->prm.particleCount, c->n);
    }
    c->dp.n = c->n;
    if (!(c->dp.cellSize > 0.0f)) return fail(c, SPH_E_INVALID, "cellSize must be > 0");
    if (needGrid && !c->gridValid) return fail(c, SPH_E_STATE, "grid not built (dispatch CS_BuildGrid first)");
    return SPH_OK;
}

// n <= 16 ints from the device, through pinned memory (a pageable destination makes the runtime stage
// the copy and costs ~0.1 ms per read); synchronises the stream.
int readInts(sph_ctx* c, const int* dev, int n, int* out) {
    HIP_TRY(c, hipMemcpyAsync(c->hostInts, dev, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < n; k++) out[k] = c->hostInts[k];
    return SPH_OK;
}

int harvestSlot(sph_ctx* c, int slot) {
    hipEvent_t* ev = c->ev[slot];
    const unsigned rec = c->slotRecorded[slot];
    c->slotPending[slot] = false;
    if (!rec) return SPH_OK;                                  // an empty slab: nothing was launched
    for (int k = kNumEvents - 1; k >= 0; k--)
        if (rec & (1u << k)) { HIP_TRY(c, hipEventSynchronize(ev[k])); break; }
    auto span = [&](int a, int b, int into) -> int {
        if (!(rec & (1u << a)) || !(rec & (1u << b))) return SPH_OK;
        float t = 0;
        HIP_TRY(c, hipEventElapsedTime(&t, ev[a], ev[b]));
        c->stageSum[into] += t;
        return SPH_OK;
    };
    int rc = span(E_GRID0, E_GRID1, 0);                       // grid build incl. the fused first half step
    if (!rc) rc = span(E_DENS0, E_DENS1, 2);
    if (!rc) rc = span(E_FORCE0, E_FORCE1, 3);
    if (!rc) rc = span(E_TAIL0, E_TAIL1, 4);                  // fused tail: full step + boundary (+ damping)
    // whole frame, first launch to last (the force kernel is the last one when it applied the tail itself)
    if (!rc) rc = span(E_GRID0, (rec & (1u << E_TAIL1)) ? E_TAIL1 : E_FORCE1, 7);
    if (rc) return rc;
    c->framesTimed++;
    return SPH_OK;
}

StateSet curSet(sph_ctx* c) {
    StateSet s = c->set[c->cur];
    s.velV = c->set[c->velCur].velV;
    s.posM = c->set[c->posCur].posM;
    return s;
}

int applyPhaseTable(sph_ctx* c) {
    if (!c->staticDirty) return SPH_OK;
    if (c->slab) {
        // live edit of a spawn box in slab mode: every local slot (owned and ghost) takes its phase's row; records
        // that arrive later take theirs when they are stored (store_record).  Every rank applies the same table.
        const StateSet st = curSet(c);
        if (!c->phaseTable.empty() && c->n > 0 && st.phase) {
            hipLaunchKernelGGL(k_apply_phase_table_slots, dim3(blocksFor(c->n)), dim3(kBlock), 0, c->stream, c->n, st,
                               (const sph_phase*)c->phaseTableDev, (int)c->phaseTable.size());
            CHECK_LAUNCH(c);
            c->packedValid = false;     // the records the density kernel packed carry nothing static, but stay on the safe side
        }
        c->staticDirty = false;
        c->slabPackedOff = false;       // local slots and table agree again
        return SPH_OK;
    }
    if (c->phaseTable.empty() || !c->havePhaseIds) { c->staticDirty = false; return SPH_OK; }
    hipLaunchKernelGGL(k_apply_phase_table, dim3(blocksFor(c->n)), dim3(kBlock), 0, c->stream, c->n,
                       curSet(c),
                       c->color, c->phaseOfId, c->phaseTableDev);
    CHECK_LAUNCH(c);
    c->staticDirty = false;
    return SPH_OK;
}

// CS_ClearGrid (compute:104-111): a bin is a range of the cumulative table here, so clearing = zeroing
// the table.  Only the kernel-granular dispatch runs it (and the empty-slab case): a grid build overwrites the
// whole table and needs no clearing pass.
int clearGrid(sph_ctx* c) {
    // T[-4 .. cells padded to int4): zero = "no particle in any cell"
    const size_t padded = ((size_t)c->tableCells + 3) / 4 * 4 + 4;
    const long long n4 = (long long)(padded / 4);
    hipLaunchKernelGGL(k_clear_table, dim3((unsigned)((n4 + kBlock * 4 - 1) / (kBlock * 4))), dim3(kBlock), 0, c->stream,
                       (int4*)c->cumOutBase, n4, (int*)nullptr);
    CHECK_LAUNCH(c);
    c->gridValid = false;
    c->maskValid = false;
    c->cachedShift = -1;
    c->cachedMisses = -1;
    return SPH_OK;
}

// CS_BuildGrid (compute:117-136): hash + counting sort by cell over the cumulative table -> id order inside
// each cell -> physical reorder of the state into (cell, id) order.
int buildGrid(sph_ctx* c, bool fuseHalfStep = false) {
    const int n = c->n;
    const int nb = blocksFor(n);
    c->gridValid = false;
    c->maskValid = false;
    c->cachedShift = -1;
    c->cachedMisses = -1;
    // per-build counters: this build and its frame use the set the last build zeroed (k_cum_block_sum)
    int* const idleCounters = c->counters;
    c->counters = c->counters == c->countersBase ? c->countersBase + SPH_CTR_COUNT : c->countersBase;
    const SlabFilter filter = c->filter;
    c->filter.active = 0;
    c->slabMismatchClean = false;
    if (c->slab) c->ownedRef = OwnedRef{c->keys[1], c->dp.rx, c->dp.ownLo, c->dp.ownHi};   // as of THIS build
    if (n == 0) {   // an empty slab: the zeroed table is its grid
        if (int rc = clearGrid(c)) return rc;
        HIP_TRY(c, hipMemsetAsync(idleCounters + SPH_CTR_BIN_OVERFLOW, 0, sizeof(int) * (SPH_CTR_COUNT - SPH_CTR_BIN_OVERFLOW), c->stream));
        c->gridValid = true;
        return SPH_OK;
    }
    StateSet src = curSet(c);
    // keys[0] = cell per old slot, vals[0] = arrival number; keys[1] / vals[1] / sid = key, source slot
    // and id per bucket slot
    LAUNCH_T(c, E_GRID0, -1, k_hash_count, dim3(nb), c->dp, (const float4*)src.posM, c->keys[0], c->vals[0], c->cumT, c->counters,
             filter, c->tileTouched);
    CHECK_LAUNCH(c);
    {   // counts -> cumulative cell table
        const long long cells4 = (c->tableCells + 3) / 4;
        const int cb = (int)((c->tableCells + kScanTile - 1) / kScanTile);
        // (counting the tile sums in k_hash_count instead -- one atomic per wave and 4096-cell tile -- was tried: the fluid
        // sits in a few dozen tiles, the atomics serialise on their addresses and the grid build went from 74 to 119 us)
        hipLaunchKernelGGL(k_cum_block_sum, dim3(cb), dim3(kBlock), 0, c->stream, (const int4*)c->cumT, cells4,
                           c->cumBlocks, (const uint8_t*)c->tileTouched, idleCounters);
        hipLaunchKernelGGL(k_cum_write, dim3(cb), dim3(kBlock), 0, c->stream, (int4*)c->cumT, (int4*)c->cumOut, cells4,
                           (const int*)c->cumBlocks, c->tileTouched);
    }
    c->sortedKeys = c->keys[1];
    uint32_t* perm = c->vals[1];
    hipLaunchKernelGGL(k_place, dim3(nb), dim3(kBlock), 0, c->stream, n, c->keys[0], c->vals[0], src.origId,
                       c->cellCum, c->sortedKeys, perm, c->sid);
    int nNew = n;
    if (filter.active) {
        // the survivors are what the table counted: one read, the only synchronisation of a slab rebuild
        // (and, same synchronisation, the particles the last in-frame pack failed to send)
        HIP_TRY(c, hipMemcpyAsync(c->hostInts, c->cellCum + c->tableCells, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        if (filter.bandWg)
            HIP_TRY(c, hipMemcpyAsync(c->hostInts + 1, c->counters + SPH_CTR_BAND_MISSES, sizeof(int), hipMemcpyDeviceToHost,
                                      c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->hostInts + 2, c->phaseMismatch, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        nNew = c->hostInts[0];
        c->cachedMisses = filter.bandWg ? c->hostInts[1] : 0;
        c->slabMismatchClean = c->hostInts[2] == 0;   // every record of this frame has been checked by now (same stream)
        c->n = nNew;
        c->dp.n = nNew;
        if (nNew == 0) { c->cur = 1 - c->cur; c->velCur = 1 - c->velCur; c->posCur = 1 - c->posCur; c->gridValid = true; return SPH_OK; }
    }
    const int nbNew = blocksFor(nNew);
    StateSet dst = c->set[1 - c->cur];
    // velocities of the new order go to the set that does NOT hold the current ones
    dst.velV = c->set[1 - c->velCur].velV;
    dst.posM = c->set[1 - c->posCur].posM;
    if (fuseHalfStep)   // sph_step_frame: rank + reorder + first CS_VvHalfStep in one pass
        LAUNCH_T(c, -1, E_GRID1, k_rank_move<true>, dim3(nbNew), c->sortedKeys, perm, c->sid, nNew, c->cap, c->dp.dt, c->cellCum, src,
                 dst, c->counters, c->pos3);
    else
        hipLaunchKernelGGL(k_rank_move<false>, dim3(nbNew), dim3(kBlock), 0, c->stream, c->sortedKeys, perm, c->sid, nNew, c->cap,
                           c->dp.dt, c->cellCum, src, dst, c->counters, (float*)nullptr);
    CHECK_LAUNCH(c);
    c->pos3Valid = fuseHalfStep;
    c->cur = 1 - c->cur;
    c->velCur = 1 - c->velCur;
    c->posCur = 1 - c->posCur;
    c->gridValid = true;
    return SPH_OK;
}

// Fold the per-workgroup collision counts of the last boundary pass into the device counter.
int foldCollisions(sph_ctx* c) {
    if (c->collideFoldBlocks <= 0) return SPH_OK;
    hipLaunchKernelGGL(k_sum_into, dim3(1), dim3(kBlock), 0, c->stream, c->blockCollide, c->collideFoldBlocks,
                       c->countersBase + SPH_CTR_COLLISIONS, c->collisionsResetPending ? 1 : 0);
    CHECK_LAUNCH(c);
    c->collideFoldBlocks = 0;
    c->collisionsResetPending = false;
    return SPH_OK;
}

// wgPhase (CS_ForceXSPH only): -1 = one launch over everything; 1 = band workgroups, then 0 = the rest -- two
// launches that together are the one, writing the same velocity buffer (flipped after the second).
int dispatchKernel(sph_ctx* c, int id, bool fused = false, bool lastSub = false, int wgPhase = -1) {
    const int n = c->n;
    c->dp.n = n;
    c->dp.wgBand = wgPhase >= 0 ? c->wgBand : (const uint8_t*)nullptr;
    c->dp.wgPhase = wgPhase;
    c->dp.ghostKeys = (id == SPH_K_FORCE_XSPH && c->slab && fused && lastSub && c->skipGhostForce) ? (const uint32_t*)c->sortedKeys
                                                                                                  : (const uint32_t*)nullptr;
    if (n == 0 && id != SPH_K_CLEAR_GRID && id != SPH_K_BUILD_GRID) {   // an empty slab has nothing to launch
        if (id < SPH_K_CLEAR || id > SPH_K_VELOCITY_DAMPING) return fail(c, SPH_E_INVALID, "bad kernel id %d", id);
        return SPH_OK;
    }
    const int nb = blocksFor(n);
    StateSet s = curSet(c);
    switch (id) {
        case SPH_K_CLEAR:
            c->packedValid = false;   // density / pressure change under the packed neighbour records
            hipLaunchKernelGGL(k_clear, dim3(nb), dim3(kBlock), 0, c->stream, n, s.acc, s.rhoP, s.rho0);
            break;
        case SPH_K_CLEAR_GRID:
            return clearGrid(c);
        case SPH_K_BUILD_GRID:
            return buildGrid(c);
        case SPH_K_DENSITY_PRESSURE: {
            // packed neighbour records: only while every particle's mass / viscosity IS its phase's row
            // (one GPU: the table has been applied; slab mode: the records came with phase ids and the
            // table has not changed since -- records that disagree with it raise phaseMismatch)
            // ... and the phase id rides in the SIGN bits of density and pressure: density >= rest density > 0 and
            // Tait pressure B ((rho/rho0)^gamma - 1) >= 0 need restDensity > 0 in every phase and gamma > 0 (both are
            // Inspector-editable in the reference); otherwise the three-array gathers are used
            bool tableOk = !c->phaseTable.empty() && c->phaseTable.size() <= 4 && c->prm.gamma > 0.0f;
            for (size_t k = 0; tableOk && k < c->phaseTable.size(); k++) tableOk = c->phaseTable[k].restDensity > 0.0f;
            const bool packed = c->useHitMask && c->hitMask && tableOk &&
                                (c->slab ? (s.phase != nullptr && !c->slabPackedOff) : (c->havePhaseIds && !c->staticDirty));
            PackedNbr nbr{packed ? c->nbrA : nullptr, c->nbrB, s.velV, s.origId, c->phaseOfId,
                          c->slab ? (const uint8_t*)s.phase : (const uint8_t*)nullptr};
            // same particle mass in every phase, every particle's mass is its phase's (`packed`), and the 12-byte
            // positions were written by the half step that produced the current positions: 12-byte candidates
            bool uniform = packed && c->pos3 && c->pos3Valid && c->uniformMassOk && (!c->slab || c->slabMismatchClean);
            for (size_t k = 1; uniform && k < c->phaseTable.size(); k++) uniform = c->phaseTable[k].mass == c->phaseTable[0].mass;
            uint32_t* const mask = c->useHitMask ? c->hitMask : (uint32_t*)nullptr;
            const uint32_t* const skeys = c->slab ? (const uint32_t*)c->sortedKeys : (const uint32_t*)nullptr;
            const int d0 = c->tevSub ? E_DENS0 : -1, d1 = c->tevSub ? E_DENS1 : -1;
            c->launchLds = c->densLdsPad;
            if (uniform && !c->slab && c->ldsDensity && mask) {
                // candidates staged in LDS per workgroup (k_density_pressure_lds): opt-in experiment, SPH_HIP_LDS_DENSITY=1
                if (c->ldsStats)
                    LAUNCH_T(c, d0, d1, k_density_pressure_lds<true>, dim3(nb), c->dp, (const float4*)s.posM, (const float*)s.rho0, s.rhoP,
                             (const int*)c->cellCum, (const int*)c->counters, mask, c->capacity, nbr, (const float*)c->pos3,
                             c->phaseTable[0].mass, LdsStats{c->ldsStats});
                else
                    LAUNCH_T(c, d0, d1, k_density_pressure_lds<false>, dim3(nb), c->dp, (const float4*)s.posM, (const float*)s.rho0, s.rhoP,
                             (const int*)c->cellCum, (const int*)c->counters, mask, c->capacity, nbr, (const float*)c->pos3,
                             c->phaseTable[0].mass, LdsStats{nullptr});
            } else if (uniform)
                LAUNCH_T(c, d0, d1, k_density_pressure_rows<true>, dim3(nb), c->dp, (const float4*)s.posM, (const float*)s.rho0, s.rhoP,
                         (const int*)c->cellCum, (const int*)c->counters, mask, c->capacity, nbr, skeys, c->blockInts,
                         (const float*)c->pos3, c->phaseTable[0].mass);
            else
                LAUNCH_T(c, d0, d1, k_density_pressure_rows<false>, dim3(nb), c->dp, (const float4*)s.posM, (const float*)s.rho0, s.rhoP,
                         (const int*)c->cellCum, (const int*)c->counters, mask, c->capacity, nbr, skeys, c->blockInts,
                         (const float*)nullptr, 0.0f);
            c->launchLds = 0;
            if (c->slab)
                hipLaunchKernelGGL(k_slab_shift_reduce, dim3((nb + kShiftPerGroup - 1) / kShiftPerGroup), dim3(kBlock), 0, c->stream, c->blockInts, nb,
                                   c->counters + SPH_CTR_MAX_PLANE_SHIFT);
            c->maskValid = c->useHitMask && c->hitMask;
            c->packedValid = packed;
            break;
        }
        case SPH_K_FORCE_XSPH: {
            float4* velOut = c->set[1 - c->velCur].velV;
            // timed sub-step: the span runs from the first launch (band workgroups, or the only one) to the last
            const int f0 = (c->tevSub && wgPhase != 0) ? E_FORCE0 : -1, f1 = (c->tevSub && wgPhase != 1) ? E_FORCE1 : -1;
            // which kernel: no fresh hit mask -> scan; else the hit path with 0 = three gathers per neighbour, 1 = packed
            // records + per-phase decode, 2 = packed records, uniform mass and viscosity (see k_force_xsph_hits).  Slab
            // mode takes a packed mode only when the mismatch flag was read as clean for this frame's records.
            int mode = -1;
            if (c->maskValid) {
                const bool packedOk = c->packedValid && (c->slab ? (!c->slabPackedOff && c->slabMismatchClean) : !c->staticDirty);
                mode = 0;
                if (packedOk) {
                    mode = 2;
                    for (size_t k = 1; k < c->phaseTable.size(); k++)
                        if (c->phaseTable[k].mass != c->phaseTable[0].mass || c->phaseTable[k].viscosity != c->phaseTable[0].viscosity)
                            mode = 1;
                    if (!c->uniformPhaseOk) mode = 1;
                }
            }
            if (c->debug && mode != c->lastForceVariant) {
                static const char* const names[4] = {"k_force_xsph_rows (scan)", "k_force_xsph_hits<0> (hit mask, three gathers)",
                                                     "k_force_xsph_hits<1> (hit mask, packed neighbour records, per-phase decode)",
                                                     "k_force_xsph_hits<2> (hit mask, packed neighbour records, uniform mass / viscosity)"};
                fprintf(stderr, "[libsph_hip] CS_ForceXSPH -> %s%s\n", names[mode + 1], c->slab ? " [slab]" : "");
                c->lastForceVariant = mode;
            }
            float m[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
            for (size_t k = 0; k < c->phaseTable.size() && k < 4; k++) {
                m[k] = c->phaseTable[k].mass;
                v[k] = c->phaseTable[k].viscosity;
            }
            const PhaseConsts ph{m[0], m[1], m[2], m[3], v[0], v[1], v[2], v[3]};
            // one GPU, sph_step_frame, hit path: the force kernel applies the frame's tail itself (TailArgs); the new
            // positions go to the other position buffer
            const bool fuseTail = fused && !c->slab && mode >= 0 && c->fuseTailOk && wgPhase < 0;
            c->tailFusedLast = fuseTail;
            TailArgs tail{nullptr, 0, nullptr, nullptr};
            if (fuseTail) {
                if (int rcFold = foldCollisions(c)) return rcFold;
                tail = TailArgs{c->set[1 - c->posCur].posM, lastSub ? 1 : 0, c->blockCollide, c->counters};
            }
#define FORCE_HITS_T(MODE_, TAIL_)                                                                                            \
    LAUNCH_T(c, f0, f1, (k_force_xsph_hits<MODE_, TAIL_>), dim3(nb), c->dp, (const float4*)s.posM, (const float4*)s.velV,     \
             (const float2*)s.rhoP, s.acc, velOut, (const int*)c->cellCum, (const int*)c->counters, (const uint32_t*)c->hitMask, \
             c->capacity, (const float4*)c->nbrA, (const float4*)c->nbrB, ph, fused ? 1 : 0, c->rowSlack, tail)
#define FORCE_HITS(MODE_) do { if (fuseTail) FORCE_HITS_T(MODE_, true); else FORCE_HITS_T(MODE_, false); } while (0)
            c->launchLds = c->forceLdsPad;
            if (mode == 2) FORCE_HITS(2);
            else if (mode == 1) FORCE_HITS(1);
            else if (mode == 0) FORCE_HITS(0);
            else
                LAUNCH_T(c, f0, f1, k_force_xsph_rows, dim3(nb), c->dp, (const float4*)s.posM, (const float4*)s.velV,
                         (const float2*)s.rhoP, s.acc, velOut, (const int*)c->cellCum, (const int*)c->counters, fused ? 1 : 0);
#undef FORCE_HITS
#undef FORCE_HITS_T
            c->launchLds = 0;
            if (wgPhase != 1) c->velCur = 1 - c->velCur;
            if (fuseTail) {   // what the separate tail pass leaves behind
                c->posCur = 1 - c->posCur;
                c->collideFoldBlocks = nb;
                c->maskValid = false;
                c->pos3Valid = false;
            }
            break;
        }
        case SPH_K_VV_HALF_STEP:
            c->maskValid = false;   // positions move
            hipLaunchKernelGGL(k_half_step, dim3(nb), dim3(kBlock), 0, c->stream, n, c->dp.dt, s.posM, s.velV, s.acc, c->pos3);
            c->pos3Valid = c->pos3 != nullptr;
            break;
        case SPH_K_VV_FULL_STEP:
            if (fused) {   // + CS_BoundObs (+ CS_VelocityDamping after the last sub-step), one pass
                c->maskValid = false;
                c->pos3Valid = false;
                if (int rcFold = foldCollisions(c)) return rcFold;
                LAUNCH_T(c, c->tevSub ? E_TAIL0 : -1, c->tevSub ? E_TAIL1 : -1, k_full_bound_damp, dim3(nb), c->dp, s.posM, s.velV,
                         (const float4*)s.acc, c->blockCollide, c->slab ? (const uint32_t*)c->sortedKeys : (const uint32_t*)nullptr,
                         lastSub ? 1 : 0, c->counters);
                c->collideFoldBlocks = nb;
                break;
            }
            c->packedValid = false;   // velocities change under the packed neighbour records (record b)
            hipLaunchKernelGGL(k_full_step, dim3(nb), dim3(kBlock), 0, c->stream, n, c->dp.dt, s.velV, s.acc);
            break;
        case SPH_K_BOUND_OBS:
            c->maskValid = false;
            c->pos3Valid = false;
            if (int rcFold = foldCollisions(c)) return rcFold;
            hipLaunchKernelGGL(k_bound_obs, dim3(nb), dim3(kBlock), 0, c->stream, c->dp, s.posM, s.velV, c->blockCollide,
                               c->slab ? (const uint32_t*)c->sortedKeys : (const uint32_t*)nullptr, c->counters);
            c->collideFoldBlocks = nb;
            break;
        case SPH_K_VELOCITY_DAMPING:
            c->packedValid = false;
            hipLaunchKernelGGL(k_velocity_damping, dim3(nb), dim3(kBlock), 0, c->stream, n, s.velV);
            break;
        default:
            return fail(c, SPH_E_INVALID, "bad kernel id %d", id);
    }
    CHECK_LAUNCH(c);
    return SPH_OK;
}

// Shared by the two list exports: device scratch for counts[n] and lists[n * max], run `launch`, copy back.
template <class Launch>
int exportLists(sph_ctx* c, int32_t* counts, int32_t* lists, int max_neighbors, Launch launch) {
    int *dCounts = nullptr, *dLists = nullptr;
    const size_t lb = sizeof(int) * (size_t)c->n * (size_t)max_neighbors;
    HIP_TRY(c, hipMalloc((void**)&dCounts, sizeof(int) * (size_t)c->n));
    if (lists && lb) {
        hipError_t e = hipMalloc((void**)&dLists, lb);
        if (e != hipSuccess) { (void)hipFree(dCounts); return fail(c, SPH_E_NOMEM, "neighbour list alloc failed"); }
        (void)hipMemsetAsync(dLists, 0xFF, lb, c->stream);   // unused tail entries read back as -1
    }
    launch(dCounts, dLists);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(counts, dCounts, sizeof(int) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && dLists) e = hipMemcpyAsync(lists, dLists, lb, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dCounts);
    if (dLists) (void)hipFree(dLists);
    if (e != hipSuccess) return fail(c, SPH_E_HIP, "neighbour export failed: %s", hipGetErrorString(e));
    return SPH_OK;
}

}  // namespace

extern "C" {

#ifdef SPH_WG_TRACE
// debug builds only: the per-wave trace of the last density (0) / force (1) launch; out = waves x 3 u64
int sph_debug_wave_trace(int kernel, unsigned long long* out, int waves) {
    if (kernel < 0 || kernel > 1 || waves < 0 || waves > kTraceWaves) return SPH_E_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return SPH_E_HIP;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_waveTrace), (size_t)waves * 3 * sizeof(unsigned long long),
                               (size_t)kernel * kTraceWaves * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess
               ? SPH_OK : SPH_E_HIP;
}
#endif

const char* sph_version(void) { return "libsph_hip 0.1 (gfx950, hand-written HIP)"; }

const char* sph_last_error(const sph_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int sph_create(sph_ctx** out, int device_id) {
    if (!out) return SPH_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SPH_E_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return SPH_E_NO_DEVICE;
    sph_ctx* c = new (std::nothrow) sph_ctx();
    if (!c) return SPH_E_NOMEM;
    c->device = device_id;
    if (const char* tl = getenv("SPH_HIP_TIMING")) c->timingLevel = atoi(tl);
    if (const char* rs = getenv("SPH_HIP_ROW_SLACK")) c->rowSlack = std::min(std::max(atoi(rs), 0), 8);
    if (const char* ef = getenv("SPH_HIP_EVENT_FENCE")) c->eventFlags = atoi(ef) ? hipEventDefault : hipEventDisableSystemFence;
    if (const char* ts = getenv("SPH_HIP_TIMING_STRIDE")) c->timingStride = std::max(1, atoi(ts));
    if (const char* hm = getenv("SPH_HIP_HITMASK")) c->useHitMask = atoi(hm) != 0;
    if (const char* xr = getenv("SPH_HIP_XCD_RUN")) c->xcdRunOverride = atoi(xr);
    if (const char* dbg = getenv("SPH_HIP_DEBUG")) c->debug = atoi(dbg) != 0;
    if (const char* um = getenv("SPH_HIP_UNIFORM_MASS")) c->uniformMassOk = atoi(um) != 0;
    if (const char* up = getenv("SPH_HIP_UNIFORM_PHASE")) c->uniformPhaseOk = atoi(up) != 0;
    if (const char* ft = getenv("SPH_HIP_FUSE_TAIL")) c->fuseTailOk = atoi(ft) != 0;
    if (const char* v = getenv("SPH_HIP_DENSITY_LDS_KB")) c->densLdsPad = (unsigned)atoi(v) * 1024u;
    if (const char* v = getenv("SPH_HIP_FORCE_LDS_KB")) c->forceLdsPad = (unsigned)atoi(v) * 1024u;
    if (const char* ld = getenv("SPH_HIP_LDS_DENSITY")) c->ldsDensity = atoi(ld) != 0;
    if (const char* gf = getenv("SPH_HIP_GHOST_FORCE")) c->skipGhostForce = atoi(gf) == 0;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return SPH_E_HIP;
    }
    if (planeScratchFor(device_id, &c->planeScratch) != hipSuccess) {   // device_id is current here
        (void)hipGetLastError();
        sph_destroy(c);
        return SPH_E_HIP;
    }
    bool eventsOk = true;
    for (int r = 0; r < kTimingRing; r++)
        // no system-scope fence (cache write-back) when the event completes: it only brackets kernels on this stream
        for (int k = 0; k < kNumEvents; k++) eventsOk = eventsOk && hipEventCreateWithFlags(&c->ev[r][k], c->eventFlags) == hipSuccess;
    eventsOk = eventsOk && hipEventCreateWithFlags(&c->packEvent, hipEventDisableTiming) == hipSuccess;
    if (!eventsOk) {   // a context without its events would silently time nothing (and hang sph_slab_pack_wait)
        (void)hipGetLastError();
        sph_destroy(c);
        return SPH_E_HIP;
    }
    if (hipHostMalloc((void**)&c->hostInts, 16 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&c->countersBase, 2 * SPH_CTR_COUNT * sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&c->slabCounts, 4 * sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&c->phaseMismatch, sizeof(int)) != hipSuccess ||
        hipMemsetAsync(c->phaseMismatch, 0, sizeof(int), c->stream) != hipSuccess ||
        hipMemsetAsync(c->countersBase, 0, 2 * SPH_CTR_COUNT * sizeof(int), c->stream) != hipSuccess) {
        sph_destroy(c);
        return SPH_E_NOMEM;
    }
    c->counters = c->countersBase;
    if (const char* ls = getenv("SPH_HIP_LDS_STATS"))
        if (atoi(ls) != 0 && (hipMalloc((void**)&c->ldsStats, 4 * sizeof(unsigned long long)) != hipSuccess ||
                              hipMemsetAsync(c->ldsStats, 0, 4 * sizeof(unsigned long long), c->stream) != hipSuccess)) {
            sph_destroy(c);
            return SPH_E_NOMEM;
        }
    *out = c;
    return SPH_OK;
}

int sph_destroy(sph_ctx* c) {
    if (!c) return SPH_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    freeParticles(c);
    freeGrid(c);
    if (c->ldsStats) {   // debug statistics of the LDS density kernel
        unsigned long long h[4] = {0, 0, 0, 0};
        if (hipMemcpy(h, c->ldsStats, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[libsph_hip] k_density_pressure_lds: lane-steps from LDS %llu, from memory %llu (%.3f %%), chunks staged %llu\n",
                    h[0], h[1], 100.0 * (double)h[1] / (double)(h[0] + h[1] + 1), h[2]);
        devFree(c->ldsStats);
    }
    devFree(c->countersBase);
    c->counters = nullptr;
    if (c->hostInts) (void)hipHostFree(c->hostInts);
    c->hostInts = nullptr;
    devFree(c->slabCounts);
    devFree(c->phaseMismatch);
    devFree(c->phaseTableDev);
    for (int r = 0; r < kTimingRing; r++)
        for (int k = 0; k < kNumEvents; k++)
            if (c->ev[r][k]) (void)hipEventDestroy(c->ev[r][k]);
    if (c->packEvent) (void)hipEventDestroy(c->packEvent);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SPH_OK;
}

static int allocCapacity(sph_ctx* c, int n) {
    if (n <= c->capacity) return SPH_OK;
    if (n > kMaxParticles) return fail(c, SPH_E_INVALID, "particle capacity %d exceeds the limit of %d", n, kMaxParticles);
    freeParticles(c);
    int rc = 0;
    for (int k = 0; k < 2 && !rc; k++) {
        rc = devAlloc(c, &c->set[k].posM, (size_t)n + kSlotPad);
        if (!rc) rc = devAlloc(c, &c->set[k].velV, n);
        if (!rc) rc = devAlloc(c, &c->set[k].acc, n);
        if (!rc) rc = devAlloc(c, &c->set[k].rhoP, n);
        if (!rc) rc = devAlloc(c, &c->set[k].rho0, n);
        if (!rc) rc = devAlloc(c, &c->set[k].origId, n);
        if (!rc && c->slab) rc = devAlloc(c, &c->set[k].phase, n);   // phases travel by slot in slab mode
        if (!rc) rc = devAlloc(c, &c->keys[k], n);
        if (!rc) rc = devAlloc(c, &c->vals[k], n);
    }
    if (!rc) rc = devAlloc(c, &c->color, n);
    if (!rc) rc = devAlloc(c, &c->phaseOfId, n);
    if (!rc) rc = devAlloc(c, &c->sid, n);
    if (!rc) rc = devAlloc(c, &c->blockInts, (size_t)blocksFor(n) + 1);
    if (!rc) rc = devAlloc(c, &c->blockCollide, (size_t)blocksFor(n) + 1);
    if (!rc && c->slab) rc = devAlloc(c, &c->wgBand, (size_t)blocksFor(n) + 1);
    if (!rc) rc = devAlloc(c, &c->hitMask, (size_t)n * kMaskWords);
    if (!rc) rc = devAlloc(c, &c->nbrA, (size_t)n);
    if (!rc) rc = devAlloc(c, &c->nbrB, (size_t)n);
    if (!rc) rc = devAlloc(c, &c->pos3, 3 * ((size_t)n + kSlotPad));
    if (!rc) rc = devAlloc(c, &c->aosStage, (size_t)n * SPH_PARTICLE_FLOATS);
    if (rc) { freeParticles(c); return rc; }
    c->capacity = n;
    return SPH_OK;
}

int sph_alloc_particles(sph_ctx* c, int n) {
    if (!c) return SPH_E_INVALID;
    if (n <= 0) return fail(c, SPH_E_INVALID, "particle count must be > 0 (got %d)", n);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int rc = allocCapacity(c, n);
    if (rc) return rc;
    c->n = n;
    c->cur = 0; c->velCur = 0; c->posCur = 0;
    c->gridValid = false; c->maskValid = false;
    c->havePhaseIds = false;
    // a fresh ComputeBuffer is zero-filled; give ids so download before upload is well defined
    HIP_TRY(c, hipMemsetAsync(c->aosStage, 0, (size_t)n * SPH_PARTICLE_BYTES, c->stream));
    hipLaunchKernelGGL(k_import_aos, dim3(blocksFor(n)), dim3(kBlock), 0, c->stream, n, c->aosStage, c->set[0], c->color);
    CHECK_LAUNCH(c);
    return SPH_OK;
}

int sph_upload_particles(sph_ctx* c, const void* aos72, int n) {
    if (!c) return SPH_E_INVALID;
    if (c->slab) return fail(c, SPH_E_STATE, "id-ordered AoS transfer is not available in slab mode (use sph_slab_upload)");
    if (!aos72) return fail(c, SPH_E_INVALID, "null particle array");
    if (n != c->n || n <= 0) return fail(c, SPH_E_INVALID, "upload count %d != allocated %d", n, c->n);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)n * SPH_PARTICLE_BYTES;
    int rc = ensurePinned(c, bytes);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // previous use of the staging buffers
    memcpy(c->pinned, aos72, bytes);
    HIP_TRY(c, hipMemcpyAsync(c->aosStage, c->pinned, bytes, hipMemcpyHostToDevice, c->stream));
    c->cur = 0; c->velCur = 0; c->posCur = 0;
    hipLaunchKernelGGL(k_import_aos, dim3(blocksFor(n)), dim3(kBlock), 0, c->stream, n, c->aosStage, c->set[0], c->color);
    CHECK_LAUNCH(c);
    c->gridValid = false; c->maskValid = false;
    c->staticDirty = true;   // SPH.cs:284-298 would overwrite the static fields on the next Update
    return SPH_OK;
}

int sph_download_particles(sph_ctx* c, void* aos72, int n) {
    if (!c) return SPH_E_INVALID;
    if (c->slab) return fail(c, SPH_E_STATE, "id-ordered AoS transfer is not available in slab mode (use sph_slab_download)");
    if (!aos72) return fail(c, SPH_E_INVALID, "null particle array");
    if (n != c->n || n <= 0) return fail(c, SPH_E_INVALID, "download count %d != allocated %d", n, c->n);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)n * SPH_PARTICLE_BYTES;
    int rc = ensurePinned(c, bytes);
    if (rc) return rc;
    hipLaunchKernelGGL(k_export_aos, dim3(blocksFor(n)), dim3(kBlock), 0, c->stream, n, c->aosStage, curSet(c), c->color);
    CHECK_LAUNCH(c);
    HIP_TRY(c, hipMemcpyAsync(c->pinned, c->aosStage, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(aos72, c->pinned, bytes);
    return SPH_OK;
}

// Render feed: the 72-byte records straight into a DEVICE buffer of the caller (what ParticleRender.shader binds as
// _ParticleBuffer), no host copy, no synchronisation.
int sph_export_particles_device(sph_ctx* c, void* dev_aos72, int n) {
    if (!c) return SPH_E_INVALID;
    if (c->slab) return fail(c, SPH_E_STATE, "id-ordered AoS transfer is not available in slab mode (use sph_slab_download)");
    if (!dev_aos72) return fail(c, SPH_E_INVALID, "null device buffer");
    if (n != c->n || n <= 0) return fail(c, SPH_E_INVALID, "export count %d != allocated %d", n, c->n);
    HIP_TRY(c, hipSetDevice(c->device));
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, dev_aos72) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != c->device) {
        (void)hipGetLastError();
        return fail(c, SPH_E_INVALID, "buffer is not device memory of GPU %d", c->device);
    }
    hipDeviceptr_t base = nullptr; size_t span = 0;
    if (hipMemGetAddressRange(&base, &span, (hipDeviceptr_t)dev_aos72) != hipSuccess ||
        (const char*)dev_aos72 + (size_t)n * SPH_PARTICLE_BYTES > (const char*)base + span || ((uintptr_t)dev_aos72 & 3)) {
        (void)hipGetLastError();
        return fail(c, SPH_E_INVALID, "device buffer is smaller than %d records of %d bytes (or misaligned)", n, SPH_PARTICLE_BYTES);
    }
    hipLaunchKernelGGL(k_export_aos, dim3(blocksFor(n)), dim3(kBlock), 0, c->stream, n, (float*)dev_aos72, curSet(c), c->color);
    CHECK_LAUNCH(c);
    return SPH_OK;
}

int sph_resize_grid(sph_ctx* c, int rx, int ry, int rz, int max_per_cell) {
    if (!c) return SPH_E_INVALID;
    if (rx <= 0 || ry <= 0 || rz <= 0 || max_per_cell <= 0)
        return fail(c, SPH_E_INVALID, "bad grid %d x %d x %d cap %d", rx, ry, rz, max_per_cell);
    const long long cells = (long long)rx * ry * rz;
    // rows are stored in tiles of kTileY x kTileZ (sph_common.h): the table covers whole tiles
    const long long table = (long long)rx * (tiles_along(ry, kTileYLog) * kTileY) * (tiles_along(rz, kTileZLog) * kTileZ);
    if (table > 0x7FFFFFF0ll) return fail(c, SPH_E_INVALID, "grid has %lld cells (table of %lld > 2^31)", cells, table);
    HIP_TRY(c, hipSetDevice(c->device));
    if (table > c->cellCapacity) {
        // grow-only, with headroom: a rocking boundary cube changes the AABB every frame
        // (SPH.cs:176-185 -> 339-345 reallocates each time; here that is a no-op almost always)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        freeGrid(c);
        long long want = table + table / 4;
        if (want > 0x7FFFFFF0ll) want = 0x7FFFFFF0ll;
        const size_t tiles = (size_t)((want + kScanTile - 1) / kScanTile) + 1;
        int rc = devAlloc(c, &c->cumBase, (size_t)want + 16);
        if (!rc) rc = devAlloc(c, &c->cumOutBase, (size_t)want + 16);
        if (!rc) { c->cumT = c->cumBase + 4; c->cumOut = c->cumOutBase + 4; c->cellCum = c->cumOut - 1; }
        if (!rc) rc = devAlloc(c, &c->cumBlocks, tiles);
        if (!rc) rc = devAlloc(c, &c->tileTouched, tiles);
        if (rc) { freeGrid(c); return rc; }
        // invariant of the grid build: the count array and the tile flags are all-zero between builds (k_cum_write
        // restores it); the cumulative table's T[-1] = 0 is never written again
        HIP_TRY(c, hipMemsetAsync(c->cumBase, 0, sizeof(int) * ((size_t)want + 16), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->cumOutBase, 0, sizeof(int) * ((size_t)want + 16), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->tileTouched, 0, tiles, c->stream));
        c->cellCapacity = want;
    }
    c->rx = rx; c->ry = ry; c->rz = rz; c->cap = max_per_cell; c->cells = cells; c->tableCells = table;
    c->gridValid = false; c->maskValid = false;   // fresh GridCounts/GridIndices buffers (SPH.cs:344-345)
    return SPH_OK;
}

int sph_set_params(sph_ctx* c, const sph_params* p) {
    if (!c) return SPH_E_INVALID;
    if (!p) return fail(c, SPH_E_INVALID, "null params");
    if (p->struct_size != sizeof(sph_params))
        return fail(c, SPH_E_INVALID, "sph_params.struct_size %u != %zu", p->struct_size, sizeof(sph_params));
    c->prm = *p;
    c->haveParams = true;
    c->maskValid = false;   // h / grid geometry may have changed
    deriveParams(c);
    return SPH_OK;
}

int sph_set_delta_time(sph_ctx* c, float dt) {
    if (!c) return SPH_E_INVALID;
    c->prm.deltaTime = dt;
    c->dp.dt = dt;
    return SPH_OK;
}

int sph_set_phase_table(sph_ctx* c, const sph_phase* table, int n_phases) {
    if (!c) return SPH_E_INVALID;
    if (!table || n_phases <= 0 || n_phases > 256) return fail(c, SPH_E_INVALID, "bad phase table (n=%d)", n_phases);
    if ((int)c->phaseTable.size() == n_phases && memcmp(c->phaseTable.data(), table, sizeof(sph_phase) * n_phases) == 0)
        return SPH_OK;   // unchanged since last frame: nothing to re-apply
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // slab mode: until sph_apply_phase_table has rewritten the local slots, the table no longer describes them, so
    // the packed neighbour records (mass / viscosity looked up by phase) stay off
    if (c->slab && c->n > 0) c->slabPackedOff = true;
    if ((int)c->phaseTable.size() != n_phases) {
        devFree(c->phaseTableDev);
        int rc = devAlloc(c, &c->phaseTableDev, (size_t)n_phases);
        if (rc) return rc;
    }
    c->phaseTable.assign(table, table + n_phases);
    HIP_TRY(c, hipMemcpy(c->phaseTableDev, table, sizeof(sph_phase) * n_phases, hipMemcpyHostToDevice));
    c->staticDirty = true;
    return SPH_OK;
}

int sph_set_phase_ids(sph_ctx* c, const uint8_t* ids, int n) {
    if (!c) return SPH_E_INVALID;
    if (!ids || n != c->n || n <= 0) return fail(c, SPH_E_INVALID, "phase id count %d != allocated %d", n, c->n);
    const int np = (int)c->phaseTable.size();
    if (np == 0) return fail(c, SPH_E_STATE, "sph_set_phase_table first");
    for (int i = 0; i < n; i++)
        if (ids[i] >= np) return fail(c, SPH_E_INVALID, "phase id %d of particle %d out of range (%d phases)", ids[i], i, np);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->phaseOfId, ids, (size_t)n, hipMemcpyHostToDevice));
    c->havePhaseIds = true;
    c->staticDirty = true;
    return SPH_OK;
}

int sph_apply_phase_table(sph_ctx* c) {
    if (!c) return SPH_E_INVALID;
    if (c->n <= 0 && !c->slab) return fail(c, SPH_E_STATE, "no particles allocated");   // an empty slab is a valid state
    HIP_TRY(c, hipSetDevice(c->device));
    return applyPhaseTable(c);
}

int sph_dispatch(sph_ctx* c, int kernel_id, int count) {
    if (!c) return SPH_E_INVALID;
    if (kernel_id < 0 || kernel_id >= SPH_K_COUNT) return fail(c, SPH_E_INVALID, "bad kernel id %d", kernel_id);
    const bool usesGrid = kernel_id == SPH_K_DENSITY_PRESSURE || kernel_id == SPH_K_FORCE_XSPH;
    int rc = checkReady(c, usesGrid);
    if (rc) return rc;
    const long long expect = kernel_id == SPH_K_CLEAR_GRID ? c->cells : c->n;
    if (count != expect)
        return fail(c, SPH_E_INVALID, "dispatch count %d != %lld (SPH.cs always dispatches the full buffer)", count, expect);
    HIP_TRY(c, hipSetDevice(c->device));
    return dispatchKernel(c, kernel_id);
}

int sph_step_frame(sph_ctx* c, int sub_steps, float time_step) {
    if (!c) return SPH_E_INVALID;
    if (sub_steps < 1) return fail(c, SPH_E_INVALID, "subSteps must be >= 1");
    int rc = checkReady(c, false);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = applyPhaseTable(c))) return rc;                               // SPH.cs:198
    const int slot = c->ringNext;
    if (c->slotPending[slot] && (rc = harvestSlot(c, slot))) return rc;
    // every timingStride-th frame is timed, starting with the first one after sph_get_timings; the events ride on the
    // frame's own launches (LAUNCH_T), so a timed frame enqueues exactly what an untimed one does
    const bool timed = c->timingLevel >= 1 && (c->timingLevel >= 2 || c->framesSinceHarvest % c->timingStride == 0);
    c->framesSinceHarvest++;
    c->slotRecorded[slot] = 0;
    c->tev = timed ? c->ev[slot] : nullptr;
    c->tevMask = &c->slotRecorded[slot];
    c->tevSub = false;
    struct Untime { sph_ctx* c; ~Untime() { c->tev = nullptr; c->tevSub = false; } } untime{c};   // also on the error returns
    sph_set_delta_time(c, time_step / (float)sub_steps);                    // SPH.cs:207-208
    // SPH.cs:201 + 204: CS_ClearGrid is the first step of the grid build here; the state reorder at
    // its end also does the first sub-step's CS_VvHalfStep
    c->dp.n = c->n;
    if ((rc = buildGrid(c, true))) return rc;
    // slab mode, sph_slab_arm_pack: this frame packs its own face messages (see slab_kernels.h)
    const sph_ctx::ArmedPack ap = c->armedPack;
    c->armedPack.armed = false;
    c->bandValid = false;
    const bool split = ap.armed && c->slab && c->n > 0 && c->wgBand;
    int bandLo = 0, bandHi = 0;
    if (ap.armed) {
        if (c->packPending) return fail(c, SPH_E_STATE, "the previous in-frame pack was not collected (sph_slab_pack_wait)");
        HIP_TRY(c, hipMemsetAsync(c->slabCounts, 0, 4 * sizeof(int), c->stream));
        // no left (right) neighbour: left_below = 0 (right_from = global resolution), nothing is near that face
        bandLo = ap.leftBelow > 0 ? ap.leftBelow + ap.margin : 0;
        bandHi = ap.rightFrom < c->rxGlobal ? ap.rightFrom - ap.margin : c->rxGlobal;
    }
    if (split) {
        hipLaunchKernelGGL(k_band_flags, dim3(blocksFor(c->n)), dim3(kBlock), 0, c->stream, c->n, c->sortedKeys, c->dp.rx,
                           c->dp.xoff, c->dp.ownLo, c->dp.ownHi, bandLo, bandHi, c->wgBand, c->slabCounts + 2);
        CHECK_LAUNCH(c);
    }
    // SPH.cs:210-222 with the dead and the purely streaming stages folded together (bit-identical
    // results): CS_Clear is elided (density/pressure/acceleration are fully overwritten by the two
    // neighbour kernels; the force kernel zeroes the acceleration of the particles it skips), and
    // CS_VvFullStep + CS_BoundObs + (after the last sub-step) CS_VelocityDamping are one pass.
    for (int s = 0; s < sub_steps; s++) {                                   // SPH.cs:210-219
        const bool last = s == sub_steps - 1;
        if (s > 0 && (rc = dispatchKernel(c, SPH_K_VV_HALF_STEP))) return rc;
        c->tevSub = last;                                                   // the sub-step stages reported are the last sub-step's
        if ((rc = dispatchKernel(c, SPH_K_DENSITY_PRESSURE))) return rc;
        if (last && split) {
            if ((rc = dispatchKernel(c, SPH_K_FORCE_XSPH, true, last, 1))) return rc;       // band workgroups
            StateSet st = curSet(c);
            st.velV = c->set[1 - c->velCur].velV;                                           // what the force launch wrote
            hipLaunchKernelGGL(k_slab_pack_final, dim3((c->n + kBlock * kKeepItems - 1) / (kBlock * kKeepItems)), dim3(kBlock), 0,
                               c->stream, c->dp, st, c->ownedRef, (const uint8_t*)c->wgBand, 1, ap.leftBelow, ap.rightFrom,
                               (SlabRecord*)ap.devL, ap.capL, (SlabRecord*)ap.devR, ap.capR, c->slabCounts);
            CHECK_LAUNCH(c);
        }
        if (last && ap.armed) {   // counts + the frame's plane shift to the host; the host may wait for this point only
            HIP_TRY(c, hipMemcpyAsync(c->hostInts + 10, c->slabCounts, 3 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->hostInts + 8, c->counters + SPH_CTR_MAX_PLANE_SHIFT, sizeof(int), hipMemcpyDeviceToHost,
                                      c->stream));
            HIP_TRY(c, hipEventRecord(c->packEvent, c->stream));
            c->packPending = true;
            c->packCapL = ap.capL; c->packCapR = ap.capR;
            c->bandValid = split;
            c->bandSendLo = ap.leftBelow; c->bandSendHi = ap.rightFrom;
        }
        if ((rc = dispatchKernel(c, SPH_K_FORCE_XSPH, true, last, last && split ? 0 : -1))) return rc;
        if (!c->tailFusedLast && (rc = dispatchKernel(c, SPH_K_VV_FULL_STEP, true, last))) return rc;
    }
    if (timed) {
        c->slotPending[slot] = true;
        c->ringNext = (slot + 1) % kTimingRing;
    }
    return SPH_OK;
}

int sph_synchronize(sph_ctx* c) {
    if (!c) return SPH_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_reset_counter(sph_ctx* c, int which) {
    if (!c) return SPH_E_INVALID;
    if (which < 0 || which >= SPH_CTR_COUNT) return fail(c, SPH_E_INVALID, "bad counter %d", which);
    HIP_TRY(c, hipSetDevice(c->device));
    if (which == SPH_CTR_COLLISIONS) {   // SPH.cs:190 does this every frame: no launch, the next boundary pass overwrites
        c->collisionsResetPending = true;
        c->collideFoldBlocks = 0;        // counts not folded yet are void too
        return SPH_OK;
    }
    HIP_TRY(c, hipMemsetAsync(c->counters + which, 0, sizeof(int), c->stream));
    if (which == SPH_CTR_MAX_PLANE_SHIFT) c->cachedShift = -1;
    return SPH_OK;
}

int sph_read_counter(sph_ctx* c, int which, int* value) {
    if (!c) return SPH_E_INVALID;
    if (which < 0 || which >= SPH_CTR_COUNT || !value) return fail(c, SPH_E_INVALID, "bad counter %d", which);
    HIP_TRY(c, hipSetDevice(c->device));
    if (which == SPH_CTR_MAX_PLANE_SHIFT && c->cachedShift >= 0) { *value = c->cachedShift; return SPH_OK; }
    if (which == SPH_CTR_BAND_MISSES && c->cachedMisses >= 0) { *value = c->cachedMisses; return SPH_OK; }
    if (which == SPH_CTR_COLLISIONS) {
        if (int rcFold = foldCollisions(c)) return rcFold;
        if (c->collisionsResetPending) { *value = 0; return SPH_OK; }
    }
    if (int rcRead = readInts(c, (which == SPH_CTR_COLLISIONS ? c->countersBase : c->counters) + which, 1, value)) return rcRead;
    return SPH_OK;
}

int sph_download_grid(sph_ctx* c, int32_t* cell_start, int32_t* cell_end, int32_t* sorted_ids,
                      uint32_t* cell_of_particle) {
    if (!c) return SPH_E_INVALID;
    if (!c->gridValid) return fail(c, SPH_E_STATE, "grid not built");
    if (c->slab && cell_of_particle) return fail(c, SPH_E_STATE, "cell_of_particle is indexed by id: not available in slab mode");
    HIP_TRY(c, hipSetDevice(c->device));
    // Parity export in the REFERENCE's cell order (compute:127): bins are stored in row tiles on the
    // device, so the table and the id sequence are re-linearised here on the host (export only).
    const size_t nb = sizeof(int) * (size_t)c->n;
    deriveParams(c);
    c->dp.n = c->n;
    std::vector<int> cum, ids;
    if (cell_start || cell_end || sorted_ids) {
        cum.resize((size_t)c->tableCells + 1);
        ids.resize((size_t)c->n);
        HIP_TRY(c, hipMemcpyAsync(cum.data(), c->cellCum, sizeof(int) * cum.size(), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(ids.data(), c->set[c->cur].origId, nb, hipMemcpyDeviceToHost, c->stream));
    }
    if (cell_of_particle) {
        hipLaunchKernelGGL(k_scatter_by_id, dim3(blocksFor(c->n)), dim3(kBlock), 0, c->stream, c->dp, c->sortedKeys,
                           c->set[c->cur].origId, c->sid);
        CHECK_LAUNCH(c);
        HIP_TRY(c, hipMemcpyAsync(cell_of_particle, c->sid, nb, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!cum.empty()) {
        int run = 0;
        size_t cell = 0;
        for (int z = 0; z < c->rz; z++)
            for (int y = 0; y < c->ry; y++) {
                const size_t base = (size_t)c->rx * row_index(c->dp.ryT, y, z);
                for (int x = 0; x < c->rx; x++, cell++) {
                    const int a = cum[base + x], e = cum[base + x + 1];
                    if (cell_start) cell_start[cell] = run;
                    if (sorted_ids) std::copy(ids.begin() + a, ids.begin() + e, sorted_ids + run);
                    run += e - a;
                    if (cell_end) cell_end[cell] = run;
                }
            }
    }
    return SPH_OK;
}

int sph_download_neighbors(sph_ctx* c, int32_t* counts, int32_t* lists, int max_neighbors) {
    if (!c) return SPH_E_INVALID;
    if (!counts || max_neighbors < 0) return fail(c, SPH_E_INVALID, "bad neighbour export arguments");
    int rc = checkReady(c, true);
    if (rc) return rc;
    if (c->slab) return fail(c, SPH_E_STATE, "neighbour export is indexed by id: not available in slab mode");
    HIP_TRY(c, hipSetDevice(c->device));
    StateSet s = curSet(c);
    return exportLists(c, counts, lists, max_neighbors, [&](int* dCounts, int* dLists) {
        hipLaunchKernelGGL(k_export_neighbors, dim3(blocksFor(c->n)), dim3(kBlock), 0, c->stream, c->dp, s.posM, s.origId,
                           c->cellCum, c->counters, dCounts, dLists, max_neighbors);
    });
}

int sph_download_hit_lists(sph_ctx* c, int32_t* counts, int32_t* lists, int max_neighbors, int* mask_state) {
    if (!c) return SPH_E_INVALID;
    if (!counts || max_neighbors < 0 || !mask_state) return fail(c, SPH_E_INVALID, "bad hit list export arguments");
    int rc = checkReady(c, true);
    if (rc) return rc;
    if (c->slab) return fail(c, SPH_E_STATE, "hit list export is indexed by id: not available in slab mode");
    HIP_TRY(c, hipSetDevice(c->device));
    *mask_state = 0;
    if (!c->maskValid || !c->hitMask) return SPH_OK;          // no fresh mask: CS_ForceXSPH would take the scan kernel
    int overflow = 0;
    if (int rcRead = readInts(c, c->counters + SPH_CTR_BIN_OVERFLOW, 1, &overflow)) return rcRead;
    if (overflow) { *mask_state = 2; return SPH_OK; }          // a bin overflowed: nothing was recorded, every wave scans
    *mask_state = 1;
    StateSet s = curSet(c);
    return exportLists(c, counts, lists, max_neighbors, [&](int* dCounts, int* dLists) {
        hipLaunchKernelGGL(k_export_hit_lists, dim3(blocksFor(c->n)), dim3(kBlock), 0, c->stream, c->dp, (const float4*)s.posM,
                           (const uint32_t*)s.origId, (const int*)c->cellCum, (const uint32_t*)c->hitMask, c->capacity, dCounts,
                           dLists, max_neighbors);
    });
}

int sph_upload_particles_in_place(sph_ctx* c, const void* aos72, int n) {
    if (!c) return SPH_E_INVALID;
    if (c->slab) return fail(c, SPH_E_STATE, "id-ordered AoS transfer is not available in slab mode (use sph_slab_upload)");
    if (!aos72) return fail(c, SPH_E_INVALID, "null particle array");
    if (n != c->n || n <= 0) return fail(c, SPH_E_INVALID, "upload count %d != allocated %d", n, c->n);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)n * SPH_PARTICLE_BYTES;
    int rc = ensurePinned(c, bytes);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // previous use of the staging buffers
    memcpy(c->pinned, aos72, bytes);
    HIP_TRY(c, hipMemcpyAsync(c->aosStage, c->pinned, bytes, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_update_aos_in_place, dim3(blocksFor(n)), dim3(kBlock), 0, c->stream, n, (const float*)c->aosStage, curSet(c),
                       c->color);
    CHECK_LAUNCH(c);
    // positions, densities and velocities changed under everything derived from them; the grid is deliberately kept
    c->maskValid = false; c->packedValid = false; c->pos3Valid = false;
    c->staticDirty = true;   // as after sph_upload_particles: SPH.cs:284-298 rewrites the static fields on the next Update
    return SPH_OK;
}

int sph_get_timings(sph_ctx* c, sph_timings* out) {
    if (!c) return SPH_E_INVALID;
    if (!out) return fail(c, SPH_E_INVALID, "null timings");
    HIP_TRY(c, hipSetDevice(c->device));
    for (int r = 0; r < kTimingRing; r++)
        if (c->slotPending[r]) {
            int rc = harvestSlot(c, r);
            if (rc) return rc;
        }
    if (c->framesTimed == 0) {
        c->framesSinceHarvest = 0;
        return fail(c, SPH_E_STATE, "no timed sph_step_frame has run since the last sph_get_timings");
    }
    const double inv = 1.0 / c->framesTimed;
    out->build_grid_ms = (float)(c->stageSum[0] * inv);
    out->half_step_ms = (float)(c->stageSum[1] * inv);
    out->density_ms = (float)(c->stageSum[2] * inv);     // includes CS_Clear
    out->force_ms = (float)(c->stageSum[3] * inv);
    out->full_step_ms = (float)(c->stageSum[4] * inv);
    out->bound_obs_ms = (float)(c->stageSum[5] * inv);
    out->damping_ms = (float)(c->stageSum[6] * inv);
    out->frame_ms = (float)(c->stageSum[7] * inv);
    out->frames = c->framesTimed;
    for (int k = 0; k < kNumEvents; k++) c->stageSum[k] = 0;
    c->framesTimed = 0;
    c->framesSinceHarvest = 0;
    return SPH_OK;
}

int sph_device_pointer(sph_ctx* c, const char* name, void** ptr, uint64_t* bytes) {
    if (!c) return SPH_E_INVALID;
    if (!name || !ptr) return fail(c, SPH_E_INVALID, "null argument");
    StateSet s = curSet(c);
    const uint64_t n = (uint64_t)c->n;
    uint64_t b = 0;
    void* p = nullptr;
    if (!strcmp(name, "posM")) { p = s.posM; b = n * 16; }
    else if (!strcmp(name, "velV")) { p = s.velV; b = n * 16; }
    else if (!strcmp(name, "acc")) { p = s.acc; b = n * 16; }
    else if (!strcmp(name, "rhoP")) { p = s.rhoP; b = n * 8; }
    else if (!strcmp(name, "rho0")) { p = s.rho0; b = n * 4; }
    else if (!strcmp(name, "origId")) { p = s.origId; b = n * 4; }
    else return fail(c, SPH_E_INVALID, "unknown buffer '%s'", name);
    *ptr = p;
    if (bytes) *bytes = b;
    return SPH_OK;
}

int sph_get_stream(sph_ctx* c, void** stream) {
    if (!c || !stream) return SPH_E_INVALID;
    *stream = (void*)c->stream;
    return SPH_OK;
}

// ---- slab decomposition -------------------------------------------------------------------------------
int sph_slab_configure(sph_ctx* c, int rx_global, int x_offset, int own_lo, int own_hi) {
    if (!c) return SPH_E_INVALID;
    if (rx_global <= 0) { c->slab = false; c->gridValid = false; c->maskValid = false; return SPH_OK; }
    if (x_offset < 0 || own_lo < x_offset || own_hi < own_lo || own_hi > rx_global)
        return fail(c, SPH_E_INVALID, "bad slab: global %d, offset %d, own [%d,%d)", rx_global, x_offset, own_lo, own_hi);
    c->slab = true;
    c->rxGlobal = rx_global; c->xOffset = x_offset; c->ownLoG = own_lo; c->ownHiG = own_hi;
    c->gridValid = false; c->maskValid = false;
    return SPH_OK;
}

int sph_slab_reserve(sph_ctx* c, int capacity) {
    if (!c) return SPH_E_INVALID;
    if (capacity <= 0) return fail(c, SPH_E_INVALID, "capacity must be > 0");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (capacity > c->capacity) {
        if (c->n > 0) return fail(c, SPH_E_STATE, "sph_slab_reserve cannot grow a populated slab (reserve before upload)");
        int rc = allocCapacity(c, capacity);
        if (rc) return rc;
        c->n = 0; c->cur = 0; c->velCur = 0; c->posCur = 0; c->gridValid = false; c->maskValid = false;
    }
    return SPH_OK;
}

static PhaseCheck phaseCheck(sph_ctx* c) {
    return PhaseCheck{c->phaseTable.empty() ? (const sph_phase*)nullptr : (const sph_phase*)c->phaseTableDev,
                      (int)c->phaseTable.size(), c->phaseMismatch};
}

static int ensureRecStage(sph_ctx* c, size_t n) {
    if (n <= c->recStageCap) return SPH_OK;
    devFree(c->recStage);
    c->recStageCap = 0;
    int rc = devAlloc(c, &c->recStage, n);
    if (rc) return rc;
    c->recStageCap = n;
    return SPH_OK;
}

int sph_slab_upload(sph_ctx* c, const sph_slab_record* records, int n) {
    if (!c) return SPH_E_INVALID;
    if (!c->slab) return fail(c, SPH_E_STATE, "sph_slab_configure first");
    if (n < 0 || (n > 0 && !records)) return fail(c, SPH_E_INVALID, "bad records");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n > c->capacity) {
        c->n = 0;
        int rc = allocCapacity(c, n);
        if (rc) return rc;
    }
    c->cur = 0; c->velCur = 0; c->posCur = 0; c->gridValid = false; c->maskValid = false;
    c->n = n;
    c->ownedRef.keys = nullptr;       // everything uploaded is owned
    c->filter.active = 0;             // a rebuild that was waiting for its grid build is void
    c->bandValid = false;
    c->armedPack.armed = false;
    c->slabPackedOff = false;
    HIP_TRY(c, hipMemsetAsync(c->phaseMismatch, 0, sizeof(int), c->stream));
    if (n == 0) return SPH_OK;
    int rc = ensureRecStage(c, (size_t)n);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpy(c->recStage, records, sizeof(SlabRecord) * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_slab_import, dim3(blocksFor(n)), dim3(kBlock), 0, c->stream, n, c->recStage, c->set[0],
                       phaseCheck(c));
    CHECK_LAUNCH(c);
    return SPH_OK;
}

int sph_slab_download(sph_ctx* c, sph_slab_record* records, int max_records, int owned_only, int* n_out) {
    if (!c) return SPH_E_INVALID;
    if (!c->slab) return fail(c, SPH_E_STATE, "sph_slab_configure first");
    if (!n_out || max_records < 0 || (max_records > 0 && !records)) return fail(c, SPH_E_INVALID, "bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    *n_out = 0;
    // between sph_slab_rebuild and the grid build the slots from filter.nOld on are records just received
    // (copies of particles their senders still hold): not part of the local set yet
    const int nLocal = c->filter.active ? c->filter.nOld : c->n;
    if (nLocal == 0) return SPH_OK;
    if (c->packPending) return fail(c, SPH_E_STATE, "collect the in-frame pack first (sph_slab_pack_wait)");
    if (c->haveParams) deriveParams(c);
    c->dp.n = nLocal;
    int rc = ensureRecStage(c, (size_t)nLocal);
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(c->slabCounts, 0, 4 * sizeof(int), c->stream));
    hipLaunchKernelGGL(k_slab_export, dim3(blocksFor(nLocal)), dim3(kBlock), 0, c->stream, c->dp, curSet(c),
                       c->ownedRef, owned_only, c->recStage,
                       nLocal, c->slabCounts);
    CHECK_LAUNCH(c);
    int cnt = 0;
    if (int rcRead = readInts(c, c->slabCounts, 1, &cnt)) return rcRead;
    if (cnt > max_records) return fail(c, SPH_E_INVALID, "%d records do not fit %d", cnt, max_records);
    if (cnt > 0) HIP_TRY(c, hipMemcpy(records, c->recStage, sizeof(SlabRecord) * (size_t)cnt, hipMemcpyDeviceToHost));
    *n_out = cnt;
    return SPH_OK;
}

int sph_slab_pack(sph_ctx* c, int left_below, int right_from, void* dev_left, int cap_left, void* dev_right,
                  int cap_right, int* n_left, int* n_right) {
    if (!c) return SPH_E_INVALID;
    if (!c->slab || !c->haveParams) return fail(c, SPH_E_STATE, "sph_slab_configure and sph_set_params first");
    if (!n_left || !n_right || cap_left < 0 || cap_right < 0 || (cap_left > 0 && !dev_left) || (cap_right > 0 && !dev_right))
        return fail(c, SPH_E_INVALID, "bad message buffers");
    HIP_TRY(c, hipSetDevice(c->device));
    deriveParams(c);
    c->dp.n = c->n;
    *n_left = *n_right = 0;
    if (c->n == 0) return SPH_OK;
    HIP_TRY(c, hipMemsetAsync(c->slabCounts, 0, 4 * sizeof(int), c->stream));
    hipLaunchKernelGGL(k_slab_pack, dim3((c->n + kBlock * kKeepItems - 1) / (kBlock * kKeepItems)), dim3(kBlock), 0, c->stream, c->dp, curSet(c),
                       c->ownedRef, left_below, right_from,
                       (SlabRecord*)dev_left, cap_left, (SlabRecord*)dev_right, cap_right, c->slabCounts);
    CHECK_LAUNCH(c);
    // the two message sizes and -- same synchronisation -- the frame's plane shift, which the driver asks for next
    int cnt[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(c->hostInts + 8, c->counters + SPH_CTR_MAX_PLANE_SHIFT, sizeof(int), hipMemcpyDeviceToHost,
                              c->stream));
    if (int rcRead = readInts(c, c->slabCounts, 2, cnt)) return rcRead;
    c->cachedShift = c->hostInts[8];
    *n_left = cnt[0];
    *n_right = cnt[1];
    if (cnt[0] > cap_left || cnt[1] > cap_right)
        return fail(c, SPH_E_NOMEM, "slab message overflow: %d/%d left, %d/%d right records", cnt[0], cap_left, cnt[1], cap_right);
    return SPH_OK;
}

int sph_slab_rebuild(sph_ctx* c, int keep_lo, int keep_hi, const void* dev_from_left, int n_from_left,
                     const void* dev_from_right, int n_from_right, int* n_local) {
    if (!c) return SPH_E_INVALID;
    if (!c->slab || !c->haveParams) return fail(c, SPH_E_STATE, "sph_slab_configure and sph_set_params first");
    if (!n_local || n_from_left < 0 || n_from_right < 0) return fail(c, SPH_E_INVALID, "bad arguments");
    if (c->capacity <= 0) return fail(c, SPH_E_STATE, "sph_slab_reserve first");
    HIP_TRY(c, hipSetDevice(c->device));
    deriveParams(c);
    c->dp.n = c->n;
    // No copy of the survivors: the received records are appended behind the current particles and the
    // next grid build (which follows in the same frame) drops what does not belong -- stale ghosts,
    // emigrants, records outside [keep_lo, keep_hi) -- while it sorts (SlabFilter).
    const int nOld = c->n;
    if ((long long)nOld + n_from_left + n_from_right > c->capacity)
        return fail(c, SPH_E_NOMEM, "slab population %d + %d + %d received exceeds the reserved capacity %d", nOld,
                    n_from_left, n_from_right, c->capacity);
    StateSet cur = curSet(c);
    if (n_from_left > 0)
        hipLaunchKernelGGL(k_slab_append_at, dim3(blocksFor(n_from_left)), dim3(kBlock), 0, c->stream,
                           (const SlabRecord*)dev_from_left, n_from_left, cur, nOld, phaseCheck(c));
    if (n_from_right > 0)
        hipLaunchKernelGGL(k_slab_append_at, dim3(blocksFor(n_from_right)), dim3(kBlock), 0, c->stream,
                           (const SlabRecord*)dev_from_right, n_from_right, cur, nOld + n_from_left, phaseCheck(c));
    CHECK_LAUNCH(c);
    c->filter = SlabFilter{1, keep_lo, keep_hi, nOld, c->ownedRef,
                           c->bandValid ? (const uint8_t*)c->wgBand : (const uint8_t*)nullptr, c->bandSendLo, c->bandSendHi};
    c->bandValid = false;
    c->n = nOld + n_from_left + n_from_right;
    c->gridValid = false; c->maskValid = false;
    *n_local = c->n;       // upper bound until the grid build; sph_particle_count afterwards
    return SPH_OK;
}

int sph_slab_arm_pack(sph_ctx* c, int left_below, int right_from, int band_margin, void* dev_left, int cap_left,
                      void* dev_right, int cap_right) {
    if (!c) return SPH_E_INVALID;
    if (!c->slab) return fail(c, SPH_E_STATE, "sph_slab_configure first");
    if (band_margin < 0 || cap_left < 0 || cap_right < 0 || (cap_left > 0 && !dev_left) || (cap_right > 0 && !dev_right))
        return fail(c, SPH_E_INVALID, "bad message buffers");
    if (c->packPending) return fail(c, SPH_E_STATE, "the previous in-frame pack was not collected (sph_slab_pack_wait)");
    c->armedPack.armed = true;
    c->armedPack.leftBelow = left_below; c->armedPack.rightFrom = right_from; c->armedPack.margin = band_margin;
    c->armedPack.devL = dev_left; c->armedPack.capL = cap_left;
    c->armedPack.devR = dev_right; c->armedPack.capR = cap_right;
    return SPH_OK;
}

int sph_slab_pack_wait(sph_ctx* c, int* n_left, int* n_right, int* n_band_workgroups) {
    if (!c) return SPH_E_INVALID;
    if (!n_left || !n_right) return fail(c, SPH_E_INVALID, "bad arguments");
    if (!c->packPending) return fail(c, SPH_E_STATE, "no in-frame pack pending (sph_slab_arm_pack + sph_step_frame first)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->packEvent));
    c->packPending = false;
    c->cachedShift = c->hostInts[8];
    *n_left = c->hostInts[10];
    *n_right = c->hostInts[11];
    if (n_band_workgroups) *n_band_workgroups = c->hostInts[12];
    if (*n_left > c->packCapL || *n_right > c->packCapR)
        return fail(c, SPH_E_NOMEM, "slab message overflow: %d/%d left, %d/%d right records", *n_left, c->packCapL, *n_right,
                    c->packCapR);
    return SPH_OK;
}

int sph_slab_plane_counts(sph_ctx* c, int32_t* counts, int n_planes) {
    if (!c) return SPH_E_INVALID;
    if (!c->slab) return fail(c, SPH_E_STATE, "sph_slab_configure first");
    if (!counts || n_planes != c->rx) return fail(c, SPH_E_INVALID, "expected %d plane counts (the local x resolution), got %d", c->rx, n_planes);
    if (!c->gridValid) return fail(c, SPH_E_STATE, "plane counts need the grid of the last frame (call before sph_slab_rebuild)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_planes > c->planeCountsCap) {
        devFree(c->planeCounts);
        c->planeCountsCap = 0;
        int rc = devAlloc(c, &c->planeCounts, (size_t)n_planes);
        if (rc) return rc;
        c->planeCountsCap = n_planes;
    }
    HIP_TRY(c, hipMemsetAsync(c->planeCounts, 0, sizeof(int) * (size_t)n_planes, c->stream));
    const long long rows = c->tableCells / c->rx;
    hipLaunchKernelGGL(k_plane_counts, dim3((unsigned)((rows + kPlaneRows - 1) / kPlaneRows)), dim3(kBlock), 0, c->stream,
                       (const int*)c->cellCum, c->rx, rows, c->planeCounts);
    CHECK_LAUNCH(c);
    HIP_TRY(c, hipMemcpyAsync(counts, c->planeCounts, sizeof(int) * (size_t)n_planes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

// The variant of sph_slab_plane_counts that allocates nothing: the counters are this GPU's copy of the static
// `scratch` array (c->planeScratch, resolved in sph_create).  The array keeps the last call's contents, so the part in
// use is cleared first.
int sph_slab_plane_counts_noalloc(sph_ctx* c, int32_t* counts, int n_planes) {
    if (!c) return SPH_E_INVALID;
    if (n_planes > SPH_PLANE_COUNTS_MAX)
        return fail(c, SPH_E_INVALID, "%d planes exceed the static scratch of %d planes (use sph_slab_plane_counts)", n_planes,
                    SPH_PLANE_COUNTS_MAX);
    if (!counts || n_planes <= 0 || n_planes != c->rx)
        return fail(c, SPH_E_INVALID, "expected %d plane counts (the local x resolution), got %d", c->rx, n_planes);
    if (!c->gridValid) return fail(c, SPH_E_STATE, "plane counts need the grid of the last frame (call before sph_slab_rebuild)");
    if (!c->planeScratch) return fail(c, SPH_E_STATE, "static scratch not resolved on GPU %d", c->device);
    HIP_TRY(c, hipSetDevice(c->device));
    int* const ptr = c->planeScratch;
    // every ctx on this GPU shares the array: one call at a time per device (the call synchronises before it returns,
    // so holding the lock over it costs nothing)
    if (c->device < 0 || c->device >= kMaxDevices) return fail(c, SPH_E_INVALID, "device %d out of range", c->device);
    std::lock_guard<std::mutex> lock(g_scratchUse[c->device]);
    HIP_TRY(c, hipMemsetAsync(ptr, 0, sizeof(int) * (size_t)n_planes, c->stream));
    const long long rows = c->tableCells / c->rx;
    hipLaunchKernelGGL(k_plane_counts, dim3((unsigned)((rows + kPlaneRows - 1) / kPlaneRows)), dim3(kBlock), 0, c->stream,
                       (const int*)c->cellCum, c->rx, rows, ptr);
    CHECK_LAUNCH(c);
    HIP_TRY(c, hipMemcpyAsync(counts, ptr, sizeof(int) * (size_t)n_planes, hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_particle_count(sph_ctx* c, int* n) {
    if (!c || !n) return SPH_E_INVALID;
    *n = c->n;
    return SPH_OK;
}

}  // extern "C"
