// physics_kernels.h -- per-particle SPH stages on the slot-ordered SoA state (gfx950).
// Each kernel states the reference kernel it stands in for (Assets/SPHCompute.compute).
#pragma once
#include <type_traits>

#include "grid_kernels.h"

namespace sph {

// dot(a,a) exactly as the oracle evaluates it (rule O5): (x*x + y*y) + z*z, every operation
// rounded, no FMA -- the in-range test rSq < h2 decides neighbour-list membership and has to
// agree bit for bit.  The TU is built with -ffp-contract=off; the pragma pins it locally too.
__device__ __forceinline__ float dist2_exact(float rx, float ry, float rz) {
#pragma clang fp contract(off)
    float a = rx * rx;
    float b = ry * ry;
    float c = rz * rz;
    return (a + b) + c;
}

// CS_Clear (compute:87-98)
__global__ __launch_bounds__(kBlock) void k_clear(int n, float4* __restrict__ acc, float2* __restrict__ rhoP,
                                                  const float* __restrict__ rho0) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    rhoP[s] = make_float2(rho0[s], 0.f);
}

// ---- the neighbour kernels -----------------------------------------------------------------------------
// The three x-neighbour cells of one (y, z) row are adjacent in the linear cell index
// (compute:127), hence adjacent in the sorted order: their bins are ONE slot range, read from the
// cumulative table with two loads.  Visiting order is unchanged: z, y, then x-1, x, x+1, then id.
struct RowRange { int a, e; };


constexpr int kMaskWords = 24;   // hit-mask words per particle (rows start on word boundaries); ~9-11 are in use at fluid densities
constexpr int kMaskStage = 16;   // of which the density kernel stages this many in LDS (the rest, rare, go straight to memory)

// XCD-aware workgroup -> tile mapping.  Workgroups are dealt round-robin over the 8 XCDs, each with
// its own 4 MiB L2; consecutive tiles of the slot order share most of their neighbour rows, so runs
// of `run` CONSECUTIVE tiles are mapped to one XCD instead of every eighth tile (MI355X guide, T1).
// Runs are dealt round-robin so that regions of different cost per particle (dense / sparse, phase
// A / phase B) spread evenly; the host picks the run length (DevParams::xcdRun): long enough that an
// XCD sweeps several whole cell rows and re-uses in its own L2 the rows it fetched for the row
// before, short enough that every XCD gets many runs.  Bijective for any grid size; affects speed
// only, never results.
__device__ __forceinline__ int xcd_tile(const DevParams& P, int bid, int nwg) {
    const int run = P.xcdRun, span = 8 * run;
    const int full = nwg / span * span;
    if (bid >= full) return bid;                     // ragged tail: identity
    const int g = bid / span, w = bid - g * span;
    return g * span + (w & 7) * run + (w >> 3);
}

__device__ __forceinline__ int3 search_cell(const DevParams& P, float px, float py, float pz) {
    int3 c = cell_unclamped(P, px, py, pz);
    c.x = min(max(c.x, -2), P.rxg + 1);
    c.y = min(max(c.y, -2), P.ry + 1);
    c.z = min(max(c.z, -2), P.rz + 1);
    return c;
}

__device__ __forceinline__ RowRange row_range(const DevParams& P, const int* __restrict__ cellCum, int3 c, int dy,
                                              int dz, bool live) {
    const int y = c.y + dy, z = c.z + dz;
    // global x range (compute:166), then the part of it this slab holds
    const int xlo = max(max(c.x - 1, 0) - P.xoff, 0), xhi = min(min(c.x + 1, P.rxg - 1) - P.xoff, P.rx - 1);
    RowRange r{0, 0};
    if (live && y >= 0 && y < P.ry && z >= 0 && z < P.rz && xlo <= xhi) {
        const int base = P.rx * row_index(P.ryT, y, z);
        r.a = cellCum[base + xlo];
        r.e = cellCum[base + xhi + 1];
    }
    return r;
}

// The t-th candidate range of a particle, in visiting order.  Normal case (no bin overflowed):
// 9 ranges, one per (dy, dz) row, each covering the three x-neighbour cells.  If some bin
// overflowed at the last grid build (capped != 0): 27 ranges, one per cell, each cut to the
// reference's bin capacity (rule O2: the first `cap` ids of a bin are visible).
__device__ __forceinline__ RowRange range_at(const DevParams& P, const int* __restrict__ cellCum, int3 c, int t,
                                             bool capped, bool live) {
    if (!capped) return row_range(P, cellCum, c, t % 3 - 1, t / 3 - 1, live && t < 9);
    RowRange r{0, 0};
    const int xg = c.x + (t % 3 - 1), y = c.y + ((t / 3) % 3 - 1), z = c.z + (t / 9 - 1);
    const int x = xg - P.xoff;
    if (live && t < 27 && xg >= 0 && xg < P.rxg && x >= 0 && x < P.rx && y >= 0 && y < P.ry && z >= 0 && z < P.rz) {
        const int ci = x + P.rx * row_index(P.ryT, y, z);
        r.a = cellCum[ci];
        r.e = min(cellCum[ci + 1], r.a + P.cap);
    }
    return r;
}

// What the force kernel needs from a NEIGHBOUR is ten floats -- position, velocity, density, pressure,
// mass, viscosity -- i.e. three scattered gathers per in-range pair, and that kernel is bound by the
// texture addresser's gather rate (~32 clk per wave-wide scattered load, measured), not by arithmetic
// or HBM.  Mass and viscosity are per-PHASE constants in the reference (SPH.cs:284-298 rewrites them
// from the spawn box of every particle each frame), so when the caller has given a phase table of at
// most four rows the density kernel leaves two 16-byte records per particle,
//     a = (x, y, z, density | phase bit 0 in the sign)    b = (vx, vy, vz, pressure | phase bit 1 in the sign)
// (density > 0 and Tait pressure >= 0 because density >= rest density, so the signs are free), and the
// force kernel does two gathers per pair and looks mass / viscosity up by phase: same bits, 1/3 fewer gathers.
struct PackedNbr {
    float4* a;                       // null = not in use
    float4* b;
    const float4* velV;
    const uint32_t* origId;
    const uint8_t* phaseOfId;        // phases by particle id ...
    const uint8_t* phaseOfSlot;      // ... or by slot (slab mode); one of the two
};
struct PhaseConsts { float m0, m1, m2, m3, v0, v1, v2, v3; };   // named, not arrays: selects must stay v_cndmask, not indexed loads

// CS_DensityPressure (compute:142-193), fast path.  Same operations in the same order as the
// per-cell kernel below, so the density is bit-identical to the oracle's.
// UNIFORM_MASS: every phase has the same particle mass (true of all the reference's scenes: 0.2) and every
// particle's mass is its phase's (the conditions of the packed neighbour records), so a candidate is its position
// only, read from `pos3` -- 12 bytes per slot, written by the kernel that did the half step -- instead of the
// 16-byte posM entry: four candidates are 48 contiguous bytes per lane instead of 64, and the kernel is bound by
// the lines its gathers touch.  (Loading 12 of the 16 bytes of posM entries gains nothing: same lines.)  In slab
// mode the host takes this path only after it has read the sticky phaseMismatch flag as clean for this frame's
// records (with the survivor count of the grid build).
struct __attribute__((packed, aligned(4))) Float4U { float x, y, z, w; };   // 16-byte load from a 4-byte aligned address
template <bool UNIFORM_MASS>
__device__ __forceinline__ void density_rows_body(const DevParams& P, const float4* __restrict__ posM,
                                                  const float* __restrict__ rho0,
                                                  float2* __restrict__ rhoP,
                                                  const int* __restrict__ cellCum,
                                                  const int* __restrict__ counters,
                                                  uint32_t* __restrict__ hitMask, int maskStride,
                                                  const PackedNbr& nbr,
                                                  const uint32_t* __restrict__ slabKeys,
                                                  int* __restrict__ slabShift,
                                                  const float* __restrict__ pos3, float uniformMass,
                                                  uint32_t (*maskStage)[kBlock]) {
    const bool capped = counters[SPH_CTR_BIN_OVERFLOW] != 0;   // wave-uniform
    const int nRanges = capped ? 27 : 9;
    const int tile = xcd_tile(P, blockIdx.x, gridDim.x);
    const int s = tile * kBlock + threadIdx.x;
    const bool valid = s < P.n;
    const float4 p = posM[valid ? s : 0];
    if (slabKeys) {
        // slab mode: how many planes has the particle moved since it was binned?  The halo width the
        // driver needs follows from the maximum (SPH_CTR_MAX_PLANE_SHIFT).  Every workgroup stores its
        // own maximum; k_slab_shift_reduce folds them.  (One atomic per WAVE on the counter -- even
        // behind a check -- serialises 32 k requests on one address: +70 us at 2 M particles.)
        __shared__ int waveShift[kBlock / 64];
        const int cx = search_cell(P, p.x, p.y, p.z).x;
        // owned particles only: a ghost is measured by its owner (the driver adds up the ranks' verdicts), and with
        // subSteps > 1 the outermost ghosts are no longer right after the first sub-step and must not raise alarms
        const int binned = valid ? (int)(slabKeys[s] % (uint32_t)P.rx) : 0;
        int shift = valid && binned >= P.ownLo && binned < P.ownHi ? abs(cx - (binned + P.xoff)) : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) shift = max(shift, __shfl_xor(shift, o));
        if ((threadIdx.x & 63) == 0) waveShift[threadIdx.x >> 6] = shift;
        __syncthreads();
        if (threadIdx.x == 0)
            slabShift[blockIdx.x] = max(max(waveShift[0], waveShift[1]), max(waveShift[2], waveShift[3]));
    }
    if (!valid) return;
    // (for the packed neighbour records written at the end; loaded here so the latency is hidden)
    const uint32_t ph = !nbr.a ? 0u : nbr.phaseOfSlot ? nbr.phaseOfSlot[s] : nbr.phaseOfId[nbr.origId[s]];
    const float4 v = nbr.a ? nbr.velV[s] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int3 c = search_cell(P, p.x, p.y, p.z);
    float density = 0.0f;
    // Hit mask for the force kernel (see k_force_xsph_hits): one bit per candidate slot, every row
    // starting a fresh 32-bit word (bit b of a row's k-th word = slot row.a + 32 k + b), so that a
    // set bit decodes to a slot with one add; kMaskWords words per particle.  Particles whose rows
    // need more words simply stop recording (the force kernel recomputes the word count from the
    // same ranges and falls back to scanning).  Words are staged in LDS and written out once at
    // the end, 16 bytes at a time: word-by-word global stores left every particle's line
    // half-written in L2 for the whole kernel and cost 150 B of HBM writes per particle.  (maskStage: LDS of the kernel.)
    // words 0..15 of a particle live in one 64-byte line; the rest (crowded particles only) in a second array
    uint32_t* const mrowG = hitMask ? hitMask + (size_t)s * kMaskStage : nullptr;
    uint32_t* const mrowHi = hitMask ? hitMask + (size_t)maskStride * kMaskStage + (size_t)s * (kMaskWords - kMaskStage) - kMaskStage : nullptr;
    const bool recording = hitMask && !capped;
    uint32_t word = 0;
    int sh = 0, wi = 0;
    // Keep gathers in flight: the next row's range is fetched while this row is walked, and
    // candidates are loaded four at a time before any is tested.  Order of the sum is unchanged.
    // (Packed v_pk_*_f32 arithmetic was tried here: on gfx950 a packed op issues at half the rate
    // of a plain one -- scripts/experiments/valu_rate.hip -- so it only adds register shuffles; the
    // TU is built with -fno-slp-vectorize for the same reason.)
    // compile-time: a wave-uniform RUN-TIME choice between the two candidate loads inside the loop (slab mode's
    // phaseMismatch flag, as the force kernel reads it) cost 0.24 instead of 0.13 ms; the host decides instead
    constexpr bool uniform = UNIFORM_MASS;
    RowRange r = range_at(P, cellCum, c, 0, capped, true);
    // One row's walk.  SELF: some lane of the wave has its own slot in this row, so candidates are also tested
    // against the lane's slot (compute:173); in the other rows (about two of three) those four compares and ANDs per
    // group are dropped -- a wave-uniform choice per row, same results.
    auto walk_row = [&](auto selfTag) {
        constexpr bool SELF = decltype(selfTag)::value;
        const int last = r.e - 1;
        for (int j = r.a; j < r.e; j += 4) {
            // one 32-bit byte offset + immediates: slots past the row's end are loaded (the arrays
            // are padded by kSlotPad entries) and rejected by the predicates below
            const char* const qb = (const char*)posM + ((uint32_t)j << 4);
            float4 q0, q1, q2, q3;
            if (UNIFORM_MASS && uniform) {
                const char* const tb = (const char*)pos3 + (uint32_t)j * 12u;
                const Float4U c0 = *(const Float4U*)tb, c1 = *(const Float4U*)(tb + 16), c2 = *(const Float4U*)(tb + 32);
                q0 = make_float4(c0.x, c0.y, c0.z, uniformMass); q1 = make_float4(c0.w, c1.x, c1.y, uniformMass);
                q2 = make_float4(c1.z, c1.w, c2.x, uniformMass); q3 = make_float4(c2.y, c2.z, c2.w, uniformMass);
            } else {
                q0 = *(const float4*)qb; q1 = *(const float4*)(qb + 16);
                q2 = *(const float4*)(qb + 32); q3 = *(const float4*)(qb + 48);
            }
            const float rs0 = dist2_exact(p.x - q0.x, p.y - q0.y, p.z - q0.z);
            const float rs1 = dist2_exact(p.x - q1.x, p.y - q1.y, p.z - q1.z);
            const float rs2 = dist2_exact(p.x - q2.x, p.y - q2.y, p.z - q2.z);
            const float rs3 = dist2_exact(p.x - q3.x, p.y - q3.y, p.z - q3.z);
            const float d0 = P.h2 - rs0, d1 = P.h2 - rs1, d2 = P.h2 - rs2, d3 = P.h2 - rs3;
            const float t0 = q0.w * P.poly6 * d0 * d0 * d0;              // compute:180, left to right
            const float t1 = q1.w * P.poly6 * d1 * d1 * d1;
            const float t2 = q2.w * P.poly6 * d2 * d2 * d2;
            const float t3 = q3.w * P.poly6 * d3 * d3 * d3;
            const bool k0 = rs0 < P.h2 && (!SELF || j != s);
            const bool k1 = j + 1 <= last && rs1 < P.h2 && (!SELF || j + 1 != s);
            const bool k2 = j + 2 <= last && rs2 < P.h2 && (!SELF || j + 2 != s);
            const bool k3 = j + 3 <= last && rs3 < P.h2 && (!SELF || j + 3 != s);
            // branch-free: a rejected candidate adds +0.0f, which leaves the running sum
            // bit-identical (every term is >= 0), so no exec-mask juggling per candidate
            density += k0 ? t0 : 0.0f;
            density += k1 ? t1 : 0.0f;
            density += k2 ? t2 : 0.0f;
            density += k3 ? t3 : 0.0f;
            const uint32_t h0 = k0, h1 = k1, h2 = k2, h3 = k3;
            if (recording) {
                word |= (h0 | (h1 << 1) | (h2 << 2) | (h3 << 3)) << sh;
                sh += 4;
                if (sh == 32) {
                    if (wi < kMaskStage) maskStage[wi][threadIdx.x] = word;
                    else if (wi < kMaskWords) mrowHi[wi] = word;
                    wi++;
                    word = 0;
                    sh = 0;
                }
            }
        }
    };
#pragma unroll 1
    for (int t = 0; t < nRanges; t++) {
        const RowRange nxt = range_at(P, cellCum, c, t + 1, capped, t + 1 < nRanges);
        if (__any(s >= r.a && s < r.e)) walk_row(std::true_type{});
        else walk_row(std::false_type{});
        if (recording && sh != 0) {                      // rows are padded to whole words
            if (wi < kMaskStage) maskStage[wi][threadIdx.x] = word;
            else if (wi < kMaskWords) mrowHi[wi] = word;
            wi++;
            word = 0;
            sh = 0;
        }
        r = nxt;
    }
    if (recording) {
        static_assert(kMaskStage == 16, "a mask row is four 16-byte quads");
        const int tile = s - (int)threadIdx.x;
        if (tile + kBlock <= P.n) {
            // Full workgroup: every wave writes its 64 rows (4 KB, contiguous) TRANSPOSED -- an instruction covers
            // 1 KB = 8 lines instead of one quad of each of 64 rows = 32 lines (the kernel is bound by the lines
            // its memory instructions touch).  Words past a particle's count are whatever LDS held: never read.
            __syncthreads();
            const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
            uint4* const wrow = (uint4*)(hitMask + (size_t)(tile + wbase) * kMaskStage);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int e = lane + 64 * k, col = wbase + (e >> 2), q4 = (e & 3) * 4;
                wrow[e] = make_uint4(maskStage[q4][col], maskStage[q4 + 1][col], maskStage[q4 + 2][col], maskStage[q4 + 3][col]);
            }
        } else {
            const int nWords = min(wi, kMaskStage);
            uint4* const row = (uint4*)mrowG;
#pragma unroll
            for (int q = 0; q < kMaskStage / 4; q++)
                if (q * 4 < nWords)
                    row[q] = make_uint4(maskStage[q * 4][threadIdx.x], maskStage[q * 4 + 1][threadIdx.x],
                                        maskStage[q * 4 + 2][threadIdx.x], maskStage[q * 4 + 3][threadIdx.x]);
        }
    }
    const float r0 = rho0[s];
    density = fmaxf(density, r0);                                          // compute:186
    const float localB = r0 * (P.soundSpeed * P.soundSpeed) / P.gamma;     // compute:189
    const float pressure = localB * (powf(density / r0, P.gamma) - 1.0f);  // compute:190
    rhoP[s] = make_float2(density, pressure);
    if (recording && nbr.a) {
        // see PackedNbr: the phase id rides in the sign bits of density (> 0) and pressure (>= 0)
        nbr.a[s] = make_float4(p.x, p.y, p.z, __uint_as_float(__float_as_uint(density) | ((ph & 1u) << 31)));
        nbr.b[s] = make_float4(v.x, v.y, v.z, __uint_as_float(__float_as_uint(pressure) | ((ph & 2u) << 30)));
    }
}

template <bool UNIFORM_MASS>
__global__ __launch_bounds__(kBlock) void k_density_pressure_rows(DevParams P, const float4* __restrict__ posM,
                                                                  const float* __restrict__ rho0,
                                                                  float2* __restrict__ rhoP,
                                                                  const int* __restrict__ cellCum,
                                                                  const int* __restrict__ counters,
                                                                  uint32_t* __restrict__ hitMask, int maskStride,
                                                                  PackedNbr nbr,
                                                                  const uint32_t* __restrict__ slabKeys,
                                                                  int* __restrict__ slabShift,
                                                                  const float* __restrict__ pos3, float uniformMass) {
    __shared__ uint32_t maskStage[kMaskStage][kBlock];
    SPH_WAVE_TRACE(0);
    density_rows_body<UNIFORM_MASS>(P, posM, rho0, rhoP, cellCum, counters, hitMask, maskStride, nbr, slabKeys, slabShift, pos3,
                                    uniformMass, maskStage);
}

// CS_DensityPressure with the candidates staged in LDS per WORKGROUP (round 3; the gather this replaces is
// compute:170-175).  The rows kernel above loads every candidate per lane: lanes of one cell load the same slots and
// neighbouring cells' ranges overlap in two of three cells, so a workgroup pulls each candidate position through the
// texture addresser ~7 times, and the addresser is what bounds the kernel.  Here the 256 lanes of a workgroup stage,
// for every one of the nine (dy, dz) steps, the union of what they need -- for each distinct CURRENT (y, z) row of
// the workgroup's particles the window [cells xmin-1 .. xmax+1] of the row (y+dy, z+dz) -- with coalesced 64-slot
// loads into LDS, and every lane walks its own range [a, e) from there: same candidates, same order, same
// arithmetic as the rows kernel, so the density stays bit-identical and the hit mask is the same.
//   * The grid is STALE (bins are frame-start positions, the search cell is the current one): the lanes of a
//     workgroup sit in ~9 distinct current rows (up to 40), not one.  Rows are keyed by their offset (-8..7)^2 from
//     the row of the workgroup's first particle; a lane further away, or a window that no longer fits the LDS
//     budget, simply takes the global loads of the rows kernel for that step (second pass; rare).
//   * Windows are placed by an LDS bump allocator (order irrelevant), 64 slots per chunk; kWinChunks chunks of
//     12-byte positions per step.
constexpr int kWinChunks = 36;                   // 2,304 slots x 12 B = 27 KB of candidate positions per step
constexpr int kWinSlots = kWinChunks * 64;
constexpr int kEntSide = 16;                     // row offsets -8 .. 7 in y and z
struct LdsStats { unsigned long long* v; };      // [0] lane-steps walked from LDS, [1] from global, [2] chunks staged (debug builds)

template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_density_pressure_lds(DevParams P, const float4* __restrict__ posM,
                                                                 const float* __restrict__ rho0,
                                                                 float2* __restrict__ rhoP,
                                                                 const int* __restrict__ cellCum,
                                                                 const int* __restrict__ counters,
                                                                 uint32_t* __restrict__ hitMask, int maskStride,
                                                                 PackedNbr nbr, const float* __restrict__ pos3,
                                                                 float uniformMass, LdsStats stats) {
    __shared__ uint32_t maskStage[kMaskStage][kBlock];
    // candidate positions, one array per coordinate, and slot j of a window at an index == j (mod 4): a lane reads the
    // four candidates of a group as three aligned 16-byte LDS loads (ds_read_b128, 256 B/clk; 12-byte records read
    // as dwords made the kernel LDS-bound: 0.26 ms)
    __shared__ __attribute__((aligned(16))) float winX[kWinSlots + 4], winY[kWinSlots + 4], winZ[kWinSlots + 4];
    __shared__ int entMin[kEntSide * kEntSide], entMax[kEntSide * kEntSide];
    __shared__ int entA[2][kEntSide * kEntSide], entLen[2][kEntSide * kEntSide], entBase[2][kEntSide * kEntSide];
    __shared__ int chunkDesc[2][kWinChunks];
    __shared__ int bump[2];
    __shared__ int ref[2];
    if (counters[SPH_CTR_BIN_OVERFLOW] != 0) {
        // a bin overflowed its capacity at the last grid build (rule O2): 27 capped cell ranges, no mask -- the rows kernel
        density_rows_body<true>(P, posM, rho0, rhoP, cellCum, counters, hitMask, maskStride, nbr, nullptr, nullptr, pos3,
                                uniformMass, maskStage);
        return;
    }
    const int tid = threadIdx.x;
    const int s = xcd_tile(P, blockIdx.x, gridDim.x) * kBlock + tid;
    const bool valid = s < P.n;
    const float4 p = posM[valid ? s : 0];
    const uint32_t ph = (!nbr.a || !valid) ? 0u : nbr.phaseOfSlot ? nbr.phaseOfSlot[s] : nbr.phaseOfId[nbr.origId[s]];
    const float4 v = (nbr.a && valid) ? nbr.velV[s] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int3 c = search_cell(P, p.x, p.y, p.z);
    // the row table: keyed by the offset from the current row of the workgroup's first particle
    if (tid == 0) { ref[0] = c.y; ref[1] = c.z; bump[0] = 0; bump[1] = 0; }
    entMin[tid] = 0x7FFFFFFF;
    entMax[tid] = -0x7FFFFFFF;
    __syncthreads();
    const int y0 = ref[0], z0 = ref[1];
    const int oy = c.y - y0 + kEntSide / 2, oz = c.z - z0 + kEntSide / 2;
    const bool regular = valid && oy >= 0 && oy < kEntSide && oz >= 0 && oz < kEntSide;
    const int ent = regular ? oy + kEntSide * oz : 0;
    if (regular) { atomicMin(&entMin[ent], c.x); atomicMax(&entMax[ent], c.x); }
    __syncthreads();
    // this thread as the OWNER of table entry `tid`
    const int ex0 = entMin[tid], ex1 = entMax[tid];
    const bool used = ex0 <= ex1;
    const int ey = y0 + (tid & (kEntSide - 1)) - kEntSide / 2, ez = z0 + (tid / kEntSide) - kEntSide / 2;
    const int wxlo = max(max(ex0 - 1, 0) - P.xoff, 0), wxhi = min(min(ex1 + 1, P.rxg - 1) - P.xoff, P.rx - 1);
    auto window_of = [&](int t, int& A, int& len) {          // the entry's window in step t: first slot (multiple of 4), length
        A = 0; len = 0;
        const int y = ey + (t % 3 - 1), z = ez + (t / 3 - 1);
        if (used && t < 9 && y >= 0 && y < P.ry && z >= 0 && z < P.rz && wxlo <= wxhi) {
            const int base = P.rx * row_index(P.ryT, y, z);
            A = cellCum[base + wxlo] & ~3;
            len = cellCum[base + wxhi + 1] - A;
        }
    };
    // entry owner: reserve chunks for the window, publish it (double-buffered by step parity)
    auto publish = [&](int t, int A, int len) {
        const int par = t & 1;
        int chunks = (len + 63) >> 6, cb = 0;
        if (chunks > 0) {
            cb = atomicAdd(&bump[par], chunks);
            if (cb + chunks > kWinChunks) {                  // does not fit: its lanes take the global loads
                for (int i = cb; i < min(cb + chunks, kWinChunks); i++) chunkDesc[par][i] = -1;
                chunks = 0; len = 0;
            }
            for (int i = 0; i < chunks; i++) chunkDesc[par][cb + i] = A + (i << 6);
        }
        entA[par][tid] = A; entLen[par][tid] = len; entBase[par][tid] = cb << 6;
    };
    float density = 0.0f;
    uint32_t* const mrowG = hitMask ? hitMask + (size_t)s * kMaskStage : nullptr;
    uint32_t* const mrowHi = hitMask ? hitMask + (size_t)maskStride * kMaskStage + (size_t)s * (kMaskWords - kMaskStage) - kMaskStage : nullptr;
    const bool recording = hitMask != nullptr && valid;
    int sh = 0, wi = 0;
    const int maxSlot = P.n - 1 + kSlotPad;
    RowRange r = row_range(P, cellCum, c, -1, -1, valid);
    {
        int A, len;
        window_of(0, A, len);
        publish(0, A, len);
    }
    // One row's walk from LDS (li = index of slot ja & ~3 in the window arrays) or from memory.  Groups are aligned to
    // multiples of four SLOTS (that is what makes the LDS loads 16-byte aligned), so the first group of a row may start
    // up to three slots before the row: those candidates are rejected (they add +0.0f like every rejected candidate,
    // the sum is unchanged) and their bits are shifted out, so the mask keeps the rows kernel's layout: bit b of a
    // row's k-th word = slot row.a + 32 k + b.
    unsigned long long mbits = 0;
    auto flush = [&]() {
        if (wi < kMaskStage) maskStage[wi][tid] = (uint32_t)mbits;
        else if (wi < kMaskWords) mrowHi[wi] = (uint32_t)mbits;
        wi++;
        mbits >>= 32;
        sh -= 32;
    };
    auto walk_row = [&](auto ldsTag, auto selfTag, int ja, int je, int li) {
        constexpr bool LDS = decltype(ldsTag)::value;
        constexpr bool SELF = decltype(selfTag)::value;
        const int last = je - 1;
        int skip = ja & 3;
        for (int j = ja & ~3; j < je; j += 4, li += 4) {
            float4 qx, qy, qz;
            if (LDS) {
                const int q4 = li >> 2;                       // li is a multiple of 4
                qx = ((const float4*)winX)[q4]; qy = ((const float4*)winY)[q4]; qz = ((const float4*)winZ)[q4];
            } else {
                const float4* const tb = (const float4*)((const char*)pos3 + (uint32_t)j * 12u);    // 48-byte aligned
                const float4 c0 = tb[0], c1 = tb[1], c2 = tb[2];
                qx = make_float4(c0.x, c0.w, c1.z, c2.y); qy = make_float4(c0.y, c1.x, c1.w, c2.z); qz = make_float4(c0.z, c1.y, c2.x, c2.w);
            }
            const float rs0 = dist2_exact(p.x - qx.x, p.y - qy.x, p.z - qz.x);
            const float rs1 = dist2_exact(p.x - qx.y, p.y - qy.y, p.z - qz.y);
            const float rs2 = dist2_exact(p.x - qx.z, p.y - qy.z, p.z - qz.z);
            const float rs3 = dist2_exact(p.x - qx.w, p.y - qy.w, p.z - qz.w);
            const float d0 = P.h2 - rs0, d1 = P.h2 - rs1, d2 = P.h2 - rs2, d3 = P.h2 - rs3;
            const float t0 = uniformMass * P.poly6 * d0 * d0 * d0;       // compute:180, left to right
            const float t1 = uniformMass * P.poly6 * d1 * d1 * d1;
            const float t2 = uniformMass * P.poly6 * d2 * d2 * d2;
            const float t3 = uniformMass * P.poly6 * d3 * d3 * d3;
            const bool k0 = j >= ja && j <= last && rs0 < P.h2 && (!SELF || j != s);
            const bool k1 = j + 1 >= ja && j + 1 <= last && rs1 < P.h2 && (!SELF || j + 1 != s);
            const bool k2 = j + 2 >= ja && j + 2 <= last && rs2 < P.h2 && (!SELF || j + 2 != s);
            const bool k3 = j + 3 <= last && rs3 < P.h2 && (!SELF || j + 3 != s);
            density += k0 ? t0 : 0.0f;
            density += k1 ? t1 : 0.0f;
            density += k2 ? t2 : 0.0f;
            density += k3 ? t3 : 0.0f;
            const uint32_t h0 = k0, h1 = k1, h2 = k2, h3 = k3;
            if (recording) {
                mbits |= (unsigned long long)((h0 | (h1 << 1) | (h2 << 2) | (h3 << 3)) >> skip) << sh;
                sh += 4 - skip;
                if (sh >= 32) flush();
            }
            skip = 0;
        }
    };
#pragma unroll 1
    for (int t = 0; t < 9; t++) {
        const int par = t & 1;
        __syncthreads();                       // step t's windows are published; everybody is done with step t-1's LDS
        // stage: wave w takes chunks w, w+4, ...; lane l loads slot d + l (coalesced 12-byte loads)
        {
            const int nCh = min(bump[par], kWinChunks);
            const int lane = tid & 63;
            for (int ci = tid >> 6; ci < nCh; ci += kBlock / 64) {
                const int d = chunkDesc[par][ci];
                if (d >= 0) {
                    const int slot = min(d + lane, maxSlot);
                    const float* const src = pos3 + 3 * (size_t)slot;
                    const float x = src[0], y = src[1], z = src[2];
                    const int di = (ci << 6) + lane;
                    winX[di] = x; winY[di] = y; winZ[di] = z;
                }
            }
            if (STATS && tid == 0) atomicAdd(&stats.v[2], (unsigned long long)nCh);
        }
        // meanwhile: next step's range of this lane and window of this entry (loads in flight across the walk)
        const RowRange nxt = row_range(P, cellCum, c, (t + 1) % 3 - 1, (t + 1) / 3 - 1, valid && t + 1 < 9);
        int nA, nLen;
        window_of(t + 1, nA, nLen);
        if (tid == 0) bump[par ^ 1] = 0;       // (nobody reads bump[par ^ 1] between the barrier above and the publish below)
        __syncthreads();                       // staged
        const int wA = entA[par][ent], wLen = entLen[par][ent], wBase = entBase[par][ent];
        const bool nonEmpty = r.a < r.e;
        const bool covered = nonEmpty && regular && r.a >= wA && r.e <= wA + wLen;
        const bool self = __any(s >= r.a && s < r.e);
        const int wiRow = wi;
        {
            const int ja = covered ? r.a : 0, je = covered ? r.e : 0;
            const int li = wBase + ((r.a & ~3) - wA);          // wA, wBase are multiples of 4: so is li
            if (self) walk_row(std::true_type{}, std::true_type{}, ja, je, li);
            else walk_row(std::true_type{}, std::false_type{}, ja, je, li);
        }
        if (__any(nonEmpty && !covered)) {     // lanes outside the table / windows that did not fit: the rows kernel's loads
            const int ja = covered ? 0 : r.a, je = covered ? 0 : r.e;
            if (self) walk_row(std::false_type{}, std::true_type{}, ja, je, 0);
            else walk_row(std::false_type{}, std::false_type{}, ja, je, 0);
        }
        if (STATS) {
            const unsigned long long mc = __ballot(covered), mu = __ballot(nonEmpty && !covered);
            if ((tid & 63) == 0) { atomicAdd(&stats.v[0], (unsigned long long)__popcll(mc)); atomicAdd(&stats.v[1], (unsigned long long)__popcll(mu)); }
        }
        if (recording) {                       // rows are padded to whole words (bits left over that lie past the row's
            if (sh != 0 && 32 * (wi - wiRow) < r.e - r.a) flush();      // end -- alignment padding -- are not a word)
            mbits = 0;
            sh = 0;
        }
        if (t + 1 < 9) publish(t + 1, nA, nLen);
        r = nxt;
    }
    if (hitMask) {
        static_assert(kMaskStage == 16, "a mask row is four 16-byte quads");
        const int tile = s - tid;
        if (tile + kBlock <= P.n) {
            __syncthreads();
            const int lane = tid & 63, wbase = tid & ~63;
            uint4* const wrow = (uint4*)(hitMask + (size_t)(tile + wbase) * kMaskStage);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int e = lane + 64 * k, col = wbase + (e >> 2), q4 = (e & 3) * 4;
                wrow[e] = make_uint4(maskStage[q4][col], maskStage[q4 + 1][col], maskStage[q4 + 2][col], maskStage[q4 + 3][col]);
            }
        } else if (valid) {
            const int nWords = min(wi, kMaskStage);
            uint4* const row = (uint4*)mrowG;
#pragma unroll
            for (int q = 0; q < kMaskStage / 4; q++)
                if (q * 4 < nWords)
                    row[q] = make_uint4(maskStage[q * 4][tid], maskStage[q * 4 + 1][tid],
                                        maskStage[q * 4 + 2][tid], maskStage[q * 4 + 3][tid]);
        }
    }
    if (!valid) return;
    const float r0 = rho0[s];
    density = fmaxf(density, r0);                                          // compute:186
    const float localB = r0 * (P.soundSpeed * P.soundSpeed) / P.gamma;     // compute:189
    const float pressure = localB * (powf(density / r0, P.gamma) - 1.0f);  // compute:190
    rhoP[s] = make_float2(density, pressure);
    if (hitMask && nbr.a) {
        nbr.a[s] = make_float4(p.x, p.y, p.z, __uint_as_float(__float_as_uint(density) | ((ph & 1u) << 31)));
        nbr.b[s] = make_float4(v.x, v.y, v.z, __uint_as_float(__float_as_uint(pressure) | ((ph & 2u) << 30)));
    }
}

// slab mode: maximum of the per-workgroup plane shifts left by the density kernel
constexpr int kShiftPerGroup = 2048;
__global__ __launch_bounds__(kBlock) void k_slab_shift_reduce(const int* __restrict__ blockShift, int nBlocks,
                                                              int* __restrict__ counter) {
    __shared__ int ws[kBlock / 64];
    int m = 0;
    // every workgroup folds kShiftPerGroup entries (one workgroup over 32 k entries took 25 us at 8.4 M particles)
    const int b0 = blockIdx.x * kShiftPerGroup, b1 = min(b0 + kShiftPerGroup, nBlocks);
    for (int b = b0 + (int)threadIdx.x; b < b1; b += kBlock) m = max(m, blockShift[b]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = max(max(ws[0], ws[1]), max(ws[2], ws[3]));
        if (tot > 0) atomicMax(counter, tot);
    }
}

// One in-range pair of CS_ForceXSPH (compute:250-277) with explicit FMAs and hardware
// reciprocals / square root (1 ulp) instead of the 12 IEEE divides of the literal form; the
// result is tolerance-checked against the oracle (1e-5 of field scale), not bit-compared.
struct ForceAcc {
    float fx, fy, fz, cgx, cgy, cgz, cl, vcx, vcy, vcz, norm;
};

template <bool UNI = false>
__device__ __forceinline__ void force_pair(const DevParams& P, ForceAcc& A, const float4 p, const float4 v,
                                           float densitySelf, float selfTerm, const float4 q, const float4 u,
                                           const float2 o) {
    // Every multiply-add is spelled out (fmaf) with contraction OFF, so that each inlined copy of
    // this function is the same instruction sequence: a pair's contribution must not depend on
    // where in a lane's FIFO it sits (slab runs are compared bit for bit with one-GPU runs).
#pragma clang fp contract(off)
    const float rx = p.x - q.x, ry = p.y - q.y, rz = p.z - q.z;
    const float rSq = dist2_exact(rx, ry, rz);
    const float dist = __builtin_amdgcn_sqrtf(rSq) + 1e-6f;               // compute:250
    const float invd = __builtin_amdgcn_rcpf(dist);
    const float invRho = __builtin_amdgcn_rcpf(o.x);
    const float diff = P.h2 - rSq;
    const float pressureTerm = fmaf(o.y * invRho, invRho, selfTerm);      // compute:255-256
    const float gs = (P.spikyGradFactor * diff) * (diff * invd);          // gradW = gs * r  (compute:257)
    const float gwx = gs * rx, gwy = gs * ry, gwz = gs * rz;
    const float pf = -q.w * pressureTerm;
    const float dvx = u.x - v.x, dvy = u.y - v.y, dvz = u.z - v.z;        // compute:261
    const float mor = q.w * invRho;                                       // m_j / rho_j
    // UNI: every phase has the same viscosity, v.w == u.w and 0.5 (a + a) == a exactly
    const float viscMean = UNI ? u.w : 0.5f * (v.w + u.w);
    const float vm = (viscMean * mor) * (P.viscLapFactor * (P.h - dist));               // compute:263-265
    A.fx = fmaf(pf, gwx, fmaf(vm, dvx, A.fx));                            // compute:258, 265
    A.fy = fmaf(pf, gwy, fmaf(vm, dvy, A.fy));
    A.fz = fmaf(pf, gwz, fmaf(vm, dvz, A.fz));
    A.cgx = fmaf(mor, gwx, A.cgx);                                        // compute:268
    A.cgy = fmaf(mor, gwy, A.cgy);
    A.cgz = fmaf(mor, gwz, A.cgz);
    const float lapW = (-P.lapColFactor * diff) * fmaf(-5.0f, rSq, P.h2); // compute:269
    A.cl = fmaf(mor, lapW, A.cl);                                         // compute:270
    const float wm = ((P.poly6 * diff) * (diff * diff)) * (q.w * __builtin_amdgcn_rcpf(0.5f * (densitySelf + o.x)));
    A.vcx = fmaf(dvx, wm, A.vcx);                                         // compute:276
    A.vcy = fmaf(dvy, wm, A.vcy);
    A.vcz = fmaf(dvz, wm, A.vcz);
    A.norm += wm;                                                         // compute:277
}

// CS_ForceXSPH (compute:199-303).  The expensive pair arithmetic must run with (nearly) full lanes
// although only ~1/4 of the candidates are in range.  Two ways to get there:
//
//  scan path (force_scan): every lane walks its candidates, does the exact rSq < h2 test and appends
//      hits to its own FIFO in LDS (queue[k][thread], conflict-free); when any lane's FIFO is nearly
//      full (wave ballot) all lanes pop their hits in order and do the pair arithmetic.
//  hit path (k_force_xsph_hits): CS_DensityPressure has just done the same tests on the same
//      positions and grid, and left one bit per candidate; the lane walks the set bits only, so the
//      rejected ~3/4 of the candidates are never loaded again.
//
// Either way a lane sums its neighbours in the reference's visiting order.
constexpr int kHitQueue = 20;   // FIFO depth per lane (20 KB of LDS per workgroup)
constexpr int kScanUnroll = 4;  // candidates loaded per lane before any is tested

struct ForceSelf {
    float4 p, v;
    float densitySelf, selfTerm;
    int s, sl;
    bool live;
};

__device__ __forceinline__ void force_scan(const DevParams& P, ForceAcc& A, const ForceSelf& me, int3 c,
                                           bool capped, const float4* __restrict__ posM,
                                           const float4* __restrict__ velIn, const float2* __restrict__ rhoP,
                                           const int* __restrict__ cellCum, int (*queue)[kBlock]) {
    const int nRanges = capped ? 27 : 9;
    int cnt = 0;
    auto drain = [&]() {
        // two pairs per trip: six gathers in flight before the arithmetic starts
#pragma unroll 1
        for (int k = 0; k < kHitQueue; k += 2) {
            if (!__any(k < cnt)) break;
            const int j0 = (k < cnt) ? queue[k][threadIdx.x] : me.sl;
            const int j1 = (k + 1 < cnt) ? queue[k + 1][threadIdx.x] : me.sl;
            const float4 q0 = posM[j0], u0 = velIn[j0];
            const float2 o0 = rhoP[j0];
            const float4 q1 = posM[j1], u1 = velIn[j1];
            const float2 o1 = rhoP[j1];
            if (k < cnt) force_pair(P, A, me.p, me.v, me.densitySelf, me.selfTerm, q0, u0, o0);
            if (k + 1 < cnt) force_pair(P, A, me.p, me.v, me.densitySelf, me.selfTerm, q1, u1, o1);
        }
        cnt = 0;
    };
    auto test = [&](const float4 q, int j) {
        const float rSq = dist2_exact(me.p.x - q.x, me.p.y - q.y, me.p.z - q.z);
        if (rSq < P.h2 && j != me.s) {
            queue[cnt][threadIdx.x] = j;
            cnt++;
        }
    };
    RowRange r = range_at(P, cellCum, c, 0, capped, me.live);
#pragma unroll 1
    for (int t = 0; t < nRanges; t++) {
        const RowRange nxt = range_at(P, cellCum, c, t + 1, capped, me.live && t + 1 < nRanges);
        const int last = r.e - 1;
        int j = r.a;
        while (__any(j < r.e)) {
            if (j < r.e) {
                const float4 q0 = posM[j];
                const float4 q1 = posM[min(j + 1, last)];
                const float4 q2 = posM[min(j + 2, last)];
                const float4 q3 = posM[min(j + 3, last)];
                test(q0, j);
                if (j + 1 <= last) test(q1, j + 1);
                if (j + 2 <= last) test(q2, j + 2);
                if (j + 3 <= last) test(q3, j + 3);
            }
            j += kScanUnroll;
            if (__any(cnt > kHitQueue - kScanUnroll)) drain();   // room for the next four
        }
        r = nxt;
    }
    drain();
}

__device__ __forceinline__ ForceSelf force_load_self(const DevParams& P, int s, const float4* __restrict__ posM,
                                                     const float4* __restrict__ velIn,
                                                     const float2* __restrict__ rhoP) {
    ForceSelf me;
    me.s = s;
    const bool valid = s < P.n;
    me.sl = valid ? s : 0;
    me.p = posM[me.sl];
    me.v = velIn[me.sl];
    const float2 dp = rhoP[me.sl];
    me.densitySelf = dp.x;
    me.live = valid && !(me.densitySelf < 1e-6f);                          // compute:207-211
    if (P.ghostKeys && valid) {                                            // slab mode: see DevParams::ghostKeys
        const int plane = (int)(P.ghostKeys[s] % (uint32_t)P.rx);
        me.live = me.live && plane >= P.ownLo && plane < P.ownHi;
    }
    me.selfTerm = dp.y / (me.densitySelf * me.densitySelf);
    return me;
}

// sph_step_frame on one GPU: the frame's tail -- CS_VvFullStep + CS_BoundObs (+ CS_VelocityDamping after the last
// sub-step), k_full_bound_damp below -- applied by the force kernel itself to the particle it has just finished, in
// registers: the same statements on the same values, so positions and velocities are bit-identical to the separate
// pass, which then does not run (one launch and 48 B read + 16 B written per particle less).  The new position goes to
// the OTHER position buffer (posOut): a wave that fell back to the scan path still reads its neighbours' positions
// from posM while other workgroups finish.
struct TailArgs {
    float4* posOut;                  // null = tail not fused
    int damp;
    int* blockCollide;               // per-workgroup OBB collision counts (block_count)
    int* counters;
};
__device__ __forceinline__ bool bound_obs_body(const DevParams& P, float4& p, float4& v);
__device__ __forceinline__ void damp_velocity(float4& v);
__device__ __forceinline__ void block_count(bool pred, int* __restrict__ blockCount);
__device__ __forceinline__ void count_non_finite(const float4& p, const float4& v, bool counted, int* __restrict__ counters);

template <bool TAIL>
__device__ __forceinline__ void force_finish(const DevParams& P, ForceAcc& A, const ForceSelf& me,
                                             float4* __restrict__ acc, float4* __restrict__ velOut,
                                             int zeroDeadAcc, const TailArgs& T) {
    const int s = me.s;
    bool collided = false;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), v = p;
    if (s < P.n) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 vo = me.v;
        if (me.live) {
            // explicit operations, contraction off: this tail is inlined into several kernels and a particle
            // must get the same bits whichever of them handles it
#pragma clang fp contract(off)
            const float gradMag = sqrtf((A.cgx * A.cgx + A.cgy * A.cgy) + A.cgz * A.cgz);   // compute:285
            if (gradMag > P.stThreshold) {
                const float curvature = A.cl / (gradMag + 1e-6f);
                const float st = -P.sigma * curvature / gradMag;
                A.fx = fmaf(st, A.cgx, A.fx); A.fy = fmaf(st, A.cgy, A.fy); A.fz = fmaf(st, A.cgz, A.fz);
            }
            const float invD = 1.0f / me.densitySelf;
            a = make_float4(A.fx * invD, A.fy * invD, A.fz * invD, 0.f);                 // compute:293
            if (A.norm > 1e-6f) {                                                        // compute:296-300
                const float sc = P.xsphEps / A.norm;
                vo.x = fmaf(A.vcx, sc, vo.x); vo.y = fmaf(A.vcy, sc, vo.y); vo.z = fmaf(A.vcz, sc, vo.z);
            }
        }
        if (!TAIL) {
            velOut[s] = vo;
            // a skipped particle keeps its acceleration unless the elided CS_Clear would have zeroed it
            if (me.live || zeroDeadAcc) acc[s] = a;
        } else {
            // (TAIL is only launched by sph_step_frame, where zeroDeadAcc is set: a skipped particle's acceleration is 0)
#pragma clang fp contract(off)
            p = me.p; v = vo;
            const float hdt = 0.5f * P.dt;
            v.x += hdt * a.x; v.y += hdt * a.y; v.z += hdt * a.z;                        // compute:332
            collided = bound_obs_body(P, p, v);
            if (T.damp) damp_velocity(v);
            T.posOut[s] = p;
            velOut[s] = v;
            acc[s] = a;
        }
    }
    if (TAIL) {
        count_non_finite(p, v, s < P.n, T.counters);
        block_count(collided, T.blockCollide);
    }
}

// Scan path only: valid for any state (used when no fresh hit mask exists).
__global__ __launch_bounds__(kBlock) void k_force_xsph_rows(DevParams P, const float4* __restrict__ posM,
                                                            const float4* __restrict__ velIn,
                                                            const float2* __restrict__ rhoP,
                                                            float4* __restrict__ acc, float4* __restrict__ velOut,
                                                            const int* __restrict__ cellCum,
                                                            const int* __restrict__ counters, int zeroDeadAcc) {
    __shared__ int queue[kHitQueue][kBlock];
    const bool capped = counters[SPH_CTR_BIN_OVERFLOW] != 0;   // wave-uniform
    const int wg = xcd_tile(P, blockIdx.x, gridDim.x);
    if (P.wgBand && P.wgBand[wg] != P.wgPhase) return;         // slab mode: the other launch of the pair does this one
    const int s = wg * kBlock + threadIdx.x;
    const ForceSelf me = force_load_self(P, s, posM, velIn, rhoP);
    ForceAcc A{0.0f, P.gravity * me.densitySelf, 0.0f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // compute:219
    const int3 c = search_cell(P, me.p.x, me.p.y, me.p.z);
    force_scan(P, A, me, c, capped, posM, velIn, rhoP, cellCum, queue);
    force_finish<false>(P, A, me, acc, velOut, zeroDeadAcc, TailArgs{nullptr, 0, nullptr, nullptr});
}

// Hit path.  PRECONDITION (kept by the host, libsph_hip.hip `maskValid`): hitMask was written by
// k_density_pressure_rows on exactly these positions and this grid.  A wave falls back to the scan
// path if a bin overflowed (27 capped ranges, no mask) or any of its particles needs more mask words
// than a row holds.
// MODE 0: a neighbour is three gathers (posM, velV, rhoP).  MODE 1: two packed records (PackedNbr), mass / viscosity
// looked up by the phase id in their sign bits.  MODE 2: packed records and EVERY phase has the same particle mass and
// the same viscosity (ph.m0, ph.v0: true of all the BASELINE configurations, whose phases differ in rest density only),
// so there is nothing to decode.  The host picks the mode (libsph_hip.hip); in slab mode it takes a packed mode only
// after it has read the record / phase-table mismatch flag as clean for this frame's records.
template <int MODE, bool TAIL>
__global__ __launch_bounds__(kBlock) void k_force_xsph_hits(DevParams P, const float4* __restrict__ posM,
                                                            const float4* __restrict__ velIn,
                                                            const float2* __restrict__ rhoP,
                                                            float4* __restrict__ acc, float4* __restrict__ velOut,
                                                            const int* __restrict__ cellCum,
                                                            const int* __restrict__ counters,
                                                            const uint32_t* __restrict__ hitMask, int maskStride,
                                                            const float4* __restrict__ nbrA,
                                                            const float4* __restrict__ nbrB, PhaseConsts ph,
                                                            int zeroDeadAcc, int rowSlack, TailArgs tail) {
    // one buffer, two uses: (a) per lane the non-empty mask words and the slot of their bit 0,
    // (b) the scan path's hit FIFO.  Every thread only touches its own column.
    static_assert(kMaskWords + 9 >= kHitQueue, "the FIFO lives in the word list's LDS");
    __shared__ int lds[kMaskWords + 9][kBlock];
    __shared__ uint8_t codeOf[kMaskWords][kBlock];       // (row << 3) | word-in-row of every kept word
    int (*const wordOf)[kBlock] = lds;                   // [kMaskWords]
    int (*const rowA)[kBlock] = lds + kMaskWords;        // [9] first slot of every row
    SPH_WAVE_TRACE(1);
    const bool capped = counters[SPH_CTR_BIN_OVERFLOW] != 0;   // wave-uniform
    const int wg = xcd_tile(P, blockIdx.x, gridDim.x);
    if (P.wgBand && P.wgBand[wg] != P.wgPhase) return;         // slab mode: the other launch of the pair does this one
    const int s = wg * kBlock + threadIdx.x;
    const ForceSelf me = force_load_self(P, s, posM, velIn, rhoP);
    ForceAcc A{0.0f, P.gravity * me.densitySelf, 0.0f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // compute:219
    const int3 c = search_cell(P, me.p.x, me.p.y, me.p.z);
    const int tid = threadIdx.x;
    RowRange rr[9];
    int nWords = 0;
    if (!capped) {
#pragma unroll
        for (int t = 0; t < 9; t++) {
            rr[t] = row_range(P, cellCum, c, t % 3 - 1, t / 3 - 1, me.live);
            const int nw = (rr[t].e - rr[t].a + 31) >> 5;
            nWords += nw > 8 ? kMaskWords + 1 : nw;      // the word-in-row code has three bits
        }
    }
    if (capped || __any(nWords > kMaskWords)) {
        force_scan(P, A, me, c, capped, posM, velIn, rhoP, cellCum, lds);
    } else {
        // 1) the particle's mask row -> LDS (16 bytes at a time, only the part in use)
        const uint4* __restrict__ mlo = (const uint4*)(hitMask + (size_t)me.sl * kMaskStage);
        const uint4* __restrict__ mhi = (const uint4*)(hitMask + (size_t)maskStride * kMaskStage +
                                                       (size_t)me.sl * (kMaskWords - kMaskStage)) - kMaskStage / 4;
        const int tile = wg * kBlock;
        if (tile + kBlock <= P.n) {
            // Full workgroup: the wave reads its 64 rows' first 16 words (4 KB, contiguous) TRANSPOSED: an
            // instruction covers 1 KB = 8 lines instead of one quad of each of 64 rows = 32 lines.  Only this wave's
            // columns of wordOf are touched, so a wave-level barrier orders the LDS writes before the reads below.
            const int lane = tid & 63, wbase = tid & ~63;
            const uint4* __restrict__ wrow = (const uint4*)(hitMask + (size_t)(tile + wbase) * kMaskStage);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int e = lane + 64 * k, col = wbase + (e >> 2), q4 = (e & 3) * 4;
                const uint4 v = wrow[e];
                wordOf[q4][col] = (int)v.x; wordOf[q4 + 1][col] = (int)v.y;
                wordOf[q4 + 2][col] = (int)v.z; wordOf[q4 + 3][col] = (int)v.w;
            }
#pragma unroll
            for (int q = kMaskStage / 4; q < kMaskWords / 4; q++)
                if (q * 4 < nWords) {
                    const uint4 v = mhi[q];
                    wordOf[q * 4][tid] = (int)v.x; wordOf[q * 4 + 1][tid] = (int)v.y;
                    wordOf[q * 4 + 2][tid] = (int)v.z; wordOf[q * 4 + 3][tid] = (int)v.w;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
#pragma unroll
            for (int q = 0; q < kMaskWords / 4; q++)
                if (q * 4 < nWords) {
                    const uint4 v = q < kMaskStage / 4 ? mlo[q] : mhi[q];
                    wordOf[q * 4][tid] = (int)v.x; wordOf[q * 4 + 1][tid] = (int)v.y;
                    wordOf[q * 4 + 2][tid] = (int)v.z; wordOf[q * 4 + 3][tid] = (int)v.w;
                }
        }
        // 2) keep the non-empty words, each with its row and its place in the row (in place: the
        //    write position never overtakes the read position)
        int cnt = 0, idx = 0;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            rowA[t][tid] = rr[t].a;
            const int nw = (rr[t].e - rr[t].a + 31) >> 5;
            for (int k = 0; k < nw; k++, idx++) {
                const int w = wordOf[idx][tid];
                if (w != 0) {
                    wordOf[cnt][tid] = w;
                    codeOf[cnt][tid] = (uint8_t)((t << 3) | k);
                    cnt++;
                }
            }
        }
        constexpr bool packed = MODE != 0;
        // 3) walk the set bits: visiting order, one add per hit to find its slot.
        //    The lanes of a wave walk their rows TOGETHER: a lane may start a word of row r only while
        //    r <= (the row the slowest lane is in) + rowSlack.  A lane's own sequence is unchanged (same hits, same
        //    order, same sums); what changes is WHERE the 64 gathers of one instruction point: into the windows of
        //    1 + rowSlack cell rows instead of all nine, i.e. ~12 lines per row instead of ~45, and a wave-wide gather
        //    costs the texture addresser ~0.8 clk per distinct line (scripts/experiments/gather_rate.hip).  The price
        //    is idle lanes while a row's busiest lane finishes; rowSlack >= 8 is the ungated walk.  Measured: the lanes
        //    of a wave have similar hit counts per row and stay nearly in step by themselves -- a loose gate (4 rows,
        //    the host's default) is worth ~1 %, a tight one costs up to 35 % (libsph_hip.hip, rowSlack).
        int li = 0, base = 0, rowOfW = 0;
        uint32_t w = 0;
        int nxt = cnt > 0 ? (int)codeOf[0][tid] : (9 << 3);      // code of the next word to start; row 9 = none left
        int curRow = 0;                                          // wave-uniform: no lane has hits left in a row below it
        const int nLast = P.n - 1;
        auto pop = [&](int& j, int limit) -> bool {
            if (w == 0 && (nxt >> 3) <= limit) {
                w = (uint32_t)wordOf[li][tid];
                rowOfW = nxt >> 3;
                base = rowA[rowOfW][tid] + ((nxt & 7) << 5);     // slot of the word's bit 0
                li++;
                nxt = li < cnt ? (int)codeOf[li][tid] : (9 << 3);
            }
            if (w == 0) { j = me.sl; return false; }
            j = min(base + __builtin_ctz(w), nLast);     // the min only guards against a stale mask
            w &= w - 1;
            return true;
        };
        // Software-pipelined, two pairs per stage: the six gathers of the next two hits are issued
        // before the arithmetic of the current two starts.
        struct Stage {
            float4 q0, u0, q1, u1;
            float2 o0, o1;
            bool h0, h1;
        };
        auto fetch = [&](Stage& S) {
            // the row the slowest lane is in (a lane between words counts with the word it starts next)
            int limit = 8;
            if (rowSlack < 8) {                                  // wave-uniform; 8 = ungated: no bookkeeping at all
                while (curRow < 9 && !__any((w != 0 ? rowOfW : (nxt >> 3)) <= curRow)) curRow++;
                limit = min(curRow + rowSlack, 8);
            }
            int j0, j1;
            S.h0 = pop(j0, limit);
            S.h1 = pop(j1, limit);
            if (packed) {
                S.q0 = nbrA[j0]; S.u0 = nbrB[j0];
                S.q1 = nbrA[j1]; S.u1 = nbrB[j1];
            } else {
                S.q0 = posM[j0]; S.u0 = velIn[j0]; S.o0 = rhoP[j0];
                S.q1 = posM[j1]; S.u1 = velIn[j1]; S.o1 = rhoP[j1];
            }
        };
        auto unpack = [&](float4& q, float4& u, float2& o) {
            const bool b0 = (__float_as_uint(q.w) >> 31) != 0, b1 = (__float_as_uint(u.w) >> 31) != 0;
            o = make_float2(fabsf(q.w), fabsf(u.w));
            if (MODE == 2) { q.w = ph.m0; u.w = ph.v0; return; }
            const float mLo = b0 ? ph.m1 : ph.m0, mHi = b0 ? ph.m3 : ph.m2;
            const float vLo = b0 ? ph.v1 : ph.v0, vHi = b0 ? ph.v3 : ph.v2;
            q.w = b1 ? mHi : mLo;
            u.w = b1 ? vHi : vLo;
        };
        auto compute = [&](Stage& S) {
            if (packed) { unpack(S.q0, S.u0, S.o0); unpack(S.q1, S.u1, S.o1); }
            if (S.h0) force_pair<MODE == 2>(P, A, me.p, me.v, me.densitySelf, me.selfTerm, S.q0, S.u0, S.o0);
            if (S.h1) force_pair<MODE == 2>(P, A, me.p, me.v, me.densitySelf, me.selfTerm, S.q1, S.u1, S.o1);
        };
        Stage SA, SB;
        fetch(SA);
#pragma unroll 1
        for (;;) {
            if (!__any(SA.h0)) break;
            fetch(SB);
            compute(SA);
            if (!__any(SB.h0)) break;
            fetch(SA);
            compute(SB);
        }
    }
    force_finish<TAIL>(P, A, me, acc, velOut, zeroDeadAcc, tail);
}

// CS_VvHalfStep (compute:309-319)
__global__ __launch_bounds__(kBlock) void k_half_step(int n, float dt, float4* __restrict__ posM,
                                                      float4* __restrict__ velV, const float4* __restrict__ acc,
                                                      float* __restrict__ pos3) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    float4 p = posM[s], v = velV[s];
    const float4 a = acc[s];
    const float hdt = 0.5f * dt;
    v.x += hdt * a.x; v.y += hdt * a.y; v.z += hdt * a.z;
    p.x += v.x * dt; p.y += v.y * dt; p.z += v.z * dt;
    velV[s] = v;
    posM[s] = p;
    if (pos3) *(Float3U*)(pos3 + 3 * (size_t)s) = Float3U{p.x, p.y, p.z};             // see k_density_pressure_rows
}

// CS_VvFullStep (compute:325-334)
__global__ __launch_bounds__(kBlock) void k_full_step(int n, float dt, float4* __restrict__ velV,
                                                      const float4* __restrict__ acc) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    float4 v = velV[s];
    const float4 a = acc[s];
    const float hdt = 0.5f * dt;
    v.x += hdt * a.x; v.y += hdt * a.y; v.z += hdt * a.z;
    velV[s] = v;
}

__device__ __forceinline__ float dot3f(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax * bx + ay * by) + az * bz;
}

// CS_BoundObs (compute:340-486) for one particle: OBB or AABB boundary, then the sphere obstacle.
// Returns true when the OBB branch collided (what compute:418 counts).
__device__ __forceinline__ bool bound_obs_body(const DevParams& P, float4& p, float4& v) {
    bool collided = false;
    if (P.obb) {
        float lp[3], pen[3] = {0.f, 0.f, 0.f};
        for (int r = 0; r < 3; r++)
            lp[r] = ((P.Minv[r * 4 + 0] * p.x + P.Minv[r * 4 + 1] * p.y) + P.Minv[r * 4 + 2] * p.z) +
                    P.Minv[r * 4 + 3] * 1.0f;                          // compute:351
        const float E = 0.5f;
#pragma unroll
        for (int k = 0; k < 3; k++) {                                  // compute:357-392
            if (lp[k] < -E) { pen[k] = -E - lp[k]; lp[k] = -E; collided = true; }
            else if (lp[k] > E) { pen[k] = lp[k] - E; lp[k] = E; collided = true; }
        }
        if (collided) {
            float wp[3];
            for (int r = 0; r < 3; r++)
                wp[r] = ((P.M[r * 4 + 0] * lp[0] + P.M[r * 4 + 1] * lp[1]) + P.M[r * 4 + 2] * lp[2]) +
                        P.M[r * 4 + 3] * 1.0f;                         // compute:396
            p.x = wp[0]; p.y = wp[1]; p.z = wp[2];
            float ln[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (pen[k] > 0.0f) ln[k] = (lp[k] < 0.0f) ? -1.0f : 1.0f;          // compute:400-405
            const float l = sqrtf(dot3f(ln[0], ln[1], ln[2], ln[0], ln[1], ln[2]));
            ln[0] /= l; ln[1] /= l; ln[2] /= l;                        // compute:406
            float wn[3];
            for (int r = 0; r < 3; r++)
                wn[r] = (P.M[r * 4 + 0] * ln[0] + P.M[r * 4 + 1] * ln[1]) + P.M[r * 4 + 2] * ln[2];  // 408
            const float wl = sqrtf(dot3f(wn[0], wn[1], wn[2], wn[0], wn[1], wn[2]));
            wn[0] /= wl; wn[1] /= wl; wn[2] /= wl;
            const float d = dot3f(v.x, v.y, v.z, wn[0], wn[1], wn[2]);
            float vv[3] = {v.x, v.y, v.z};
#pragma unroll
            for (int k = 0; k < 3; k++) {                              // compute:411-415
                float vN = d * wn[k];
                float vT = vv[k] - vN;
                vv[k] = (-vN) + vT;
            }
            v.x = vv[0]; v.y = vv[1]; v.z = vv[2];
        }
    } else {                                                           // compute:423-463
        float pp[3] = {p.x, p.y, p.z}, vv[3] = {v.x, v.y, v.z};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float halfSize = P.boundsSize[k] * 0.5f;
            const float mn = P.boundsCenter[k] - halfSize, mx = P.boundsCenter[k] + halfSize;
            if (pp[k] < mn) {
                pp[k] = mn;
                if (vv[k] < 0) vv[k] = -vv[k] * 0.9f;
            } else if (pp[k] > mx) {
                pp[k] = mx;
                if (vv[k] > 0) vv[k] = -vv[k] * 0.9f;
            }
        }
        p.x = pp[0]; p.y = pp[1]; p.z = pp[2];
        v.x = vv[0]; v.y = vv[1]; v.z = vv[2];
    }
    if (P.obstacleRadius > 0.0f) {                                     // compute:467-483
        const float tx = p.x - P.obstaclePos[0], ty = p.y - P.obstaclePos[1], tz = p.z - P.obstaclePos[2];
        const float dist = sqrtf(dot3f(tx, ty, tz, tx, ty, tz));
        if (dist < P.obstacleRadius && dist > 1e-6f) {
            const float penetration = P.obstacleRadius - dist;
            const float nx = tx / dist, ny = ty / dist, nz = tz / dist;
            p.x += nx * penetration; p.y += ny * penetration; p.z += nz * penetration;
            const float d = dot3f(v.x, v.y, v.z, nx, ny, nz);
            v.x = (v.x - 2.0f * nx * d) * P.collisionDamping;          // reflect(v, n) * damping
            v.y = (v.y - 2.0f * ny * d) * P.collisionDamping;
            v.z = (v.z - 2.0f * nz * d) * P.collisionDamping;
        } else if (dist <= 1e-6f) {
            p.y += P.obstacleRadius + 0.01f;
            const float d = v.y;                                       // dot(v, (0,1,0))
            v.x = (v.x - 2.0f * 0.0f * d) * P.collisionDamping;
            v.y = (v.y - 2.0f * 1.0f * d) * P.collisionDamping;
            v.z = (v.z - 2.0f * 0.0f * d) * P.collisionDamping;
        }
    }
    return collided;
}

__device__ __forceinline__ bool slot_is_owned(const DevParams& P, const uint32_t* __restrict__ sortedKeys, int s) {
    if (!sortedKeys) return true;
    const int plane = (int)(sortedKeys[s] % (uint32_t)P.rx);
    return plane >= P.ownLo && plane < P.ownHi;
}

// The per-collision InterlockedAdd (compute:418) becomes a count per WORKGROUP (wave ballots + LDS),
// stored -- not added -- into blockCount[workgroup]; k_sum_into folds the counts into the counter.  Even
// one atomic per wave on the counter serialises on its address (~2 ns each): in a settled tank, where
// most waves have a particle touching a wall, that tripled the time of the fused tail kernel.
__device__ __forceinline__ void block_count(bool pred, int* __restrict__ blockCount) {
    __shared__ int waveCnt[kBlock / 64];
    const unsigned long long m = __ballot(pred);
    if ((threadIdx.x & 63) == 0) waveCnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) blockCount[blockIdx.x] = (waveCnt[0] + waveCnt[1]) + (waveCnt[2] + waveCnt[3]);
}

// overwrite != 0: the counter was reset by the host since it was last written (sph_reset_counter defers
// the zeroing to here instead of launching a 4-byte fill every frame)
__global__ __launch_bounds__(kBlock) void k_sum_into(const int* __restrict__ blockCount, int nBlocks,
                                                     int* __restrict__ counter, int overwrite) {
    __shared__ int ws[kBlock / 64];
    int m = 0;
    for (int b = threadIdx.x; b < nBlocks; b += kBlock) m += blockCount[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = m;
    __syncthreads();
    const int tot = (ws[0] + ws[1]) + (ws[2] + ws[3]);
    if (threadIdx.x == 0) {
        if (overwrite) *counter = tot;
        else if (tot) atomicAdd(counter, tot);
    }
}

// Failure detection (SPH_CTR_NAN): non-finite positions / velocities leaving a boundary pass.  A healthy run never
// takes the atomic (one per wave that holds such a particle).
__device__ __forceinline__ void count_non_finite(const float4& p, const float4& v, bool counted, int* __restrict__ counters) {
    const bool bad = counted && !(__builtin_isfinite(p.x) && __builtin_isfinite(p.y) && __builtin_isfinite(p.z) &&
                                  __builtin_isfinite(v.x) && __builtin_isfinite(v.y) && __builtin_isfinite(v.z));
    const unsigned long long m = __ballot(bad);
    if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(&counters[SPH_CTR_NAN], __popcll(m));
}

// CS_BoundObs kernel; in slab mode ghost copies are counted by their owner.
__global__ __launch_bounds__(kBlock) void k_bound_obs(DevParams P, float4* __restrict__ posM,
                                                      float4* __restrict__ velV, int* __restrict__ blockCount,
                                                      const uint32_t* __restrict__ sortedKeys, int* __restrict__ counters) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    bool collided = false, owned = false;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), v = p;
    if (s < P.n) {
        p = posM[s]; v = velV[s];
        owned = slot_is_owned(P, sortedKeys, s);
        collided = bound_obs_body(P, p, v) && owned;
        posM[s] = p;
        velV[s] = v;
    }
    count_non_finite(p, v, owned, counters);
    block_count(collided, blockCount);
}

__device__ __forceinline__ void damp_velocity(float4& v) {                 // compute:499-507
    const float speed = sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z);
    const float k = (speed < 0.2f) ? 0.98f : 0.999f;
    v.x *= k; v.y *= k; v.z *= k;
}

// sph_step_frame's tail: CS_VvFullStep + CS_BoundObs (+ CS_VelocityDamping after the last sub-step)
// in one pass over the particles -- the same operations in the same order per particle, so the
// result is bit-identical to the three separate dispatches; 1 read + 1 write instead of 3 + 3.
__global__ __launch_bounds__(kBlock) void k_full_bound_damp(DevParams P, float4* __restrict__ posM,
                                                            float4* __restrict__ velV,
                                                            const float4* __restrict__ acc,
                                                            int* __restrict__ blockCount,
                                                            const uint32_t* __restrict__ sortedKeys, int damp,
                                                            int* __restrict__ counters) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    bool collided = false, owned = false;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), v = p;
    if (s < P.n) {
        p = posM[s]; v = velV[s];
        const float4 a = acc[s];
        const float hdt = 0.5f * P.dt;
        v.x += hdt * a.x; v.y += hdt * a.y; v.z += hdt * a.z;              // compute:332
        owned = slot_is_owned(P, sortedKeys, s);
        collided = bound_obs_body(P, p, v) && owned;
        if (damp) damp_velocity(v);
        posM[s] = p;
        velV[s] = v;
    }
    count_non_finite(p, v, owned, counters);
    block_count(collided, blockCount);
}

// CS_VelocityDamping (compute:492-509)
__global__ __launch_bounds__(kBlock) void k_velocity_damping(int n, float4* __restrict__ velV) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    float4 v = velV[s];
    const float speed = sqrtf(dot3f(v.x, v.y, v.z, v.x, v.y, v.z));
    const float k = (speed < 0.2f) ? 0.98f : 0.999f;
    v.x *= k; v.y *= k; v.z *= k;
    velV[s] = v;
}

// ---- AoS <-> SoA at the ABI (the reference's 72-byte record, id order) ------------------------------
__global__ __launch_bounds__(kBlock) void k_import_aos(int n, const float* __restrict__ aos, StateSet st,
                                                       float4* __restrict__ color) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float* r = aos + (size_t)i * SPH_PARTICLE_FLOATS;
    st.posM[i] = make_float4(r[0], r[1], r[2], r[13]);
    st.velV[i] = make_float4(r[3], r[4], r[5], r[12]);
    st.acc[i] = make_float4(r[6], r[7], r[8], 0.f);
    st.rhoP[i] = make_float2(r[9], r[10]);
    st.rho0[i] = r[11];
    st.origId[i] = (uint32_t)i;
    color[i] = make_float4(r[14], r[15], r[16], r[17]);
}

__global__ __launch_bounds__(kBlock) void k_export_aos(int n, float* __restrict__ aos, StateSet st,
                                                       const float4* __restrict__ color) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const uint32_t id = st.origId[s];
    float* r = aos + (size_t)id * SPH_PARTICLE_FLOATS;
    const float4 p = st.posM[s], v = st.velV[s], a = st.acc[s], c = color[id];
    const float2 d = st.rhoP[s];
    r[0] = p.x; r[1] = p.y; r[2] = p.z;
    r[3] = v.x; r[4] = v.y; r[5] = v.z;
    r[6] = a.x; r[7] = a.y; r[8] = a.z;
    r[9] = d.x; r[10] = d.y; r[11] = st.rho0[s]; r[12] = v.w; r[13] = p.w;
    r[14] = c.x; r[15] = c.y; r[16] = c.z; r[17] = c.w;
}

// SPH.UpdateParticleStaticProperties (SPH.cs:284-298) on the device: per-particle phase id ->
// phase table row -> restDensity / viscosity / mass / colour.
__global__ __launch_bounds__(kBlock) void k_apply_phase_table(int n, StateSet st, float4* __restrict__ color,
                                                              const uint8_t* __restrict__ phaseOfId,
                                                              const sph_phase* __restrict__ table) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const uint32_t id = st.origId[s];
    const sph_phase ph = table[phaseOfId[id]];
    st.rho0[s] = ph.restDensity;
    st.velV[s].w = ph.viscosity;
    st.posM[s].w = ph.mass;
    color[id] = make_float4(ph.color[0], ph.color[1], ph.color[2], ph.color[3]);
}

// The same refresh in slab mode: phases are kept per SLOT there (ids are global, a slab holds a subset), colours are
// not part of a slab's state.  Slots whose phase the table does not know keep their values.
__global__ __launch_bounds__(kBlock) void k_apply_phase_table_slots(int n, StateSet st, const sph_phase* __restrict__ table,
                                                                    int rows) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const int p = st.phase[s];
    if (p >= rows) return;
    const sph_phase ph = table[p];
    st.rho0[s] = ph.restDensity;
    st.velV[s].w = ph.viscosity;
    st.posM[s].w = ph.mass;
}

// Neighbour-list export (parity only): what CS_DensityPressure's loop accepts right now, walked the
// same way the production kernels walk it (row ranges when no bin overflowed, capped cells otherwise).
__global__ __launch_bounds__(kBlock) void k_export_neighbors(DevParams P, const float4* __restrict__ posM,
                                                             const uint32_t* __restrict__ origId,
                                                             const int* __restrict__ cellCum,
                                                             const int* __restrict__ counters,
                                                             int* __restrict__ counts, int* __restrict__ lists,
                                                             int maxNeighbors) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= P.n) return;
    const float4 p = posM[s];
    const uint32_t id = origId[s];
    const bool capped = counters[SPH_CTR_BIN_OVERFLOW] != 0;
    const int3 c = search_cell(P, p.x, p.y, p.z);
    int cnt = 0;
    for (int t = 0; t < (capped ? 27 : 9); t++) {
        const RowRange r = range_at(P, cellCum, c, t, capped, true);
        for (int j = r.a; j < r.e; j++) {
            if (j == s) continue;
            const float4 q = posM[j];
            const float rSq = dist2_exact(p.x - q.x, p.y - q.y, p.z - q.z);
            if (rSq < P.h2) {
                if (cnt < maxNeighbors && lists) lists[(size_t)id * maxNeighbors + cnt] = (int)origId[j];
                cnt++;
            }
        }
    }
    counts[id] = cnt;
}

// Hit-mask export (parity only): decodes what k_density_pressure_rows recorded for the force kernel -- the bit rows
// k_force_xsph_hits walks -- into id lists in visiting order, with the decode of that kernel: row t of a particle
// starts a fresh word, bit b of the row's k-th word is slot row.a + 32 k + b; words 0..15 live in the particle's
// 64-byte line, words 16..23 in the second array.  counts[id] = -1: the particle has no usable mask (it needs more
// than kMaskWords words, or a row of more than 8 words: its wave takes the scan path).
__global__ __launch_bounds__(kBlock) void k_export_hit_lists(DevParams P, const float4* __restrict__ posM,
                                                             const uint32_t* __restrict__ origId,
                                                             const int* __restrict__ cellCum,
                                                             const uint32_t* __restrict__ hitMask, int maskStride,
                                                             int* __restrict__ counts, int* __restrict__ lists,
                                                             int maxNeighbors) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= P.n) return;
    const float4 p = posM[s];
    const uint32_t id = origId[s];
    const int3 c = search_cell(P, p.x, p.y, p.z);
    RowRange rr[9];
    int nWords = 0;
    for (int t = 0; t < 9; t++) {
        rr[t] = row_range(P, cellCum, c, t % 3 - 1, t / 3 - 1, true);
        const int nw = (rr[t].e - rr[t].a + 31) >> 5;
        nWords += nw > 8 ? kMaskWords + 1 : nw;
    }
    if (nWords > kMaskWords) { counts[id] = -1; return; }
    const uint32_t* const lo = hitMask + (size_t)s * kMaskStage;
    const uint32_t* const hi = hitMask + (size_t)maskStride * kMaskStage + (size_t)s * (kMaskWords - kMaskStage) - kMaskStage;
    int cnt = 0, idx = 0;
    for (int t = 0; t < 9; t++) {
        const int nw = (rr[t].e - rr[t].a + 31) >> 5;
        for (int k = 0; k < nw; k++, idx++) {
            uint32_t w = idx < kMaskStage ? lo[idx] : hi[idx];
            while (w) {
                const int j = rr[t].a + (k << 5) + __builtin_ctz(w);
                w &= w - 1;
                if (cnt < maxNeighbors && lists) lists[(size_t)id * maxNeighbors + cnt] = j < P.n ? (int)origId[j] : -2;
                cnt++;
            }
        }
    }
    counts[id] = cnt;
}

// ComputeBuffer.SetData in the MIDDLE of a frame (parity / debugging): the reference's SetData leaves GridCounts and
// GridIndices as they are, so the records are written over the particles where they sit -- slot order and grid stay.
__global__ __launch_bounds__(kBlock) void k_update_aos_in_place(int n, const float* __restrict__ aos, StateSet st,
                                                                float4* __restrict__ color) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const uint32_t id = st.origId[s];
    const float* r = aos + (size_t)id * SPH_PARTICLE_FLOATS;
    st.posM[s] = make_float4(r[0], r[1], r[2], r[13]);
    st.velV[s] = make_float4(r[3], r[4], r[5], r[12]);
    st.acc[s] = make_float4(r[6], r[7], r[8], 0.f);
    st.rhoP[s] = make_float2(r[9], r[10]);
    st.rho0[s] = r[11];
    color[id] = make_float4(r[14], r[15], r[16], r[17]);
}

}  // namespace sph
