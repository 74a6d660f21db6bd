// sph_common.h -- shared host/device declarations of libsph_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sph_hip.h"

namespace sph {

constexpr int kBlock = 256;        // threads per workgroup = 4 wave64
constexpr int kWave = 64;
// The density kernel loads candidates four slots at a time from a 32-bit byte offset without
// clamping to the row's end: posM is allocated kSlotPad entries longer than the particle capacity,
// and capacities stay below 2^28 particles (16-byte entries, 4 GiB of offsets).
constexpr int kSlotPad = 4;
constexpr int kMaxParticles = (1 << 28) - 8;

// Kernel-side copy of the cbuffer (compute:47-81) plus the O4 derived constants, passed by
// value as a kernel argument (lands in SGPRs / the kernarg segment; no constant-buffer upload).
struct DevParams {
    int   n;
    float dt;
    float soundSpeed, gamma, h, gravity;
    float sigma, stThreshold, xsphEps, collisionDamping;
    float obstacleRadius;
    float obstaclePos[3];
    float boundsCenter[3], boundsSize[3];
    int   rx, ry, rz, cap;   // LOCAL grid (what the cell tables index); rx == rxg on one GPU
    // slab decomposition: the local grid is planes [xoff, xoff + rx) of a global grid rxg wide;
    // planes [ownLo, ownHi) (local coordinates) are owned, the rest are ghost planes
    int   rxg, xoff, ownLo, ownHi;
    int   ryT;               // number of row tiles along y (see row_index)
    int   xcdRun;            // consecutive workgroups of a neighbour kernel kept on one XCD (xcd_tile)
    float cellSize;
    float minBound[3];
    float M[12];      // rows 0..2 of _BoundaryMatrix    (row 3 is never read: compute:351, 396 take .xyz)
    float Minv[12];   // rows 0..2 of _BoundaryInvMatrix
    int   obb;        // length(_BoundaryHalfExtents) > 0.001f  (compute:349)
    // derived once on the host in fp32 (ORACLE rule O4)
    float h2, poly6, spikyGradFactor, viscLapFactor, lapColFactor;
    // slab mode, exchange overlapped with compute: workgroup wg of CS_ForceXSPH runs only if
    // wgBand[wg] == wgPhase (null: every workgroup runs)
    const uint8_t* wgBand;
    int   wgPhase;
    // slab mode, last sub-step of a fused frame: keys of the last grid build; CS_ForceXSPH skips the ghost slots
    // (their state is dropped by the next exchange, nobody reads what the force kernel would write)
    const uint32_t* ghostKeys;
};

// ---- memory order of the cells ("row tiles") --------------------------------------------------------
// The reference linearises cells as x + rx*(y + ry*z) (compute:127).  Keeping that order in memory
// puts the rows z-1 / z+1 of a particle's 27-cell neighbourhood a whole z-plane away (tens of MB at
// 8 M particles): fine while everything sits in the 256 MB Infinity Cache, 2-3x slower per particle
// once it does not.  Here a (y, z) row keeps its x-contiguity (the neighbour search reads the three
// x-cells of a row as ONE slot range), but the rows are laid out in tiles of kTileY x kTileZ rows:
//   key = x + rx * row_index(y, z),  row_index = tile(y>>a, z>>b) * (kTileY*kTileZ) + (z&mz)*kTileY + (y&my)
// so that eight of the nine rows a particle reads are (mostly) within a few hundred KB of its own.
// Only the layout changes: bins, visiting order and every exported quantity are the reference's.
#ifndef SPH_TILE_Y_LOG
#define SPH_TILE_Y_LOG 3
#endif
#ifndef SPH_TILE_Z_LOG
#define SPH_TILE_Z_LOG 3
#endif
constexpr int kTileYLog = SPH_TILE_Y_LOG, kTileZLog = SPH_TILE_Z_LOG;
constexpr int kTileY = 1 << kTileYLog, kTileZ = 1 << kTileZLog;

__host__ __device__ __forceinline__ int row_index(int ryT, int y, int z) {
    return (((z >> kTileZLog) * ryT + (y >> kTileYLog)) << (kTileYLog + kTileZLog)) +
           ((z & (kTileZ - 1)) << kTileYLog) + (y & (kTileY - 1));
}
__host__ __device__ __forceinline__ int tiles_along(int r, int log) { return (r + (1 << log) - 1) >> log; }

struct __attribute__((packed, aligned(4))) Float3U { float x, y, z; };   // 12-byte access at a 4-byte aligned address

// SoA particle state, one "set".  Everything is indexed by SLOT (position in the
// (cell, id)-sorted order of the last grid build; slot == id right after an upload).
struct StateSet {
    float4*   posM;    // x, y, z, mass
    float4*   velV;    // vx, vy, vz, viscosity
    float4*   acc;     // ax, ay, az, (unused)
    float2*   rhoP;    // density, pressure
    float*    rho0;    // rest density
    uint32_t* origId;  // particle id living in this slot
    uint8_t*  phase;   // slab mode: phase (spawn box) of the particle in this slot; elsewhere phases are kept by id
};

// Debug builds only (-DSPH_WG_TRACE, scripts/gpu_wave_trace.sh): every wave of the two neighbour kernels leaves its
// begin / end time (100 MHz wall clock) and where it ran, to see how much of a launch is tail.
#ifdef SPH_WG_TRACE
constexpr int kTraceWaves = 1 << 17;
__device__ unsigned long long g_waveTrace[2][kTraceWaves][3];
struct WaveTrace {
    int k;
    unsigned long long t0;
    __device__ __forceinline__ explicit WaveTrace(int kernel) : k(kernel), t0(wall_clock64()) {}
    __device__ __forceinline__ ~WaveTrace() {
        if ((threadIdx.x & 63) == 0) {
            const size_t w = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
            if (w < (size_t)kTraceWaves) {
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
                const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
                g_waveTrace[k][w][0] = t0;
                g_waveTrace[k][w][1] = wall_clock64();
                g_waveTrace[k][w][2] = ((unsigned long long)xcc << 32) | hw;
            }
        }
    }
};
#define SPH_WAVE_TRACE(k) WaveTrace waveTrace_(k)
#else
#define SPH_WAVE_TRACE(k)
#endif

}  // namespace sph
