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
    float cellSize;
    float minBound[3];
    float M[12];      // rows 0..2 of _BoundaryMatrix    (row 3 is never read: compute:351, 396 take .xyz)
    float Minv[12];   // rows 0..2 of _BoundaryInvMatrix
    int   obb;        // length(_BoundaryHalfExtents) > 0.001f  (compute:349)
    // derived once on the host in fp32 (ORACLE rule O4)
    float h2, poly6, spikyGradFactor, viscLapFactor, lapColFactor;
};

// SoA particle state, one "set".  Everything is indexed by SLOT (position in the
// (cell, id)-sorted order of the last grid build; slot == id right after an upload).
struct StateSet {
    float4*   posM;    // x, y, z, mass
    float4*   velV;    // vx, vy, vz, viscosity
    float4*   acc;     // ax, ay, az, (unused)
    float2*   rhoP;    // density, pressure
    float*    rho0;    // rest density
    uint32_t* origId;  // particle id living in this slot
};

}  // namespace sph
