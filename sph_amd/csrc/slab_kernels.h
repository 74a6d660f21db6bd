// slab_kernels.h -- multi-GPU slab decomposition support (no counterpart in the reference, which is
// single-GPU).  The domain is cut into x-slabs of whole cell planes, one
// per GPU.  A rank's local particle set = the particles it owns + read-only ghost copies of the
// planes next to its slab.  Once per frame each rank sends ONE message to each slab neighbour
// (its particles near that face: emigrants and ghosts alike, the receiver tells them apart by
// plane) and rebuilds its local set; the frame itself then runs with no further communication.
#pragma once
#include "physics_kernels.h"

namespace sph {

struct SlabRecord {      // == struct sph_slab_record (64 B)
    float4 posM;
    float4 velV;
    float4 accR;         // ax, ay, az, restDensity
    uint32_t id;         // global particle id
    float density, pressure;
    uint32_t phase;
};
static_assert(sizeof(SlabRecord) == 64, "64-byte slab record");

// global cell plane of a position, clamped like CS_BuildGrid clamps it (compute:125-126)
__device__ __forceinline__ int global_plane(const DevParams& P, float x) {
    const int c = (int)floorf((x - P.minBound[0]) / P.cellSize);
    return min(max(c, 0), P.rxg - 1);
}

// wave-aggregated append: one atomic per wave, slot = base + rank among the appending lanes
__device__ __forceinline__ int wave_append(bool pred, int* counter) {
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader);
    return pred ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

__device__ __forceinline__ SlabRecord make_record(const StateSet& st, int s) {
    SlabRecord r;
    r.posM = st.posM[s];
    r.velV = st.velV[s];
    const float4 a = st.acc[s];
    r.accR = make_float4(a.x, a.y, a.z, st.rho0[s]);
    const float2 d = st.rhoP[s];
    r.id = st.origId[s];
    r.density = d.x;
    r.pressure = d.y;
    r.phase = st.phase ? (uint32_t)st.phase[s] : 0u;
    return r;
}

// Records arrive with a phase id.  The packed neighbour records of the force kernel (PackedNbr) take
// mass and viscosity from the phase table, which is only right if every record's own values ARE its
// phase's row: any record that disagrees raises a sticky flag; the host reads it with the survivor count of the next grid
// build and launches the three-gather mode of the force kernel (and the 16-byte candidates of the density kernel).
struct PhaseCheck {
    const sph_phase* table;   // null = no table set
    int rows;
    int* mismatch;
};

__device__ __forceinline__ void store_record(const StateSet& st, int k, const SlabRecord& r, const PhaseCheck& pc) {
    if (st.phase) st.phase[k] = (uint8_t)(r.phase & 255u);
    float mass = r.posM.w, visc = r.velV.w, rho0 = r.accR.w;
    if (pc.table) {
        // A phase table in force is authoritative, like the spawn boxes are in the reference (SPH.cs:284-298 rewrites
        // the static fields of every particle from its box each frame): a record packed before a live edit of its
        // phase arrives with the old values and takes the current ones here.  A phase the table does not know keeps
        // the record's own values and raises the flag (three-gather force kernel, 16-byte density candidates).
        if (r.phase < (uint32_t)pc.rows) {
            const sph_phase ph = pc.table[r.phase];
            mass = ph.mass; visc = ph.viscosity; rho0 = ph.restDensity;
        } else {
            *pc.mismatch = 1;
        }
    }
    st.posM[k] = make_float4(r.posM.x, r.posM.y, r.posM.z, mass);
    st.velV[k] = make_float4(r.velV.x, r.velV.y, r.velV.z, visc);
    st.acc[k] = make_float4(r.accR.x, r.accR.y, r.accR.z, 0.f);
    st.rhoP[k] = make_float2(r.density, r.pressure);
    st.rho0[k] = rho0;
    st.origId[k] = r.id;
}

// Messages to the slab neighbours: previously-owned particles whose CURRENT plane is below
// `leftBelow` (-> left message) or at/above `rightFrom` (-> right message).  counts[0..1].
// A workgroup handles kKeepItems x 256 particles and reserves its part of each message with one
// atomic (the band particles are the two ends of every cell row, so most waves hold a few; one
// atomic per wave would again serialise on the two counters).
constexpr int kKeepItems = 8;
__global__ __launch_bounds__(kBlock) void k_slab_pack(DevParams P, StateSet st, OwnedRef owned,
                                                      int leftBelow, int rightFrom, SlabRecord* __restrict__ left,
                                                      int capLeft, SlabRecord* __restrict__ right, int capRight,
                                                      int* __restrict__ counts) {
    __shared__ int waveTot[2][kBlock / 64];
    __shared__ int blockBase[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = blockIdx.x * (kBlock * kKeepItems) + threadIdx.x;
    unsigned long long mL[kKeepItems], mR[kKeepItems];
    int totL = 0, totR = 0;
#pragma unroll
    for (int it = 0; it < kKeepItems; it++) {
        const int s = first + it * kBlock;
        bool toL = false, toR = false;
        if (s < P.n && was_owned(owned, s)) {
            const int g = global_plane(P, st.posM[s].x);
            toL = g < leftBelow;
            toR = g >= rightFrom;
        }
        mL[it] = __ballot(toL);
        mR[it] = __ballot(toR);
        totL += __popcll(mL[it]);
        totR += __popcll(mR[it]);
    }
    if (lane == 0) { waveTot[0][wave] = totL; waveTot[1][wave] = totR; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int t = (waveTot[threadIdx.x][0] + waveTot[threadIdx.x][1]) + (waveTot[threadIdx.x][2] + waveTot[threadIdx.x][3]);
        blockBase[threadIdx.x] = t ? atomicAdd(&counts[threadIdx.x], t) : 0;
    }
    __syncthreads();
    int kL = blockBase[0], kR = blockBase[1];
    for (int w = 0; w < wave; w++) { kL += waveTot[0][w]; kR += waveTot[1][w]; }
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int it = 0; it < kKeepItems; it++) {
        const int s = first + it * kBlock;
        if ((mL[it] >> lane) & 1ull) {
            const int d = kL + __popcll(mL[it] & below);
            if (d < capLeft) left[d] = make_record(st, s);
        }
        if ((mR[it] >> lane) & 1ull) {
            const int d = kR + __popcll(mR[it] & below);
            if (d < capRight) right[d] = make_record(st, s);
        }
        kL += __popcll(mL[it]);
        kR += __popcll(mR[it]);
    }
}

// ---- exchange overlapped with compute -------------------------------------------------------------------
// The face messages need the END-of-frame state of the particles near a face, and feed the NEXT grid build.
// To send them while the frame is still running, the last sub-step's CS_ForceXSPH is launched twice: first
// the "band" workgroups (256 consecutive slots holding at least one owned particle binned within
// ghost + margin planes of a face), then the rest.  Between the two, k_slab_pack_final computes the tail of
// the frame (full step, boundary, damping) for the band particles IN REGISTERS and writes the records; the
// state arrays are not touched, so the second force launch reads exactly what an unsplit launch reads, and
// the ordinary tail kernel runs over everything afterwards (the band's tail is computed twice: cheap).
__global__ __launch_bounds__(kBlock) void k_band_flags(int n, const uint32_t* __restrict__ sortedKeys, int rx, int xoff,
                                                       int ownLo, int ownHi, int bandLo, int bandHi,
                                                       uint8_t* __restrict__ wgBand, int* __restrict__ nBand) {
    const int s = blockIdx.x * kBlock + threadIdx.x;
    int band = 0;
    if (s < n) {
        const int plane = (int)(sortedKeys[s] % (uint32_t)rx);
        const int g = plane + xoff;
        band = plane >= ownLo && plane < ownHi && (g < bandLo || g >= bandHi);
    }
    band = __syncthreads_or(band);
    if (threadIdx.x == 0) {
        wgBand[blockIdx.x] = band ? 1 : 0;
        if (band) atomicAdd(nBand, 1);      // statistic only (how much of the force kernel runs before the pack)
    }
}

// state of slot s after the fused tail of the frame (k_full_bound_damp), computed without writing it
__device__ __forceinline__ void final_state(const DevParams& P, const StateSet& st, int s, int damp, float4& p, float4& v) {
    p = st.posM[s];
    v = st.velV[s];
    const float4 a = st.acc[s];
    const float hdt = 0.5f * P.dt;
    v.x += hdt * a.x; v.y += hdt * a.y; v.z += hdt * a.z;
    (void)bound_obs_body(P, p, v);
    if (damp) damp_velocity(v);
}

// k_slab_pack on the band workgroups, with the tail of the frame applied on the fly.  `st.velV` = the
// velocities CS_ForceXSPH has just written.
__global__ __launch_bounds__(kBlock) void k_slab_pack_final(DevParams P, StateSet st, OwnedRef owned,
                                                            const uint8_t* __restrict__ wgBand, int damp, int leftBelow,
                                                            int rightFrom, SlabRecord* __restrict__ left, int capLeft,
                                                            SlabRecord* __restrict__ right, int capRight,
                                                            int* __restrict__ counts) {
    __shared__ int waveTot[2][kBlock / 64];
    __shared__ int blockBase[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = blockIdx.x * (kBlock * kKeepItems) + threadIdx.x;
    unsigned long long mL[kKeepItems], mR[kKeepItems];
    int totL = 0, totR = 0;
#pragma unroll
    for (int it = 0; it < kKeepItems; it++) {
        const int s = first + it * kBlock;
        bool toL = false, toR = false;
        if (s < P.n && wgBand[s >> 8] && was_owned(owned, s)) {     // wave-uniform: 64 | 256
            float4 p, v;
            final_state(P, st, s, damp, p, v);
            const int g = global_plane(P, p.x);
            toL = g < leftBelow;
            toR = g >= rightFrom;
        }
        mL[it] = __ballot(toL);
        mR[it] = __ballot(toR);
        totL += __popcll(mL[it]);
        totR += __popcll(mR[it]);
    }
    if (lane == 0) { waveTot[0][wave] = totL; waveTot[1][wave] = totR; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int t = (waveTot[threadIdx.x][0] + waveTot[threadIdx.x][1]) + (waveTot[threadIdx.x][2] + waveTot[threadIdx.x][3]);
        blockBase[threadIdx.x] = t ? atomicAdd(&counts[threadIdx.x], t) : 0;
    }
    __syncthreads();
    int kL = blockBase[0], kR = blockBase[1];
    for (int w = 0; w < wave; w++) { kL += waveTot[0][w]; kR += waveTot[1][w]; }
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int it = 0; it < kKeepItems; it++) {
        const int s = first + it * kBlock;
        const bool l = (mL[it] >> lane) & 1ull, r = (mR[it] >> lane) & 1ull;
        if (l || r) {
            SlabRecord rec = make_record(st, s);
            float4 p, v;
            final_state(P, st, s, damp, p, v);
            rec.posM = p;
            rec.velV = v;
            if (l) { const int d = kL + __popcll(mL[it] & below); if (d < capLeft) left[d] = rec; }
            if (r) { const int d = kR + __popcll(mR[it] & below); if (d < capRight) right[d] = rec; }
        }
        kL += __popcll(mL[it]);
        kR += __popcll(mR[it]);
    }
}

// Received records become source slots base .. base+nrec-1 of the current set; which of them (and of
// the old slots) survive is decided by the next grid build (SlabFilter in k_hash_count).
__global__ __launch_bounds__(kBlock) void k_slab_append_at(const SlabRecord* __restrict__ rec, int nrec, StateSet dst,
                                                           int base, PhaseCheck pc) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < nrec) store_record(dst, base + i, rec[i], pc);
}

// Export of the local set as records (owned only, or everything), compacted.
__global__ __launch_bounds__(kBlock) void k_slab_export(DevParams P, StateSet st, OwnedRef owned,
                                                        int ownedOnly, SlabRecord* __restrict__ out, int cap,
                                                        int* __restrict__ count) {
    const int s = blockIdx.x * kBlock + threadIdx.x;
    const bool take = s < P.n && (!ownedOnly || was_owned(owned, s));
    const int k = wave_append(take, count);
    if (take && k < cap) out[k] = make_record(st, s);
}

__global__ __launch_bounds__(kBlock) void k_slab_import(int n, const SlabRecord* __restrict__ rec, StateSet dst,
                                                        PhaseCheck pc) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) store_record(dst, i, rec[i], pc);
}

}  // namespace sph
