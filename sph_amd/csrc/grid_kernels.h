// grid_kernels.h -- spatial-hash grid build for gfx950: cell hash, counting sort by cell over the
// cumulative cell table, (cell, id) canonicalisation and the state reorder.
//
// Replaces CS_ClearGrid + CS_BuildGrid (compute:104-136).  The reference appends ids to
// fixed-capacity bins with InterlockedAdd; here particles are sorted by cell so that a bin is a
// contiguous range [cellCum[c], cellCum[c+1]) of slots.  Filling the reference's bins in ascending
// particle id (ORACLE rule O1) yields exactly the (cell, id)-sorted sequence built here.
#pragma once
#include "sph_common.h"

namespace sph {

// ---- cell hash (compute:125-127) --------------------------------------------------------------
// sub -> IEEE divide -> floor -> int -> clamp -> integer linearise, verbatim (rule O5); this TU is
// compiled with -ffp-contract=off and hipcc's default correctly-rounded fp32 divide.
__device__ __forceinline__ int3 cell_unclamped(const DevParams& P, float x, float y, float z) {
    int3 c;
    c.x = (int)floorf((x - P.minBound[0]) / P.cellSize);   // v_cvt_i32_f32 saturates, NaN -> 0
    c.y = (int)floorf((y - P.minBound[1]) / P.cellSize);
    c.z = (int)floorf((z - P.minBound[2]) / P.cellSize);
    return c;
}

// ---- counting sort by cell ---------------------------------------------------------------------------
// The table the neighbour search needs anyway -- T[c] = number of particles whose cell index is
// <= c -- IS the offset table of a counting sort by cell, so the build is:
//   k_hash_count : cell of every slot; one returning atomicAdd per particle on T[cell] gives its
//                  arrival number inside the cell (order of arrival is arbitrary);
//   k_cum_*      : inclusive prefix sum over the cells (int4, 4096 cells per workgroup) into the cumulative table;
//   k_place      : bucket slot = cellCum[cell] + arrival;
//   k_rank_move (below) then orders every cell by particle id, which removes the arbitrariness:
//                  the result is the unique (cell, id) order whatever order the atomics ran in.
// Particles are still in last frame's (cell, id) order when this runs, so consecutive lanes hit the
// same or neighbouring table entries; runs of equal cells inside a wave share one atomic.
// Kernels index the table through cellCum = T - 1 (cellCum[c] = #particles with cell < c): the bin
// of cell c is [cellCum[c], cellCum[c+1]) and any run of consecutive cells [c0, c1] is the single
// slot range [cellCum[c0], cellCum[c1+1]) -- what the 3-cells-per-row search reads with two loads.
// Slab mode: which of the source slots take part in this build.  Slots below nOld held particles at
// the last build: those that were OWNED then (their ghosts are stale copies) and whose plane is now in
// [keepLo, keepHi) stay; slots from nOld on are records just received from the slab neighbours and
// stay if their plane is in range.  Everything else gets no key, is not counted and is not placed:
// the counting sort itself is the compaction, no separate copy of the survivors.
// Who owned a slot at the last grid build: its key there, decoded with THAT build's local grid (the slab's
// cut may have moved since: sph_slab_configure before the next build).  keys == null: everything was owned.
constexpr int kScanVec = 4;                              // int4 per thread per round
constexpr int kScanRounds = 4;
constexpr int kScanTile = kBlock * kScanVec * kScanRounds;   // 4096 cells per block
constexpr int kScanTileLog = 12;
static_assert(1 << kScanTileLog == kScanTile, "tile of the table scan");

struct OwnedRef {
    const uint32_t* keys;
    int rx, ownLo, ownHi;            // local resolution and owned local planes of the build that wrote `keys`
};

__device__ __forceinline__ bool was_owned(const OwnedRef& R, int s) {
    if (!R.keys) return true;
    const int plane = (int)(R.keys[s] % (uint32_t)R.rx);
    return plane >= R.ownLo && plane < R.ownHi;
}

struct SlabFilter {
    int active, keepLo, keepHi, nOld;
    OwnedRef old;                    // the last build (keys null before the first one)
    // overlap mode: the last frame packed its face messages from the band workgroups only (bandWg[slot >> 8]);
    // an owned particle outside them that ended in a plane < sendLo or >= sendHi was NOT sent: counted
    const uint8_t* bandWg;
    int sendLo, sendHi;
};

__global__ __launch_bounds__(kBlock) void k_hash_count(DevParams P, const float4* __restrict__ posM,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ arrival,
                                                       int* __restrict__ T, int* __restrict__ counters, SlabFilter F,
                                                       uint8_t* __restrict__ tileTouched) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool oog = false;
    uint32_t key = 0xFFFFFFFFu;                          // lanes past the end: a run of their own, no atomic
    bool take = s < P.n;
    bool missed = false;
    if (take && F.active) {
        if (s < F.nOld) take = was_owned(F.old, s);
        if (take) {
            const int g = min(max((int)floorf((posM[s].x - P.minBound[0]) / P.cellSize), 0), P.rxg - 1);
            missed = F.bandWg && s < F.nOld && !F.bandWg[s >> 8] && (g < F.sendLo || g >= F.sendHi);
            take = g >= F.keepLo && g < F.keepHi;
        }
        if (!take) keys[s] = 0xFFFFFFFFu;
    }
    if (F.bandWg) {
        const unsigned long long mm = __ballot(missed);
        if (mm && lane == 0) atomicAdd(&counters[SPH_CTR_BAND_MISSES], __popcll(mm));
    }
    if (take) {
        float4 p = posM[s];
        int3 c = cell_unclamped(P, p.x, p.y, p.z);
        oog = c.x < 0 || c.x >= P.rxg || c.y < 0 || c.y >= P.ry || c.z < 0 || c.z >= P.rz;
        int x = min(max(c.x, 0), P.rxg - 1) - P.xoff;   // global clamp (compute:126), then local plane
        x = min(max(x, 0), P.rx - 1);                    // memory safety only: a slab never holds such a particle
        int y = min(max(c.y, 0), P.ry - 1);
        int z = min(max(c.z, 0), P.rz - 1);
        key = (uint32_t)(x + P.rx * row_index(P.ryT, y, z));
        keys[s] = key;
    }
    // Wave-aggregated: consecutive lanes with the same cell (the usual case -- the particles are
    // still in last frame's cell order) share ONE atomic, issued by the first lane of the run.
    const uint32_t prev = __shfl_up(key, 1);
    const bool head = lane == 0 || prev != key;
    const unsigned long long heads = __ballot(head);
    const int headLane = 63 - __builtin_clzll(heads & ((2ull << lane) - 1ull));      // bit 0 is always set
    const unsigned long long above = (headLane == 63) ? 0ull : heads & ~((2ull << headLane) - 1ull);
    const int runLen = (above ? __builtin_ctzll(above) : 64) - headLane;
    int base = 0;
    if (head && key != 0xFFFFFFFFu) {
        base = atomicAdd(&T[key], runLen);
        tileTouched[key >> kScanTileLog] = 1;            // plain store: every writer stores the same value
    }
    base = __shfl(base, headLane);
    if (s < P.n) arrival[s] = (uint32_t)(base + (lane - headLane));
    unsigned long long m = __ballot(oog);
    if (m && lane == 0) atomicAdd(&counters[SPH_CTR_OUT_OF_GRID], __popcll(m));
}

// CS_ClearGrid (compute:104-111) as a dispatchable kernel: zeroes the cumulative table (int4 at a time).  A grid
// build does not need it: the COUNT array the build accumulates into is left all-zero by k_cum_write (which reads
// every touched tile anyway), the cumulative table is a second array that k_cum_write overwrites completely, and the
// per-build counters alternate between two sets, the idle one zeroed by k_cum_block_sum.
__global__ __launch_bounds__(kBlock) void k_clear_table(int4* __restrict__ T4, long long n4, int* __restrict__ counters) {
    const long long base = (long long)blockIdx.x * (kBlock * 4) + threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const long long i = base + (long long)r * kBlock;
        if (i < n4) T4[i] = make_int4(0, 0, 0, 0);
    }
    if (counters && blockIdx.x == 0 && threadIdx.x >= SPH_CTR_BIN_OVERFLOW && threadIdx.x < SPH_CTR_COUNT)
        counters[threadIdx.x] = 0;
}


__device__ __forceinline__ int wave_incl_sum(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// cells4 = number of int4 groups (the table is padded to a multiple of 4 cells with zeros)
// Tiles no particle was counted into (tileTouched == 0: k_hash_count marks the tiles it adds to) are all-zero and
// are not read.  Workgroup 0 also zeroes the per-build counters of the NEXT build (the idle set).
__global__ __launch_bounds__(kBlock) void k_cum_block_sum(const int4* __restrict__ T4, long long cells4,
                                                          int* __restrict__ blockSum,
                                                          const uint8_t* __restrict__ tileTouched,
                                                          int* __restrict__ idleCounters) {
    __shared__ int ws[4];
    if (blockIdx.x == 0 && threadIdx.x >= SPH_CTR_BIN_OVERFLOW && threadIdx.x < SPH_CTR_COUNT)
        idleCounters[threadIdx.x] = 0;
    if (!tileTouched[blockIdx.x]) {                      // workgroup-uniform
        if (threadIdx.x == 0) blockSum[blockIdx.x] = 0;
        return;
    }
    const long long base = (long long)blockIdx.x * (kBlock * kScanRounds) + threadIdx.x;
    int m = 0;
#pragma unroll
    for (int r = 0; r < kScanRounds; r++) {
        const long long i = base + (long long)r * kBlock;
        if (i < cells4) {
            const int4 v = T4[i];
            m += (v.x + v.y) + (v.z + v.w);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o);
    if (lane == 0) ws[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) blockSum[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// Inclusive prefix sum of one 4096-cell tile of the COUNT array into the cumulative table `out`; the counts are
// zeroed on the way (so the next build finds the array all-zero and no clearing pass exists).  The tile's seed -- the
// sum of all earlier tiles -- is reduced here from the per-tile sums (a few hundred to a few thousand ints,
// L2-resident) instead of by a separate one-workgroup scan kernel.  An untouched tile (no particle counted into it)
// is not read: its 4096 entries are the seed.
__global__ __launch_bounds__(kBlock) void k_cum_write(int4* __restrict__ T4, int4* __restrict__ out4, long long cells4,
                                                      const int* __restrict__ blockSum,
                                                      uint8_t* __restrict__ tileTouched) {
    __shared__ int ws[4];
    __shared__ int seed[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += kBlock) part += blockSum[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) seed[wave] = part;
    const bool touched = tileTouched[blockIdx.x] != 0;                      // workgroup-uniform
    __syncthreads();
    int carry = (seed[0] + seed[1]) + (seed[2] + seed[3]);
    const long long base = (long long)blockIdx.x * (kBlock * kScanRounds) + threadIdx.x;
    if (!touched) {
#pragma unroll
        for (int r = 0; r < kScanRounds; r++) {
            const long long i = base + (long long)r * kBlock;
            if (i < cells4) out4[i] = make_int4(carry, carry, carry, carry);
        }
        return;
    }
#pragma unroll 1
    for (int r = 0; r < kScanRounds; r++) {
        const long long i = base + (long long)r * kBlock;
        int4 v = (i < cells4) ? T4[i] : make_int4(0, 0, 0, 0);
        if (i < cells4) T4[i] = make_int4(0, 0, 0, 0);
        v.y += v.x; v.z += v.y; v.w += v.z;                                 // thread-local
        const int incl = wave_incl_sum(v.w, lane);
        __syncthreads();                                                    // ws free again
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        int off = carry + (incl - v.w);
        for (int w = 0; w < wave; w++) off += ws[w];
        if (i < cells4) out4[i] = make_int4(v.x + off, v.y + off, v.z + off, v.w + off);
        carry += (ws[0] + ws[1]) + (ws[2] + ws[3]);
    }
    if (threadIdx.x == 0) tileTouched[blockIdx.x] = 0;                      // only this workgroup reads it
}

// bucket slot of every particle; the slot's key, source slot and id travel with it
__global__ __launch_bounds__(kBlock) void k_place(int n, const uint32_t* __restrict__ keys,
                                                  const uint32_t* __restrict__ arrival,
                                                  const uint32_t* __restrict__ origId,
                                                  const int* __restrict__ cellCum,
                                                  uint32_t* __restrict__ sortedKeys, uint32_t* __restrict__ perm,
                                                  uint32_t* __restrict__ sid) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const uint32_t key = keys[s];
    if (key == 0xFFFFFFFFu) return;                      // slab mode: not part of this build (SlabFilter)
    const int t = cellCum[key] + (int)arrival[s];
    sortedKeys[t] = key;
    perm[t] = (uint32_t)s;
    sid[t] = origId[s];
}

// Within each cell, order by particle id (bucket order is the atomics' arrival order): rank = #ids in
// the cell smaller than mine; cells hold ~1-15 particles.  Also counts cells whose population
// exceeds the reference's bin capacity (compute:131).  The thread of bucket slot k then carries its
// particle's state from the old slot straight to the final one (a scatter inside the cell's range).
// HALF (sph_step_frame) = also do the first CS_VvHalfStep (compute:309-319) on the way: the same two
// statements as k_half_step, so positions and velocities are bit-identical to reorder-then-half-step;
// acceleration is consumed here and density / pressure are dead (the neighbour kernels overwrite all
// three before anything reads them), so they are not carried over.
constexpr int kRankSerialMax = 64;   // cells up to this many particles are ranked by a serial scan per thread
template <bool HALF>
__global__ __launch_bounds__(kBlock) void k_rank_move(const uint32_t* __restrict__ sortedKeys,
                                                      const uint32_t* __restrict__ perm,
                                                      const uint32_t* __restrict__ sid, int n, int cap, float dt,
                                                      const int* __restrict__ cellCum, StateSet src, StateSet dst,
                                                      int* __restrict__ counters, float* __restrict__ pos3) {
    const int k = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = k < n;                            // lanes past the end stay active for the wave-wide ranking below
    const int kk = valid ? k : n - 1;
    const uint32_t c = sortedKeys[kk];
    const int a = cellCum[c], e = cellCum[c + 1];
    const uint32_t me = sid[kk];
    int r = 0;
    // Usual cells (a handful of particles): every thread scans its own cell.  CROWDED cells -- every escaped particle
    // is clamped into a boundary cell (compute:126) and a NaN position hashes to cell 0, so a diverged run can put
    // 10^5..10^6 particles into one bin -- are ranked by the wave together: the cell's ids are read 64 at a time
    // with ONE coalesced load per chunk and every lane of that cell counts the chunk's ids below its own through
    // lane broadcasts (O(pop^2 / 64) instructions and O(pop^2 / 4096) loads per cell instead of O(pop^2) dependent
    // loads per thread, which at 10^6 particles in a bin was effectively a hang).
    const bool crowded = valid && e - a > kRankSerialMax;
    if (!crowded)
        for (int j = a; j < e; j++) r += (sid[j] < me) ? 1 : 0;
    const int lane = threadIdx.x & 63;
    unsigned long long pending = __ballot(crowded);
    while (pending) {                                    // wave-uniform: one trip per distinct crowded cell of this wave
        const int src = __builtin_ctzll(pending);
        const int la = __shfl(a, src), le = __shfl(e, src);
        const bool mine = crowded && a == la;
        int rb = 0;
        for (int base = la; base < le; base += 64) {
            const int idx = base + lane;
            const uint32_t v = idx < le ? sid[idx] : 0xFFFFFFFFu;     // ids are < 2^28: the filler is below nobody
#pragma unroll
            for (int l = 0; l < 64; l++) rb += (uint32_t)__builtin_amdgcn_readlane((int)v, l) < me ? 1 : 0;
        }
        if (mine) r = rb;
        pending &= ~__ballot(mine);
    }
    if (!valid) return;
    if (k == a && e - a > cap) atomicAdd(&counters[SPH_CTR_BIN_OVERFLOW], 1);
    const uint32_t s = perm[k];
    const int d = a + r;
    float4 p = src.posM[s], v = src.velV[s];
    if (HALF) {
        const float4 acc = src.acc[s];
        const float hdt = 0.5f * dt;
        v.x += hdt * acc.x; v.y += hdt * acc.y; v.z += hdt * acc.z;
        p.x += v.x * dt; p.y += v.y * dt; p.z += v.z * dt;
        // positions alone, 12 bytes per slot: what the density kernel's candidate loads read (physics_kernels.h)
        if (pos3) *(Float3U*)(pos3 + 3 * (size_t)d) = Float3U{p.x, p.y, p.z};        // one 12-byte store
    } else {
        dst.acc[d] = src.acc[s];
        dst.rhoP[d] = src.rhoP[s];
    }
    dst.posM[d] = p;
    dst.velV[d] = v;
    dst.rho0[d] = src.rho0[s];
    dst.origId[d] = me;
    if (src.phase) dst.phase[d] = src.phase[s];
}

// cell_of_particle export: keys are per slot and in the tiled memory order; the caller wants the
// reference's cell index (compute:127) per id
__global__ __launch_bounds__(kBlock) void k_scatter_by_id(DevParams P, const uint32_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ origId,
                                                          uint32_t* __restrict__ out) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= P.n) return;
    const uint32_t key = keys[s];
    const int x = (int)(key % (uint32_t)P.rx), row = (int)(key / (uint32_t)P.rx);
    const int tile = row >> (kTileYLog + kTileZLog), in = row & (kTileY * kTileZ - 1);
    const int y = ((tile % P.ryT) << kTileYLog) + (in & (kTileY - 1));
    const int z = ((tile / P.ryT) << kTileZLog) + (in >> kTileYLog);
    out[origId[s]] = (uint32_t)(x + y * P.rx + z * P.rx * P.ry);
}

// Particles per local cell plane at the last grid build, from the cumulative table alone (slab mode: what the
// driver cuts the slabs by).  rows = table rows (rows of the padding tiles are empty); a workgroup sums
// kPlaneRows rows, its threads striding over x (coalesced: x is the fastest index of a key).
constexpr int kPlaneRows = 64;
__global__ __launch_bounds__(kBlock) void k_plane_counts(const int* __restrict__ cellCum, int rx, long long rows,
                                                         int* __restrict__ out) {
    const long long r0 = (long long)blockIdx.x * kPlaneRows;
    const long long r1 = r0 + kPlaneRows < rows ? r0 + kPlaneRows : rows;
    for (int x = threadIdx.x; x < rx; x += kBlock) {
        int sum = 0;
        for (long long r = r0; r < r1; r++) {
            const long long b = r * rx + x;
            sum += cellCum[b + 1] - cellCum[b];
        }
        if (sum) atomicAdd(&out[x], sum);
    }
}

}  // namespace sph
