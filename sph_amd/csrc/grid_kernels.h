// grid_kernels.h -- spatial-hash grid build for gfx950: cell hash, stable LSD radix sort by
// cell, cell start/end ranges, (cell, id) canonicalisation and the state reorder.
//
// Replaces CS_ClearGrid + CS_BuildGrid (compute:104-136).  The reference appends ids to
// fixed-capacity bins with InterlockedAdd; here particles are sorted by cell so that a bin is a
// contiguous range [cellStart, cellEnd) of slots.  Filling the reference's bins in ascending
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

__global__ __launch_bounds__(kBlock) void k_hash(DevParams P, const float4* __restrict__ posM,
                                                 uint32_t* __restrict__ keys, int* __restrict__ counters) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    bool oog = false;
    if (s < P.n) {
        float4 p = posM[s];
        int3 c = cell_unclamped(P, p.x, p.y, p.z);
        oog = c.x < 0 || c.x >= P.rxg || c.y < 0 || c.y >= P.ry || c.z < 0 || c.z >= P.rz;
        int x = min(max(c.x, 0), P.rxg - 1) - P.xoff;   // global clamp (compute:126), then local plane
        x = min(max(x, 0), P.rx - 1);                    // memory safety only: a slab never holds such a particle
        int y = min(max(c.y, 0), P.ry - 1);
        int z = min(max(c.z, 0), P.rz - 1);
        keys[s] = (uint32_t)(x + y * P.rx + z * P.rx * P.ry);
    }
    unsigned long long m = __ballot(oog);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(&counters[SPH_CTR_OUT_OF_GRID], __popcll(m));
}

// ---- LSD radix sort, 8-bit digits, stable ---------------------------------------------------------
// One pass = histogram -> scan -> scatter.  A block owns a tile of kSortTile consecutive keys;
// wave w owns keys [w*512, w*512+512) of the tile and walks them 64 at a time, so (wave,
// iteration, lane) order is memory order and the ranking below is stable.
constexpr int kSortItems = 8;
constexpr int kSortTile = kBlock * kSortItems;   // 2048 keys per block
constexpr int kRadix = 256;

__global__ __launch_bounds__(kBlock) void k_sort_hist(const uint32_t* __restrict__ keys, int n, int shift,
                                                      uint32_t* __restrict__ blockHist, int numBlocks) {
    __shared__ uint32_t hist[kRadix];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int base = blockIdx.x * kSortTile + wave * (kSortItems * kWave) + lane;
#pragma unroll
    for (int it = 0; it < kSortItems; it++) {
        int i = base + it * kWave;
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    blockHist[(size_t)threadIdx.x * numBlocks + blockIdx.x] = hist[threadIdx.x];   // digit-major
}

// Exclusive scan of blockHist along blocks, one workgroup per digit; digitTotal[d] = row sum.
__global__ __launch_bounds__(kBlock) void k_sort_scan_rows(uint32_t* __restrict__ blockHist, int numBlocks,
                                                           uint32_t* __restrict__ digitTotal) {
    __shared__ uint32_t waveSum[4];
    __shared__ uint32_t carry;
    uint32_t* row = blockHist + (size_t)blockIdx.x * numBlocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < numBlocks; base += kBlock) {
        int i = base + threadIdx.x;
        uint32_t v = (i < numBlocks) ? row[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) waveSum[wave] = incl;
        __syncthreads();
        uint32_t off = carry;
        for (int w = 0; w < wave; w++) off += waveSum[w];
        if (i < numBlocks) row[i] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) digitTotal[blockIdx.x] = carry;
}

// Stable scatter.  vals_in == nullptr means "value = index" (first pass).
__global__ __launch_bounds__(kBlock) void k_sort_scatter(const uint32_t* __restrict__ keys_in,
                                                         const uint32_t* __restrict__ vals_in,
                                                         uint32_t* __restrict__ keys_out,
                                                         uint32_t* __restrict__ vals_out, int n, int shift,
                                                         const uint32_t* __restrict__ blockHist, int numBlocks,
                                                         const uint32_t* __restrict__ digitTotal) {
    // per-wave digit counters: only the owning wave touches its row, and a wave's LDS operations
    // execute in issue order, so the ranking loop needs no workgroup barrier (volatile keeps the
    // compiler from reordering the read-modify-write across iterations)
    __shared__ volatile uint32_t waveCnt[4][kRadix];
    __shared__ uint32_t digitBase[kRadix];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int w = 0; w < 4; w++) waveCnt[w][threadIdx.x] = 0;
    {   // exclusive scan of the 256 digit totals (tiny; every block redoes it)
        uint32_t v = digitTotal[threadIdx.x];
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        __shared__ uint32_t ws[4];
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        uint32_t off = 0;
        for (int w = 0; w < wave; w++) off += ws[w];
        digitBase[threadIdx.x] = off + incl - v;
    }
    __syncthreads();

    uint32_t key[kSortItems], val[kSortItems], rank[kSortItems];
    const int base = blockIdx.x * kSortTile + wave * (kSortItems * kWave) + lane;
    const unsigned long long ltMask = (1ull << lane) - 1ull;
#pragma unroll
    for (int it = 0; it < kSortItems; it++) {
        int i = base + it * kWave;
        bool valid = i < n;
        key[it] = valid ? keys_in[i] : 0xFFFFFFFFu;
        val[it] = valid ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
    }
#pragma unroll
    for (int it = 0; it < kSortItems; it++) {
        const bool valid = base + it * kWave < n;
        const uint32_t d = (key[it] >> shift) & 255u;
        // lanes holding the same digit (wave-wide match via 8 ballots)
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            bool bit = (d >> b) & 1u;
            unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        uint32_t before = 0;
        if (valid) {
            before = waveCnt[wave][d];
            rank[it] = before + (uint32_t)__popcll(peers & ltMask);
        }
        __builtin_amdgcn_wave_barrier();   // every lane has read its counter before the leaders bump them
        if (valid && (peers & ltMask) == 0ull) waveCnt[wave][d] = before + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // digit t: turn per-wave counts into global destinations
        const int d = threadIdx.x;
        uint32_t run = digitBase[d] + blockHist[(size_t)d * numBlocks + blockIdx.x];
        for (int w = 0; w < 4; w++) {
            uint32_t c = waveCnt[w][d];
            waveCnt[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kSortItems; it++) {
        int i = base + it * kWave;
        if (i < n) {
            uint32_t d = (key[it] >> shift) & 255u;
            uint32_t dst = waveCnt[wave][d] + rank[it];
            keys_out[dst] = key[it];
            vals_out[dst] = val[it];
        }
    }
}

// ---- cumulative cell table ---------------------------------------------------------------------------
// T[c] = number of particles whose cell index is <= c.  Built in two steps: (1) the last slot of every
// run of equal keys marks T[key] = slot + 1 (everything else was zeroed), the neighbour keys coming
// from wave shuffles; (2) the marks grow with the cell index, so an inclusive prefix MAX over cells
// fills the empty cells.  Kernels index it through cellCum = T - 1 (cellCum[c] = #particles with key
// < c): the bin of cell c is [cellCum[c], cellCum[c+1]) and any run of consecutive cells [c0, c1] is the
// single slot range [cellCum[c0], cellCum[c1+1]) -- what the 3-cells-per-row neighbour search reads
// with two loads; empty cells cost nothing.  T is 16-byte aligned so the scans move int4.
__global__ __launch_bounds__(kBlock) void k_cell_marks(const uint32_t* __restrict__ sortedKeys, int n,
                                                       int* __restrict__ T) {
    int k = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t key = (k < n) ? sortedKeys[k] : 0xFFFFFFFFu;
    uint32_t next = __shfl_down(key, 1);
    if (k < n) {
        if (lane == 63) next = (k + 1 < n) ? sortedKeys[k + 1] : 0xFFFFFFFFu;
        if (k + 1 >= n) next = 0xFFFFFFFFu;
        if (next != key) T[key] = k + 1;
    }
}

constexpr int kScanVec = 4;                              // int4 per thread per round
constexpr int kScanRounds = 4;
constexpr int kScanTile = kBlock * kScanVec * kScanRounds;   // 4096 cells per block

__device__ __forceinline__ int wave_incl_max(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (lane >= o) v = max(v, t);
    }
    return v;
}

// cells4 = number of int4 groups (the table is padded to a multiple of 4 cells with zeros)
__global__ __launch_bounds__(kBlock) void k_cum_block_max(const int4* __restrict__ T4, long long cells4,
                                                          int* __restrict__ blockMax) {
    __shared__ int ws[4];
    const long long base = (long long)blockIdx.x * (kBlock * kScanRounds) + threadIdx.x;
    int m = 0;
#pragma unroll
    for (int r = 0; r < kScanRounds; r++) {
        const long long i = base + (long long)r * kBlock;
        if (i < cells4) {
            const int4 v = T4[i];
            m = max(max(m, max(v.x, v.y)), max(v.z, v.w));
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if (lane == 0) ws[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) blockMax[blockIdx.x] = max(max(ws[0], ws[1]), max(ws[2], ws[3]));
}

// exclusive prefix max over the block maxima, one workgroup
__global__ __launch_bounds__(kBlock) void k_cum_scan_blocks(int* __restrict__ blockMax, int numBlocks) {
    __shared__ int ws[4];
    __shared__ int carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < numBlocks; base += kBlock) {
        const int i = base + threadIdx.x;
        const int v = (i < numBlocks) ? blockMax[i] : 0;
        const int incl = wave_incl_max(v, lane);
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; w++) off = max(off, ws[w]);
        int excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 0;
        excl = max(excl, off);
        if (i < numBlocks) blockMax[i] = excl;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry = max(off, incl);
        __syncthreads();
    }
}

// in-place inclusive prefix max of one 4096-cell tile, seeded with the tile's exclusive prefix
__global__ __launch_bounds__(kBlock) void k_cum_write(int4* __restrict__ T4, long long cells4,
                                                      const int* __restrict__ blockPrefix) {
    __shared__ int ws[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = blockPrefix[blockIdx.x];
    const long long base = (long long)blockIdx.x * (kBlock * kScanRounds) + threadIdx.x;
#pragma unroll 1
    for (int r = 0; r < kScanRounds; r++) {
        const long long i = base + (long long)r * kBlock;
        int4 v = (i < cells4) ? T4[i] : make_int4(0, 0, 0, 0);
        v.y = max(v.y, v.x); v.z = max(v.z, v.y); v.w = max(v.w, v.z);      // thread-local
        const int incl = wave_incl_max(v.w, lane);
        __syncthreads();                                                    // ws free again
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; w++) off = max(off, ws[w]);
        int excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 0;
        off = max(off, excl);
        if (i < cells4) T4[i] = make_int4(max(v.x, off), max(v.y, off), max(v.z, off), max(v.w, off));
        carry = max(carry, max(max(ws[0], ws[1]), max(ws[2], ws[3])));
    }
}

// sid[k] = id of the particle that the cell-sorted rank k refers to
__global__ __launch_bounds__(kBlock) void k_gather_ids(const uint32_t* __restrict__ perm,
                                                       const uint32_t* __restrict__ origId, int n,
                                                       uint32_t* __restrict__ sid) {
    int k = blockIdx.x * kBlock + threadIdx.x;
    if (k < n) sid[k] = origId[perm[k]];
}

// Within each cell, order by particle id (the radix sort is stable w.r.t. the PREVIOUS slot
// order, which is (previous cell, id), not id).  rank = #ids in the cell smaller than mine.
// Also counts cells whose population exceeds the reference's bin capacity (compute:131).
__global__ __launch_bounds__(kBlock) void k_rank_in_cell(const uint32_t* __restrict__ sortedKeys,
                                                         const uint32_t* __restrict__ perm,
                                                         const uint32_t* __restrict__ sid, int n, int cap,
                                                         const int* __restrict__ cellStart,
                                                         const int* __restrict__ cellEnd,
                                                         uint32_t* __restrict__ perm2, uint32_t* __restrict__ newOrigId,
                                                         int* __restrict__ counters) {
    int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    uint32_t c = sortedKeys[k];
    int a = cellStart[c], e = cellEnd[c];
    uint32_t me = sid[k];
    int r = 0;
    for (int j = a; j < e; j++) r += (sid[j] < me) ? 1 : 0;
    perm2[a + r] = perm[k];
    newOrigId[a + r] = me;
    if (k == a && e - a > cap) atomicAdd(&counters[SPH_CTR_BIN_OVERFLOW], 1);
}

// state[new slot k] = state[old slot perm2[k]]
__global__ __launch_bounds__(kBlock) void k_permute_state(int n, const uint32_t* __restrict__ perm2, StateSet src,
                                                          StateSet dst) {
    int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    uint32_t s = perm2[k];
    dst.posM[k] = src.posM[s];
    dst.velV[k] = src.velV[s];
    dst.acc[k] = src.acc[s];
    dst.rhoP[k] = src.rhoP[s];
    dst.rho0[k] = src.rho0[s];
}

// cell_of_particle export: keys are per slot, the caller wants them per id
__global__ __launch_bounds__(kBlock) void k_scatter_by_id(int n, const uint32_t* __restrict__ in,
                                                          const uint32_t* __restrict__ origId,
                                                          uint32_t* __restrict__ out) {
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s < n) out[origId[s]] = in[s];
}

}  // namespace sph
