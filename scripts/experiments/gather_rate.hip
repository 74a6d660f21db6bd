// Micro-benchmark: what does a wave-wide 16-byte-per-lane load cost the vector memory pipe, as a function of
// WHERE the 64 lanes read?  Data stays in a 64 KB window (vector-L1 / L2 resident), eight waves per SIMD, so
// the time is the texture addresser / vector-L1 rate, not memory.  Prints ns per wave-load per CU and the
// equivalent clocks at 2.4 GHz.  Patterns:
//   0 contiguous          lane l reads 16 B at 16 l                         (1 KB, 8 lines)
//   1 same address        every lane reads the same 16 B
//   2 runs of 11          lanes in runs of 11 share an address, runs 132 B apart   (the density kernel's candidates)
//   3 runs of 11, 176 B   same with the runs 176 B apart                            (16-byte records: 11 x 16)
//   4 stride 64           lane l reads 16 B at 64 l                          (one quad of each of 64 rows: the mask rows before)
//   5 random              lane l reads 16 B at a pseudo-random 16-byte slot of the window
//   6 runs of 11 random   runs of 11 lanes share an address, the runs at random slots
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k(const char* __restrict__ buf, float* out, int iters, int mode) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    uint32_t off;
    switch (mode) {
        case 0: off = lane * 16; break;
        case 1: off = 0; break;
        case 2: off = (lane / 11) * 132; break;
        case 3: off = (lane / 11) * 176; break;
        case 4: off = lane * 64; break;
        case 5: off = ((lane * 2654435761u) >> 20) % 4000 * 16; break;
        default: off = (((lane / 11) * 2654435761u) >> 20) % 4000 * 16; break;
    }
    off = (off + (wave % 7) * 16) & 0xFFF0u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t o = off;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float4 v = *(const float4*)(buf + o);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            o = (o + 4096 + 16) & 0xFFF0u;       // walk the 64 KB window, keep the pattern's shape
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
int main() {
    char* d; float* o;
    (void)hipMalloc(&d, 1 << 17); (void)hipMemset(d, 0, 1 << 17); (void)hipMalloc(&o, 4);
    const char* names[7] = {"contiguous", "same address", "runs of 11 / 132 B", "runs of 11 / 176 B", "stride 64 B", "random", "runs of 11 random"};
    const int blocks = 256 * 8, iters = 512;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int mode = 0; mode < 7; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, o, iters, mode);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        const double loadsPerCu = (double)blocks * 4 * iters * 8 / 256;
        const double ns = best * 1e6 / loadsPerCu;
        printf("%-20s %.3f ms  %.2f ns per wave-load per CU = %.1f clk at 2.4 GHz\n", names[mode], best, ns, ns * 2.4);
    }
    return 0;
}
