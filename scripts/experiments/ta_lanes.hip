// Micro-benchmark (round 3): what does a wave-wide vector load cost the texture addresser as a function of
//   (a) how many lanes are active and where they sit (first K lanes vs every (64/K)-th lane), 16 B per lane,
//   (b) the width per lane (4 / 8 / 12 / 16 B), all lanes active, lanes in runs of 3 sharing an address (the density
//       kernel's pattern at ~3 particles per cell),
//   (c) an LDS read of 16 B per lane from a 4-byte-aligned address (is ds_read_b128 usable on 12-byte records?).
// Data stays in a 64 KB window (L1 / L2 resident), eight waves per SIMD: the time is the addresser's, not memory's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) F3U { float x, y, z; };
struct __attribute__((packed, aligned(4))) F2U { float x, y; };

template <int BYTES>
__global__ __launch_bounds__(256) void k_width(const char* __restrict__ buf, float* out, int iters, int active, int spread) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    const bool on = spread ? (lane % (64 / active) == 0) : (lane < active);
    uint32_t o = ((lane / 3) * 36 + (wave % 7) * 16) & 0xFFFCu;
    float acc = 0.f;
    if (on) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (BYTES == 16) { const F4U v = *(const F4U*)(buf + o); acc += v.x + v.y + v.z + v.w; }
                if (BYTES == 12) { const F3U v = *(const F3U*)(buf + o); acc += v.x + v.y + v.z; }
                if (BYTES == 8) { const F2U v = *(const F2U*)(buf + o); acc += v.x + v.y; }
                if (BYTES == 4) { acc += *(const float*)(buf + o); }
                o = (o + 4096 + 12) & 0xFFFCu;
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_lds(float* out, int iters, int misalign) {
    __shared__ float lds[4096 + 64];
    for (int i = threadIdx.x; i < 4096 + 64; i += 256) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t o = ((lane / 3) * 9 + misalign) & 4095u;          // in floats
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const F4U v = *(const F4U*)(lds + o);
            acc += v.x + v.y + v.z + v.w;
            o = (o + 3 * 37) & 4095u;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    char* d; float* o;
    (void)hipMalloc(&d, 1 << 17); (void)hipMemset(d, 0, 1 << 17); (void)hipMalloc(&o, 4 * 256 * 2048);
    const int blocks = 256 * 8, iters = 256;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto time = [&](auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        return best * 1e6 / ((double)blocks * 4 * iters * 8 / 256);     // ns per wave-load per CU
    };
    for (int spread = 0; spread < 2; spread++)
        for (int act : {64, 32, 16, 8, 4}) {
            const double ns = time([&] { hipLaunchKernelGGL(k_width<16>, dim3(blocks), dim3(256), 0, 0, d, o, iters, act, spread); });
            printf("16 B/lane, %2d lanes active (%s): %.2f ns per wave-load per CU = %.1f clk at 2.4 GHz\n", act, spread ? "spread" : "first", ns, ns * 2.4);
        }
    { const double ns = time([&] { hipLaunchKernelGGL(k_width<12>, dim3(blocks), dim3(256), 0, 0, d, o, iters, 64, 0); }); printf("12 B/lane, all lanes: %.2f ns = %.1f clk\n", ns, ns * 2.4); }
    { const double ns = time([&] { hipLaunchKernelGGL(k_width<8>, dim3(blocks), dim3(256), 0, 0, d, o, iters, 64, 0); }); printf(" 8 B/lane, all lanes: %.2f ns = %.1f clk\n", ns, ns * 2.4); }
    { const double ns = time([&] { hipLaunchKernelGGL(k_width<4>, dim3(blocks), dim3(256), 0, 0, d, o, iters, 64, 0); }); printf(" 4 B/lane, all lanes: %.2f ns = %.1f clk\n", ns, ns * 2.4); }
    for (int mis : {0, 1}) {
        const double ns = time([&] { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, o, iters, mis); });
        printf("LDS 16 B/lane read, %s: %.2f ns per wave-read per CU = %.1f clk\n", mis ? "4-byte aligned" : "in runs of 3 lanes, 36 B apart", ns, ns * 2.4);
    }
    float h[4]; (void)hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    printf("lds check value %.1f\n", h[0]);
    return 0;
}
