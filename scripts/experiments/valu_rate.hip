// Micro-benchmark: sustained VALU issue rate (scalar v_fma_f32 vs v_pk_fma_f32 vs v_pk_mul_f32)
// with every SIMD of the chip busy; prints wave-instructions per second per SIMD = effective
// clock / cycles-per-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 y0 = {x0, x1}, y1 = {x2, x3}, y2 = {x4, x5}, y3 = {x6, x7}, y4 = y0 + 1.f, y5 = y1 + 1.f, y6 = y2 + 1.f, y7 = y3 + 1.f;
    const f2 av = {a, a}, bv = {b, b};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                y0 = __builtin_elementwise_fma(y0, av, bv); y1 = __builtin_elementwise_fma(y1, av, bv);
                y2 = __builtin_elementwise_fma(y2, av, bv); y3 = __builtin_elementwise_fma(y3, av, bv);
                y4 = __builtin_elementwise_fma(y4, av, bv); y5 = __builtin_elementwise_fma(y5, av, bv);
                y6 = __builtin_elementwise_fma(y6, av, bv); y7 = __builtin_elementwise_fma(y7, av, bv);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                y0 = y0 * av; y1 = y1 * av; y2 = y2 * av; y3 = y3 * av; y4 = y4 * av; y5 = y5 * av; y6 = y6 * av; y7 = y7 * av;
            }
        }
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0.x + y0.y + y1.x + y1.y + y2.x + y2.y + y3.x + y3.y + y4.x + y4.y + y5.x +
              y5.y + y6.x + y6.y + y7.x + y7.y;
    if (r == 12345.678f) out[0] = r;
}
template <int MODE>
void run(const char* name, float* d) {
    const int blocks = 256 * 8, iters = 4096;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        double instr = (double)blocks * 4 * iters * 64;          // wave-instructions
        printf("%-14s rep %d: %.3f ms, %.2f G wave-instr/s per SIMD (1024 SIMDs)\n", name, rep, ms, instr / (ms * 1e-3) / 1024 / 1e9);
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 4);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_pk_mul_f32", d);
    return 0;
}
