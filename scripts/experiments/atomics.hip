// Micro-benchmark: cost of one returning atomicAdd per particle on a cells-sized table
// (question: can a counting sort replace the 3-pass radix sort of the grid build?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__global__ void k_count(const unsigned* __restrict__ keys, int n, int* __restrict__ table, int* __restrict__ arrival) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    arrival[i] = atomicAdd(&table[keys[i]], 1);
}
__global__ void k_count_noret(const unsigned* __restrict__ keys, int n, int* __restrict__ table) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicAdd(&table[keys[i]], 1);
}
int main() {
    const int n = 1 << 20; const int cells = 3515136;
    std::vector<unsigned> h(n);
    // nearly sorted keys: like particles in last frame's order, ~10 per occupied cell
    for (int i = 0; i < n; i++) h[i] = (unsigned)(((long long)i * cells / n / 3) * 3 % cells + (rand() % 3));
    unsigned* dk; int *tab, *arr;
    hipMalloc(&dk, n * 4); hipMalloc(&tab, cells * 4); hipMalloc(&arr, n * 4);
    hipMemcpy(dk, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; mode++)
      for (int rep = 0; rep < 5; rep++) {
        hipMemset(tab, 0, cells * 4);
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(k_count, dim3(n / 256), dim3(256), 0, 0, dk, n, tab, arr);
        else hipLaunchKernelGGL(k_count_noret, dim3(n / 256), dim3(256), 0, 0, dk, n, tab);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("mode %d (%s) rep %d: %.1f us\n", mode, mode ? "no return" : "returning", rep, ms * 1e3);
      }
    std::random_shuffle(h.begin(), h.end());
    hipMemcpy(dk, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(tab, 0, cells * 4);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_count, dim3(n / 256), dim3(256), 0, 0, dk, n, tab, arr);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("shuffled returning rep %d: %.1f us\n", rep, ms * 1e3);
    }
    return 0;
}
