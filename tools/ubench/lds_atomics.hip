// Micro-benchmark: LDS atomic throughput on gfx950 (what bounds the grouped scan).
// Each wave issues N atomic instructions; address pattern: distinct per lane (stride 1 element),
// or K lanes per address.  Prints cycles per wave-instruction (per CU, 4 waves resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k(T *out, int iters, int lanes_per_addr, long long *cycles) {
    __shared__ T tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = wave * 1024 + (lane / lanes_per_addr) * (MODE == 2 ? 9 : 1);
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0 || MODE == 2) atomicAdd(&tab[(idx + i * 64) & 4095], (T)1);
        else tab[(idx + i * 64) & 4095] += (T)1;          // plain read-modify-write (no atomic)
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    T s = 0;
    for (int i = threadIdx.x; i < 4096; i += 256) s += tab[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename T, int MODE>
void run(const char *name, int lpa) {
    T *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * sizeof(T)); hipMalloc(&cyc, 256 * sizeof(long long));
    const int iters = 4096;
    k<T, MODE><<<256, 256>>>(out, iters, lpa, cyc);
    k<T, MODE><<<256, 256>>>(out, iters, lpa, cyc);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; i++) avg += h[i]; avg /= 256;
    // 4 waves per block share the CU's LDS: wave-instructions per block = 4 * iters
    printf("%-28s lanes/addr=%2d : %7.1f clock64 ticks per wave-instruction (block of 4 waves)\n", name, lpa, avg / (4.0 * iters));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int lpa : {1, 2, 4, 16, 64}) {
        run<unsigned int, 0>("ds_add_u32 atomic", lpa);
        run<unsigned long long, 0>("ds_add_u64 atomic", lpa);
    }
    run<unsigned int, 1>("u32 plain RMW", 1);
    run<unsigned long long, 1>("u64 plain RMW", 1);
    run<unsigned long long, 2>("ds_add_u64 stride 9", 1);
    run<unsigned long long, 2>("ds_add_u64 stride 9", 8);
    return 0;
}
