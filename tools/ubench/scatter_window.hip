// Micro-benchmark: inverting a permutation (out[perm[i]] = i: the last step of a sparse-domain Partition, whose final radix
// pass hands every rank its originating slot) as Q passes over the input, pass q storing only the elements whose destination
// lies in the q-th window of the output.  One pass = 60 M random 8-byte stores over 480 MB: every store is a partial line that
// travels to memory on its own.  With windows that fit the caches (XCD L2s 32 MB in all, Infinity Cache 256 MB) the stores of a
// line can meet before the line is written back -- at the price of reading the input Q times.  Prints the time per Q.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_make(int64_t *perm, int64_t n, uint64_t a, uint64_t b) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        perm[i] = (int64_t)(((unsigned __int128)(uint64_t)i * a + b) % (uint64_t)n);         // a coprime to n: a permutation
}
__global__ __launch_bounds__(256) void k_invert(const int64_t *__restrict__ perm, int64_t n, int64_t lo, int64_t hi, int64_t *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 2;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += stride) {
        const int64_t p0 = perm[i], p1 = i + 1 < n ? perm[i + 1] : -1;
        if (p0 >= lo && p0 < hi) out[p0] = i;
        if (p1 >= lo && p1 < hi) out[p1] = i + 1;
    }
}
__global__ void k_check(const int64_t *perm, const int64_t *out, int64_t n, unsigned long long *bad) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (out[perm[i]] != i) atomicAdd(bad, 1ull);
}
int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 59986052;
    int64_t *perm, *out; unsigned long long *bad;
    CHECK(hipMalloc(&perm, n * 8)); CHECK(hipMalloc(&out, n * 8)); CHECK(hipMalloc(&bad, 8));
    k_make<<<2048, 256>>>(perm, n, 2654435761ull * 40503ull + 1ull | 1ull, 12345);   // (odd multiplier; n below is not a multiple of its factors)
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int Q : {1, 2, 4, 8, 16, 32}) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipMemset(out, 0xff, n * 8));
            CHECK(hipEventRecord(e0));
            for (int q = 0; q < Q; q++) k_invert<<<2048, 256>>>(perm, n, n * q / Q, n * (q + 1) / Q, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        CHECK(hipMemset(bad, 0, 8));
        k_check<<<2048, 256>>>(perm, out, n, bad);
        unsigned long long h; CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        printf("Q=%2d windows of %6.1f MB: %8.1f us in all (%7.1f us per pass)  wrong %llu\n", Q, n * 8.0 / Q / 1e6, best * 1e3, best * 1e3 / Q, h);
    }
    return 0;
}
