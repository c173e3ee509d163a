// Micro-benchmark of the one-sweep radix Partition (mplan2vdl_amd/csrc/vdl_partition.hip, included as source so that its tile-shape
// macros can be swept): 60 M keys = product of two uniform factors (what tools/op_roofline.py's statement 32 sorts: a 36-bit range
// inside a 2^40 domain), lazy output form (slots in rank order + the sorted keys), checked (keys ascend, slots are a permutation
// whose keys match, ties in slot order).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I mplan2vdl_amd/csrc [-DVDL_PART_BLOCK=.. -DVDL_PART_STEPS=.. -DVDL_PART_EU=.. -DVDL_PART_NOWAIT] \
//         tools/ubench/part_bench.hip -o tools/ubench/part_bench
//   rocprofv3 --kernel-trace --stats -- tools/ubench/part_bench [n]
#include "../../mplan2vdl_amd/csrc/vdl_partition.hip"
#include <cstdio>
#include <vector>
namespace vdl { hipError_t launch_compact_scan(int64_t *, int64_t, hipStream_t, int64_t *) { return hipErrorNotSupported; } int64_t compact_tile() { return 4096; } }   // (launch_prefix_sum's helper lives in vdl_ops.hip; not used here)
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ void k_make(int64_t *k, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        k[i] = (int64_t)((90091 + mix(i) % 10404860) * (100 * (1 + mix(i ^ 0x5555555555ull) % 50)));
}
__global__ void k_check(const int64_t *key, const int64_t *order, const int64_t *sorted, int64_t n, unsigned long long *bad, unsigned long long *sum) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long s = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        const int64_t slot = order[r];
        bool wrong = slot < 0 || slot >= n || key[slot < 0 || slot >= n ? 0 : slot] != sorted[r];
        if (r + 1 < n) wrong |= sorted[r] > sorted[r + 1] || (sorted[r] == sorted[r + 1] && order[r] >= order[r + 1]);
        if (wrong) atomicAdd(bad, 1ull);
        s += (unsigned long long)slot;
    }
    atomicAdd(sum, s);
}
int main(int argc, char **argv) {
    using namespace vdl;
    const int64_t n = argc > 1 ? atoll(argv[1]) : 59986052;
    const int64_t pcount = (int64_t)1 << 40, maxb = (int64_t)10494950 * 5000;
    int64_t *key, *ka, *kb, *order, *sorted, *nvalid; void *scr; unsigned long long *res;
    CHECK(hipMalloc(&key, n * 8)); CHECK(hipMalloc(&ka, n * 8)); CHECK(hipMalloc(&kb, n * 8)); CHECK(hipMalloc(&order, n * 8)); CHECK(hipMalloc(&sorted, n * 8));
    CHECK(hipMalloc(&nvalid, 8)); CHECK(hipMalloc(&res, 16));
    const size_t sb = partition_scratch_bytes(n, pcount);
    CHECK(hipMalloc(&scr, sb));
    k_make<<<2048, 256>>>(key, n);
    Src d; d.p = key; d.kind = SRC_I64;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0));
        CHECK(launch_partition(d, nullptr, n, 0, pcount, scr, (uint64_t *)ka, nullptr, (uint64_t *)kb, nullptr, nvalid, nullptr, 0, maxb, order, sorted));
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
#ifdef VDL_PART_TIMING
    {   // phase timeline of the third pass: averages over the tiles, in us (wall_clock64 ticks at 100 MHz)
        const int64_t nt = partition_tiles(n);
        CHECK(hipMalloc(&g_part_timing, nt * 64)); CHECK(hipMemset(g_part_timing, 0, nt * 64));
        CHECK(launch_partition(d, nullptr, n, 0, pcount, scr, (uint64_t *)ka, nullptr, (uint64_t *)kb, nullptr, nvalid, nullptr, 0, maxb, order, sorted));
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(nt * 8);
        CHECK(hipMemcpy(h.data(), g_part_timing, nt * 64, hipMemcpyDeviceToHost));
        double ph[8] = {0}; unsigned long long t0 = ~0ull, t1 = 0;
        for (int64_t i = 0; i < nt; i++) { for (int k = 1; k < 8; k++) ph[k] += (double)(h[i * 8 + k] - h[i * 8 + k - 1]); t0 = std::min(t0, h[i * 8]); t1 = std::max(t1, h[i * 8 + 7]); }
        const char *nm[8] = {"", "fetch", "count", "publish", "rank", "offsets+stage", "prefix", "store"};
        printf("  pass %d timeline:", VDL_PART_TIMING); printf(" %lld tiles over %.1f us; per tile:", (long long)nt, (t1 - t0) / 100.0);
        double tot = 0;
        for (int k = 1; k < 8; k++) { printf("  %s %.2f", nm[k], ph[k] / nt / 100.0); tot += ph[k] / nt / 100.0; }
        printf("  = %.2f us\n", tot);
        g_part_timing = nullptr;
    }
#endif
    CHECK(hipMemset(res, 0, 16));
    k_check<<<2048, 256>>>(key, order, sorted, n, res, res + 1);
    unsigned long long h[2]; CHECK(hipMemcpy(h, res, 16, hipMemcpyDeviceToHost));
    const unsigned long long want = (unsigned long long)n * (unsigned long long)(n - 1) / 2;
    printf("block %d steps %d (tile %d) eu %d%s: %lld keys, scratch %.1f MB: %8.1f us  wrong %llu  slot sum %s\n", VDL_PART_BLOCK, VDL_PART_STEPS, VDL_PART_BLOCK * VDL_PART_STEPS, VDL_PART_EU,
#ifdef VDL_PART_NOWAIT
           " NOWAIT",
#else
           "",
#endif
           (long long)n, sb / 1e6, best * 1e3, h[0], h[1] == want ? "ok" : "BAD");
    return 0;
}
