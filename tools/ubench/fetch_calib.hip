// Micro-benchmark: what a MASKED 16-byte-per-lane load costs in HBM traffic on gfx950, and what rocprofv3's FETCH_SIZE says
// about it.  The staged scans (csrc/vdl_mscan_body.h, "late materialisation") read the columns behind the first filter with
// 16-byte loads under an exec mask -- a lane loads its row pair only while one of its two rows is still in -- so the bytes
// they move depend on the granule at which the memory side fetches (32 B, 64 B or a whole 128-B line) and FETCH_SIZE needs
// its own calibration for that access pattern (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated").
//
// Every configuration sweeps the same buffer (default 2 GiB, far beyond the 256 MiB Infinity Cache) once, lane l of a wave
// owning bytes [16 l, 16 l + 16) of each 1 KiB piece, exactly like the scans; what differs is WHICH lanes load:
//   stream          every lane                                   (the reference point: FETCH_SIZE x 2 = bytes)
//   sector32 /k     lanes whose 32-byte half-sector index is a multiple of k
//   sector64 /k     lanes whose 64-byte sector index is a multiple of k
//   line128  /k     lanes whose 128-byte line index is a multiple of k
//   rows p          a lane loads when one of its two 8-byte rows is "alive"; rows are alive independently with probability p
//                   (a hash of the row number) -- the staged scan's pattern on uniformly random data
// The kernel counts, from the wave's ballot, the distinct 32-B / 64-B / 128-B units and the lanes it asked for; the host prints
// them beside the kernel time.  Run it once plainly (times) and once under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- ./fetch_calib
// and join the counter rows with the printed table by dispatch order (tools/ubench/fetch_calib_join.py).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                 \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

typedef long long2v __attribute__((ext_vector_type(2)));

// (a 32-bit finaliser: cheap enough that the mask arithmetic stays far below the memory time)
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

enum Mode { STREAM = 0, SECTOR32 = 1, SECTOR64 = 2, LINE128 = 3, ROWS = 4 };

// counts[0..3] = lanes, 32-B units, 64-B units, 128-B units asked for; counts[4] = checksum (keeps the loads alive)
// COUNT = false: the same sweep without the bookkeeping -- the one whose TIME means something (k_sweep)
template <int MODE, int U, bool COUNT>
__device__ __forceinline__ void calib_body(const long2v *__restrict__ buf, int64_t npairs, int k, uint32_t thresh, unsigned long long *counts) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = 256 * U;                               // pairs per block iteration
    int64_t acc = 0;
    unsigned long long c_lanes = 0, c32 = 0, c64 = 0, c128 = 0;
    for (int64_t base = (int64_t)blockIdx.x * tile; base + tile <= npairs; base += (int64_t)gridDim.x * tile) {
        long2v v[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t pair = base + (int64_t)u * 256 + threadIdx.x;        // 16 bytes each: byte address = 16 * pair
            if (MODE == STREAM) on[u] = true;
            else if (MODE == SECTOR32) on[u] = ((pair >> 1) % k) == 0;
            else if (MODE == SECTOR64) on[u] = ((pair >> 2) % k) == 0;
            else if (MODE == LINE128) on[u] = ((pair >> 3) % k) == 0;
            else on[u] = mix32((uint32_t)pair * 2u) < thresh || mix32((uint32_t)pair * 2u + 1u) < thresh;
            v[u] = long2v{0, 0};
            if (on[u]) v[u] = __builtin_nontemporal_load(buf + pair);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            acc += v[u].x ^ v[u].y;
            const uint64_t m = COUNT ? __ballot(on[u]) : 0ull;
            if (COUNT && lane == 0) {
                c_lanes += __popcll(m);
                uint64_t a = m | (m >> 1);                      // bit 2j: any of lanes 2j, 2j+1 (one 32-B unit)
                c32 += __popcll(a & 0x5555555555555555ull);
                a |= a >> 2;                                    // bit 4j: any of lanes 4j..4j+3 (one 64-B sector)
                c64 += __popcll(a & 0x1111111111111111ull);
                a |= a >> 4;                                    // bit 8j: any of lanes 8j..8j+7 (one 128-B line)
                c128 += __popcll(a & 0x0101010101010101ull);
            }
        }
    }
    if (COUNT && lane == 0) {
        atomicAdd(&counts[0], c_lanes); atomicAdd(&counts[1], c32); atomicAdd(&counts[2], c64); atomicAdd(&counts[3], c128);
    }
    for (int off = 32; off; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0 && acc == 0x7fffffffffffffffll) atomicAdd(&counts[4], 1ull);
}
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_calib(const long2v *__restrict__ buf, int64_t npairs, int k, uint32_t thresh, unsigned long long *counts) {
    calib_body<MODE, U, true>(buf, npairs, k, thresh, counts);
}
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_sweep(const long2v *__restrict__ buf, int64_t npairs, int k, uint32_t thresh, unsigned long long *counts) {
    calib_body<MODE, U, false>(buf, npairs, k, thresh, counts);
}

struct Config { const char *name; int mode; int k; double p; };

int main(int argc, char **argv) {
    const size_t bytes = (argc > 1 ? (size_t)atof(argv[1]) : 2.0) * (1ull << 30);
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const int64_t npairs = (int64_t)(bytes / 16);
    long2v *buf;
    unsigned long long *counts;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&counts, 5 * sizeof(unsigned long long)));
    CHECK(hipMemset(buf, 1, bytes));
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    CHECK(hipGetDevice(&dev));
    CHECK(hipGetDeviceProperties(&prop, dev));
    if (prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    const int grid = cus * 8;
    constexpr int U = 4, UT = 8;                                // (the timed sweep keeps 8 x 16 B per lane in flight, 4 blocks per CU)
    std::vector<Config> cfgs = {{"stream", STREAM, 1, 0}};
    for (int k : {2, 4, 8}) cfgs.push_back({"sector32", SECTOR32, k, 0});
    for (int k : {2, 4, 8}) cfgs.push_back({"sector64", SECTOR64, k, 0});
    for (int k : {2, 4, 8}) cfgs.push_back({"line128", LINE128, k, 0});
    for (double p : {0.5, 0.1445, 0.0394, 0.0181, 0.005}) cfgs.push_back({"rows", ROWS, 1, p});
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("# buffer %.3f GiB, grid %d x 256, %d counting launches (k_calib) per configuration, then timed sweeps without bookkeeping (k_sweep: best_us); dispatch order = row order\n",
           bytes / (double)(1ull << 30), grid, reps);
    printf("# idx name k p lanes_x16B units32_x32B units64_x64B units128_x128B best_us GBps_lanes GBps_64 GBps_128\n");
    int idx = 0;
    for (const Config &c : cfgs) {
        const uint32_t thresh = c.p >= 1 ? ~0u : (uint32_t)(c.p * 4294967295.0);
        float best = 1e30f;
        unsigned long long h[5] = {};
        for (int r = 0; r < reps; r++) {
            CHECK(hipMemset(counts, 0, 5 * sizeof(unsigned long long)));
            CHECK(hipEventRecord(e0));
            switch (c.mode) {
                case STREAM: k_calib<STREAM, U><<<grid, 256>>>(buf, npairs, c.k, thresh, counts); break;
                case SECTOR32: k_calib<SECTOR32, U><<<grid, 256>>>(buf, npairs, c.k, thresh, counts); break;
                case SECTOR64: k_calib<SECTOR64, U><<<grid, 256>>>(buf, npairs, c.k, thresh, counts); break;
                case LINE128: k_calib<LINE128, U><<<grid, 256>>>(buf, npairs, c.k, thresh, counts); break;
                default: k_calib<ROWS, U><<<grid, 256>>>(buf, npairs, c.k, thresh, counts); break;
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 || reps == 1) best = ms < best ? ms : best;
            CHECK(hipMemcpy(h, counts, sizeof h, hipMemcpyDeviceToHost));
        }
        // the timed sweep: no ballots, no counters
        float tbest = 1e30f;
        for (int r = 0; r < reps + 2; r++) {
            CHECK(hipEventRecord(e0));
            switch (c.mode) {
                case STREAM: k_sweep<STREAM, UT><<<cus * 4, 256>>>(buf, npairs, c.k, thresh, counts); break;
                case SECTOR32: k_sweep<SECTOR32, UT><<<cus * 4, 256>>>(buf, npairs, c.k, thresh, counts); break;
                case SECTOR64: k_sweep<SECTOR64, UT><<<cus * 4, 256>>>(buf, npairs, c.k, thresh, counts); break;
                case LINE128: k_sweep<LINE128, UT><<<cus * 4, 256>>>(buf, npairs, c.k, thresh, counts); break;
                default: k_sweep<ROWS, UT><<<cus * 4, 256>>>(buf, npairs, c.k, thresh, counts); break;
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) tbest = ms < tbest ? ms : tbest;
        }
        best = tbest;
        const double s = best * 1e-3;
        printf("%d %s %d %.4f %llu %llu %llu %llu %.1f %.1f %.1f %.1f\n", idx++, c.name, c.k, c.p, h[0] * 16, h[1] * 32, h[2] * 64, h[3] * 128, best * 1e3,
               h[0] * 16 / s / 1e9, h[2] * 64 / s / 1e9, h[3] * 128 / s / 1e9);
    }
    return 0;
}
