// vdl_partition.hip -- device-wide prefix sum, Partition (stable LSD radix ranks) and the row exchange of a sharded
// Partition.
#include "vdl_device.h"

#include <cstdlib>

namespace vdl {

// ------------------------------------------------------------------------------------------
// device-wide exclusive prefix sum over int64 (in place): block sums -> one-block scan of the
// sums -> per-block scan with carry.  Tile = 1024 threads x 4 consecutive items.
// ------------------------------------------------------------------------------------------
constexpr int kPsBlock = 1024, kPsItems = 4, kPsTile = kPsBlock * kPsItems;
int64_t prefix_sum_blocks(int64_t n) { return (n + kPsTile - 1) / kPsTile; }

__global__ __launch_bounds__(kPsBlock) void k_ps_block_sums(const int64_t *x, int64_t n, int64_t *sums) {
    __shared__ int64_t red[kPsBlock / kWave];
    const int64_t base = (int64_t)blockIdx.x * kPsTile + (int64_t)threadIdx.x * kPsItems;
    int64_t t = 0;
#pragma unroll
    for (int k = 0; k < kPsItems; k++) if (base + k < n) t += x[base + k];
    t = wave_reduce(t, R_SUM);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = t;
    __syncthreads();
    if (threadIdx.x == 0) { int64_t a = 0; for (int w = 0; w < kPsBlock / kWave; w++) a += red[w]; sums[blockIdx.x] = a; }
}

__global__ __launch_bounds__(kPsBlock) void k_ps_apply(int64_t *x, int64_t n, const int64_t *block_excl) {
    __shared__ int64_t wsum[kPsBlock / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int64_t base = (int64_t)blockIdx.x * kPsTile + (int64_t)tid * kPsItems;
    int64_t v[kPsItems], t = 0;
#pragma unroll
    for (int k = 0; k < kPsItems; k++) { v[k] = (base + k < n) ? x[base + k] : 0; t += v[k]; }
    int64_t incl = t;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) { int64_t y = __shfl_up(incl, off, kWave); if (lane >= off) incl += y; }
    if (lane == kWave - 1) wsum[wave] = incl;
    __syncthreads();
    int64_t run = block_excl[blockIdx.x] + incl - t;
    for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int k = 0; k < kPsItems; k++) { if (base + k < n) x[base + k] = run; run += v[k]; }
}

// sums: prefix_sum_blocks(n) + 1 int64 of scratch; the grand total is left in sums[nblocks]
hipError_t launch_prefix_sum(int64_t *x, int64_t n, int64_t *sums, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    const int64_t nb = prefix_sum_blocks(n);
    if (nb <= 0) return hipSuccess;
    k_ps_block_sums<<<(int)nb, kPsBlock, 0, s>>>(x, n, sums);
    if (launch_compact_scan(sums, nb, s) != hipSuccess) return hipGetLastError();      // the one-block scan lives in vdl_ops.hip
    k_ps_apply<<<(int)nb, kPsBlock, 0, s>>>(x, n, sums);
    return launch_status();
}

// ------------------------------------------------------------------------------------------
// Partition (/root/reference/src/Vdl.hs:130,266-269; Vlite.hs:358-366,508,1082-1098): positions
// that stably group `data` by pivot bucket.  Pivots are the emitted RangeC min cnt 1, so
// bucket = clamp(data - min, 0, cnt).  Implemented as an LSD radix sort of (bucket, slot) over the
// non-EPS slots, 8 bits per pass; each pass = tile histogram (LDS atomics) -> device-wide prefix
// sum in digit-major order -> stable scatter (see k_part_scatter).  The last pass writes out[slot] = rank instead of the sorted pair.
// Dense group-by domains (Q1: 32 buckets) need one pass, Q3's 2^38 domain five.
// ------------------------------------------------------------------------------------------
constexpr int kPartBlock = 256, kPartSteps = 16, kPartTile = kPartBlock * kPartSteps, kRadix = 256;
int64_t partition_tiles(int64_t n) { return (n + kPartTile - 1) / kPartTile; }

struct PartIn {
    Src data;                    // first pass: raw data column
    const uint64_t *valid;       // first pass: validity of data
    int64_t pmin, pcount;        // pivots = RangeC pmin pcount 1
    const uint64_t *keys;        // later passes: bucket values of the previous pass
    const int64_t *slots;        // later passes: originating slot
    const int64_t *n_dev;        // later passes: number of elements (device scalar)
    int64_t n;                   // first pass: number of slots; later: upper bound for the grid
    int shift;
    int64_t *sorted_keys;        // with order_out: also the key VALUES in rank order (bucket + pmin: the caller knows every key lies inside the pivots)
    int order_out;               // this (last) pass writes the slots in rank order to slots_out instead of the sorted pairs / the ranks
    int slot_bits;               // > 0: packed form -- a pair travels as ONE word (bucket << slot_bits) | slot, `slots` is unused:
                                 // every later pass moves 8 instead of 16 bytes per row (taken when bits(pcount) + bits(n) <= 64)
};

// A wave's share of a tile: kPartSteps x 64 consecutive slots starting at a multiple of 64, fetched with every load
// issued before the first use (the validity word of a step is the same for all lanes).
template <bool FIRST>
__device__ __forceinline__ void part_fetch_share(const PartIn &in, int64_t n, int64_t share0 /* multiple of 64 */, int lane,
                                                 uint64_t (&keys)[kPartSteps], int64_t (&slots)[kPartSteps], bool (&oks)[kPartSteps]) {
    if (FIRST) {
        by_kind(in.data.kind, [&](auto kd) {
#pragma unroll
            for (int st = 0; st < kPartSteps; st++) {
                const int64_t i = share0 + st * kWave + lane;
                oks[st] = i < n;
                slots[st] = i;
                keys[st] = (uint64_t)ldk<decltype(kd)::value>(in.data, oks[st] ? i : 0);
            }
        });
        if (in.valid) {
            const int64_t nw = (n + 63) >> 6;
            uint64_t t[kPartSteps];
#pragma unroll
            for (int st = 0; st < kPartSteps; st++) { const int64_t w = (share0 >> 6) + st; t[st] = in.valid[w < nw ? w : nw - 1]; }
#pragma unroll
            for (int st = 0; st < kPartSteps; st++) oks[st] = oks[st] & (((t[st] >> lane) & 1ull) != 0);
        }
#pragma unroll
        for (int st = 0; st < kPartSteps; st++) {                // bucket = clamp(data - min, 0, cnt)
            const int64_t x = (int64_t)keys[st];
            int64_t b = 0;
            if (x > in.pmin) { b = (int64_t)((uint64_t)x - (uint64_t)in.pmin); if (b < 0 || b > in.pcount) b = in.pcount; }
            keys[st] = in.slot_bits ? (((uint64_t)b << in.slot_bits) | (uint64_t)slots[st]) : (uint64_t)b;
        }
    } else if (in.slot_bits) {
#pragma unroll
        for (int st = 0; st < kPartSteps; st++) {
            const int64_t i = share0 + st * kWave + lane;
            oks[st] = i < n;
            keys[st] = in.keys[oks[st] ? i : 0];
            slots[st] = (int64_t)(keys[st] & ((1ull << in.slot_bits) - 1));
        }
    } else {
#pragma unroll
        for (int st = 0; st < kPartSteps; st++) {
            const int64_t i = share0 + st * kWave + lane;
            oks[st] = i < n;
            const int64_t ii = oks[st] ? i : 0;                  // n > 0 here
            keys[st] = in.keys[ii]; slots[st] = in.slots[ii];
        }
    }
}

template <bool FIRST>
__global__ __launch_bounds__(kPartBlock) void k_part_hist(PartIn in, int64_t ntiles, int64_t *hist /*[256][ntiles]*/) {
    __shared__ unsigned int h[kRadix];
    const int64_t n = FIRST ? in.n : *in.n_dev;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    h[tid] = 0;
    __syncthreads();
    if ((int64_t)blockIdx.x * kPartTile < n) {                   // (n = 0 leaves the key buffers unwritten)
        uint64_t keys[kPartSteps];
        int64_t slots[kPartSteps];
        bool oks[kPartSteps];
        part_fetch_share<FIRST>(in, n, (int64_t)blockIdx.x * kPartTile + (int64_t)wave * (kPartSteps * kWave), lane, keys, slots, oks);
#pragma unroll
        for (int st = 0; st < kPartSteps; st++)
            if (oks[st]) atomicAdd(&h[(keys[st] >> in.shift) & (kRadix - 1)], 1u);
    }
    __syncthreads();
    hist[(int64_t)tid * ntiles + blockIdx.x] = h[tid];
}

// Each wave owns a contiguous quarter of the tile (16 steps of 64 slots), so the stable order inside a tile is wave,
// step, lane.  A wave ranks its slots on its own (peer masks from 8 ballots, its running digit counts in its row of
// whist: LDS operations of one wave execute in order); one barrier later the rows are turned into per-wave offsets
// and every slot knows its destination.  (A version that kept the block in step order needed three barriers per
// step, 48 per tile, and was twice as slow.)
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(kPartBlock) void k_part_scatter(PartIn in, int64_t ntiles, const int64_t *offsets,
                                                              uint64_t *keys_out, int64_t *slots_out, int64_t *pos_out) {
    constexpr int NW = kPartBlock / kWave;
    __shared__ int64_t woff[NW][kRadix];
    __shared__ unsigned int whist[NW][kRadix];
    const int64_t n = FIRST ? in.n : *in.n_dev;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
#pragma unroll
    for (int w = 0; w < NW; w++) whist[w][tid] = 0;
    __syncthreads();
    uint64_t keys[kPartSteps];
    int64_t slots[kPartSteps];
    bool oks[kPartSteps];
#pragma unroll
    for (int st = 0; st < kPartSteps; st++) { keys[st] = 0; slots[st] = 0; oks[st] = false; }
    if ((int64_t)blockIdx.x * kPartTile < n)                    // the whole share is fetched up front
        part_fetch_share<FIRST>(in, n, (int64_t)blockIdx.x * kPartTile + (int64_t)wave * (kPartSteps * kWave), lane, keys, slots, oks);
    unsigned int local[kPartSteps];                             // rank among this wave's slots with the same digit
    volatile unsigned int *mine = whist[wave];
#pragma unroll
    for (int st = 0; st < kPartSteps; st++) {
        const unsigned d = (unsigned)((keys[st] >> in.shift) & (kRadix - 1));
        uint64_t peers = __ballot(oks[st]);                     // lanes of this wave holding the same digit (and a value)
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const unsigned rank = (unsigned)__popcll(peers & ((1ull << lane) - 1));
        const unsigned pre = oks[st] ? mine[d] : 0u;
        if (oks[st] && rank == 0) mine[d] = pre + (unsigned)__popcll(peers);     // one leader per digit
        local[st] = pre + rank;
    }
    __syncthreads();
    if (LAST) {                                                 // ranks go to out[slot]: scattered whatever the order
        {
            int64_t run = offsets[(int64_t)tid * ntiles + blockIdx.x];          // where this tile's slots of digit `tid` start
#pragma unroll
            for (int w = 0; w < NW; w++) { woff[w][tid] = run; run += whist[w][tid]; }
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < kPartSteps; st++) {
            if (oks[st]) {
                const unsigned d = (unsigned)((keys[st] >> in.shift) & (kRadix - 1));
                pos_out[slots[st]] = woff[wave][d] + local[st];
            }
        }
        return;
    }
    // The tile is put into digit order in LDS first (keys, then slots through the same buffer): a digit's slots of
    // one tile are neighbours at the destination, so consecutive lanes then store consecutive words instead of 64
    // scattered ones.
    __shared__ uint64_t stage[kPartTile];
    __shared__ unsigned int wtot[NW];
    __shared__ int64_t gdelta[kRadix];                          // destination of sorted index idx with digit d = gdelta[d] + idx
    unsigned total;
    {
        unsigned cnt = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) cnt += whist[w][tid];
        unsigned incl = cnt;                                    // exclusive scan of the tile's digit counts over the block
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) { const unsigned y = __shfl_up(incl, off, kWave); if (lane >= off) incl += y; }
        if (lane == kWave - 1) wtot[wave] = incl;
        __syncthreads();
        unsigned pre = 0;
        for (int w = 0; w < wave; w++) pre += wtot[w];
        total = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) total += wtot[w];
        const unsigned tstart = pre + incl - cnt;               // where digit `tid` starts inside the sorted tile
        unsigned run = tstart;
#pragma unroll
        for (int w = 0; w < NW; w++) { woff[w][tid] = run; run += whist[w][tid]; }                            // tile-local
        gdelta[tid] = offsets[(int64_t)tid * ntiles + blockIdx.x] - (int64_t)tstart;
        __syncthreads();
    }
    unsigned lpos[kPartSteps];
#pragma unroll
    for (int st = 0; st < kPartSteps; st++) {
        const unsigned d = (unsigned)((keys[st] >> in.shift) & (kRadix - 1));
        lpos[st] = (unsigned)woff[wave][d] + local[st];
        if (oks[st]) stage[lpos[st]] = keys[st];
    }
    __syncthreads();
    int64_t dest[kPartSteps];
#pragma unroll
    for (int k = 0; k < kPartSteps; k++) {
        const unsigned idx = (unsigned)k * kPartBlock + tid;
        dest[k] = -1;
        if (idx < total) {
            const uint64_t key = stage[idx];
            dest[k] = gdelta[(key >> in.shift) & (kRadix - 1)] + idx;
            if (!in.order_out) keys_out[dest[k]] = key;
            else {
                if (in.slot_bits) slots_out[dest[k]] = (int64_t)(key & ((1ull << in.slot_bits) - 1));     // the last pass of a lazy Partition: slots in rank order
                if (in.sorted_keys) in.sorted_keys[dest[k]] = (int64_t)((uint64_t)in.pmin + (in.slot_bits ? key >> in.slot_bits : key));
            }
        }
    }
    if (in.slot_bits) return;                                  // packed: the slot travelled inside the key word
    __syncthreads();
#pragma unroll
    for (int st = 0; st < kPartSteps; st++)
        if (oks[st]) stage[lpos[st]] = (uint64_t)slots[st];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPartSteps; k++)
        if (dest[k] >= 0) slots_out[dest[k]] = (int64_t)stage[(unsigned)k * kPartBlock + tid];
}

// Is the (fully valid) data already in non-decreasing order?  Then its stable partition ranks are 0, 1, 2, ... and none of
// the radix passes is needed: TPC-H lineitems are clustered by order key, so the composite group keys of Q3 / Q18 arrive
// sorted.  flag[0] (pre-zeroed) is set when a descent is found.
// (flag[1], pre-zeroed, receives the largest value: when the data is not in order, the radix passes only have to cover the
// buckets that occur, not the whole declared domain -- and a smaller bucket range may let a pair travel as one word)
__global__ __launch_bounds__(256) void k_sorted_check(Src d, int64_t n, int64_t *flag) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    int64_t mx = INT64_MIN;
    by_kind(d.kind, [&](auto kd) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            const int64_t x = ldk<decltype(kd)::value>(d, i);
            if (i + 1 < n) bad |= x > ldk<decltype(kd)::value>(d, i + 1);
            mx = x > mx ? x : mx;
        }
    });
    if (__ballot(bad) != 0 && (threadIdx.x & (kWave - 1)) == 0) flag[0] = 1;
    mx = wave_reduce(mx, R_MAX);
    // (one atomic per block, after a plain look: in ascending data every wave holds a new maximum)
    __shared__ int64_t wmax[256 / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) wmax[threadIdx.x / kWave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / kWave; w++) mx = wmax[w] > mx ? wmax[w] : mx;
        if (mx > *(volatile int64_t *)&flag[1]) atomicMax((long long *)&flag[1], (long long)mx);
    }
}
hipError_t launch_sorted_check(Src d, int64_t n, int64_t *flag, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 1) return hipSuccess;
    k_sorted_check<<<grid_for(n, 256, 8), 256, 0, s>>>(d, n, flag);
    return launch_status();
}

// One wave per 64 entries: neighbour compare through a shuffle (lane 0 reads its predecessor), heads word by ballot.  flag[0] /
// flag[1] are initialised by a first tiny launch (no host copy).  One atomic per BLOCK for the maximum: data in ascending order
// -- the case this pass exists for -- gives every wave a new maximum, and 50 K atomics on one word took 80 us (3 M entries).
__global__ void k_sorted_init(int64_t *flag) { flag[0] = 0; flag[1] = INT64_MIN; flag[2] = INT64_MAX; }      // {descends, largest, smallest}
__global__ __launch_bounds__(256) void k_sorted_heads(Src d, int64_t n, uint64_t *heads, int64_t *flag) {
    __shared__ int64_t wmax[256 / kWave], wmin[256 / kWave];
    __shared__ int wbad[256 / kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int64_t nw = (n + 63) >> 6;
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave);
    bool bad = false;
    int64_t mx = INT64_MIN, mn = INT64_MAX;
    by_kind(d.kind, [&](auto kd) {
        for (int64_t w = (int64_t)blockIdx.x * (blockDim.x / kWave) + wave; w < nw; w += wstride) {
            const int64_t i = (w << 6) + lane;
            const bool in = i < n;
            const int64_t x = ldk<decltype(kd)::value>(d, in ? i : n - 1);
            int64_t prev = __shfl_up(x, 1, kWave);
            if (lane == 0) prev = i > 0 ? ldk<decltype(kd)::value>(d, i - 1) : x;
            bad |= in && x < prev;
            mx = in && x > mx ? x : mx;
            mn = in && x < mn ? x : mn;
            const uint64_t hm = __ballot(in && (i == 0 || x != prev));
            if (lane == 0) heads[w] = hm;
        }
    });
    const bool anybad = __ballot(bad) != 0;
    mx = wave_reduce(mx, R_MAX);
    mn = wave_reduce(mn, R_MIN);
    if (lane == 0) { wmax[wave] = mx; wmin[wave] = mn; wbad[wave] = anybad ? 1 : 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t m = wmax[0], lo = wmin[0];
        int b = wbad[0];
        for (int w = 1; w < 256 / kWave; w++) { m = wmax[w] > m ? wmax[w] : m; lo = wmin[w] < lo ? wmin[w] : lo; b |= wbad[w]; }
        if (b) flag[0] = 1;
        if (m > *(volatile int64_t *)&flag[1]) atomicMax((long long *)&flag[1], (long long)m);
        if (lo < *(volatile int64_t *)&flag[2]) atomicMin((long long *)&flag[2], (long long)lo);
    }
}
// The same for an int64 vector at a 16-byte aligned address (the entries of a GROUP BY key): four consecutive entries per lane (two
// 16-byte loads), the left neighbour's last value by shuffle, the four head bits of every lane gathered into the wave's four bitmap
// words by an OR across each group of 16 lanes; two groups of 256 entries in flight per wave.  3 M entries: 24 -> ~10 us.
__global__ __launch_bounds__(256) void k_sorted_heads_dense(const int64_t *__restrict__ d, int64_t n, uint64_t *__restrict__ heads, int64_t *flag) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    constexpr int R = 2;
    __shared__ int64_t wmax[256 / kWave], wmin[256 / kWave];
    __shared__ int wbad[256 / kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int64_t nq = (n + 255) >> 8;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    bool bad = false;
    int64_t mx = INT64_MIN, mn = INT64_MAX;
    for (int64_t q0 = ((int64_t)blockIdx.x * (blockDim.x / kWave) + wave) * R; q0 < nq; q0 += nwaves * R) {
        int64_t v[R][4], left[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int64_t q = q0 + r, i0 = (q << 8) + 4 * lane;
            if (i0 + 4 <= n) {
                const i64x2 a = *(const i64x2 *)(d + i0), b = *(const i64x2 *)(d + i0 + 2);
                v[r][0] = a.x; v[r][1] = a.y; v[r][2] = b.x; v[r][3] = b.y;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) v[r][k] = i0 + k < n ? d[i0 + k] : 0;
            }
            left[r] = (lane == 0 && q > 0 && q < nq) ? d[(q << 8) - 1] : 0;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int64_t q = q0 + r, i0 = (q << 8) + 4 * lane;
            if (q >= nq) break;                                       // wave-uniform
            const int64_t up = __shfl_up(v[r][3], 1, kWave);
            int64_t prev = lane == 0 ? left[r] : up;
            unsigned nib = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool in = i0 + k < n;
                const int64_t x = v[r][k];
                const bool first = i0 + k == 0;
                if (in) {
                    bad |= !first && x < prev;
                    mx = x > mx ? x : mx;
                    mn = x < mn ? x : mn;
                    if (first || x != prev) nib |= 1u << k;
                }
                prev = x;
            }
            uint64_t m = (uint64_t)nib << (4 * (lane & 15));
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) m |= __shfl_xor(m, off, kWave);
            const int64_t w = (q << 2) + (lane >> 4);
            if ((lane & 15) == 0 && w < ((n + 63) >> 6)) heads[w] = m;
        }
    }
    const bool anybad = __ballot(bad) != 0;
    mx = wave_reduce(mx, R_MAX);
    mn = wave_reduce(mn, R_MIN);
    if (lane == 0) { wmax[wave] = mx; wmin[wave] = mn; wbad[wave] = anybad ? 1 : 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t m = wmax[0], lo = wmin[0];
        int b = wbad[0];
        for (int w = 1; w < 256 / kWave; w++) { m = wmax[w] > m ? wmax[w] : m; lo = wmin[w] < lo ? wmin[w] : lo; b |= wbad[w]; }
        if (b) flag[0] = 1;
        if (m > *(volatile int64_t *)&flag[1]) atomicMax((long long *)&flag[1], (long long)m);
        if (lo < *(volatile int64_t *)&flag[2]) atomicMin((long long *)&flag[2], (long long)lo);
    }
}
hipError_t launch_sorted_heads(Src d, int64_t n, uint64_t *heads, int64_t *flag, hipStream_t s) {
    (void)hipGetLastError();
    k_sorted_init<<<1, 1, 0, s>>>(flag);
    if (n <= 0) return launch_status();
    int grid = grid_for(n, 256, 4);
    if (grid > 2048) grid = 2048;
    if (d.kind == SRC_I64 && ((uintptr_t)d.p & 15u) == 0 && !getenv("VDL_NO_DENSE_HEADS")) {
        int g4 = grid_for((n + 3) / 4, 256, 2);
        if (g4 > 2048) g4 = 2048;
        k_sorted_heads_dense<<<g4, 256, 0, s>>>((const int64_t *)d.p, n, heads, flag);
    } else k_sorted_heads<<<grid, 256, 0, s>>>(d, n, heads, flag);
    return launch_status();
}

// scratch layout is owned by the caller (vdl_engine.cpp); see launch_partition's arguments.
hipError_t launch_partition(Src data, const uint64_t *valid, int64_t n, int64_t pmin, int64_t pcount,
                            int64_t *hist /* 256*ntiles + 1 */, int64_t *scan_scratch /* prefix_sum_blocks(256*ntiles)+1 */,
                            uint64_t *keys_a, int64_t *slots_a, uint64_t *keys_b, int64_t *slots_b /* n each, or null if one pass */,
                            int64_t *n_valid_dev /* 1 word */, int64_t *pos_out, hipStream_t s, int64_t max_bucket, int64_t *order_out, int64_t *sorted_keys_out) {
    (void)hipGetLastError();   // see launch_status()
    if (n <= 0) return hipSuccess;
    int bits = 0;
    const int64_t top = (max_bucket >= 0 && max_bucket < pcount) ? max_bucket : pcount;      // buckets 0..pcount, or those known to occur
    while (bits < 63 && ((uint64_t)top >> bits) != 0) bits++;
    const int passes = bits <= 8 ? 1 : (bits + 7) / 8;
    const int64_t ntiles = partition_tiles(n);
    const int64_t hn = (int64_t)kRadix * ntiles;
    PartIn in{};
    in.data = data; in.valid = valid; in.pmin = pmin; in.pcount = pcount; in.n = n; in.n_dev = n_valid_dev;
    int nbits = 1;
    while (nbits < 63 && ((uint64_t)(n - 1) >> nbits) != 0) nbits++;    // slots 0..n-1
    in.slot_bits = (passes > 1 && bits + nbits <= 64 && !getenv("VDL_NO_PACKED_PARTITION")) ? nbits : 0;
    uint64_t *kin = nullptr, *kout = keys_a; int64_t *sin = nullptr, *sout = slots_a;
    for (int p = 0; p < passes; p++) {
        in.shift = 8 * p + in.slot_bits; in.keys = kin; in.slots = sin;
        const bool first = p == 0, last = p == passes - 1;
        if (first) k_part_hist<true><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist);
        else k_part_hist<false><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist);
        hipError_t e = launch_prefix_sum(hist, hn, scan_scratch, s);
        if (e != hipSuccess) return e;
        if (first) {   // number of non-EPS slots = grand total of the first histogram
            e = hipMemcpyAsync(n_valid_dev, scan_scratch + prefix_sum_blocks(hn), sizeof(int64_t), hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return e;
        }
        // order_out: the caller wants the slots in rank order (the inverse of the positions) -- the last pass then stores like a middle
        // pass, consecutive lanes on consecutive words, instead of one 8-byte store per slot at the slot's own address
        in.order_out = (last && order_out) ? 1 : 0;
        in.sorted_keys = in.order_out ? sorted_keys_out : nullptr;
        if (in.order_out) {
            if (first) k_part_scatter<true, false><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist, nullptr, order_out, nullptr);
            else k_part_scatter<false, false><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist, nullptr, order_out, nullptr);
        }
        else if (first && last) k_part_scatter<true, true><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist, nullptr, nullptr, pos_out);
        else if (first) k_part_scatter<true, false><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist, kout, sout, nullptr);
        else if (last) k_part_scatter<false, true><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist, nullptr, nullptr, pos_out);
        else k_part_scatter<false, false><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist, kout, sout, nullptr);
        kin = kout; sin = sout;
        kout = (kout == keys_a) ? keys_b : keys_a;
        sout = (sout == slots_a) ? slots_b : slots_a;
    }
    return launch_status();
}
int partition_passes(int64_t pcount) {
    int bits = 0;
    while (bits < 63 && ((uint64_t)pcount >> bits) != 0) bits++;
    return bits <= 8 ? 1 : (bits + 7) / 8;
}

// ------------------------------------------------------------------------------------------
// Row exchange for sharded Partition (Partition / join redistribution over
// xGMI).  Each rank sends every row of the partition key and of the vectors scattered by it to the
// rank that owns the row's key range; afterwards Partition / Scatter / Fold run locally on the
// received rows and the outputs of the ranks concatenate in rank order (keys ascend across ranks).
//   k_ex_dest   : destination rank of each row = (key - pmin) * world / pcount, EPS for rows that do
//                 not take part; per-destination row counts by 64-bit atomics (world <= 256)
//   (stable order inside each destination: launch_partition over the destination vector)
//   k_ex_pack   : scatter a column into send order; k_ex_mask: validity word of the source vectors
//   k_ex_unmask : received validity words -> one bitmap per source vector
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ex_dest(Src key, const uint64_t *vkey, int64_t n, int64_t pmin, int64_t pcount, int world,
                                                 int64_t *dest, uint64_t *vdest, int64_t *counts, int64_t *oob) {
    __shared__ unsigned long long cnt[kMaxExWorld + 1];       // per-block row counts per destination (+ out-of-range keys)
    for (int i = threadIdx.x; i <= kMaxExWorld; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave);
    for (int64_t w = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; w < nw; w += wstride) {
        const int64_t i = (w << 6) + lane;
        bool ok = i < n && bit(vkey, i);
        bool out_of_range = false;
        int64_t d = 0;
        if (ok) {
            const int64_t b = (int64_t)((uint64_t)ld(key, i) - (uint64_t)pmin);
            if (b < 0 || b >= pcount) { out_of_range = true; ok = false; }
            else d = (int64_t)(((unsigned __int128)(uint64_t)b * (uint64_t)world) / (uint64_t)pcount);
        }
        if (i < n) dest[i] = d;
        const uint64_t m = __ballot(ok);
        if (lane == 0) vdest[w] = m;
        const uint64_t bad = __ballot(out_of_range);
        if (bad && lane == 0) atomicAdd(&cnt[kMaxExWorld], (unsigned long long)__popcll(bad));
        // one LDS atomic per (wave, destination present in the wave)
        uint64_t todo = m;
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int64_t dl = __shfl(d, leader, kWave);
            const uint64_t same = __ballot(ok && d == dl) & todo;
            if (lane == leader) atomicAdd(&cnt[dl], (unsigned long long)__popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= kMaxExWorld; i += blockDim.x) {
        const unsigned long long c = cnt[i];
        if (c) atomicAdd((unsigned long long *)(i == kMaxExWorld ? oob : &counts[i]), c);
    }
}
hipError_t launch_ex_dest(Src key, const uint64_t *vkey, int64_t n, int64_t pmin, int64_t pcount, int world, int64_t *dest,
                          uint64_t *vdest, int64_t *counts, int64_t *oob, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    k_ex_dest<<<grid_for(n, 256, 4), 256, 0, s>>>(key, vkey, n, pmin, pcount, world, dest, vdest, counts, oob);
    return launch_status();
}

__global__ __launch_bounds__(256) void k_ex_pack(Src src, const uint64_t *vdest, const int64_t *pos, int64_t n, int64_t *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (bit(vdest, i)) out[pos[i]] = ld(src, i);
}
hipError_t launch_ex_pack(Src src, const uint64_t *vdest, const int64_t *pos, int64_t n, int64_t *out, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    k_ex_pack<<<grid_for(n, 256, 4), 256, 0, s>>>(src, vdest, pos, n, out);
    return launch_status();
}

// mask word per row: bit j = source vector j holds a value in that row (j < 63)
__global__ __launch_bounds__(256) void k_ex_mask(ExValid v, const uint64_t *vdest, const int64_t *pos, int64_t n, int64_t *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!bit(vdest, i)) continue;
        uint64_t m = 0;
        for (int j = 0; j < v.n; j++) m |= (uint64_t)bit(v.valid[j], i) << j;
        out[pos[i]] = (int64_t)m;
    }
}
hipError_t launch_ex_mask(const ExValid &v, const uint64_t *vdest, const int64_t *pos, int64_t n, int64_t *out, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    k_ex_mask<<<grid_for(n, 256, 4), 256, 0, s>>>(v, vdest, pos, n, out);
    return launch_status();
}

__global__ __launch_bounds__(256) void k_ex_unmask(const int64_t *mask, int64_t n, int j, uint64_t *valid) {
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave);
    for (int64_t w = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; w < nw; w += wstride) {
        const int64_t i = (w << 6) + lane;
        const uint64_t m = __ballot(i < n && (((uint64_t)mask[i < n ? i : 0] >> j) & 1ull));
        if (lane == 0) valid[w] = m;
    }
}
hipError_t launch_ex_unmask(const int64_t *mask, int64_t n, int j, uint64_t *valid, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    k_ex_unmask<<<grid_for(n, 256, 4), 256, 0, s>>>(mask, n, j, valid);
    return launch_status();
}

}  // namespace vdl
