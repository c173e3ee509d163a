// vdl_partition.hip -- device-wide prefix sum, Partition (stable LSD radix ranks) and the row exchange of a sharded
// Partition.
#include "vdl_device.h"

#include <algorithm>
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
// non-EPS slots, 8 bits per pass, as a ONE-SWEEP sort (round 4): ONE pass over the input up front counts every
// digit of every pass (k_part_digits), and each pass is then a single kernel that reads its tile once, ranks it, learns where
// its digits start from the tiles before it *while they run* (decoupled look-back, below) and stores the tile in digit order.
// Per pass the keys are read once and written once; the separate histogram pass and the device-wide prefix sum of the previous
// design (one more read of the keys + three launches per pass) are gone.
// The last pass writes out[slot] = rank, or -- lazy positions -- the slots in rank order.
// Dense group-by domains (Q1: 32 buckets) need one pass, Q3's 2^38 domain five.
//
// Look-back.  A tile's slots of digit d go to  goff[d] + (slots of digit d in all EARLIER tiles) + rank inside the tile.  The
// middle term is a prefix over tiles that are in flight at the same time.  The textbook chained scan walks back tile by tile
// until it meets a tile that knows its own prefix; on this chip a polled word costs 1-3 us under streaming load
// (MI355X_MICROARCH.md, handoff-1to1), ~500 tiles are in flight and a new one starts every few tens of ns, so such a walk is
// hundreds of tiles deep.  Here the tiles keep a FENWICK TREE of fan-out 16 instead: a node is one ROW of 256 status words
// {ready bit, count} -- row (u, 0) = the digit counts of tile u, row (u, j) = those of the 16^j tiles ending at u, made by tile u
// when 16^j divides u + 1 out of its own row (u, j-1) and its 15 siblings'.  The prefix of tile t is the sum of at most 15 rows
// per level (one per unit of each hex digit of t); which rows follows from t alone, so every load goes out at once and nothing is
// walked.  Rows are read by whole waves (a lane takes 4 digits), a block's waves share the rows out and add their partial sums in
// LDS.  A tile publishes its counts BEFORE it ranks its slots (a plain LDS histogram costs 16 atomics per lane) and asks for its
// prefix only after it has ranked and staged them, so what it asks for has usually been published for microseconds.  Measured at
// 60 M keys (tools/ubench/part_bench.hip): a binary tree polled right after ranking cost 392 us per pass against 250 us with the
// waits compiled out (VDL_PART_NOWAIT); this form waits 1.5-2 of a tile's 16 us and a pass takes 280 us.
// A status word is written once by one agent-scope relaxed store and polled with agent-scope relaxed loads (value and flag
// travel together: no fence).  Tiles take their number from a counter, so a tile only waits for tiles that already run.
// ------------------------------------------------------------------------------------------
// (tile shape: -DVDL_PART_BLOCK / -DVDL_PART_STEPS / -DVDL_PART_EU for tools/ubench/part_bench.hip's sweeps)
#ifndef VDL_PART_BLOCK
#define VDL_PART_BLOCK 512
#endif
#ifndef VDL_PART_STEPS
#define VDL_PART_STEPS 16
#endif
#ifndef VDL_PART_EU
#define VDL_PART_EU 4
#endif
constexpr int kPartBlock = VDL_PART_BLOCK, kPartSteps = VDL_PART_STEPS, kPartTile = kPartBlock * kPartSteps, kRadix = 256, kPartWaves = kPartBlock / kWave;
static_assert(kPartBlock >= kRadix, "threads 0..255 own a digit each");
constexpr int kPartMaxPasses = 8;
constexpr int kFanBits = 4, kFan = 1 << kFanBits, kFenLevels = 6;            // tiles < 16^6 (launch_partition checks)
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
    // one-sweep state of this pass
    const int64_t *goff;         // [256] where each digit starts in the output (exclusive scan of the up-front histogram)
    void *nodes;                 // Fenwick rows of this pass (partition_rows(ntiles) rows of 256 status words, zeroed)
    unsigned int *ticket;        // tile numbers are handed out in starting order (zeroed)
    int64_t ntiles;
#ifdef VDL_PART_TIMING
    unsigned long long *timing;  // [ntiles][8] phase timestamps (tools/ubench/part_bench.hip)
#endif
};
#ifdef VDL_PART_TIMING
#define PART_STAMP(k) do { if (tid == 0 && in.timing) in.timing[tile * 8 + (k)] = wall_clock64(); } while (0)
#else
#define PART_STAMP(k) do { } while (0)
#endif

// ---- status rows -------------------------------------------------------------------------------------------------------------
// row of node (u, level) in a pass's rows: level j holds the tiles u with 16^j | u + 1, in order
__host__ __device__ static inline int64_t partition_rows(int64_t ntiles) { int64_t r = 0; for (int j = 0; j < kFenLevels; j++) r += ntiles >> (kFanBits * j); return r; }
__device__ __forceinline__ int64_t fen_row(int64_t ntiles, int64_t u, int level) {
    int64_t base = 0;
    for (int j = 0; j < level; j++) base += ntiles >> (kFanBits * j);
    return base + ((u + 1) >> (kFanBits * level)) - 1;
}
// the i-th row of the prefix of tile t: one row per unit of each hex digit of t
__device__ __forceinline__ int fen_prefix_rows(int64_t t) { int r = 0; for (int j = 0; j < kFenLevels; j++) r += (int)((t >> (kFanBits * j)) & (kFan - 1)); return r; }
__device__ __forceinline__ int64_t fen_prefix_row(int64_t ntiles, int64_t t, int i) {
    for (int j = 0; j < kFenLevels; j++) {
        const int v = (int)((t >> (kFanBits * j)) & (kFan - 1));
        if (i < v) {
            const int64_t hi = t & ~(((int64_t)kFan << (kFanBits * j)) - 1);
            return fen_row(ntiles, hi + ((int64_t)(i + 1) << (kFanBits * j)) - 1, j);
        }
        i -= v;
    }
    return 0;   // not reached
}
// A wave reads a row with every lane taking 4 of its 256 words.  32-bit words: two 8-byte loads, the lane's digits are
// 2l, 2l+1, 128+2l, 129+2l; 64-bit words (counts beyond 2^31): four loads, digits l, 64+l, 128+l, 192+l.
template <class ST> struct StRow;
template <> struct StRow<uint32_t> {
    static constexpr uint32_t kReady = 0x80000000u;
    uint64_t a, b;
    __device__ __forceinline__ void fetch(const uint32_t *row, int lane) {
        const uint64_t *p = (const uint64_t *)row;
        a = __hip_atomic_load(p + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b = __hip_atomic_load(p + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ bool ready() const { return ((a & b) & 0x8000000080000000ull) == 0x8000000080000000ull; }
    __device__ __forceinline__ void add(int64_t (&acc)[4]) const {
        acc[0] += (int64_t)(a & 0x7fffffffull); acc[1] += (int64_t)((a >> 32) & 0x7fffffffull);
        acc[2] += (int64_t)(b & 0x7fffffffull); acc[3] += (int64_t)((b >> 32) & 0x7fffffffull);
    }
    __device__ static __forceinline__ int digit(int lane, int q) { return (q >> 1) * 128 + 2 * lane + (q & 1); }
};
template <> struct StRow<uint64_t> {
    static constexpr uint64_t kReady = 0x8000000000000000ull;
    uint64_t w[4];
    __device__ __forceinline__ void fetch(const uint64_t *row, int lane) {
#pragma unroll
        for (int q = 0; q < 4; q++) w[q] = __hip_atomic_load(row + 64 * q + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ bool ready() const { return ((w[0] & w[1] & w[2] & w[3]) & kReady) != 0; }
    __device__ __forceinline__ void add(int64_t (&acc)[4]) const {
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] += (int64_t)(w[q] & ~kReady);
    }
    __device__ static __forceinline__ int digit(int lane, int q) { return 64 * q + lane; }
};
// sum[d] += the counts of digit d in the `nrows` rows rowf(0) .. rowf(nrows - 1), for every d: the block's waves take four rows
// each per sweep, all of a wave's loads go out before the first is looked at, a row that is not there yet is polled.
template <class ST, class RowF, class AT>
__device__ __forceinline__ void add_rows(const ST *nodes, int nrows, RowF rowf, int wave, int lane, AT *sum /* LDS [256] */) {
    int64_t acc[4] = {0, 0, 0, 0};
    for (int base = wave * 4; base < nrows; base += kPartWaves * 4) {
        StRow<ST> r[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (base + q < nrows) r[q].fetch(nodes + rowf(base + q) * kRadix, lane);
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (base + q < nrows) {
#ifndef VDL_PART_NOWAIT                                           // (ablation, tools/ubench/part_bench.hip: wrong results, the data path's own time)
                while (!r[q].ready()) { __builtin_amdgcn_s_sleep(2); r[q].fetch(nodes + rowf(base + q) * kRadix, lane); }
#endif
                r[q].add(acc);
            }
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (acc[q]) atomicAdd(&sum[StRow<ST>::digit(lane, q)], (AT)acc[q]);
}

__device__ __forceinline__ int64_t part_bucket(const PartIn &in, int64_t x) {     // bucket = clamp(data - min, 0, cnt)
    int64_t b = 0;
    if (x > in.pmin) { b = (int64_t)((uint64_t)x - (uint64_t)in.pmin); if (b < 0 || b > in.pcount) b = in.pcount; }
    return b;
}
// histogram update of one step of a wave: when every lane holds the same digit (the high digits of neighbouring keys mostly do; 64
// lanes adding to one LDS word would serialise) lane 0 adds the wave's population once
template <bool FULL>
__device__ __forceinline__ void part_count(unsigned int *h, unsigned d, bool ok, int lane) {
    const unsigned d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
    if (__ballot(FULL ? d != d0 : (!ok || d != d0)) == 0) { if (lane == 0) atomicAdd(&h[d0], (unsigned)kWave); }      // (wave-uniform branch)
    else if (FULL || ok) atomicAdd(&h[d], 1u);
}
// the 8-bit digit at `shift` of a 64-bit word, from whichever half holds it (one or two 32-bit operations instead of a 64-bit shift)
__device__ __forceinline__ unsigned part_digit(uint64_t key, int shift) {
    const unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
    const unsigned x = shift >= 32 ? hi >> (shift - 32) : __builtin_amdgcn_alignbit(hi, lo, (unsigned)shift);     // (shift is uniform: a scalar select)
    return x & (kRadix - 1);
}

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
        for (int st = 0; st < kPartSteps; st++) {
            const int64_t b = part_bucket(in, (int64_t)keys[st]);
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

// The up-front histogram: every 8-bit digit of every bucket, counted in ONE pass over the data (ghist[pass][digit], pre-zeroed).
// A wave takes four 64-slot words per trip.
__global__ __launch_bounds__(256) void k_part_digits(PartIn in, int passes, unsigned long long *ghist) {
    constexpr int U = 4;
    __shared__ unsigned int h[kPartMaxPasses][kRadix];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    for (int i = tid; i < kPartMaxPasses * kRadix; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    const int64_t n = in.n, nw = (n + 63) >> 6;
    by_kind(in.data.kind, [&](auto kd) {
        for (int64_t w0 = ((int64_t)blockIdx.x * 4 + wave) * U; w0 < nw; w0 += (int64_t)gridDim.x * 4 * U) {
            int64_t x[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t i = ((w0 + u) << 6) + lane;
                ok[u] = i < n;
                x[u] = ldk_stream<decltype(kd)::value>(in.data, ok[u] ? i : 0);
            }
            if (in.valid) {
                uint64_t t[U];
#pragma unroll
                for (int u = 0; u < U; u++) t[u] = in.valid[w0 + u < nw ? w0 + u : nw - 1];
#pragma unroll
                for (int u = 0; u < U; u++) ok[u] = ok[u] & (((t[u] >> lane) & 1ull) != 0);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t b = (uint64_t)part_bucket(in, x[u]);
                for (int p = 0; p < passes; p++) part_count<false>(h[p], (unsigned)((b >> (8 * p)) & (kRadix - 1)), ok[u], lane);
            }
        }
    });
    __syncthreads();
    for (int i = tid; i < passes * kRadix; i += 256) {
        const unsigned c = (&h[0][0])[i];
        if (c) atomicAdd(&ghist[i], (unsigned long long)c);
    }
}
// one block per pass: exclusive scan of its 256 counts; block 0 also leaves the number of non-EPS slots
__global__ __launch_bounds__(kRadix) void k_part_digit_offsets(const unsigned long long *ghist, int64_t *goff, int64_t *n_valid) {
    __shared__ int64_t wsum[kRadix / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int64_t c = (int64_t)ghist[blockIdx.x * kRadix + tid];
    int64_t incl = c;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) { const int64_t y = __shfl_up(incl, off, kWave); if (lane >= off) incl += y; }
    if (lane == kWave - 1) wsum[wave] = incl;
    __syncthreads();
    int64_t pre = 0;
    for (int w = 0; w < wave; w++) pre += wsum[w];
    goff[blockIdx.x * kRadix + tid] = pre + incl - c;
    if (blockIdx.x == 0 && tid == kRadix - 1) *n_valid = pre + incl;
}

// One radix pass over one tile per block.  Order of events: fetch the tile; count its digits (LDS histogram) and PUBLISH the row --
// a tile that completes a group of 16 / 256 / ... also publishes the wider rows; rank the slots (each wave owns a contiguous share
// of the tile -- 16 steps of 64 slots --, so the stable order inside a tile is wave, step, lane: a wave ranks its slots on its own,
// match masks in LDS, its running digit counts in its row of whist: LDS operations of one wave execute in order); turn
// the rows of whist into offsets inside the sorted tile; stage the tile in LDS in digit order; only then collect the prefix over
// the earlier tiles (add_rows) and store.
// The pass is bound by instruction issue, not by memory (a timeline of phase stamps per tile, tools/ubench/part_bench.hip
// -DVDL_PART_TIMING: 18 us per tile of which 6 waiting for loads and stores, two tiles per CU): hence a body without validity
// tests for full tiles (FULL), digits extracted once, 32-bit destinations while the Partition has fewer than 2^31 slots.
// MODE 0: middle pass (sorted pairs out); 1: last pass of a lazy Partition (slots in rank order, optionally the key values);
// 2: last pass writing pos_out[slot] = rank.
template <int MODE, class ST> struct PartLds {
    using IT = typename std::conditional<sizeof(ST) == 4, unsigned int, unsigned long long>::type;     // destinations: modulo 2^32 while they fit
    unsigned int whist[kPartWaves][kRadix];                     // per wave and digit: count, later offset inside the sorted tile
    unsigned int chist[kRadix];                                 // the tile's digit counts
    IT gdelta[kRadix];                                          // destination of sorted index idx with digit d = gdelta[d] + idx
    IT before[kRadix];                                          // slots of digit d in the earlier tiles
    unsigned int wtot[kRadix / kWave];
    unsigned int tile;
    // The tile is put into digit order in LDS before it is stored (keys, then slots through the same buffer): a digit's slots of
    // one tile are neighbours at the destination, so consecutive lanes then store consecutive words instead of 64 scattered ones.
    // Until then the area holds the waves' match masks (NW x 256 words).
    uint64_t stage[MODE == 2 ? kPartWaves * kRadix : kPartTile];
};
static_assert(kPartTile >= kPartWaves * kRadix, "the match masks live in the staging area");

template <bool FIRST, bool PACKED, int MODE, class ST, bool FULL>
__device__ __forceinline__ void part_pass_tile(const PartIn &in, PartLds<MODE, ST> &L, const int64_t tile, const int64_t n,
                                               uint64_t *keys_out, int64_t *slots_out, int64_t *pos_out) {
    constexpr int NW = kPartWaves;
    using IT = typename PartLds<MODE, ST>::IT;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    PART_STAMP(0);
    uint64_t keys[kPartSteps];
    int64_t slots[kPartSteps];
    bool oks[kPartSteps];
    part_fetch_share<FIRST>(in, n, tile * kPartTile + (int64_t)wave * (kPartSteps * kWave), lane, keys, slots, oks);
#define OK(st) (FULL || oks[st])
#ifdef VDL_PART_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#endif
    PART_STAMP(1);
    unsigned int dg[kPartSteps];
#pragma unroll
    for (int st = 0; st < kPartSteps; st++) dg[st] = part_digit(keys[st], in.shift);
    ST *const nodes = (ST *)in.nodes;
#pragma unroll
    for (int st = 0; st < kPartSteps; st++) part_count<FULL>(L.chist, dg[st], OK(st), lane);
    __syncthreads();
    const unsigned cnt = tid < kRadix ? L.chist[tid] : 0u;
    PART_STAMP(2);
    // the tile's row, and the wider rows of a tile that ends a group of 16 / 256 / ...
    if (tid < kRadix)
        __hip_atomic_store(nodes + fen_row(in.ntiles, tile, 0) * kRadix + tid, (ST)((ST)cnt | StRow<ST>::kReady), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (((tile + 1) & (kFan - 1)) == 0) {                       // (one tile in 16; block-uniform)
        ST acc = (ST)cnt;
        for (int j = 1; j < kFenLevels && ((tile + 1) & (((int64_t)1 << (kFanBits * j)) - 1)) == 0; j++) {
            const int64_t step = (int64_t)1 << (kFanBits * (j - 1));
            add_rows<ST>(nodes, kFan - 1, [&](int i) { return fen_row(in.ntiles, tile - (int64_t)(i + 1) * step, j - 1); }, wave, lane, L.before);
            __syncthreads();
            if (tid < kRadix) {
                acc += (ST)L.before[tid];
                L.before[tid] = 0;
                __hip_atomic_store(nodes + fen_row(in.ntiles, tile, j) * kRadix + tid, (ST)(acc | StRow<ST>::kReady), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
    }
    PART_STAMP(3);
    // Which lanes of the wave hold the same digit?  Every lane ORs its bit into the wave's mask of its digit (LDS, one 64-bit word per
    // digit; the table lies in the staging area, which is not in use yet) and reads the word back: LDS operations of one wave execute
    // in order, so the read sees all 64 contributions.  (Eight ballots per step with per-lane 64-bit mask arithmetic -- 70 vector
    // instructions for every 64 keys -- were the first version.)  Three sweeps over the 16 steps, each a train of LDS operations that
    // never waits for the one before (a dependent round trip per step made ranking 5.3 of a tile's 18 us):
    //   1. OR the lane's bit in, read the word back, write 0 over it -- issued back to back, the LDS runs them in that order;
    //   2. the lowest lane of each digit adds the digit's population to the wave's running count (atomic with return: the
    //      returned values are the counts before each step, in step order);
    //   3. its peers fetch that value from it (a cross-lane read; through the digit's LDS word it took half as long again).
    unsigned int local[kPartSteps];                             // rank among this wave's slots with the same digit
    {
        unsigned int *mine = L.whist[wave];
        volatile unsigned long long *mm = (volatile unsigned long long *)L.stage + wave * kRadix;
        unsigned int info[kPartSteps];                          // rank | population << 8 | lowest peer lane << 16
#pragma unroll
        for (int st = 0; st < kPartSteps; st++) {
            if (OK(st)) __hip_atomic_fetch_or((unsigned long long *)mm + dg[st], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __builtin_amdgcn_wave_barrier();
            const unsigned long long peers = OK(st) ? mm[dg[st]] : (1ull << lane);
            __builtin_amdgcn_wave_barrier();
            if (OK(st)) mm[dg[st]] = 0ull;
            __builtin_amdgcn_wave_barrier();
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0u));
            info[st] = rank | ((unsigned)__popcll(peers) << 8) | ((unsigned)(__ffsll((long long)peers) - 1) << 16);
        }
#pragma unroll
        for (int st = 0; st < kPartSteps; st++) {
            unsigned pre = 0;
            if (OK(st) && (info[st] & 0xffu) == 0) pre = __hip_atomic_fetch_add(mine + dg[st], (info[st] >> 8) & 0xffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __builtin_amdgcn_wave_barrier();
            local[st] = pre;
        }
#pragma unroll
        for (int st = 0; st < kPartSteps; st++) local[st] = (unsigned)__shfl((int)local[st], (int)(info[st] >> 16), kWave) + (info[st] & 0xffu);
    }
    PART_STAMP(4);
    unsigned incl = cnt;                                        // exclusive scan of the tile's digit counts over threads 0 .. 255
    if (tid < kRadix) {
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) { const unsigned y = __shfl_up(incl, off, kWave); if (lane >= off) incl += y; }
        if (lane == kWave - 1) L.wtot[wave] = incl;
    }
    __syncthreads();
    unsigned total = 0;
#pragma unroll
    for (int w = 0; w < kRadix / kWave; w++) total += L.wtot[w];
    if (tid < kRadix) {
        unsigned pre = 0;
        for (int w = 0; w < wave; w++) pre += L.wtot[w];
        const unsigned tstart = pre + incl - cnt;               // where digit `tid` starts inside the sorted tile
        unsigned run = tstart;
#pragma unroll
        for (int w = 0; w < NW; w++) { const unsigned c = L.whist[w][tid]; L.whist[w][tid] = run; run += c; }
        L.gdelta[tid] = (IT)in.goff[tid] - (IT)tstart;
    }
    __syncthreads();
    unsigned lpos[kPartSteps];
#pragma unroll
    for (int st = 0; st < kPartSteps; st++) lpos[st] = L.whist[wave][dg[st]] + local[st];
    const int nrows = fen_prefix_rows(tile);
    auto prefix_row = [&](int i) { return fen_prefix_row(in.ntiles, tile, i); };
    if (MODE != 2) {
#pragma unroll
        for (int st = 0; st < kPartSteps; st++)
            if (OK(st)) L.stage[lpos[st]] = keys[st];
    }
    PART_STAMP(5);
    add_rows<ST>(nodes, nrows, prefix_row, wave, lane, L.before);
    __syncthreads();
    if (tid < kRadix) L.gdelta[tid] += L.before[tid];
    __syncthreads();
    PART_STAMP(6);
    if (MODE == 2) {                                            // ranks go to out[slot]: scattered whatever the order
#pragma unroll
        for (int st = 0; st < kPartSteps; st++)
            if (OK(st)) pos_out[slots[st]] = (int64_t)(IT)(L.gdelta[dg[st]] + (IT)lpos[st]);
        return;
    }
    IT dest[kPartSteps];
#pragma unroll
    for (int k = 0; k < kPartSteps; k++) {
        const unsigned idx = (unsigned)k * kPartBlock + tid;
        dest[k] = 0;
        if (FULL || idx < total) {
            const uint64_t key = L.stage[idx];
            dest[k] = L.gdelta[part_digit(key, in.shift)] + (IT)idx;
            if (MODE == 0) keys_out[dest[k]] = key;
            else {
                if (PACKED) slots_out[dest[k]] = (int64_t)(key & ((1ull << in.slot_bits) - 1));     // the last pass of a lazy Partition: slots in rank order
                if (in.sorted_keys) in.sorted_keys[dest[k]] = (int64_t)((uint64_t)in.pmin + (PACKED ? key >> in.slot_bits : key));
            }
        }
    }
#ifdef VDL_PART_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PART_STAMP(7);
    if (PACKED) return;                                         // packed: the slot travelled inside the key word
    __syncthreads();
#pragma unroll
    for (int st = 0; st < kPartSteps; st++)
        if (OK(st)) L.stage[lpos[st]] = (uint64_t)slots[st];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPartSteps; k++)
        if (FULL || (unsigned)k * kPartBlock + tid < total) slots_out[dest[k]] = (int64_t)L.stage[(unsigned)k * kPartBlock + tid];
#undef OK
}

template <bool FIRST, bool PACKED, int MODE, class ST>
__global__ __launch_bounds__(kPartBlock, VDL_PART_EU) void k_part_pass(PartIn in, uint64_t *keys_out, int64_t *slots_out, int64_t *pos_out) {
    __shared__ PartLds<MODE, ST> L;
    const int tid = threadIdx.x;
#ifdef VDL_PART_NOTICKET                                        // (ablation, tools/ubench/part_sweep.sh: tiles by block number -- no guarantee that earlier tiles run)
    if (tid == 0) L.tile = blockIdx.x;
#else
    if (tid == 0) L.tile = atomicAdd(in.ticket, 1u);
#endif
#pragma unroll
    for (int k = 0; k < kPartWaves * kRadix / kPartBlock; k++) { (&L.whist[0][0])[k * kPartBlock + tid] = 0; L.stage[k * kPartBlock + tid] = 0; }
    if (tid < kRadix) { L.chist[tid] = 0; L.before[tid] = 0; }
    __syncthreads();
    const int64_t tile = L.tile;
    const int64_t n = FIRST ? in.n : *in.n_dev;
    if (tile * kPartTile >= n) return;                          // (whole block; the tiles before it are all full: nobody waits for this one)
    if ((tile + 1) * kPartTile <= n && !(FIRST && in.valid)) part_pass_tile<FIRST, PACKED, MODE, ST, true>(in, L, tile, n, keys_out, slots_out, pos_out);
    else part_pass_tile<FIRST, PACKED, MODE, ST, false>(in, L, tile, n, keys_out, slots_out, pos_out);
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
// The sortedness pass, the heads' tile counts, their scan and the report to the host in ONE launch (round 4: the five launches it replaces
// -- init, heads, count, scan, post -- were 40 us of a 490 us Q3 at SF10, most of it the gaps between them).  A block owns whole compaction
// tiles (4096 entries = 16 groups of 256, two per wave in flight) and STORES its tiles' head counts, so nothing has to be zeroed first; the
// block that finishes last (two levels of arrival counters in `state`, which lives with the context and is left all zero as it was found)
// gathers the blocks' verdicts -- {descends, largest, smallest}, three words per block behind the counts --, scans the counts in place,
// appends {total, descends, largest, smallest} and posts those four words and the sequence number into pinned host memory with
// system-scope stores (vdl_ctx::wait_seq polls it).  Counts, verdicts and arrivals cross the blocks as agent-scope atomics: the XCDs' L2s
// do not see one another's plain stores inside a kernel.
constexpr int kHeadMaxGrid = 2048;
constexpr int kHeadArrivals = 64;          // arrival counters of k_sorted_heads_counted (+ the one their last arrivals meet on): sorted_heads_state_words() words of state
constexpr int kHeadTileGroups = 16;         // 256-entry groups per compaction tile (compact_tile() = 4096, vdl_ops.hip)
__global__ __launch_bounds__(256) void k_sorted_heads_counted(const int64_t *__restrict__ d, int64_t n, uint64_t *__restrict__ heads, int64_t *counts, int64_t nb,
                                                              int64_t *state, int64_t *pin, int64_t *pflag, int64_t seq) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    constexpr int R = 2;
    __shared__ int64_t wmax[256 / kWave], wmin[256 / kWave];
    __shared__ int wbad[256 / kWave];
    __shared__ int last_block;
    __shared__ int64_t wcount[256 / kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int64_t nq = (n + 255) >> 8;
    bool bad = false;
    int64_t mx = INT64_MIN, mn = INT64_MAX;
    // a BLOCK owns a compaction tile (its four waves take two neighbouring groups each, twice), blocks take neighbouring tiles: what is in
    // flight at any moment is one contiguous stretch of the vector.  (A wave per tile -- 2 000 streams 32 KB apart -- ran at a third of
    // the rate: 112-184 us for 260 MB.)
    for (int64_t t = blockIdx.x; t < nb; t += gridDim.x) {
        int found = 0;
        for (int gi = 0; gi < kHeadTileGroups; gi += R * (256 / kWave)) {
            const int64_t q0 = t * kHeadTileGroups + gi + wave * R;
            if (q0 >= nq) continue;                                       // wave-uniform
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
                found += __popc(nib);
                uint64_t m = (uint64_t)nib << (4 * (lane & 15));
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) m |= __shfl_xor(m, off, kWave);
                const int64_t w = (q << 2) + (lane >> 4);
                if ((lane & 15) == 0 && w < ((n + 63) >> 6)) heads[w] = m;
            }
        }
        const int64_t total = wave_reduce((int64_t)found, R_SUM);
        if (lane == 0) wcount[wave] = total;
        __syncthreads();
        if (threadIdx.x == 0) {
            int64_t sum = 0;
            for (int w = 0; w < 256 / kWave; w++) sum += wcount[w];
            __hip_atomic_store(&counts[t], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    // What crosses the blocks -- the tile counts, the flags, the arrival counter -- are agent-scope atomics, which are performed where every
    // XCD sees them; a wave waits for ITS stores to be acknowledged (a workgroup-scope release is that wait and nothing else) before the block
    // announces its arrival.  An agent-scope release fence instead writes back the XCD's whole L2 (buffer_wbl2) once per block: 2 000 blocks
    // at 32 M entries spent 240 us in a pass that moves 260 MB.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const bool anybad = __ballot(bad) != 0;
    mx = wave_reduce(mx, R_MAX);
    mn = wave_reduce(mn, R_MIN);
    if (lane == 0) { wmax[wave] = mx; wmin[wave] = mn; wbad[wave] = anybad ? 1 : 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t m = wmax[0], lo = wmin[0];
        int b = wbad[0];
        for (int w = 1; w < 256 / kWave; w++) { m = wmax[w] > m ? wmax[w] : m; lo = wmin[w] < lo ? wmin[w] : lo; b |= wbad[w]; }
        // this block's verdict in its own three words behind the counts; its arrival on one of kHeadArrivals counters, the last of each on
        // the top one.  (One counter and one pair of atomic max / min words for everybody: same-address atomics are carried out one after
        // the other where all XCDs meet, 50-70 ns each -- 4 096 blocks spent 290 us on a pass that moves 260 MB.)
        int64_t *rec = counts + nb + 4 + 3 * (int64_t)blockIdx.x;
        __hip_atomic_store(&rec[0], (int64_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&rec[1], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&rec[2], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // this block's counts and verdict have arrived before it says so
        const int64_t groups = (int64_t)gridDim.x < kHeadArrivals ? (int64_t)gridDim.x : kHeadArrivals;
        const int64_t mine = blockIdx.x % kHeadArrivals, members = ((int64_t)gridDim.x - mine + kHeadArrivals - 1) / kHeadArrivals;
        int last = 0;
        if (__hip_atomic_fetch_add(&state[mine], (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            __hip_atomic_store(&state[mine], (int64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           // as the next launch wants to find it
            if (__hip_atomic_fetch_add(&state[kHeadArrivals], (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1) {
                __hip_atomic_store(&state[kHeadArrivals], (int64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        last_block = last;
    }
    __syncthreads();
    if (!last_block) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- the last block: everybody's verdicts
    {
        int64_t m = INT64_MIN, lo = INT64_MAX;
        bool b = false;
        for (int64_t k = threadIdx.x; k < (int64_t)gridDim.x; k += 256) {
            const int64_t *rec = counts + nb + 4 + 3 * k;
            b |= __hip_atomic_load(&rec[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            const int64_t x = __hip_atomic_load(&rec[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), y = __hip_atomic_load(&rec[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            m = x > m ? x : m; lo = y < lo ? y : lo;
        }
        const bool anyb = __ballot(b) != 0;
        m = wave_reduce(m, R_MAX); lo = wave_reduce(lo, R_MIN);
        __syncthreads();                                                  // (wmax / wmin / wbad are read above by thread 0 only, before the barrier)
        if (lane == 0) { wmax[wave] = m; wmin[wave] = lo; wbad[wave] = anyb ? 1 : 0; }
        __syncthreads();
    }
    // ---- the last block: exclusive scan of the tile counts in place, total behind them
    constexpr int K = 8;
    __shared__ int64_t wsum[256 / kWave];
    __shared__ int64_t carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 256 * K) {
        const int64_t i0 = base + (int64_t)tid * K;
        int64_t x[K], sum = 0;
#pragma unroll
        for (int k = 0; k < K; k++) { x[k] = i0 + k < nb ? __hip_atomic_load(&counts[i0 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0; sum += x[k]; }
        int64_t incl = sum;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int64_t y = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += y;
        }
        if (lane == kWave - 1) wsum[wave] = incl;
        __syncthreads();
        int64_t wprefix = 0;
        for (int w = 0; w < wave; w++) wprefix += wsum[w];
        int64_t run = carry + wprefix + incl - sum;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++) { if (i0 + k < nb) counts[i0 + k] = run; run += x[k]; }
        if (tid == 255) carry = run;
        __syncthreads();
    }
    if (tid == 0) {
        int64_t m = wmax[0], lo = wmin[0];
        int b = wbad[0];
        for (int w = 1; w < 256 / kWave; w++) { m = wmax[w] > m ? wmax[w] : m; lo = wmin[w] < lo ? wmin[w] : lo; b |= wbad[w]; }
        const int64_t out[4] = {carry, (int64_t)b, m, lo};
        for (int k = 0; k < 4; k++) { counts[nb + k] = out[k]; if (pin) __hip_atomic_store(pin + k, out[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        if (pflag) {
            __threadfence_system();
            __hip_atomic_store(pflag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void k_sorted_state_init(int64_t *state) { if (threadIdx.x <= kHeadArrivals) state[threadIdx.x] = 0; }
hipError_t launch_sorted_state_init(int64_t *state, hipStream_t s) {
    (void)hipGetLastError();
    k_sorted_state_init<<<1, 128, 0, s>>>(state);
    return launch_status();
}
int64_t sorted_heads_state_words() { return kHeadArrivals + 1; }
int64_t sorted_heads_counts_words(int64_t n) {                            // tile counts, {total, descends, largest, smallest}, a verdict per block
    const int64_t nb = (n + compact_tile() - 1) / compact_tile();
    return nb + 4 + 3 * std::min<int64_t>(std::max<int64_t>(nb, 1), kHeadMaxGrid);
}
bool sorted_heads_counted_serves(Src d, int64_t n) { return n > 0 && d.kind == SRC_I64 && ((uintptr_t)d.p & 15u) == 0 && !getenv("VDL_NO_DENSE_HEADS"); }
hipError_t launch_sorted_heads_counted(const int64_t *d, int64_t n, uint64_t *heads, int64_t *counts, int64_t *state, int64_t *pinned_dst, int64_t *pinned_flag,
                                       int64_t seq, hipStream_t s) {
    (void)hipGetLastError();
    const int64_t nb = (n + compact_tile() - 1) / compact_tile();
    int64_t grid = nb;                                                    // a tile per block and trip
    if (grid > kHeadMaxGrid) grid = kHeadMaxGrid;
    k_sorted_heads_counted<<<(int)grid, 256, 0, s>>>(d, n, heads, counts, nb, state, pinned_dst, pinned_flag, seq);
    return launch_status();
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

// Scratch of one Partition: tickets, the up-front histogram and its scan, and per pass the Fenwick nodes of its tiles
// (partition_rows(tiles) rows of 256 status words: 32-bit words while a count fits 31 bits).
static int partition_bits(int64_t top) { int bits = 0; while (bits < 63 && ((uint64_t)top >> bits) != 0) bits++; return bits; }
static bool partition_narrow_status(int64_t n) { return n < ((int64_t)1 << 31); }
static size_t partition_head_bytes() { return 64 + 2 * sizeof(int64_t) * (size_t)kPartMaxPasses * kRadix; }
static size_t partition_node_bytes(int64_t n) { return (size_t)partition_rows(partition_tiles(n)) * kRadix * (partition_narrow_status(n) ? 4 : 8); }
size_t partition_scratch_bytes(int64_t n, int64_t pcount) {
    return partition_head_bytes() + (size_t)partition_passes(pcount) * partition_node_bytes(n);
}

#ifdef VDL_PART_TIMING
unsigned long long *g_part_timing = nullptr;
#endif
template <bool FIRST, bool PACKED, int MODE>
static void launch_part_pass(const PartIn &in, bool narrow, uint64_t *keys_out, int64_t *slots_out, int64_t *pos_out, hipStream_t s) {
    if (narrow) k_part_pass<FIRST, PACKED, MODE, uint32_t><<<(int)in.ntiles, kPartBlock, 0, s>>>(in, keys_out, slots_out, pos_out);
    else k_part_pass<FIRST, PACKED, MODE, uint64_t><<<(int)in.ntiles, kPartBlock, 0, s>>>(in, keys_out, slots_out, pos_out);
}
template <bool FIRST, bool PACKED>
static void launch_part_pass_mode(const PartIn &in, int mode, bool narrow, uint64_t *keys_out, int64_t *slots_out, int64_t *pos_out, hipStream_t s) {
    if (mode == 0) launch_part_pass<FIRST, PACKED, 0>(in, narrow, keys_out, slots_out, pos_out, s);
    else if (mode == 1) launch_part_pass<FIRST, PACKED, 1>(in, narrow, keys_out, slots_out, pos_out, s);
    else launch_part_pass<FIRST, PACKED, 2>(in, narrow, keys_out, slots_out, pos_out, s);
}

hipError_t launch_partition(Src data, const uint64_t *valid, int64_t n, int64_t pmin, int64_t pcount, void *scratch /* partition_scratch_bytes(n, pcount) */,
                            uint64_t *keys_a, int64_t *slots_a, uint64_t *keys_b, int64_t *slots_b /* n each, or null if one pass */,
                            int64_t *n_valid_dev /* 1 word */, int64_t *pos_out, hipStream_t s, int64_t max_bucket, int64_t *order_out, int64_t *sorted_keys_out) {
    (void)hipGetLastError();   // see launch_status()
    if (n <= 0) return hipSuccess;
    if (partition_tiles(n) >= ((int64_t)1 << (kFanBits * kFenLevels))) return hipErrorInvalidValue;      // (2^37 slots: the look-back's level count)
    const int64_t top = (max_bucket >= 0 && max_bucket < pcount) ? max_bucket : pcount;      // buckets 0..pcount, or those known to occur
    const int bits = partition_bits(top);
    const int passes = bits <= 8 ? 1 : (bits + 7) / 8;
    const int64_t ntiles = partition_tiles(n);
    const bool narrow = partition_narrow_status(n);
    PartIn in{};
    in.data = data; in.valid = valid; in.pmin = pmin; in.pcount = pcount; in.n = n; in.n_dev = n_valid_dev; in.ntiles = ntiles;
    int nbits = 1;
    while (nbits < 63 && ((uint64_t)(n - 1) >> nbits) != 0) nbits++;    // slots 0..n-1
    in.slot_bits = (passes > 1 && bits + nbits <= 64 && !getenv("VDL_NO_PACKED_PARTITION")) ? nbits : 0;
    // scratch: tickets | ghist | goff | nodes of pass 0, 1, ...   (tickets, ghist and the nodes start at zero)
    unsigned int *tickets = (unsigned int *)scratch;
    unsigned long long *ghist = (unsigned long long *)((char *)scratch + 64);
    int64_t *goff = (int64_t *)(ghist + (size_t)kPartMaxPasses * kRadix);
    char *nodes = (char *)scratch + partition_head_bytes();
    hipError_t e = hipMemsetAsync(scratch, 0, partition_head_bytes() + (size_t)passes * partition_node_bytes(n), s);
    if (e != hipSuccess) return e;
    {
        const int grid = (int)std::min<int64_t>((n + 1023) / 1024, 2048);
        k_part_digits<<<grid, 256, 0, s>>>(in, passes, ghist);
        k_part_digit_offsets<<<passes, kRadix, 0, s>>>(ghist, goff, n_valid_dev);
    }
    uint64_t *kin = nullptr, *kout = keys_a; int64_t *sin = nullptr, *sout = slots_a;
    for (int p = 0; p < passes; p++) {
        in.shift = 8 * p + in.slot_bits; in.keys = kin; in.slots = sin;
        in.goff = goff + (size_t)p * kRadix; in.nodes = nodes + (size_t)p * partition_node_bytes(n); in.ticket = tickets + p;
#ifdef VDL_PART_TIMING
        in.timing = p == VDL_PART_TIMING ? g_part_timing : nullptr;
#endif
        const bool first = p == 0, last = p == passes - 1;
        // order_out: the caller wants the slots in rank order (the inverse of the positions) -- the last pass then stores like a middle
        // pass, consecutive lanes on consecutive words, instead of one 8-byte store per slot at the slot's own address
        in.order_out = (last && order_out) ? 1 : 0;
        in.sorted_keys = in.order_out ? sorted_keys_out : nullptr;
        const int mode = in.order_out ? 1 : last ? 2 : 0;
        uint64_t *ko = mode == 0 ? kout : nullptr;
        int64_t *so = mode == 1 ? order_out : mode == 0 ? sout : nullptr;
        if (first) { if (in.slot_bits) launch_part_pass_mode<true, true>(in, mode, narrow, ko, so, pos_out, s); else launch_part_pass_mode<true, false>(in, mode, narrow, ko, so, pos_out, s); }
        else { if (in.slot_bits) launch_part_pass_mode<false, true>(in, mode, narrow, ko, so, pos_out, s); else launch_part_pass_mode<false, false>(in, mode, narrow, ko, so, pos_out, s); }
        kin = kout; sin = sout;
        kout = (kout == keys_a) ? keys_b : keys_a;
        sout = (sout == slots_a) ? slots_b : slots_a;
    }
    return launch_status();
}
int partition_passes(int64_t pcount) {
    const int bits = partition_bits(pcount);
    return bits <= 8 ? 1 : (bits + 7) / 8;
}

// ------------------------------------------------------------------------------------------
// Row exchange for sharded Partition (Partition / join redistribution over
// xGMI).  Each rank sends every row of the partition key and of the vectors scattered by it to the
// rank that owns the row's key range; afterwards Partition / Scatter / Fold run locally on the
// received rows and the outputs of the ranks concatenate in rank order (keys ascend across ranks).
//   destination rank of a row = (key - pmin) * world / pcount, or the balanced cut's slice -> rank table; rows without a key do not
//   take part; the order inside each destination is the rows' own (stable)
//   k_ex_count / k_ex_offsets / k_ex_bases / k_ex_pack_all : "routing" below
//   k_ex_unmask : received validity words -> one bitmap per source vector
// ------------------------------------------------------------------------------------------
//   k_ex_hist   : (round 4) how the keys spread over kExBins equal slices of the pivots' domain: the ranks all-gather these histograms and
//                 cut the domain where the DATA is, so that every rank receives about as many rows (`owner`: slice -> rank).  With the
//                 declared domain cut evenly, TPC-H's order keys -- which reach 0.56 of their power-of-two domain -- left the last three
//                 of eight ranks without a row.
// slice of the pivots' domain a key offset b in [0, pcount) falls into: slices are a power of two wide -- the narrowest that leaves at most
// kExBins of them (so between kExBins / 2 + 1 and kExBins are in use) --, which makes the slice a shift instead of a 128-bit division per row
__host__ __device__ inline int ex_slice_shift(int64_t pcount) {
    int sh = 0;
    while (sh < 62 && ((pcount - 1) >> sh) >= kExBins) sh++;
    return sh;
}
__global__ __launch_bounds__(256) void k_ex_hist(Src key_src, const uint64_t *vkey, int64_t n, int64_t pmin, int64_t pcount, int shift, unsigned long long *hist /* kExBins + 1 */) {
    __shared__ unsigned int cnt[kExBins + 1];
    for (int i = threadIdx.x; i <= kExBins; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave);
    constexpr int U = 4;                                        // bitmap words per wave and trip (their loads in flight together)
    for (int64_t w0 = ((int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave) * U; w0 < nw; w0 += wstride * U) {
        int64_t key[U];
        bool okv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = ((w0 + u) << 6) + lane;
            okv[u] = i < n && bit(vkey, i);
            key[u] = ld(key_src, okv[u] ? i : 0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool ok = okv[u];
            int slot = kExBins;
            if (ok) {
                const int64_t b = (int64_t)((uint64_t)key[u] - (uint64_t)pmin);
                if (b >= 0 && b < pcount) slot = (int)(b >> shift);
            }
            // a table clustered by the key puts a whole wave into one slice (64 LDS atomics on one address, one after the other): count once then
            const uint64_t live = __ballot(ok);
            if (!live) continue;
            const int first = __shfl(slot, __ffsll((long long)live) - 1, kWave);
            const uint64_t same = __ballot(ok && slot == first);
            if (same == live) { if (lane == 0) atomicAdd(&cnt[first], (unsigned)__popcll(live)); }
            else if (ok) atomicAdd(&cnt[slot], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= kExBins; i += blockDim.x) { const unsigned c = cnt[i]; if (c) atomicAdd(&hist[i], (unsigned long long)c); }
}
hipError_t launch_ex_hist(Src key, const uint64_t *vkey, int64_t n, int64_t pmin, int64_t pcount, int64_t *hist, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    int grid = grid_for(n, 256, 16);
    if (grid > 2048) grid = 2048;
    k_ex_hist<<<grid, 256, 0, s>>>(key, vkey, n, pmin, pcount, ex_slice_shift(pcount), (unsigned long long *)hist);
    return launch_status();
}

// ---- routing (round 4, second form): no destination vector, no positions, no sort ----------------------------------------------
// A row's place in the send buffer = rows of its destination in earlier tiles + earlier rows of its destination in its own tile.
//   k_ex_count    : a block per tile of kExTile rows: the rows per destination of the tile (tilecnt[destination][tile]) and of the rank
//   k_ex_offsets  : a block per destination: its tile counts scanned in place, the rows of the destinations before it added
//   k_ex_pack_all : a block per tile again: the rows' destinations once more (a shift and a table lookup), their rank inside the tile
//                   (per wave by matching ballots, across the tile's 64 wave-slices by a small table in LDS), and every column --
//                   and the mask word of the vectors' validity -- written to its place.
// The first form wrote a destination per row, ran one pass of the radix Partition over it for the stable positions, and packed column
// by column through them: 2.0 ms per 60 M rows of three columns (TPC-H Q18's GROUP BY over all of lineitem, profiles/r04/q18_chain.txt),
// more than the query it served.
int64_t ex_route_tiles(int64_t n);
constexpr int kExRows = 16, kExTile = 256 * kExRows;       // a thread's rows of a tile: row = tile * kExTile + j * 256 + tid
// (the keys of a thread's rows are loaded first, all of them in flight at once; what follows -- ballots, LDS -- waits for them once)
__device__ __forceinline__ void ex_load_keys(const ExRoute &R, int64_t tile, int tid, int64_t (&key)[kExRows], unsigned &present) {
    present = 0;
#pragma unroll
    for (int j = 0; j < kExRows; j++) {
        const int64_t i = tile * kExTile + (int64_t)j * 256 + tid;
        const bool ok = i < R.n && bit(R.vkey, i);
        key[j] = R.pcount > 0 ? ld(R.key, ok ? i : 0) : 0;
        if (ok) present |= 1u << j;
    }
}
__device__ __forceinline__ int ex_destination(const ExRoute &R, int64_t key, bool &ok, bool &outside) {
    outside = false;
    if (!ok || R.pcount <= 0) return 0;
    const int64_t b = (int64_t)((uint64_t)key - (uint64_t)R.pmin);
    if (b < 0 || b >= R.pcount) { ok = false; outside = true; return 0; }
    return R.owner ? R.owner[(int)(b >> R.shift)] : (int)(((unsigned __int128)(uint64_t)b * (uint64_t)R.world) / (uint64_t)R.pcount);
}
__global__ __launch_bounds__(256) void k_ex_count(ExRoute R, int64_t ntiles, int64_t *tilecnt, unsigned long long *counts /* world, then keys outside the pivots */) {
    __shared__ unsigned int cnt[kMaxExWorld + 1];
    for (int i = threadIdx.x; i <= kMaxExWorld; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t tile = blockIdx.x;
    int64_t key[kExRows];
    unsigned present;
    ex_load_keys(R, tile, threadIdx.x, key, present);
#pragma unroll
    for (int j = 0; j < kExRows; j++) {
        if (tile * kExTile + (int64_t)j * 256 >= R.n) break;          // block-uniform
        bool ok = (present >> j) & 1u, outside;
        const int d = ex_destination(R, key[j], ok, outside);
        const uint64_t bad = __ballot(outside);
        if (bad && lane == 0) atomicAdd(&cnt[kMaxExWorld], (unsigned)__popcll(bad));
        uint64_t todo = __ballot(ok);                                   // one LDS atomic per (wave, destination present in the wave)
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int dl = __shfl(d, leader, kWave);
            const uint64_t same = __ballot(ok && d == dl) & todo;
            if (lane == leader) atomicAdd(&cnt[dl], (unsigned)__popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d <= kMaxExWorld; d += blockDim.x) {
        const unsigned c = cnt[d];
        if (d < R.world) tilecnt[(int64_t)d * ntiles + tile] = c;
        // (the rows per destination are what k_ex_offsets' scan ends with: one atomic per block and destination on `world` words was 15 000
        // same-address atomics for 60 M rows -- a third of this kernel's time)
        if (c && d == kMaxExWorld) atomicAdd(&counts[R.world], (unsigned long long)c);
    }
}
__global__ __launch_bounds__(1024) void k_ex_offsets(int64_t *tilecnt, int64_t ntiles, unsigned long long *counts) {
    constexpr int K = 8;
    __shared__ int64_t wsum[1024 / kWave];
    __shared__ int64_t carry;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    int64_t *c = tilecnt + (int64_t)blockIdx.x * ntiles;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < ntiles; base += 1024 * K) {
        const int64_t i0 = base + (int64_t)tid * K;
        int64_t x[K], sum = 0;
#pragma unroll
        for (int k = 0; k < K; k++) { x[k] = i0 + k < ntiles ? c[i0 + k] : 0; sum += x[k]; }
        int64_t incl = sum;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int64_t y = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += y;
        }
        if (lane == kWave - 1) wsum[wave] = incl;
        __syncthreads();
        int64_t wprefix = 0;
        for (int w = 0; w < wave; w++) wprefix += wsum[w];
        int64_t run = carry + wprefix + incl - sum;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++) { if (i0 + k < ntiles) c[i0 + k] = run; run += x[k]; }
        if (tid == 1023) carry = run;
        __syncthreads();
    }
    if (tid == 0) counts[blockIdx.x] = (unsigned long long)carry;          // the rows this destination receives from here
}
// counts[world + 1 + d] = rows of the destinations before d: where d's rows begin in the send buffer
__global__ void k_ex_bases(unsigned long long *counts, int world) {
    if (threadIdx.x == 0) {
        unsigned long long before = 0;
        for (int d = 0; d < world; d++) { counts[world + 1 + d] = before; before += counts[d]; }
    }
}
hipError_t launch_ex_route(const ExRoute &R, int64_t *tileoff, int64_t *counts, hipStream_t s) {
    (void)hipGetLastError();
    if (R.n <= 0) return hipSuccess;
    if (R.world < 1 || R.world > kMaxExWorld) return hipErrorInvalidValue;
    const int64_t ntiles = ex_route_tiles(R.n);
    k_ex_count<<<(unsigned)ntiles, 256, 0, s>>>(R, ntiles, tileoff, (unsigned long long *)counts);
    hipError_t e = launch_status();
    if (e != hipSuccess) return e;
    k_ex_offsets<<<R.world, 1024, 0, s>>>(tileoff, ntiles, (unsigned long long *)counts);
    e = launch_status();
    if (e != hipSuccess) return e;
    k_ex_bases<<<1, 64, 0, s>>>((unsigned long long *)counts, R.world);
    return launch_status();
}
int64_t ex_route_tiles(int64_t n) { return (n + kExTile - 1) / kExTile; }
int ex_route_shift(int64_t pcount) { return ex_slice_shift(pcount); }

__global__ __launch_bounds__(256) void k_ex_pack_all(ExRoute R, ExCols C, const int64_t *tileoff, const int64_t *bases, int64_t ntiles, int64_t n_send, int64_t *out) {
    constexpr int NW = 256 / kWave, CELLS = kExRows * NW;
    __shared__ unsigned short cell[CELLS][kMaxExWorld];     // rows of destination d in wave-slice c of the tile; then: in the slices before c
    __shared__ int64_t base[kMaxExWorld];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int64_t tile = blockIdx.x;
    for (int i = tid; i < CELLS * R.world; i += 256) cell[i / R.world][i % R.world] = 0;
    __syncthreads();
    int dest[kExRows], rank[kExRows];
    unsigned okbits = 0;
    const uint64_t below = (1ull << lane) - 1;
    {
        int64_t key[kExRows];
        unsigned present;
        ex_load_keys(R, tile, tid, key, present);
#pragma unroll
        for (int j = 0; j < kExRows; j++) {
            bool ok = (present >> j) & 1u, outside;
            dest[j] = ex_destination(R, key[j], ok, outside);
            if (ok) okbits |= 1u << j;
        }
    }
#pragma unroll
    for (int j = 0; j < kExRows; j++) {
        const bool ok = (okbits >> j) & 1u;
        const int d = dest[j];
        rank[j] = 0;
        uint64_t todo = __ballot(ok);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int dl = __shfl(d, leader, kWave);
            const uint64_t same = __ballot(ok && d == dl) & todo;
            if (ok && d == dl) rank[j] = __popcll(same & below);
            if (lane == leader) cell[j * NW + wave][dl] = (unsigned short)__popcll(same);
            todo &= ~same;
        }
    }
    __syncthreads();
    for (int d = tid; d < R.world; d += 256) {
        unsigned pre = 0;
        for (int c = 0; c < CELLS; c++) { const unsigned t = cell[c][d]; cell[c][d] = (unsigned short)pre; pre += t; }
        base[d] = bases[d] + tileoff[(int64_t)d * ntiles + tile];
    }
    __syncthreads();
    int64_t where[kExRows];
#pragma unroll
    for (int j = 0; j < kExRows; j++) where[j] = base[dest[j]] + cell[j * NW + wave][dest[j]] + rank[j];
    for (int k = 0; k < C.ncol; k++) {
        const Src src = C.src[k];
        int64_t *o = out + (int64_t)(C.first + k) * n_send;
#pragma unroll
        for (int j = 0; j < kExRows; j++)
            if ((okbits >> j) & 1u) o[where[j]] = ld(src, tile * kExTile + (int64_t)j * 256 + tid);
    }
    if (C.mask_at >= 0) {
        int64_t *o = out + (int64_t)C.mask_at * n_send;
#pragma unroll
        for (int j = 0; j < kExRows; j++) {
            if (!((okbits >> j) & 1u)) continue;
            const int64_t i = tile * kExTile + (int64_t)j * 256 + tid;
            uint64_t m = 0;
            for (int v = 0; v < C.nvalid; v++) m |= (uint64_t)bit(C.valid[v], i) << v;
            o[where[j]] = (int64_t)m;
        }
    }
}
hipError_t launch_ex_pack_all(const ExRoute &R, const ExCols &C, const int64_t *tileoff, const int64_t *counts, int64_t n_send, int64_t *out, hipStream_t s) {
    (void)hipGetLastError();
    if (R.n <= 0 || n_send <= 0) return hipSuccess;
    if (R.world < 1 || R.world > kMaxExWorld || C.ncol < 0 || C.ncol > kExPackCols) return hipErrorInvalidValue;
    const int64_t ntiles = ex_route_tiles(R.n);
    k_ex_pack_all<<<(unsigned)ntiles, 256, 0, s>>>(R, C, tileoff, counts + R.world + 1, ntiles, n_send, out);
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
