// vdl_device.h -- device-side helpers shared by the kernel files: wave64 reductions, operand loads, the batched
// gather front end, grid sizing.  Everything here is inline / static: each .hip file is its own code object.
#pragma once
#include "vdl_kernels.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace vdl {


typedef long long ll2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef char i8x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;

// hipGetLastError() is sticky per thread: a failed call made earlier by anybody in this process
// (e.g. an advisory query) would be reported by the next launch check.  Launchers therefore clear
// the slot before launching and read it right after (launch_status).
static inline hipError_t launch_status() { return hipGetLastError(); }

// ------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------
enum { R_SUM = 0, R_MIN = 1, R_MAX = 2 };

__device__ __forceinline__ int64_t r_identity(int kind) {
    return kind == R_SUM ? 0 : kind == R_MIN ? INT64_MAX : INT64_MIN;
}
__device__ __forceinline__ int64_t r_combine(int kind, int64_t a, int64_t b) {
    if (kind == R_SUM) return (int64_t)((uint64_t)a + (uint64_t)b);
    if (kind == R_MIN) return a < b ? a : b;
    return a > b ? a : b;
}
__device__ __forceinline__ int64_t wave_reduce(int64_t x, int kind) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        int64_t y = __shfl_down(x, off, kWave);
        x = r_combine(kind, x, y);
    }
    return x;
}

__device__ __forceinline__ int64_t ld(const Src &s, int64_t i) {
    switch (s.kind) {                                     // wave-uniform
    case SRC_I64: return ((const int64_t *)s.p)[i];
    case SRC_I32: return ((const int32_t *)s.p)[i];
    case SRC_I16: return ((const int16_t *)s.p)[i];
    case SRC_I8: return ((const int8_t *)s.p)[i];
    default: return (int64_t)((uint64_t)s.from + (uint64_t)i * (uint64_t)s.step);
    }
}
__device__ __forceinline__ bool bit(const uint64_t *v, int64_t i) { return v ? ((v[i >> 6] >> (i & 63)) & 1ull) : true; }

// Element loads with the representation fixed at compile time, and a dispatcher that runs a body once with the
// wave-uniform kind turned into a constant.  Written with ld()/bit() and `&&`, a gather is a chain of exec-masked
// regions, each load waited for before the next test; with unconditional loads (masked lanes read slot 0) and bitwise
// tests the loads of several bitmap words are in flight together.
template <int K> __device__ __forceinline__ int64_t ldk(const Src &s, int64_t i) {
    if (K == SRC_I64) return ((const int64_t *)s.p)[i];
    if (K == SRC_I32) return ((const int32_t *)s.p)[i];
    if (K == SRC_I16) return ((const int16_t *)s.p)[i];
    if (K == SRC_I8) return ((const int8_t *)s.p)[i];
    return (int64_t)((uint64_t)s.from + (uint64_t)i * (uint64_t)s.step);
}
// the same for data that is read once (streams through a fold): non-temporal, so that it does not displace what the
// following statements reuse
template <int K> __device__ __forceinline__ int64_t ldk_stream(const Src &s, int64_t i) {
    if (K == SRC_I64) return __builtin_nontemporal_load((const int64_t *)s.p + i);
    if (K == SRC_I32) return __builtin_nontemporal_load((const int32_t *)s.p + i);
    if (K == SRC_I16) return __builtin_nontemporal_load((const int16_t *)s.p + i);
    if (K == SRC_I8) return __builtin_nontemporal_load((const int8_t *)s.p + i);
    return (int64_t)((uint64_t)s.from + (uint64_t)i * (uint64_t)s.step);
}
template <class F> __device__ __forceinline__ void by_kind(int kind, F f) {
    switch (kind) {
    case SRC_I64: f(std::integral_constant<int, SRC_I64>{}); break;
    case SRC_I32: f(std::integral_constant<int, SRC_I32>{}); break;
    case SRC_I16: f(std::integral_constant<int, SRC_I16>{}); break;
    case SRC_I8: f(std::integral_constant<int, SRC_I8>{}); break;
    default: f(std::integral_constant<int, SRC_RANGE>{}); break;
    }
}
template <class F> __device__ __forceinline__ void by_reduction(int rk, F f) {        // same for R_SUM / R_MIN / R_MAX
    if (rk == R_SUM) f(std::integral_constant<int, R_SUM>{});
    else if (rk == R_MIN) f(std::integral_constant<int, R_MIN>{});
    else f(std::integral_constant<int, R_MAX>{});
}
constexpr int kGatherUnroll = 4;      // bitmap words (64 positions each) a wave has in flight

// positions of U consecutive words -> clamped source slots pc[] and lane flags ok[] (position present, in range,
// source slot holds a value); `extra` = a second validity bitmap over the positions (may be null)
template <int KP, bool VS>
__device__ __forceinline__ void gather_slots(const Src &pos, const uint64_t *vpos, const uint64_t *extra, const uint64_t *vsrc, int64_t nsrc,
                                             int64_t n, int64_t nw, int64_t w0, int lane, int64_t (&pc)[kGatherUnroll], bool (&ok)[kGatherUnroll]) {
    constexpr int U = kGatherUnroll;
    int64_t p[U], w[U];
    uint64_t a[U];
    bool in[U];
#pragma unroll
    for (int u = 0; u < U; u++) {                                  // the position loads go out first ...
        w[u] = w0 + u < nw ? w0 + u : nw - 1;                      // wave-uniform; spare words repeat the last one and are not stored
        a[u] = w0 + u < nw ? ~0ull : 0ull;
        const int64_t i = (w[u] << 6) + lane;
        in[u] = i < n;
        p[u] = ldk<KP>(pos, in[u] ? i : 0);
    }
    if (vpos) {                                                    // ... then the bitmap words, U loads under one branch
        uint64_t t[U];
#pragma unroll
        for (int u = 0; u < U; u++) t[u] = vpos[w[u]];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] &= t[u];
    }
    if (extra) {
        uint64_t t[U];
#pragma unroll
        for (int u = 0; u < U; u++) t[u] = extra[w[u]];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] &= t[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) ok[u] = in[u] & (((a[u] >> lane) & 1ull) != 0);
#pragma unroll
    for (int u = 0; u < U; u++) {
        ok[u] = ok[u] & (p[u] >= 0) & (p[u] < nsrc);
        pc[u] = ok[u] ? p[u] : 0;
    }
    if (VS) {
        uint64_t vw[U];
#pragma unroll
        for (int u = 0; u < U; u++) vw[u] = vsrc[pc[u] >> 6];
#pragma unroll
        for (int u = 0; u < U; u++) ok[u] = ok[u] & (((vw[u] >> (pc[u] & 63)) & 1ull) != 0);
    }
}
__device__ __forceinline__ int64_t wave_index() {               // in an SGPR: the bitmap words of a wave are scalar loads
    return (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave));
}

static inline int grid_for(int64_t n, int block, int per_thread) {
    int64_t g = (n + (int64_t)block * per_thread - 1) / ((int64_t)block * per_thread);
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace vdl
