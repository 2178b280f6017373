// Shared device/host helpers for libsaunet_hip.so (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include "../../include/saunet_hip.h"

namespace saunet {

typedef __hip_bfloat16 bf16;
typedef unsigned short u16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

int set_error(int code, const char* fmt, ...);
// thread-local log of the kernels the current thread's API calls launched (name = kernel symbol without "_kernel"); saunet_launch_log()
// returns and clears it -- how bench.py attributes the HIP-event time of a call to a kernel family without guessing the dispatch
void note_launch(const char* name);

#define SAUNET_CHECK_LAUNCH(name)                                                     \
    do {                                                                              \
        saunet::note_launch(name);                                                    \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess)                                                        \
            return saunet::set_error(SAUNET_LAUNCH_FAILED, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// ---- bf16 <-> f32 bit tricks (round-to-nearest-even on the way down) -----------------------------
__device__ __forceinline__ float bf16_lo(unsigned int packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int packed) { return __uint_as_float(packed & 0xffff0000u); }
__device__ __forceinline__ unsigned int f32_to_bf16_bits(float f)
{
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// gfx950 has a hardware RNE convert (v_cvt_pk_bf16_f32): let the compiler pick it through the native type
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi)
{
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}

struct f32s { float v; };                     // float32 storage whose matrix products run as 3 x bf16 splits (see mma_f32_chunk_split below)
constexpr long SAUNET_F32_SPLIT_MINPIX = 16384;
inline bool f32_split_wanted(long pixels) { return pixels >= SAUNET_F32_SPLIT_MINPIX; }
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int dtype = SAUNET_F32;
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Elem<f32s> {
    static constexpr int dtype = SAUNET_F32;
    __device__ static __forceinline__ float load(const f32s* p) { return p->v; }
    __device__ static __forceinline__ void store(f32s* p, float v) { p->v = v; }
};
template <> struct Elem<u16> {  // bf16 stored as raw 16-bit words
    static constexpr int dtype = SAUNET_BF16;
    __device__ static __forceinline__ float load(const u16* p) { return __uint_as_float(((unsigned int)*p) << 16); }
    __device__ static __forceinline__ void store(u16* p, float v) { *p = __builtin_bit_cast(u16, (__bf16)v); }
};

// ---- float32 operands on the bf16 matrix cores -------------------------------------------------------------------------------------
// gfx950 has no reduced-precision fast path for f32 inputs: mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate, 1/16 of the bf16 MFMA.
// A float is the sum of three bf16 pieces to 2^-27 (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 9 signed bits each), and the products
// of bf16 pieces are exact in the float32 accumulator.  With the six products
// hh, hm, mh, mm, hl, lh (what is dropped is below 2^-24 of |a||b|: float32 round-off class) one K = 8 step of the float path -- a lane
// holds 4 floats of each operand -- is THREE mfma_f32_32x32x16_bf16 instead of four mfma_f32_32x32x2_f32: the 16 K slots of a bf16 MFMA
// carry two product kinds at once (slots 0-3 / 4-7 of each lane half).  96 instead of 256 matrix-pipe cycles; the splits are VALU work
// that runs beside the MFMAs and is shared by all tiles a fragment feeds.
// The split result has the same rms error against float64 as the exact MFMA (scripts/f32_accuracy.py: 1.0e-6 vs 1.2e-6 of the output scale at
// K = 4608) but a small same-signed bias (-2e-7: the bf16 MFMA's adder does not round its addends to nearest), and on launches with a few
// hundred pixels per channel -- where BatchNorm amplifies float32 noise a thousand-fold -- that cost parity margin (smoke() at 64 x 64: gradient
// error 3.4e-4 -> 8.4e-4 of scale; the two-rank SGD test's second loss 0.24 % off).  Those launches are latency-bound anyway, so the kernels
// are instantiated for BOTH element tags: `float` = exact f32 MFMA, `f32s` = split, and the dispatchers take `f32s` from
// SAUNET_F32_SPLIT_MINPIX pixels per launch upwards (f32_split_wanted).
struct F32Split { unsigned h01, h23, m01, m23, l01, l23; };
// round-to-nearest splits (v_cvt_pk_bf16_f32): x = h + m + l + e with |e| <= 2^-27 |x| and NO bias -- truncation splits (and / perm) cost the
// same number of VALU operations but drop a same-signed remainder from every product, which a K = 10^4 reduction turns into a relative error
// of 1e-5 (measured: smoke()'s gradient error against the oracle doubled to 6e-4 of the gradient scale)
__device__ __forceinline__ F32Split f32_split3(const u32x4& v)
{
    F32Split s;
    float x[4], r[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = __uint_as_float(v[i]);
    s.h01 = pack_bf16x2(x[0], x[1]); s.h23 = pack_bf16x2(x[2], x[3]);
    r[0] = x[0] - bf16_lo(s.h01); r[1] = x[1] - bf16_hi(s.h01); r[2] = x[2] - bf16_lo(s.h23); r[3] = x[3] - bf16_hi(s.h23);
    // Non-finite operands: an Inf element (or |x| above the largest bf16, 3.39e38) has h = Inf and r = x - h = NaN, so every output it touches is
    // NaN here where the exact f32 MFMA of the small launches gives +-Inf.  Zeroing r for a non-finite h costs four compare + select per split:
    // measured +6 % on the whole float32 step (87.8 -> 93.2 ms) -- not paid for an input that is already outside the network's domain; the
    // behaviour is pinned by tests/test_hip_ops.py::test_f32_split_nonfinite_operand_poisons_only_its_outputs.
    s.m01 = pack_bf16x2(r[0], r[1]); s.m23 = pack_bf16x2(r[2], r[3]);
    q[0] = r[0] - bf16_lo(s.m01); q[1] = r[1] - bf16_hi(s.m01); q[2] = r[2] - bf16_lo(s.m23); q[3] = r[3] - bf16_hi(s.m23);
    s.l01 = pack_bf16x2(q[0], q[1]); s.l23 = pack_bf16x2(q[2], q[3]);
    return s;
}
// c += A * B for one 16-byte chunk pair of float32 fragments (K = 8: 4 floats per lane and half)
__device__ __forceinline__ void mma_f32_chunk_split(const u32x4& a, const u32x4& b, f32x16& c)
{
    const F32Split A = f32_split3(a), B = f32_split3(b);
    const u32x4 a3 = {A.h01, A.h23, A.l01, A.l23}, b3 = {B.l01, B.l23, B.h01, B.h23};     // h*l + l*h   (smallest terms first)
    const u32x4 a2 = {A.m01, A.m23, A.m01, A.m23}, b12 = {B.h01, B.h23, B.m01, B.m23};    // m*h + m*m
    const u32x4 a1 = {A.h01, A.h23, A.h01, A.h23};                                         // h*h + h*m
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a3), __builtin_bit_cast(bf16x8_t, b3), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a2), __builtin_bit_cast(bf16x8_t, b12), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a1), __builtin_bit_cast(bf16x8_t, b12), c, 0, 0, 0);
}
// the same with both operands split beforehand (a fragment that feeds several products is split once)
__device__ __forceinline__ void mma_f32_split_pre(const F32Split& A, const F32Split& B, f32x16& c)
{
    const u32x4 a3 = {A.h01, A.h23, A.l01, A.l23}, b3 = {B.l01, B.l23, B.h01, B.h23};
    const u32x4 a2 = {A.m01, A.m23, A.m01, A.m23}, b12 = {B.h01, B.h23, B.m01, B.m23};
    const u32x4 a1 = {A.h01, A.h23, A.h01, A.h23};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a3), __builtin_bit_cast(bf16x8_t, b3), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a2), __builtin_bit_cast(bf16x8_t, b12), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a1), __builtin_bit_cast(bf16x8_t, b12), c, 0, 0, 0);
}
__device__ __forceinline__ void mma_f32_chunk_exact(const u32x4& a, const u32x4& b, f32x16& c)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
}

// 16-byte vector of T as floats: 4 f32 or 8 bf16
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f)
    {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(f[i]);
        return v;
    }
};
template <> struct Vec16<f32s> : Vec16<float> {};
template <> struct Vec16<u16> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_lo(v[i]); f[2 * i + 1] = bf16_hi(v[i]); }
    }
    __device__ static __forceinline__ u32x4 pack(const float* f)
    {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
        return v;
    }
};

// ---- wave64 reductions ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// fast unsigned divide by a runtime constant (host builds the magic): q = (n * mul) >> 32 >> shr
struct FastDiv {
    unsigned int mul, shr, d;
    __host__ static FastDiv make(unsigned int d)
    {
        FastDiv f; f.d = d;
        if (d == 1) { f.mul = 0; f.shr = 0; return f; }
        unsigned int l = 0; while ((1ull << l) < d) ++l;            // ceil(log2 d)
        unsigned long long m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
        f.mul = (unsigned int)m; f.shr = l; return f;
    }
    __device__ __forceinline__ unsigned int div(unsigned int n) const
    {
        if (d == 1) return n;
        unsigned int t = __umulhi(n, mul);
        return (t + ((n - t) >> 1)) >> (shr - 1);
    }
};

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// kernel symbol the way rocprofv3 prints it -- "name_kernel<unsigned short, 64, 64, 32, 32, 8, false>" -- for the launch log
// (saunet_launch_log): built once per templated launch site (function-local static)
template <typename T> inline const char* type_name();
template <> inline const char* type_name<float>() { return "float"; }
template <> inline const char* type_name<u16>() { return "unsigned short"; }
template <> inline const char* type_name<f32s>() { return "saunet::f32s"; }
inline void kn_put(char* o, size_t cap, int v) { snprintf(o + strlen(o), cap - strlen(o), "%d", v); }
inline void kn_put(char* o, size_t cap, bool v) { snprintf(o + strlen(o), cap - strlen(o), "%s", v ? "true" : "false"); }
inline void kn_put(char* o, size_t cap, const char* v) { snprintf(o + strlen(o), cap - strlen(o), "%s", v); }
struct KName {
    char s[192];
    template <typename... A> KName(const char* base, A... a)
    {
        snprintf(s, sizeof(s), "%s<", base);
        int i = 0;
        ((i++ ? (void)kn_put(s, sizeof(s), ", ") : (void)0, kn_put(s, sizeof(s), a)), ...);
        kn_put(s, sizeof(s), ">");
    }
};

// ---- A/B switches.  The PRODUCT build has none: the library's behaviour is a function of its arguments only (include/saunet_hip.h:
// "no global mutable state").  A variant build (scripts/build_variant.sh ... -DSAUNET_AB_SWITCHES) reads the named environment variables
// once per process so that two kernel selections can be compared on one box.
#ifdef SAUNET_AB_SWITCHES
#include <stdlib.h>
inline bool ab_env_on(const char* name) { const char* v = getenv(name); return !(v && v[0] == '0'); }      // default on, "0" = off
inline int ab_env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
inline bool ab_env_on(const char*) { return true; }
inline int ab_env_int(const char*, int dflt) { return dflt; }
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: first() is true once per (call site, device).  A lost race sets the
// attribute twice, which is harmless.
struct DeviceOnce {
    unsigned long long done = 0;
    bool first()
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (__atomic_load_n(&done, __ATOMIC_RELAXED) & bit) return false;
        __atomic_fetch_or(&done, bit, __ATOMIC_RELAXED);
        return true;
    }
};
// the same for launch sites whose requirement grows with the problem: true when `lds` exceeds what this device was last given
struct DeviceMaxLds {
    int seen[64] = {0};
    bool raise(int lds)
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        int& s = seen[dev & 63];
        if (lds <= s) return false;
        s = lds;
        return true;
    }
};

// sum over the replicated accumulators (saunet_bn_epilogue.sums_replicas): 8 loads in flight at a time -- these are pure latency chains
// (one thread per channel), a one-load-per-iteration loop costs 16 memory round trips
__device__ __forceinline__ double rep_sum(const double* __restrict__ s, int reps, int rstride, int i)
{
    double v = 0.0;
    int r = 0;
    for (; r + 8 <= reps; r += 8) {
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = s[(size_t)(r + j) * rstride + i];
        v += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    for (; r < reps; ++r) v += s[(size_t)r * rstride + i];
    return v;
}

// the same for the (sum, sum of squares) pair of one channel: both accumulators' replicas travel together (two round trips, not four)
__device__ __forceinline__ void rep_sum2(const double* __restrict__ a, const double* __restrict__ b, int reps, int rstride, int i, double& sa, double& sb)
{
    double va = 0.0, vb = 0.0;
    int r = 0;
    // sixteen replicas (the usual count) in ONE round trip: these chains sit in the prologue of every small-map kernel, where a second
    // dependent batch is a microsecond of a 10 us launch; the summation order (pairs, then groups of eight in replica order) is unchanged
    for (; r + 16 <= reps; r += 16) {
        double t[16], u[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { t[j] = a[(size_t)(r + j) * rstride + i]; u[j] = b[(size_t)(r + j) * rstride + i]; }
        va += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        vb += ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
        va += ((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15]));
        vb += ((u[8] + u[9]) + (u[10] + u[11])) + ((u[12] + u[13]) + (u[14] + u[15]));
    }
    for (; r + 8 <= reps; r += 8) {
        double t[8], u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { t[j] = a[(size_t)(r + j) * rstride + i]; u[j] = b[(size_t)(r + j) * rstride + i]; }
        va += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        vb += ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
    }
    for (; r < reps; ++r) { va += a[(size_t)r * rstride + i]; vb += b[(size_t)r * rstride + i]; }
    sa = va; sb = vb;
}

template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(moved);
}
// sum over the 32 lanes of each wave half; valid in lanes 16..31 and 48..63
__device__ __forceinline__ float half_wave_sum(float v)
{
    v = dpp_add<0xB1>(v);          // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);          // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);         // row_half_mirror
    v = dpp_add<0x140>(v);         // row_mirror  -> every lane of a 16-lane row holds the row sum
    v = dpp_add<0x142>(v);         // row_bcast:15: rows 1 and 3 += the sum of the row before (bound_ctrl: row 0 adds 0; row 2 is not used)
    return v;
}

// keep + (send of the partner lane under the DPP permutation CTRL)
template <int CTRL> __device__ __forceinline__ float dpp_exchange_add(float keep, float send)
{
    return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL, 0xf, 0xf, true));
}
// TRANSPOSING reduction of the two BN-backward sums of 8 channels over the 16 lanes (pixels) of a DPP row: every stage pairs two values, a lane
// keeps one of the pair and receives the partner lane's copy of the same one, so the 16 inputs shrink 8 -> 4 -> 2 -> 1 while the lanes
// spread over the outputs: 15 DPP adds (+ 30 selects) instead of the 80 DPP adds of sixteen separate 5-step reductions -- a DPP add issues
// every 8 cycles per wave, an LDS float atomic costs ~12 cycles per active lane (scripts/probes/valu_probe.hip).  Stage order = row_mirror,
// row_half_mirror, quad_perm xor 1, quad_perm xor 2, so that the lanes a later stage pairs made the same choices in all earlier stages.
// Result: lane l of the row holds the 16-lane partial sum of  (l & 8 ? e2 : e1)[4 * ((l >> 1) & 1) + 2 * (l & 1) + ((l >> 2) & 1)].
__device__ __forceinline__ float row_transpose_sum(const float (&e1)[8], const float (&e2)[8], bool s0, bool s1, bool s2, bool s3)
{
    float w0[8], w1[4], w2[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) w0[j] = dpp_exchange_add<0x140>(s0 ? e2[j] : e1[j], s0 ? e1[j] : e2[j]);                 // row_mirror
#pragma unroll
    for (int k = 0; k < 4; ++k) w1[k] = dpp_exchange_add<0x141>(s1 ? w0[2 * k + 1] : w0[2 * k], s1 ? w0[2 * k] : w0[2 * k + 1]);   // row_half_mirror
#pragma unroll
    for (int k = 0; k < 2; ++k) w2[k] = dpp_exchange_add<0xB1>(s2 ? w1[2 * k + 1] : w1[2 * k], s2 ? w1[2 * k] : w1[2 * k + 1]);    // quad_perm [1,0,3,2]
    return dpp_exchange_add<0x4E>(s3 ? w2[1] : w2[0], s3 ? w2[0] : w2[1]);                                                // quad_perm [2,3,0,1]
}

// Consumer-side BatchNorm finalize (saunet_bn_prologue): fills s_pro[0 .. cpad) = scale and s_pro[cpad .. 2*cpad) = shift of the input
// channels (zero beyond Cin) with the arithmetic of bn_finalize_kernel; `writer` (one workgroup of the launch) also publishes the xhat rows
// of the channels finalised here, the [4][Cin] parameter block the backward pass reads and the running statistics.
// cbias: bias of the convolution that produced the tensor (its statistics were taken before the bias was added; bn_finalize_kernel's `cbias`).
template <int NT> __device__ __forceinline__ void bn_prologue_fill(const saunet_bn_prologue& p, int Cin, int cpad, float* s_pro, bool writer,
                                                                   const float* __restrict__ cbias = nullptr)
{
    for (int c = threadIdx.x; c < cpad; c += NT) {
        float sc = 0.f, sh = 0.f;
        if (c < Cin) {
            float mean, is, var;
            if (c >= p.c_lo) {
                double s1, s2;
                rep_sum2(p.sum, p.sumsq, p.replicas, p.rstride, c, s1, s2);
                double m = s1 / p.count;
                double v = s2 / p.count - m * m;
                if (v < 0.0) v = 0.0;
                if (cbias) m += (double)cbias[c];
                mean = (float)m; is = (float)(1.0 / sqrt(v + (double)p.eps)); var = (float)v;
                if (writer && p.xhat) {
                    p.xhat[c] = is; p.xhat[p.ld_xhat + c] = -mean * is; p.xhat[2 * p.ld_xhat + c] = mean; p.xhat[3 * p.ld_xhat + c] = is;
                    p.xhat[4 * p.ld_xhat + c] = var;
                }
            } else { mean = p.xhat[2 * p.ld_xhat + c]; is = p.xhat[3 * p.ld_xhat + c]; var = p.xhat[4 * p.ld_xhat + c]; }
            sc = p.gamma[c] * is; sh = p.beta[c] - mean * sc;
            if (writer) {
                p.params[c] = sc; p.params[Cin + c] = sh; p.params[2 * Cin + c] = mean; p.params[3 * Cin + c] = is;
                if (p.running_mean) {
                    const double unb = p.count > 1.0 ? (double)var * p.count / (p.count - 1.0) : (double)var;
                    p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * mean;
                    p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)unb;
                }
            }
        }
        s_pro[c] = sc; s_pro[cpad + c] = sh;
    }
}

// ---- LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes land lane-linear at the LDS address in M0) and the counted waits that go with it.
// Inline asm: hipcc neither counts these requests nor waits for them -- the kernels do, with vmcnt(N) + s_barrier (cdna guide 5.7).
__device__ __forceinline__ void mm_dma16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void mm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }
__device__ __forceinline__ void mm_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


// ---- phase timing (profiling builds only: python -m saunet_amd._build --timing -> scripts/_ab/libsaunet_timing.so) -------------------
// TSTAMP(slot) records (slot, s_memtime) from thread 0 of ONE block into a per-translation-unit device array that
// saunet_debug_timing_<unit>() copies out; scripts/phase_timing.py prints the per-phase cycle deltas.  This is how the serialised
// waits inside a kernel are found when the PMC counters only say "waiting".  In the product build every macro expands to nothing.
#ifdef SAUNET_TIMING
static __device__ unsigned long long g_timing[2048];
#define TSTAMP_INIT() int tcount__ = 0
#ifndef SAUNET_TIMING_TID
#define SAUNET_TIMING_TID 0
#endif
#define TSTAMP(slot) do { if (blockIdx.x == SAUNET_TIMING_BLOCK && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == SAUNET_TIMING_TID && tcount__ < 2000) \
        g_timing[tcount__++] = ((unsigned long long)(slot) << 56) | (__builtin_readcyclecounter() & 0x00ffffffffffffffull); } while (0)
#define SAUNET_TIMING_READER(unit) extern "C" int saunet_debug_timing_##unit(unsigned long long* out, int n) \
    { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(saunet::g_timing), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost); }
#ifndef SAUNET_TIMING_BLOCK
#define SAUNET_TIMING_BLOCK 7
#endif
#else
#define TSTAMP_INIT() do {} while (0)
#define TSTAMP(slot) do {} while (0)
#define SAUNET_TIMING_READER(unit)
#endif

}  // namespace saunet
