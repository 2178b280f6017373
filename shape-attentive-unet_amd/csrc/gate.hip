// Fused GatedSpatialConv2d for the shape stream (reference: /root/reference/models/GSConv.py:16-57, call sites
// /root/reference/models/models.py:341-352).  With C = 32/16/8 feature channels plus ONE gating channel the module is
//
//     cat   = [feat, gate]                                   (C+1 channels, full resolution)
//     a0    = BatchNorm(cat)                                 _gate_conv[0]
//     h     = relu(W1 a0 + b1)                               _gate_conv[1..2]   (C+1 -> C+1)
//     z     = W2 h + b2                                      _gate_conv[3]      (C+1 -> 1)
//     alpha = sigmoid(BatchNorm(z))                          _gate_conv[4..5]
//     y     = Wm (feat * (alpha + 1))                        the module's own 1x1 weight
//
// Unfused this is ~13 full-resolution passes over odd-width (33/17/9 channel) tensors.  Here the chain is recomputed per
// pixel in registers and only feat / gate / z / y touch HBM.  The two batch-norm statistics are the only global
// dependencies, so forward = 2 passes (z, then y) and backward = 3 passes that RECOMPUTE the chain instead of storing it:
//     pass 1: q = dL/d(bn1 out) per pixel,  sums for BN1 backward,  dWm
//     pass 2: all remaining parameter gradients and the two BN0 backward sums
//     pass 3: dfeat, dgate
//
// Round 3: the per-pixel (C+1)x(C+1) matrix-vector products run on the MATRIX CORES (rounds 1-2: one thread per pixel, 2 x 1089
// scalar-weight FMAs per pixel and pass -- VALU / scalar-load bound at 0.45 TB/s).  A wave owns 32 pixels per tile:
//     B operand = the pixel's NHWC row as it lies in memory: lane (pixel = lane % 32, half = lane / 32) loads the 8 consecutive
//                 channels  16 ks + 8 half .. + 7  of its pixel with one 16-byte load per k-step -- no transposition;
//     A operand = the weights, built once per wave (BatchNorm-0 folded in: W1' = W1 diag(s0), b1' = b1 + W1 t0), rows PERMUTED so
//                 that the accumulator of output row-tile mt, register r, lane-half h holds position
//                        k = 32 mt + 16 (r / 8) + 8 h + r % 8            ("K-layout"),
//                 i.e. the 16 accumulator registers of a lane are, 8 by 8, exactly the B fragments of the NEXT product (h -> W1^T dh)
//                 and line up with the channels the lane loaded (dot products with feat, 16-byte stores of y / dfeat);
//     positions: k < C feature channel k, k = C the gating channel, k = C+1 a constant one (carries the folded bias).
// float32 weights enter as bf16 hi + lo pairs (two MFMAs), activations are bf16 in memory already: the products are exact to
// ~2^-17, the same arithmetic as the scalar version up to summation order.
// Cross-pixel sums  sum_p a_p[i] * b_p[j]  (weight gradients, bias gradients as products with the ones row, BN sums) also run on the
// matrix cores: every wave transposes its 64 pixels through LDS into [position][pixel] bf16 tiles and issues mfma_f32_32x32x16_bf16
// with K = pixels; per-block partial tiles go to a workspace and a small reduce + finalize pair turns them into gradients and the
// per-channel coefficients of the next pass.  The BN0 backward sums are not accumulated per pixel at all:
// sum_p da0_j = sum_u W1[u][j] sum_p dh_u and sum_p da0_j cat_j = sum_u W1[u][j] sum_p dh_u cat_j follow from the dW1 products.
// bf16 storage only.
#include "common.h"
#include "mma_tiles.h"

namespace saunet {

constexpr int Q_WS = 1056;              // floats per block of pass 1: 32x32 dWm tile + {sum q, sum q*zhat} (+ pad)
constexpr int S_VEC = 4 * 1024;         // pass 2: four 32x32 product tiles, then sum_p dz*h_u at [S_VEC + u], sum_p dz at [S_VEC + 64]
constexpr int S_WS = S_VEC + 128;

template <int C> struct GL {
    static constexpr int C1 = C + 1;
    static constexpr int NF = C >= 16 ? C / 16 : 1;          // k-steps holding feature channels
    static constexpr int NK = C >= 16 ? C / 16 + 1 : 1;      // k-steps covering [feat, gate, one]
    static constexpr int MT = C == 32 ? 2 : 1;               // 32-row tiles covering the C+1 output rows
    static constexpr int RU = C == 32 ? 16 : (C == 16 ? 9 : 8);   // accumulator registers of row-tile 0 that can hold a position <= C
    // where position C (the gating channel) sits in the accumulators
    static constexpr int G_MT = C / 32, G_REG = 8 * ((C % 32) / 16), G_HALF = (C % 16) / 8;
};
// MFMA output row m of row-tile mt holds position kpos_row(mt, m); equivalently register r of lane-half h holds kpos_reg(mt, r, h)
__device__ __forceinline__ int kpos_row(int mt, int m) { const int i = m >> 3, h = (m >> 2) & 1, t = m & 3; return 32 * mt + 16 * (i >> 1) + 8 * h + 4 * (i & 1) + t; }
__device__ __forceinline__ int kpos_reg(int mt, int r, int half) { return 32 * mt + 16 * (r >> 3) + 8 * half + (r & 7); }

// A-operand fragment of row `rowk` (a position), k-step ks: f(rowk, k) for the 8 positions this lane holds, as bf16 hi + lo
template <class F> __device__ __forceinline__ void make_frag(F f, int rowk, int ks, int lane, bf16x8_t& hi, bf16x8_t& lo)
{
    const int k0 = 16 * ks + 8 * (lane >> 5);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = f(rowk, k0 + e);
        const __bf16 h = (__bf16)v;
        hi[e] = h; lo[e] = (__bf16)(v - (float)h);
    }
}

__device__ __forceinline__ bf16x8_t as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8_t, v); }
__device__ __forceinline__ u32x4 as_words(const bf16x8_t& v) { return __builtin_bit_cast(u32x4, v); }
__device__ __forceinline__ u16 frag_elem(const bf16x8_t& v, int e) { const u32x4 w = as_words(v); return (u16)(e & 1 ? w[e >> 1] >> 16 : w[e >> 1] & 0xffffu); }

// B fragments of a pixel's feature row (NF k-steps; C = 8: the upper lane half holds zeros)
template <int C> __device__ __forceinline__ void load_feat(const u16* __restrict__ base, int ld, size_t pp, int half, bf16x8_t* xf)
{
    if constexpr (C >= 16) {
#pragma unroll
        for (int ks = 0; ks < GL<C>::NF; ++ks) xf[ks] = as_frag(*(const u32x4*)(base + pp * ld + 16 * ks + 8 * half));
    } else {
        u32x4 v = *(const u32x4*)(base + pp * ld);
        if (half) v = u32x4{0u, 0u, 0u, 0u};
        xf[0] = as_frag(v);
    }
}
// B fragments of the extended row [feat, gate, 1] (NK k-steps)
template <int C> __device__ __forceinline__ void load_xt(const u16* __restrict__ feat, int ldf, const u16* __restrict__ gate, int ldg, size_t pp, int half, bf16x8_t* xb)
{
    const unsigned int g1 = (unsigned int)gate[pp * ldg] | 0x3F800000u;      // {gate, 1.0} as two bf16
    if constexpr (C >= 16) {
#pragma unroll
        for (int ks = 0; ks < GL<C>::NF; ++ks) xb[ks] = as_frag(*(const u32x4*)(feat + pp * ldf + 16 * ks + 8 * half));
        xb[GL<C>::NF] = as_frag(u32x4{half ? 0u : g1, 0u, 0u, 0u});
    } else {
        u32x4 v = *(const u32x4*)(feat + pp * ldf);
        if (half) v = u32x4{g1, 0u, 0u, 0u};
        xb[0] = as_frag(v);
    }
}
__device__ __forceinline__ void unpack_frag(const bf16x8_t& v, float* f) { Vec16<u16>::unpack(as_words(v), f); }

// hpre = W1 (s0*cat + t0) + b1 for 32 pixels:  rows = hidden units (K-layout), k = [feat, gate, 1]
template <int C> struct HiddenW {
    bf16x8_t hi[GL<C>::MT][GL<C>::NK], lo[GL<C>::MT][GL<C>::NK];
    __device__ __forceinline__ void build(const float* __restrict__ bn0, const float* __restrict__ w1, const float* __restrict__ b1, int lane)
    {
        constexpr int C1 = C + 1;
#pragma unroll
        for (int mt = 0; mt < GL<C>::MT; ++mt) {
            const int u = kpos_row(mt, lane & 31);
            float bias = 0.f;
            if (u < C1) {
                bias = b1[u];
                for (int j = 0; j < C1; ++j) bias = fmaf(w1[u * C1 + j], bn0[C1 + j], bias);
            }
#pragma unroll
            for (int ks = 0; ks < GL<C>::NK; ++ks)
                make_frag([&](int uu, int k) { return uu >= C1 ? 0.f : (k <= C ? w1[uu * C1 + k] * bn0[k] : (k == C + 1 ? bias : 0.f)); }, u, ks, lane,
                          hi[mt][ks], lo[mt][ks]);
        }
    }
    __device__ __forceinline__ void apply(const bf16x8_t* xb, f32x16* acc) const
    {
#pragma unroll
        for (int mt = 0; mt < GL<C>::MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < GL<C>::NK; ++ks) {
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi[mt][ks], xb[ks], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo[mt][ks], xb[ks], acc[mt], 0, 0, 0);
            }
        }
    }
};
// per-position constants in the K-layout of this lane: v[k] for k <= C, zero elsewhere (row-tile 0: 16 registers; row-tile 1 of C = 32: one)
template <int C> __device__ __forceinline__ void load_kvec(const float* __restrict__ v, int half, float* k0, float& k1)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int k = kpos_reg(0, r, half); k0[r] = k <= C ? v[k] : 0.f; }
    k1 = 0.f;
    if constexpr (C == 32) { if (half == 0) k1 = v[32]; }
}

// block-cooperative copy of a small parameter array into LDS (coalesced; the per-lane fragment builders then read LDS, not L2)
__device__ __forceinline__ const float* stage(float*& cursor, const float* __restrict__ src, int n)
{
    float* dst = cursor;
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
    cursor += (n + 3) & ~3;
    return dst;
}

// block 0 of a producing kernel clears the reduction target (no memset launch)
__device__ __forceinline__ void clear_reduction(float* __restrict__ gred)
{
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < S_WS; i += 256) gred[i] = 0.f;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }
// a wave's LDS tiles are private to it: ordering its own writes before its own (cross-lane) reads needs no workgroup barrier
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
// red[row][col] += acc (one wave at a time: plain read-modify-write, fixed order)
__device__ __forceinline__ void tile_add(float* red, const f32x16& acc, int lane)
{
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] += acc[r];
}
__device__ __forceinline__ float half_sum(float v)      // sum over the 32 lanes of a lane half
{
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Two-deep software pipeline over a wave's pixel pairs (64 pixels each): the loads of the next pair are in flight while the current one
// is computed.  Two statically named buffers (no register-set copies); loads past the end re-read the current pair and are never used.
// DEEP = false: plain loop.
template <class B, bool DEEP, class L, class F> __device__ __forceinline__ void pair_pipeline(unsigned first, unsigned npairs, unsigned nw, L load, F compute)
{
    if (first >= npairs) return;
    if constexpr (!DEEP) {          // register-heavy variants (C = 32 backward passes): one buffer, occupancy hides the latency
        for (unsigned pair = first; pair < npairs; pair += nw) { B a; load(a, pair); compute(a, pair); }
        return;
    }
    B a, b;
    load(a, first);
    for (unsigned pair = first; pair < npairs; pair += 2 * nw) {
        const unsigned p1 = pair + nw, p2 = pair + 2 * nw;
        load(b, p1 < npairs ? p1 : pair);
        compute(a, pair);
        load(a, p2 < npairs ? p2 : pair);
        if (p1 < npairs) compute(b, p1);
    }
}

// ------------------------------------------------------------------------------------------------ forward, pass 1: z
template <int C> __global__ __launch_bounds__(256, 3)
void gate_fwd_z_kernel(const u16* __restrict__ feat, int ldf, const u16* __restrict__ gate, int ldg, unsigned P,
                       const float* __restrict__ bn0, const float* __restrict__ w1, const float* __restrict__ b1,
                       const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ z,
                       double* __restrict__ zsum, double* __restrict__ zsq, int reps, int rstride)
{
    using G = GL<C>;
    __shared__ float s_red[4][2];
    __shared__ float s_par[(C + 1) * (C + 1) + 6 * (C + 1) + 16];
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31, wave = threadIdx.x >> 6;
    {
        float* cur = s_par;
        w1 = stage(cur, w1, (C + 1) * (C + 1)); b1 = stage(cur, b1, C + 1); bn0 = stage(cur, bn0, 2 * (C + 1)); w2 = stage(cur, w2, C + 1);
        __syncthreads();
    }
    HiddenW<C> W; W.build(bn0, w1, b1, lane);
    float w2k[16], w2x; load_kvec<C>(w2, half, w2k, w2x);
    const float bias2 = b2[0];
    float ls = 0.f, lq = 0.f;
    const unsigned npairs = (P + 63u) >> 6, nw = gridDim.x * 4u;
    struct Buf { bf16x8_t xb[2][G::NK]; };
    auto load = [&](Buf& b, unsigned pair) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned p = pair * 64u + 32u * s + col;
            load_xt<C>(feat, ldf, gate, ldg, p < P ? p : P - 1, half, b.xb[s]);
        }
    };
    auto compute = [&](Buf& b, unsigned pair) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned p = pair * 64u + 32u * s + col;
            f32x16 hp[G::MT]; W.apply(b.xb[s], hp);
            float zz = 0.f;
#pragma unroll
            for (int r = 0; r < G::RU; ++r) zz = fmaf(w2k[r], fmaxf(hp[0][r], 0.f), zz);
            if constexpr (G::MT == 2) zz = fmaf(w2x, fmaxf(hp[1][0], 0.f), zz);
            zz += __shfl_xor(zz, 32, 64);
            zz += bias2;
            if (p < P && half == 0) { z[p] = zz; ls += zz; lq = fmaf(zz, zz, lq); }
        }
    };
    pair_pipeline<Buf, true>(blockIdx.x * 4u + wave, npairs, nw, load, compute);
    if (zsum != nullptr) {
        ls = wave_sum(ls); lq = wave_sum(lq);
        if (lane == 0) { s_red[wave][0] = ls; s_red[wave][1] = lq; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const size_t ro = (size_t)(blockIdx.x % reps) * rstride;
            atomicAdd(&zsum[ro], (double)((s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0])));
            atomicAdd(&zsq[ro], (double)((s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1])));
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward, pass 2: y, alpha
// y = (alpha + 1) * (Wm feat): rows = output channels (K-layout), k = input channels
template <int C> __global__ __launch_bounds__(256)
void gate_fwd_out_kernel(const u16* __restrict__ feat, int ldf, const float* __restrict__ z, unsigned P, const float* __restrict__ bn1,
                         const float* __restrict__ wm, u16* __restrict__ y, int ldy, u16* __restrict__ alpha)
{
    using G = GL<C>;
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31, wave = threadIdx.x >> 6;
    const float s1 = bn1[0], t1 = bn1[1];
    __shared__ float s_par[C * C];
    { float* cur = s_par; wm = stage(cur, wm, C * C); __syncthreads(); }
    bf16x8_t ahi[G::NF], alo[G::NF];
#pragma unroll
    for (int ks = 0; ks < G::NF; ++ks)
        make_frag([&](int o, int j) { return (o < C && j < C) ? wm[o * C + j] : 0.f; }, kpos_row(0, col), ks, lane, ahi[ks], alo[ks]);
    const unsigned npairs = (P + 63u) >> 6, nw = gridDim.x * 4u;
    struct Buf { bf16x8_t xf[2][G::NF]; float zz[2]; };
    auto load = [&](Buf& b, unsigned pair) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned p = pair * 64u + 32u * s + col;
            const size_t pp = p < P ? p : P - 1;
            load_feat<C>(feat, ldf, pp, half, b.xf[s]);
            b.zz[s] = z[pp];
        }
    };
    auto compute = [&](Buf& b, unsigned pair) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned p = pair * 64u + 32u * s + col;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < G::NF; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[ks], b.xf[s][ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[ks], b.xf[s][ks], acc, 0, 0, 0);
            }
            const float al = sigmoidf_(fmaf(b.zz[s], s1, t1));
            if (p < P) {
                float o[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = acc[r] * (al + 1.f);
                u16* yr = y + (size_t)p * ldy + 8 * half;
                if (C >= 16 || half == 0) *(u32x4*)yr = Vec16<u16>::pack(o);
                if constexpr (C == 32) *(u32x4*)(yr + 16) = Vec16<u16>::pack(o + 8);
                if (half == 0) alpha[p] = to_bf16(al);
            }
        }
    };
    pair_pipeline<Buf, true>(blockIdx.x * 4u + wave, npairs, nw, load, compute);
}

// ------------------------------------------------------------------------------------------------ backward, pass 1
// q = dL/d(BN1 output) = (sum_c du_c * feat_c [+ dalpha_ext]) * alpha * (1 - alpha),  du = Wm^T dy
// workspace[block] = { dWm tile [32][32] (row = out channel i, col = in channel j) , sum q , sum q * zhat }
template <int C> __global__ __launch_bounds__(256)
void gate_bwd_q_kernel(const u16* __restrict__ dy, int lddy, const u16* __restrict__ feat, int ldf, const float* __restrict__ z,
                       const u16* __restrict__ dalpha_ext, unsigned P, const float* __restrict__ bn1, const float* __restrict__ wm,
                       float* __restrict__ q, float* __restrict__ ws, float* __restrict__ gred)
{
    using G = GL<C>;
    extern __shared__ u16 g_lds[];
    clear_reduction(gred);
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31, wave = threadIdx.x >> 6;
    u16* tA = g_lds + wave * 2 * G_TILE; u16* tB = tA + G_TILE;
    const float s1 = bn1[0], t1 = bn1[1], mu1 = bn1[2], is1 = bn1[3];
    { float* cur = (float*)g_lds; wm = stage(cur, wm, C * C); __syncthreads(); }       // staged in the (not yet used) tile area
    bf16x8_t ahi[G::NF], alo[G::NF];          // Wm^T: rows = input channels j (K-layout), k = output channels o
#pragma unroll
    for (int ks = 0; ks < G::NF; ++ks)
        make_frag([&](int j, int o) { return (j < C && o < C) ? wm[o * C + j] : 0.f; }, kpos_row(0, col), ks, lane, ahi[ks], alo[ks]);
    __syncthreads();                           // fragments built: the staging area becomes tile space
    f32x16 accw;
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[r] = 0.f;
    float sq = 0.f, sqz = 0.f;
    const unsigned npairs = (P + 63u) >> 6, nw = gridDim.x * 4u;
    struct Buf { bf16x8_t gb[2][G::NF], xf[2][G::NF]; float zz[2], dext[2]; };
    auto load = [&](Buf& b, unsigned pair) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned p = pair * 64u + 32u * s + col;
            const size_t pp = p < P ? p : P - 1;
            load_feat<C>(dy, lddy, pp, half, b.gb[s]);
            load_feat<C>(feat, ldf, pp, half, b.xf[s]);
            b.zz[s] = z[pp];
            b.dext[s] = dalpha_ext != nullptr ? Elem<u16>::load(dalpha_ext + pp) : 0.f;
        }
    };
    auto compute = [&](Buf& b, unsigned pair) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned p = pair * 64u + 32u * s + col;
            const bool live = p < P;
            bf16x8_t gb[G::NF];
#pragma unroll
            for (int ks = 0; ks < G::NF; ++ks) gb[ks] = live ? b.gb[s][ks] : as_frag(u32x4{0u, 0u, 0u, 0u});
            f32x16 du;
#pragma unroll
            for (int r = 0; r < 16; ++r) du[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < G::NF; ++ks) {
                du = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[ks], gb[ks], du, 0, 0, 0);
                du = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[ks], gb[ks], du, 0, 0, 0);
            }
            float f[8 * G::NF];
#pragma unroll
            for (int ks = 0; ks < G::NF; ++ks) unpack_frag(b.xf[s][ks], f + 8 * ks);
            float dal = 0.f;
#pragma unroll
            for (int r = 0; r < 8 * G::NF; ++r) dal = fmaf(du[r], f[r], dal);
            dal += __shfl_xor(dal, 32, 64);
            const float al = sigmoidf_(fmaf(b.zz[s], s1, t1));
            const float qq = live ? (dal + b.dext[s]) * al * (1.f - al) : 0.f;
            if (live && half == 0) q[p] = qq;
            if (half == 0) { sq += qq; sqz = fmaf(qq, (b.zz[s] - mu1) * is1, sqz); }
            const int px = 32 * s + col;
#pragma unroll
            for (int ks = 0; ks < G::NF; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ch = 16 * ks + 8 * half + e;
                    tA[ch * GP + px] = frag_elem(gb[ks], e);
                    tB[ch * GP + px] = to_bf16(f[8 * ks + e] * (al + 1.f));
                }
        }
        wave_lds_fence();
        tile_mma(tA, C, tB, C, lane, accw);
        wave_lds_fence();
    };
    pair_pipeline<Buf, C != 32>(blockIdx.x * 4u + wave, npairs, nw, load, compute);
    __syncthreads();
    float* red = (float*)g_lds;
    for (int i = threadIdx.x; i < Q_WS; i += 256) red[i] = 0.f;
    __syncthreads();
    sq = wave_sum(sq); sqz = wave_sum(sqz);
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
            tile_add(red, accw, lane);
            if (lane == 0) { red[1024] += sq; red[1025] += sqz; }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < Q_WS; i += 256) ws[(size_t)blockIdx.x * Q_WS + i] = red[i];
}

// red = reduced pass-1 workspace.  Outputs: dwm [C][C], dbn1 = {dgamma1, dbeta1}, K = {K0, K1, K2} with dz = K0*q + K1 + K2*z
struct GateQFinalize {
    int C; const float* bn1; float count; float* dwm; float* dbn1; float* K;
    __device__ void operator()(const float* red) const
    {
        for (int i = threadIdx.x; i < C * C; i += blockDim.x) dwm[i] = red[(i / C) * 32 + (i % C)];
        if (threadIdx.x == 0) {
            const float s1 = bn1[0], mu1 = bn1[2], is1 = bn1[3];
            const float sq = red[1024], sqz = red[1025];
            dbn1[0] = sqz; dbn1[1] = sq;
            const float mq = sq / count, mqz = sqz / count;
            const float K2 = -s1 * mqz * is1;
            K[0] = s1; K[1] = -s1 * mq - K2 * mu1; K[2] = K2;
        }
    }
};

// out[e] += sum_b ws[b][e]  (out zeroed by block 0 of the producing kernel; blockIdx.y strides the partial blocks)
__global__ __launch_bounds__(256) void gate_reduce_kernel(const float* __restrict__ ws, int nblocks, int stride, int n, float* __restrict__ out)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    for (int b = blockIdx.y; b < nblocks; b += gridDim.y) s += ws[(size_t)b * stride + e];
    atomicAdd(&out[e], s);
}
// one block: the finished sums -> LDS (one parallel sweep), then the pass's finalize.
// (Measured: folding this into the reduce kernel's last-arriving block costs 51 us instead of 8 + 7 -- the agent-scope fences of 544 blocks.)
template <class FIN> __global__ __launch_bounds__(256) void gate_finalize_kernel(const float* __restrict__ red, int n, FIN fin)
{
    __shared__ float s_fin[S_WS];
    for (int i = threadIdx.x; i < n; i += 256) s_fin[i] = red[i];
    __syncthreads();
    fin(s_fin);
}

// ------------------------------------------------------------------------------------------------ backward, pass 2
// Products over pixels  D[u][k] = sum_p dh_u * x~_k  (u: hidden unit, x~ = [feat, gate, 1]) as 32x32 tiles
//   C < 32:  tile 0 = D (17 x 18 or 9 x 10 used)
//   C = 32:  tile 0 = D[0:32][0:32],  tile 1 = D[0:32][32:34],  tile 2 = D[32][0:32] (row 0),  tile 3 = D[32][32:34] (row 0)
// plus sum_p dz * h_u at [S_VEC + u] and sum_p dz at [S_VEC + 64].
template <int C> __global__ __launch_bounds__(256, 2)
void gate_bwd_sums_kernel(const u16* __restrict__ feat, int ldf, const u16* __restrict__ gate, int ldg, const float* __restrict__ q,
                          const float* __restrict__ z, unsigned P, const float* __restrict__ K, const float* __restrict__ bn0,
                          const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, float* __restrict__ ws,
                          float* __restrict__ gred)
{
    using G = GL<C>;
    constexpr int C1 = C + 1;
    extern __shared__ u16 g_lds[];
    clear_reduction(gred);
    constexpr int WAVE_LDS = 2 * G_TILE + (C == 32 ? 2 * G_MISC : 0);
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31, wave = threadIdx.x >> 6;
    u16* DH = g_lds + wave * WAVE_LDS; u16* XT = DH + G_TILE; u16* MDH = XT + G_TILE; u16* MX = MDH + G_MISC;
    {       // parameters staged in the (not yet used) tile area
        float* cur = (float*)g_lds;
        w1 = stage(cur, w1, C1 * C1); b1 = stage(cur, b1, C1); bn0 = stage(cur, bn0, 2 * C1);
        __syncthreads();
    }
    HiddenW<C> W; W.build(bn0, w1, b1, lane);
    const float K0 = K[0], K1 = K[1], K2 = K[2];
    __syncthreads();
    f32x16 acc[C == 32 ? 4 : 1];
#pragma unroll
    for (int t = 0; t < (C == 32 ? 4 : 1); ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float dw2a[16], dw2x = 0.f, sdz = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) dw2a[r] = 0.f;
    const unsigned npairs = (P + 63u) >> 6, nw = gridDim.x * 4u;
    struct Buf { bf16x8_t xb[2][G::NK]; float qq[2], zz[2]; };
    auto load = [&](Buf& b, unsigned pair) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned p = pair * 64u + 32u * s + col;
            const size_t pp = p < P ? p : P - 1;
            load_xt<C>(feat, ldf, gate, ldg, pp, half, b.xb[s]);
            b.qq[s] = q[pp]; b.zz[s] = z[pp];
        }
    };
    auto compute = [&](Buf& b, unsigned pair) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned p = pair * 64u + 32u * s + col;
            f32x16 hp[G::MT]; W.apply(b.xb[s], hp);
            const float dz = p < P ? fmaf(K0, b.qq[s], fmaf(K2, b.zz[s], K1)) : 0.f;
            if (half == 0) sdz += dz;
            const int px = 32 * s + col;
            // DH rows hold [hpre_u > 0] * dz (one bf16 conversion per pixel); w2_u is applied to the finished products in the finalize kernel
            const u16 dzb = to_bf16(dz);
            const float dzr = __uint_as_float((unsigned int)dzb << 16);
#pragma unroll
            for (int r = 0; r < G::RU; ++r) {
                const bool on = hp[0][r] > 0.f;
                dw2a[r] = fmaf(on ? dzr : 0.f, hp[0][r], dw2a[r]);          // dz * relu(hpre_u)
                DH[kpos_reg(0, r, half) * GP + px] = on ? dzb : (u16)0;
            }
            if constexpr (C == 32) {
                const bool on = hp[1][0] > 0.f;
                dw2x = fmaf(on ? dzr : 0.f, hp[1][0], dw2x);
                if (half == 0) MDH[px] = on ? dzb : (u16)0;
            }
#pragma unroll
            for (int ks = 0; ks < G::NK; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 16 * ks + 8 * half + e;
                    if (16 * ks + e < 32) XT[k * GP + px] = frag_elem(b.xb[s][ks], e);
                    else if (e < 2 && half == 0) MX[e * GP + px] = frag_elem(b.xb[s][ks], e);
                }
        }
        wave_lds_fence();
        if constexpr (C == 32) {
            tile_mma(DH, 32, XT, 32, lane, acc[0]);
            tile_mma(DH, 32, MX, 2, lane, acc[1]);
            tile_mma(MDH, 1, XT, 32, lane, acc[2]);
            tile_mma(MDH, 1, MX, 2, lane, acc[3]);
        } else {
            tile_mma(DH, C1, XT, C + 2, lane, acc[0]);
        }
        wave_lds_fence();
    };
    pair_pipeline<Buf, C != 32>(blockIdx.x * 4u + wave, npairs, nw, load, compute);
    __syncthreads();
    float* red = (float*)g_lds;
    for (int i = threadIdx.x; i < S_WS; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) dw2a[r] = half_sum(dw2a[r]);
    dw2x = half_sum(dw2x); sdz = half_sum(sdz);
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < (C == 32 ? 4 : 1); ++t) tile_add(red + t * 1024, acc[t], lane);
            if (col == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[S_VEC + kpos_reg(0, r, half)] += dw2a[r];
                if (half == 0) { red[S_VEC + 32] += dw2x; red[S_VEC + 64] += sdz; }
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < S_WS; i += 256) ws[(size_t)blockIdx.x * S_WS + i] = red[i];
}

// red = reduced pass-2 workspace.  Outputs: dw1 [C1][C1], db1 [C1], dw2 [C1], db2 [1], dbn0 = {dgamma0 [C1], dbeta0 [C1]},
// E [3][C1] with dcat_c = E0*da0_c + E1 + E2*cat_c
struct GateSumsFinalize {
    int C; const float* bn0; const float* w1; const float* w2; float count;
    float* dw1; float* db1; float* dw2; float* db2; float* dbn0; float* E;
    __device__ void operator()(const float* red) const
    {
        const int C1 = C + 1;
        auto D = [&](int u, int k) {                 // sum_p dh_u * x~_k ; the tiles hold sum_p [hpre_u > 0] dz x~_k
            const float t = C < 32 ? red[u * 32 + k]
                                   : (u < 32 ? (k < 32 ? red[u * 32 + k] : red[1024 + u * 32 + (k - 32)]) : (k < 32 ? red[2048 + k] : red[3072 + (k - 32)]));
            return w2[u] * t;
        };
        for (int e = threadIdx.x; e < C1 * C1; e += blockDim.x) {
            const int i = e / C1, j = e - i * C1;
            dw1[e] = bn0[j] * D(i, j) + bn0[C1 + j] * D(i, C1);       // a0_j = s0_j*cat_j + t0_j ; D(i, C+1) = sum_p dh_i
        }
        for (int c = threadIdx.x; c < C1; c += blockDim.x) {
            db1[c] = D(c, C1);
            dw2[c] = red[S_VEC + c];
            float S0 = 0.f, S1 = 0.f;                                  // sum_p da0_c and sum_p da0_c*cat_c with da0 = W1^T dh
            for (int u = 0; u < C1; ++u) { const float w = w1[u * C1 + c]; S0 = fmaf(w, D(u, C1), S0); S1 = fmaf(w, D(u, c), S1); }
            const float s0 = bn0[c], mu = bn0[2 * C1 + c], is = bn0[3 * C1 + c];
            const float dgamma = is * (S1 - mu * S0);
            dbn0[c] = dgamma; dbn0[C1 + c] = S0;
            const float E2 = -s0 * dgamma / count * is;
            E[c] = s0; E[C1 + c] = -s0 * S0 / count - E2 * mu; E[2 * C1 + c] = E2;
        }
        if (threadIdx.x == 0) db2[0] = red[S_VEC + 64];
    }
};

// ------------------------------------------------------------------------------------------------ backward, pass 3
// dcat = E0 * (W1^T dh) + E1 + E2 * cat with dh = [hpre > 0] * w2 * dz:  second product  rows = cat channels (K-layout), k = hidden units,
// A = diag(E0) W1^T diag(w2) (bf16), B = [hpre > 0] * dz packed from the first product's accumulators (already in B layout).
// dfeat = dcat[0:C] + (alpha + 1) * (Wm^T dy),  dgate = dcat[C].
template <int C> __global__ __launch_bounds__(256, 2)
void gate_bwd_apply_kernel(const u16* __restrict__ dy, int lddy, const u16* __restrict__ feat, int ldf, const u16* __restrict__ gate, int ldg,
                           const float* __restrict__ q, const float* __restrict__ z, unsigned P, const float* __restrict__ K,
                           const float* __restrict__ E, const float* __restrict__ bn0, const float* __restrict__ w1, const float* __restrict__ b1,
                           const float* __restrict__ w2, const float* __restrict__ bn1, const float* __restrict__ wm,
                           u16* __restrict__ dfeat, int lddf, u16* __restrict__ dgate, int lddg)
{
    using G = GL<C>;
    constexpr int C1 = C + 1;
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31, wave = threadIdx.x >> 6;
    const float K0 = K[0], K1 = K[1], K2 = K[2], s1 = bn1[0], t1 = bn1[1];
    __shared__ float s_par[C1 * C1 + C * C + 8 * C1 + 32];
    {
        float* cur = s_par;
        w1 = stage(cur, w1, C1 * C1); b1 = stage(cur, b1, C1); bn0 = stage(cur, bn0, 2 * C1); w2 = stage(cur, w2, C1);
        E = stage(cur, E, 3 * C1); wm = stage(cur, wm, C * C);
        __syncthreads();
    }
    HiddenW<C> W; W.build(bn0, w1, b1, lane);
    bf16x8_t a2[G::MT][G::NK];                   // diag(E0) W1^T diag(w2)
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < G::NK; ++ks) {
            bf16x8_t lo;
            make_frag([&](int c, int u) { return (c <= C && u <= C) ? E[c] * w1[u * C1 + c] * w2[u] : 0.f; }, kpos_row(mt, col), ks, lane, a2[mt][ks], lo);
        }
    bf16x8_t a3h[G::NF], a3l[G::NF];             // Wm^T
#pragma unroll
    for (int ks = 0; ks < G::NF; ++ks)
        make_frag([&](int j, int o) { return (j < C && o < C) ? wm[o * C + j] : 0.f; }, kpos_row(0, col), ks, lane, a3h[ks], a3l[ks]);
    float e1k[16], e1x, e2k[16], e2x;
    load_kvec<C>(E + C1, half, e1k, e1x); load_kvec<C>(E + 2 * C1, half, e2k, e2x);
    const unsigned npairs = (P + 63u) >> 6, nw = gridDim.x * 4u;
    struct Buf { bf16x8_t xb[2][G::NK], gb[2][G::NF]; float qq[2], zz[2]; };
    auto load = [&](Buf& b, unsigned pair) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned p = pair * 64u + 32u * s + col;
            const size_t pp = p < P ? p : P - 1;
            load_xt<C>(feat, ldf, gate, ldg, pp, half, b.xb[s]);
            load_feat<C>(dy, lddy, pp, half, b.gb[s]);
            b.qq[s] = q[pp]; b.zz[s] = z[pp];
        }
    };
    auto compute = [&](Buf& b, unsigned pair) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned p = pair * 64u + 32u * s + col;
            f32x16 hp[G::MT]; W.apply(b.xb[s], hp);
            const float dz = fmaf(K0, b.qq[s], fmaf(K2, b.zz[s], K1));
            bf16x8_t db[G::NK];                  // [hpre > 0] * dz in B layout
            {
                float m[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) m[r] = hp[0][r] > 0.f ? dz : 0.f;
                db[0] = as_frag(Vec16<u16>::pack(m));
                if constexpr (G::NK > 1 && C < 32) db[1] = as_frag(Vec16<u16>::pack(m + 8));
                if constexpr (C == 32) {
                    db[1] = as_frag(Vec16<u16>::pack(m + 8));
                    db[2] = as_frag(u32x4{pack_bf16x2(hp[1][0] > 0.f ? dz : 0.f, 0.f), 0u, 0u, 0u});
                }
            }
            f32x16 da[G::MT];
#pragma unroll
            for (int mt = 0; mt < G::MT; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) da[mt][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < G::NK; ++ks) da[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[mt][ks], db[ks], da[mt], 0, 0, 0);
            }
            f32x16 du;
#pragma unroll
            for (int r = 0; r < 16; ++r) du[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < G::NF; ++ks) {
                du = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3h[ks], b.gb[s][ks], du, 0, 0, 0);
                du = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3l[ks], b.gb[s][ks], du, 0, 0, 0);
            }
            float cat[16];
            unpack_frag(b.xb[s][0], cat);
            if constexpr (G::NK > 1) unpack_frag(b.xb[s][1], cat + 8);
            else {
#pragma unroll
                for (int r = 8; r < 16; ++r) cat[r] = 0.f;
            }
            const float a1 = sigmoidf_(fmaf(b.zz[s], s1, t1)) + 1.f;
            float o[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dcat = da[0][r] + fmaf(e2k[r], cat[r], e1k[r]);
                o[r] = fmaf(du[r], a1, dcat);         // positions >= C: du = 0, so o = dcat there
            }
            if (p < P) {
                u16* dr = dfeat + (size_t)p * lddf + 8 * half;
                if (C >= 16 || half == 0) *(u32x4*)dr = Vec16<u16>::pack(o);
                if constexpr (C == 32) *(u32x4*)(dr + 16) = Vec16<u16>::pack(o + 8);
                float dg;
                if constexpr (C == 32) {
                    float g32[8]; unpack_frag(b.xb[s][2], g32);
                    dg = da[1][0] + fmaf(e2x, g32[0], e1x);
                } else {
                    dg = o[G::G_REG];
                }
                if (half == G::G_HALF) dgate[(size_t)p * lddg] = to_bf16(dg);
            }
        }
    };
    pair_pipeline<Buf, C != 32>(blockIdx.x * 4u + wave, npairs, nw, load, compute);
}

// backward passes 1 and 2 (per-block partial tiles in the workspace): 4 waves x 64 pixels per block iteration
static int gate_blocks(int64_t pixels)
{
    long b = (pixels + 255) / 256;
    if (b > 768) b = 768;      // 3 resident blocks per CU (LDS tiles: 37-46 KB per block); partial tiles per block stay small
    return (int)(b < 1 ? 1 : b);
}
// streaming passes: a wave takes >= 8 pixel pairs (64 pixels each) so that building its weight fragments is amortised
static unsigned gate_stream_blocks(int64_t pixels)
{
    constexpr long cap = 512;    // measured: 512 < 1024 < 2048 (the per-wave fragment build)
    long b = (pixels + 2047) / 2048;
    if (b > cap) b = cap;
    return (unsigned)(b < 1 ? 1 : b);
}

static int gate_check(const char* what, int dtype, int C, int64_t pixels, const void* feat, int ldf)
{
    if (dtype != SAUNET_BF16) return set_error(SAUNET_BAD_DTYPE, "%s: bf16 storage only (dtype %d)", what, dtype);
    if (C != 8 && C != 16 && C != 32) return set_error(SAUNET_UNSUPPORTED, "%s: C=%d (8, 16 or 32)", what, C);
    if (pixels <= 0 || pixels >= (1LL << 32) - 256 * 512) return set_error(SAUNET_BAD_SHAPE, "%s: %lld pixels", what, (long long)pixels);
    if (ldf % 8 || ((uintptr_t)feat & 15)) return set_error(SAUNET_BAD_ALIGN, "%s: feature rows must be 16-byte aligned", what);
    return SAUNET_OK;
}

}  // namespace saunet

using namespace saunet;

#define GATE_C(C, CALL)                          \
    do {                                         \
        if ((C) == 32) { CALL(32); }             \
        else if ((C) == 16) { CALL(16); }        \
        else { CALL(8); }                        \
    } while (0)

extern "C" {

int saunet_gate_forward_z(int dtype, int C, const void* feat, int ldf, const void* gate, int ldg, int64_t pixels, const float* bn0,
                          const float* w1, const float* b1, const float* w2, const float* b2, float* z, double* zsum, double* zsq,
                          int replicas, int rstride, void* stream)
{
    if (int rc = gate_check("gate_forward_z", dtype, C, pixels, feat, ldf)) return rc;
    const unsigned blocks = gate_stream_blocks(pixels);
#define CALL(CC) hipLaunchKernelGGL(gate_fwd_z_kernel<CC>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)feat, ldf, (const u16*)gate, ldg, \
                                    (unsigned)pixels, bn0, w1, b1, w2, b2, z, zsum, zsq, replicas > 0 ? replicas : 1, rstride)
    GATE_C(C, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("gate_forward_z");
    return SAUNET_OK;
}

int saunet_gate_forward_out(int dtype, int C, const void* feat, int ldf, const float* z, int64_t pixels, const float* bn1, const float* wm,
                            void* y, int ldy, void* alpha, void* stream)
{
    if (int rc = gate_check("gate_forward_out", dtype, C, pixels, feat, ldf)) return rc;
    if (ldy % 8 || ((uintptr_t)y & 15)) return set_error(SAUNET_BAD_ALIGN, "gate_forward_out: output rows must be 16-byte aligned");
    const unsigned blocks = gate_stream_blocks(pixels);
#define CALL(CC) hipLaunchKernelGGL(gate_fwd_out_kernel<CC>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)feat, ldf, z, (unsigned)pixels, \
                                    bn1, wm, (u16*)y, ldy, (u16*)alpha)
    GATE_C(C, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("gate_forward_out");
    return SAUNET_OK;
}

int64_t saunet_gate_backward_workspace(int64_t pixels)
{
    return (int64_t)sizeof(float) * ((int64_t)gate_blocks(pixels) * S_WS + S_WS);
}

int saunet_gate_backward_q(int dtype, int C, const void* dy, int lddy, const void* feat, int ldf, const float* z, const void* dalpha,
                           int64_t pixels, const float* bn1, const float* wm, float* q, float* dwm, float* dbn1, float* K,
                           void* workspace, int64_t workspace_bytes, void* stream)
{
    if (int rc = gate_check("gate_backward_q", dtype, C, pixels, feat, ldf)) return rc;
    if (lddy % 8 || ((uintptr_t)dy & 15)) return set_error(SAUNET_BAD_ALIGN, "gate_backward_q: dy rows must be 16-byte aligned");
    if (workspace_bytes < saunet_gate_backward_workspace(pixels)) return set_error(SAUNET_BAD_SHAPE, "gate_backward_q: workspace too small");
    const int blocks = gate_blocks(pixels);
    float* ws = (float*)workspace; float* red = ws + (size_t)blocks * S_WS;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = sizeof(u16) * 4 * 2 * G_TILE;
#define CALL(CC) hipLaunchKernelGGL(gate_bwd_q_kernel<CC>, dim3(blocks), dim3(256), lds, st, (const u16*)dy, lddy, (const u16*)feat, ldf, z, (const u16*)dalpha, \
                                    (unsigned)pixels, bn1, wm, q, ws, red)
    GATE_C(C, CALL);
#undef CALL
    const GateQFinalize fin{C, bn1, (float)pixels, dwm, dbn1, K};
    hipLaunchKernelGGL(gate_reduce_kernel, dim3((Q_WS + 255) / 256, 32), dim3(256), 0, st, ws, blocks, Q_WS, Q_WS, red);
    hipLaunchKernelGGL(gate_finalize_kernel<GateQFinalize>, dim3(1), dim3(256), 0, st, red, Q_WS, fin);
    SAUNET_CHECK_LAUNCH("gate_backward_q");
    return SAUNET_OK;
}

int saunet_gate_backward_sums(int dtype, int C, const void* feat, int ldf, const void* gate, int ldg, const float* q, const float* z,
                              int64_t pixels, const float* K, const float* bn0, const float* w1, const float* b1, const float* w2,
                              float* dw1, float* db1, float* dw2, float* db2, float* dbn0, float* E,
                              void* workspace, int64_t workspace_bytes, void* stream)
{
    if (int rc = gate_check("gate_backward_sums", dtype, C, pixels, feat, ldf)) return rc;
    if (workspace_bytes < saunet_gate_backward_workspace(pixels)) return set_error(SAUNET_BAD_SHAPE, "gate_backward_sums: workspace too small");
    const int blocks = gate_blocks(pixels);
    float* ws = (float*)workspace; float* red = ws + (size_t)blocks * S_WS;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = sizeof(u16) * 4 * (2 * G_TILE + (C == 32 ? 2 * G_MISC : 0));
#define CALL(CC) hipLaunchKernelGGL(gate_bwd_sums_kernel<CC>, dim3(blocks), dim3(256), lds, st, (const u16*)feat, ldf, (const u16*)gate, ldg, q, z, \
                                    (unsigned)pixels, K, bn0, w1, b1, w2, ws, red)
    GATE_C(C, CALL);
#undef CALL
    const GateSumsFinalize fin{C, bn0, w1, w2, (float)pixels, dw1, db1, dw2, db2, dbn0, E};
    hipLaunchKernelGGL(gate_reduce_kernel, dim3((S_WS + 255) / 256, 32), dim3(256), 0, st, ws, blocks, S_WS, S_WS, red);
    hipLaunchKernelGGL(gate_finalize_kernel<GateSumsFinalize>, dim3(1), dim3(256), 0, st, red, S_WS, fin);
    SAUNET_CHECK_LAUNCH("gate_backward_sums");
    return SAUNET_OK;
}

int saunet_gate_backward_apply(int dtype, int C, const void* dy, int lddy, const void* feat, int ldf, const void* gate, int ldg,
                               const float* q, const float* z, int64_t pixels, const float* K, const float* E, const float* bn0,
                               const float* w1, const float* b1, const float* w2, const float* bn1, const float* wm,
                               void* dfeat, int lddf, void* dgate, int lddg, void* stream)
{
    if (int rc = gate_check("gate_backward_apply", dtype, C, pixels, feat, ldf)) return rc;
    if (lddy % 8 || lddf % 8 || (((uintptr_t)dy | (uintptr_t)dfeat) & 15)) return set_error(SAUNET_BAD_ALIGN, "gate_backward_apply: rows must be 16-byte aligned");
    const unsigned blocks = gate_stream_blocks(pixels);
#define CALL(CC) hipLaunchKernelGGL(gate_bwd_apply_kernel<CC>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)dy, lddy, (const u16*)feat, ldf, \
                                    (const u16*)gate, ldg, q, z, (unsigned)pixels, K, E, bn0, w1, b1, w2, bn1, wm, (u16*)dfeat, lddf, (u16*)dgate, lddg)
    GATE_C(C, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("gate_backward_apply");
    return SAUNET_OK;
}

}  // extern "C"
