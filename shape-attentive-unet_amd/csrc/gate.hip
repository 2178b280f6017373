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
// Unfused this is ~13 full-resolution passes over odd-width (33/17/9 channel) tensors.  Here ONE THREAD OWNS ONE PIXEL:
// the whole chain runs in registers, weights are wave-uniform (scalar loads), and only feat / gate / z / y touch HBM.
// The two batch-norm statistics are the only global dependencies, so forward = 2 passes (z, then y) and backward =
// 3 passes that RECOMPUTE the chain instead of storing it:
//     pass 1: q = dL/d(bn1 out) per pixel,  sums for BN1 backward,  dWm
//     pass 2: all remaining parameter gradients and the two BN0 backward sums
//     pass 3: dfeat, dgate
// Cross-pixel sums  sum_p a_p[i] * b_p[j]  (weight gradients, bias gradients as products with a ones column, BN sums) run on
// the matrix cores: every wave transposes its 64 pixels through LDS into [channel][pixel] bf16 tiles and issues
// mfma_f32_32x32x16_bf16 with K = pixels; per-block partial tiles go to a workspace and a small reduce + finalize pair
// turns them into gradients and the per-channel coefficients of the next pass.  bf16 storage only.
#include "common.h"
#include "mma_tiles.h"

namespace saunet {

constexpr int Q_WS = 1056;        // floats per block of pass 1: 32x32 dWm tile + {sum q, sum q*zhat} (+ pad)
constexpr int S_WS = 7 * 1024;    // floats per block of pass 2: seven 32x32 product tiles

template <int C> __device__ __forceinline__ void load_cat(const u16* __restrict__ feat, int ldf, const u16* __restrict__ gate, int ldg, size_t p, float* cat)
{
    load_row<C>(feat + p * ldf, cat);
    cat[C] = Elem<u16>::load(gate + p * ldg);
}

// Weights are wave-uniform and read with scalar loads.  Left alone the compiler hoists every one of the ~2000 loads out of
// the pixel loop and spills the SGPRs; re-materialising the row pointer through an empty asm pins each row's loads to the
// place the row is used (one s_load_dwordx16 burst per row, a couple of rows in flight).
// (the OFFSET is laundered, not the pointer, so the loads keep their global / noalias provenance and stay scalar)
__device__ __forceinline__ const float* row_ptr(const float* __restrict__ p, int off = 0)
{
    asm volatile("" : "+s"(off));
    return p + off;
}
// same, and additionally ordered after the computation of `dep` (keeps the scheduler from issuing all rows' loads up front)
__device__ __forceinline__ const float* row_ptr(const float* __restrict__ p, int off, float& dep)
{
    asm volatile("" : "+s"(off), "+v"(dep));
    return p + off;
}
// y[i] = b[i] + sum_j W[i][j] x[j]      (W row-major [M][N])
template <int M, int N> __device__ __forceinline__ void matvec(const float* __restrict__ W, const float* x, float* y)
{
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const float* wr = i >= 2 ? row_ptr(W, i * N, y[i - 2]) : row_ptr(W, i * N);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < N; ++j) s = fmaf(wr[j], x[j], s);
        y[i] = s;
    }
}
// y[j] = sum_i W[i][j] x[i]
template <int M, int N> __device__ __forceinline__ void matvec_t(const float* __restrict__ W, const float* x, float* y)
{
#pragma unroll
    for (int j = 0; j < N; ++j) y[j] = 0.f;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const float* wr = i >= 2 ? row_ptr(W, i * N, y[(i & 1) ? N - 1 : 0]) : row_ptr(W, i * N);   // y[.] as of row i-1 / i-2
#pragma unroll
        for (int j = 0; j < N; ++j) y[j] = fmaf(wr[j], x[i], y[j]);
    }
}

// hpre = W1 * (s0*cat + t0) + b1
template <int C> __device__ __forceinline__ void gate_hidden(const float* cat, const float* __restrict__ bn0, const float* __restrict__ w1,
                                                             const float* __restrict__ b1, float* hpre)
{
    constexpr int C1 = C + 1;
    float a0[C1];
    const float* sc = row_ptr(bn0);
#pragma unroll
    for (int j = 0; j < C1; ++j) a0[j] = fmaf(cat[j], sc[j], sc[C1 + j]);
#pragma unroll
    for (int i = 0; i < C1; ++i) {
        const float* wr = i >= 2 ? row_ptr(w1, i * C1, hpre[i - 2]) : row_ptr(w1, i * C1);
        const float* br = row_ptr(b1, i);
        float s = br[0];
#pragma unroll
        for (int j = 0; j < C1; ++j) s = fmaf(wr[j], a0[j], s);
        hpre[i] = s;
    }
}

// (defined below gate_hidden)
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// ------------------------------------------------------------------------------------------------ forward, pass 1: z
template <int C> __global__ __launch_bounds__(256)
void gate_fwd_z_kernel(const u16* __restrict__ feat, int ldf, const u16* __restrict__ gate, int ldg, unsigned P,
                       const float* __restrict__ bn0, const float* __restrict__ w1, const float* __restrict__ b1,
                       const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ z,
                       double* __restrict__ zsum, double* __restrict__ zsq, int reps, int rstride)
{
    constexpr int C1 = C + 1;
    __shared__ float s_red[2];
    if (threadIdx.x < 2) s_red[threadIdx.x] = 0.f;
    __syncthreads();
    float ls = 0.f, lq = 0.f;
    for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < P; p += gridDim.x * 256u) {
        float cat[C1], hpre[C1];
        load_cat<C>(feat, ldf, gate, ldg, p, cat);
        gate_hidden<C>(cat, bn0, w1, b1, hpre);
        const float* w2r = row_ptr(w2);
        float zz = row_ptr(b2)[0];
#pragma unroll
        for (int i = 0; i < C1; ++i) zz = fmaf(w2r[i], fmaxf(hpre[i], 0.f), zz);
        z[p] = zz;
        ls += zz; lq = fmaf(zz, zz, lq);
    }
    if (zsum != nullptr) {
        ls = wave_sum(ls); lq = wave_sum(lq);
        if ((threadIdx.x & 63) == 0) { atomicAdd(&s_red[0], ls); atomicAdd(&s_red[1], lq); }
        __syncthreads();
        if (threadIdx.x == 0) {
            const size_t ro = (size_t)(blockIdx.x % reps) * rstride;
            atomicAdd(&zsum[ro], (double)s_red[0]); atomicAdd(&zsq[ro], (double)s_red[1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward, pass 2: y, alpha
template <int C> __global__ __launch_bounds__(256)
void gate_fwd_out_kernel(const u16* __restrict__ feat, int ldf, const float* __restrict__ z, unsigned P, const float* __restrict__ bn1,
                         const float* __restrict__ wm, u16* __restrict__ y, int ldy, u16* __restrict__ alpha)
{
    const float s1 = bn1[0], t1 = bn1[1];
    for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < P; p += gridDim.x * 256u) {
        float u[C], o[C];
        load_row<C>(feat + (size_t)p * ldf, u);
        const float al = sigmoidf_(fmaf(z[p], s1, t1));
#pragma unroll
        for (int j = 0; j < C; ++j) u[j] *= al + 1.f;
        matvec<C, C>(wm, u, o);
        store_row<C>(y + (size_t)p * ldy, o);
        alpha[p] = to_bf16(al);
    }
}

// ------------------------------------------------------------------------------------------------ backward, pass 1
// q = dL/d(BN1 output) = (sum_c du_c * feat_c [+ dalpha_ext]) * alpha * (1 - alpha),  du = Wm^T dy
// workspace[block] = { dWm tile [32][32] (row = out channel i, col = in channel j) , sum q , sum q * zhat }
template <int C> __global__ __launch_bounds__(256)
void gate_bwd_q_kernel(const u16* __restrict__ dy, int lddy, const u16* __restrict__ feat, int ldf, const float* __restrict__ z,
                       const u16* __restrict__ dalpha_ext, unsigned P, const float* __restrict__ bn1, const float* __restrict__ wm,
                       float* __restrict__ q, float* __restrict__ ws)
{
    extern __shared__ u16 g_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u16* tA = g_lds + wave * 2 * G_TILE; u16* tB = tA + G_TILE;
    const float s1 = bn1[0], t1 = bn1[1], mu1 = bn1[2], is1 = bn1[3];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float sq = 0.f, sqz = 0.f;
    for (unsigned base = blockIdx.x * 256u; base < P; base += gridDim.x * 256u) {
        const unsigned p = base + threadIdx.x; const bool live = p < P; const size_t pp = live ? p : 0;
        float g[C], f[C];
        load_row<C>(dy + pp * lddy, g); load_row<C>(feat + pp * ldf, f);
        const float zz = z[pp];
        const float al = sigmoidf_(fmaf(zz, s1, t1));
        float dal = (dalpha_ext != nullptr) ? Elem<u16>::load(dalpha_ext + pp) : 0.f;
        if (!live) {
            dal = 0.f;
#pragma unroll
            for (int i = 0; i < C; ++i) g[i] = 0.f;
        }
        float du[C];
        matvec_t<C, C>(wm, g, du);
#pragma unroll
        for (int j = 0; j < C; ++j) dal = fmaf(du[j], f[j], dal);
        const float qq = dal * al * (1.f - al);
        if (live) q[p] = qq;
        sq += qq; sqz = fmaf(qq, (zz - mu1) * is1, sqz);
#pragma unroll
        for (int i = 0; i < C; ++i) { tA[i * GP + lane] = to_bf16(g[i]); tB[i * GP + lane] = to_bf16(f[i] * (al + 1.f)); }
        __syncthreads();
        tile_mma(tA, C, tB, C, lane, acc);
        __syncthreads();
    }
    float* red = (float*)g_lds;
    for (int i = threadIdx.x; i < Q_WS; i += 256) red[i] = 0.f;
    __syncthreads();
    tile_flush(red, acc, lane);
    sq = wave_sum(sq); sqz = wave_sum(sqz);
    if (lane == 0) { atomicAdd(&red[1024], sq); atomicAdd(&red[1025], sqz); }
    __syncthreads();
    for (int i = threadIdx.x; i < Q_WS; i += 256) ws[(size_t)blockIdx.x * Q_WS + i] = red[i];
}

// out[e] += sum_b ws[b][e]  (out zeroed by the caller; blockIdx.y strides the partial blocks)
__global__ __launch_bounds__(256) void gate_reduce_kernel(const float* __restrict__ ws, int nblocks, int stride, int n, float* __restrict__ out)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    for (int b = blockIdx.y; b < nblocks; b += gridDim.y) s += ws[(size_t)b * stride + e];
    atomicAdd(&out[e], s);
}

// red = reduced pass-1 workspace.  Outputs: dwm [C][C], dbn1 = {dgamma1, dbeta1}, K = {K0, K1, K2} with dz = K0*q + K1 + K2*z
__global__ void gate_bwd_q_finalize_kernel(int C, const float* __restrict__ red, const float* __restrict__ bn1, float count,
                                           float* __restrict__ dwm, float* __restrict__ dbn1, float* __restrict__ K)
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

// ------------------------------------------------------------------------------------------------ backward, pass 2
// Product tiles per block (A rows x B cols, both indexed by LDS tile row):
//   A tiles: DH = dh[0:32], DA = da0[0:32], DE = (da0*cat)[0:32], misc MA = {dh[32], da0[32], (da0*cat)[32], dz}
//   B tiles: CT = cat[0:32], HH = h[0:32], misc MB = {cat[32], h[32], 1}
//   0: DH x CT   1: DH x MB   2: DA x MB   3: DE x MB   4: MA x CT   5: MA x HH   6: MA x MB
template <int C> __global__ __launch_bounds__(256)
void gate_bwd_sums_kernel(const u16* __restrict__ feat, int ldf, const u16* __restrict__ gate, int ldg, const float* __restrict__ q,
                          const float* __restrict__ z, unsigned P, const float* __restrict__ K, const float* __restrict__ bn0,
                          const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, float* __restrict__ ws)
{
    constexpr int C1 = C + 1;
    constexpr int RM = C1 < 32 ? C1 : 32;       // rows used in the 32-row tiles
    extern __shared__ u16 g_lds[];
    constexpr int WAVE_LDS = 3 * G_TILE + 2 * G_MISC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u16* CT = g_lds + wave * WAVE_LDS; u16* HH = CT + G_TILE; u16* RA = HH + G_TILE; u16* MB = RA + G_TILE; u16* MA = MB + G_MISC;
    for (int i = lane; i < WAVE_LDS; i += 64) CT[i] = 0;
    __syncthreads();
    const float K0 = K[0], K1 = K[1], K2 = K[2];
    f32x16 acc[7];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (unsigned base = blockIdx.x * 256u; base < P; base += gridDim.x * 256u) {
        const unsigned p = base + threadIdx.x; const bool live = p < P; const size_t pp = live ? p : 0;
        float cat[C1], hd[C1];
        load_cat<C>(feat, ldf, gate, ldg, pp, cat);
        const float dz = live ? fmaf(K0, q[pp], fmaf(K2, z[pp], K1)) : 0.f;
        gate_hidden<C>(cat, bn0, w1, b1, hd);
        const float* w2r = row_ptr(w2);
#pragma unroll
        for (int c = 0; c < C1; ++c) {
            const u16 cv = to_bf16(cat[c]), hv = to_bf16(fmaxf(hd[c], 0.f));
            hd[c] = hd[c] > 0.f ? w2r[c] * dz : 0.f;                      // hd becomes dh
            const u16 dv = to_bf16(hd[c]);
            if (c < 32) { CT[c * GP + lane] = cv; HH[c * GP + lane] = hv; RA[c * GP + lane] = dv; }
            else { MB[0 * GP + lane] = cv; MB[1 * GP + lane] = hv; MA[0 * GP + lane] = dv; }
        }
        MB[2 * GP + lane] = to_bf16(live ? 1.f : 0.f);
        MA[3 * GP + lane] = to_bf16(dz);
        __syncthreads();
        tile_mma(RA, RM, CT, RM, lane, acc[0]);
        tile_mma(RA, RM, MB, 3, lane, acc[1]);
        float da[C1];
        matvec_t<C1, C1>(w1, hd, da);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < C1; ++j) { if (j < 32) RA[j * GP + lane] = to_bf16(da[j]); else MA[1 * GP + lane] = to_bf16(da[j]); }
        __syncthreads();
        tile_mma(RA, RM, MB, 3, lane, acc[2]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < C1; ++j) { const u16 ev = to_bf16(da[j] * cat[j]); if (j < 32) RA[j * GP + lane] = ev; else MA[2 * GP + lane] = ev; }
        __syncthreads();
        tile_mma(RA, RM, MB, 3, lane, acc[3]);
        tile_mma(MA, 4, CT, RM, lane, acc[4]);
        tile_mma(MA, 4, HH, RM, lane, acc[5]);
        tile_mma(MA, 4, MB, 3, lane, acc[6]);
        __syncthreads();
    }
    float* red = (float*)g_lds;                   // 7 x 1024 floats = 28 KB <= 4 * WAVE_LDS * 2 B
    for (int i = threadIdx.x; i < S_WS; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 7; ++t) tile_flush(red + t * 1024, acc[t], lane);
    __syncthreads();
    for (int i = threadIdx.x; i < S_WS; i += 256) ws[(size_t)blockIdx.x * S_WS + i] = red[i];
}

// red = reduced pass-2 workspace.  Outputs: dw1 [C1][C1], db1 [C1], dw2 [C1], db2 [1], dbn0 = {dgamma0 [C1], dbeta0 [C1]},
// E [3][C1] with dcat_c = E0*da0_c + E1 + E2*cat_c
__global__ void gate_bwd_sums_finalize_kernel(int C, const float* __restrict__ red, const float* __restrict__ bn0, float count,
                                              float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
                                              float* __restrict__ dbn0, float* __restrict__ E)
{
    const int C1 = C + 1;
    auto T = [&](int t, int r, int c) { return red[t * 1024 + r * 32 + c]; };
    auto dhcat = [&](int i, int j) { return i < 32 ? (j < 32 ? T(0, i, j) : T(1, i, 0)) : (j < 32 ? T(4, 0, j) : T(6, 0, 0)); };
    auto sdh = [&](int i) { return i < 32 ? T(1, i, 2) : T(6, 0, 2); };
    for (int e = threadIdx.x; e < C1 * C1; e += blockDim.x) {
        const int i = e / C1, j = e - i * C1;
        dw1[e] = bn0[j] * dhcat(i, j) + bn0[C1 + j] * sdh(i);     // a0_j = s0_j*cat_j + t0_j
    }
    for (int c = threadIdx.x; c < C1; c += blockDim.x) {
        db1[c] = sdh(c);
        dw2[c] = c < 32 ? T(5, 3, c) : T(6, 3, 1);
        const float S0 = c < 32 ? T(2, c, 2) : T(6, 1, 2);
        const float S1 = c < 32 ? T(3, c, 2) : T(6, 2, 2);
        const float s0 = bn0[c], mu = bn0[2 * C1 + c], is = bn0[3 * C1 + c];
        const float dgamma = is * (S1 - mu * S0);
        dbn0[c] = dgamma; dbn0[C1 + c] = S0;
        const float E2 = -s0 * dgamma / count * is;
        E[c] = s0; E[C1 + c] = -s0 * S0 / count - E2 * mu; E[2 * C1 + c] = E2;
    }
    if (threadIdx.x == 0) db2[0] = T(6, 3, 2);
}

// ------------------------------------------------------------------------------------------------ backward, pass 3
template <int C> __global__ __launch_bounds__(256)
void gate_bwd_apply_kernel(const u16* __restrict__ dy, int lddy, const u16* __restrict__ feat, int ldf, const u16* __restrict__ gate, int ldg,
                           const float* __restrict__ q, const float* __restrict__ z, unsigned P, const float* __restrict__ K,
                           const float* __restrict__ E, const float* __restrict__ bn0, const float* __restrict__ w1, const float* __restrict__ b1,
                           const float* __restrict__ w2, const float* __restrict__ bn1, const float* __restrict__ wm,
                           u16* __restrict__ dfeat, int lddf, u16* __restrict__ dgate, int lddg)
{
    constexpr int C1 = C + 1;
    const float K0 = K[0], K1 = K[1], K2 = K[2], s1 = bn1[0], t1 = bn1[1];
    for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < P; p += gridDim.x * 256u) {
        float cat[C1], hd[C1], g[C];
        load_cat<C>(feat, ldf, gate, ldg, p, cat);
        const float zz = z[p];
        const float dz = fmaf(K0, q[p], fmaf(K2, zz, K1));
        gate_hidden<C>(cat, bn0, w1, b1, hd);
        {
            const float* w2r = row_ptr(w2);
#pragma unroll
            for (int c = 0; c < C1; ++c) hd[c] = hd[c] > 0.f ? w2r[c] * dz : 0.f;
        }
        float da[C1];
        matvec_t<C1, C1>(w1, hd, da);
        {
            const float* e = row_ptr(E);
#pragma unroll
            for (int j = 0; j < C1; ++j) da[j] = fmaf(e[j], da[j], fmaf(e[2 * C1 + j], cat[j], e[C1 + j]));   // da becomes dcat
        }
        dgate[(size_t)p * lddg] = to_bf16(da[C]);
        const float a1 = sigmoidf_(fmaf(zz, s1, t1)) + 1.f;
        load_row<C>(dy + (size_t)p * lddy, g);
        float du[C];
        matvec_t<C, C>(wm, g, du);
#pragma unroll
        for (int j = 0; j < C; ++j) du[j] = fmaf(du[j], a1, da[j]);
        store_row<C>(dfeat + (size_t)p * lddf, du);
    }
}

static int gate_blocks(int64_t pixels)
{
    long b = (pixels + 255) / 256;
    if (b > 512) b = 512;      // 2 resident blocks per CU; partial tiles per block stay small
    return (int)(b < 1 ? 1 : b);
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
    long blocks = (pixels + 255) / 256; if (blocks > 4096) blocks = 4096;
#define CALL(CC) hipLaunchKernelGGL(gate_fwd_z_kernel<CC>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)feat, ldf, (const u16*)gate, ldg, \
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
    long blocks = (pixels + 255) / 256; if (blocks > 4096) blocks = 4096;
#define CALL(CC) hipLaunchKernelGGL(gate_fwd_out_kernel<CC>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)feat, ldf, z, (unsigned)pixels, \
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
                                    (unsigned)pixels, bn1, wm, q, ws)
    GATE_C(C, CALL);
#undef CALL
    if (hipMemsetAsync(red, 0, sizeof(float) * Q_WS, st) != hipSuccess) return set_error(SAUNET_LAUNCH_FAILED, "gate_backward_q: memset");
    hipLaunchKernelGGL(gate_reduce_kernel, dim3((Q_WS + 255) / 256, 32), dim3(256), 0, st, ws, blocks, Q_WS, Q_WS, red);
    hipLaunchKernelGGL(gate_bwd_q_finalize_kernel, dim3(1), dim3(256), 0, st, C, red, bn1, (float)pixels, dwm, dbn1, K);
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
    const size_t lds = sizeof(u16) * 4 * (3 * G_TILE + 2 * G_MISC);
#define CALL(CC) hipLaunchKernelGGL(gate_bwd_sums_kernel<CC>, dim3(blocks), dim3(256), lds, st, (const u16*)feat, ldf, (const u16*)gate, ldg, q, z, \
                                    (unsigned)pixels, K, bn0, w1, b1, w2, ws)
    GATE_C(C, CALL);
#undef CALL
    if (hipMemsetAsync(red, 0, sizeof(float) * S_WS, st) != hipSuccess) return set_error(SAUNET_LAUNCH_FAILED, "gate_backward_sums: memset");
    hipLaunchKernelGGL(gate_reduce_kernel, dim3((S_WS + 255) / 256, 32), dim3(256), 0, st, ws, blocks, S_WS, S_WS, red);
    hipLaunchKernelGGL(gate_bwd_sums_finalize_kernel, dim3(1), dim3(256), 0, st, C, red, bn0, (float)pixels, dw1, db1, dw2, db2, dbn0, E);
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
    long blocks = (pixels + 255) / 256; if (blocks > 4096) blocks = 4096;
#define CALL(CC) hipLaunchKernelGGL(gate_bwd_apply_kernel<CC>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)dy, lddy, (const u16*)feat, ldf, \
                                    (const u16*)gate, ldg, q, z, (unsigned)pixels, K, E, bn0, w1, b1, w2, bn1, wm, (u16*)dfeat, lddf, (u16*)dgate, lddg)
    GATE_C(C, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("gate_backward_apply");
    return SAUNET_OK;
}

}  // extern "C"
