// Implicit-GEMM convolution on MFMA for gfx950 (CDNA4): forward (also used for dgrad with
// flipped/transposed weights and for ConvTranspose2d as four 2x2 phase convolutions) and wgrad.
//
// GEMM view (forward):  D[m = output pixel][n = cout] = sum_k A[m][k] * B[n][k],
//   k = (tap, cin): NHWC makes the cin slice of one tap contiguous, so a K-step is one 64/128-byte
//   row segment per pixel.  A rows are gathered (halo / stride / zero padding) through registers,
//   where the consumer-side BatchNorm+ReLU prologue  a = max(x*scale[c]+shift[c], 0)  is applied,
//   then staged in LDS with a 16-byte XOR swizzle so the per-lane ds_read_b128 fragment reads of
//   mfma_f32_32x32x16_bf16 / 4x mfma_f32_32x32x2_f32 are bank-conflict free.
//   Epilogue: + bias, per-channel sum / sum-of-squares for the following BatchNorm (float64 atomics),
//   tile transposed through LDS and written as whole 16-byte row segments.
//
// Replaces F.conv2d / F.conv_transpose2d and their autograd backward in the reference
// (/root/reference/models/models.py:118-123,203-237; attention_blocks.py:179-220; torchvision DenseNet).
#include "common.h"

namespace saunet {

struct IgemmArgs {
    const void* x; const void* w; void* y;
    const float* bias; const float* pro_scale; const float* pro_shift;
    double* stat_sum; double* stat_sumsq; int stat_replicas, stat_rstride;
    int N, H, W, Cin, ldx;
    int Ho, Wo, Cout, ldy;
    int KH, KW, stride, pad;
    int transposed, pro_relu, act_relu;
    int M;            // GEMM rows per launch (per phase when transposed)
    int Mh, Mw;       // row decode: m -> (n, q, r) with q < Mh, r < Mw
    int kpt;          // K-steps per tap = ceil(Cin / KC)
    saunet_bn_epilogue epi;   // epi.bn_x == nullptr: plain store
    saunet_bn_prologue bnp;   // bnp.gamma != nullptr: the prologue coefficients are derived in the kernel from the producer's statistics
};

template <typename T> struct Mma;
template <> struct Mma<u16> {
    __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c)
    {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) { mma_f32_chunk_exact(a, b, c); }
};
template <> struct Mma<f32s> {
    __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) { mma_f32_chunk_split(a, b, c); }
};

// byte offset of 16-byte chunk `c` of row `r` in a [rows][CPR] chunk image, XOR-swizzled so that
// 16 consecutive rows at the same logical chunk fall on 16 different 16-byte bank slots.
template <int CPR> __device__ __forceinline__ int lds_off(int r, int c)
{
    constexpr int RPB = 16 / CPR;  // rows per 256-byte bank row
    return (r * CPR + (c ^ ((r / RPB) & (CPR - 1)))) * 16;
}

template <typename T, int BM, int BN, int WM, int WN, int CPR, bool BNEPI>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, 2) void conv_igemm_fwd_kernel(IgemmArgs a)
{
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    constexpr int EPC = 16 / sizeof(T);       // elements per 16-byte chunk
    constexpr int KC = CPR * EPC;             // channels per K-step
    constexpr int A_PER_T = (BM * CPR) / NT;  // A pieces per thread
    constexpr int B_ITERS = (BN * CPR + NT - 1) / NT;
    constexpr int STAGE = (BM + BN) * CPR * 16;
    constexpr int TI = WM / 32, TJ = WN / 32;
    static_assert((BM * CPR) % NT == 0, "A tile must divide over threads");
    static_assert(NT % CPR == 0, "chunk index must be thread-invariant");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int phase = blockIdx.z, ph = phase >> 1, pw = phase & 1;
    const int taps = a.KH * a.KW;
    const T* __restrict__ xg = (const T*)a.x;
    const T* __restrict__ wg = (const T*)a.w + (size_t)phase * a.Cout * taps * a.Cin;

    // ---- per-thread A rows: decode once
    const int chunk = tid % CPR;
    int rbase[A_PER_T], rih[A_PER_T], riw[A_PER_T];
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
        int row = tid / CPR + i * (NT / CPR);
        int m = m0 + row;
        if (m < a.M) {
            int n = m / (a.Mh * a.Mw), rem = m - n * (a.Mh * a.Mw);
            int q = rem / a.Mw, r = rem - q * a.Mw;
            rbase[i] = n * a.H * a.W;
            if (a.transposed) { rih[i] = q + ph; riw[i] = r + pw; }
            else { rih[i] = q * a.stride - a.pad; riw[i] = r * a.stride - a.pad; }
        } else { rbase[i] = 0; rih[i] = -(1 << 28); riw[i] = -(1 << 28); }
    }
    const int sgn = a.transposed ? -1 : 1;
    const bool has_pro = a.pro_scale != nullptr || a.bnp.gamma != nullptr;

    const int nk = taps * a.kpt;
    const float relu_lo = a.pro_relu ? 0.f : -__builtin_inff();
    // BatchNorm prologue vectors in LDS (behind the two stages): fetching them from global memory inside the K loop would put their
    // loads BEHIND the prefetched A/B pieces in the in-order vmcnt queue -- waiting for them then drains the prefetch (vmcnt(0) before
    // every commit).  LDS reads retire on lgkmcnt.
    float* s_pro = (float*)(smem + 2 * STAGE);
    const int cpad = a.kpt * KC;

    // Two register stages: the loads of K-step k+2 are issued before the MFMAs of step k, and only converted
    // (BN+ReLU prologue, zero padding) and written to LDS after the MFMAs of step k+1 -- a load has two MFMA phases to land.
    // Loads are UNCONDITIONAL (out-of-range pieces read a safe in-bounds address and are zeroed by a select): a branch
    // around each load would make the compiler wait for every load separately.
    struct Stage { u32x4 a[A_PER_T]; u32x4 b[B_ITERS]; unsigned okmask; int cs; };
    int itap = 0, icstep = 0;     // (tap, channel step) of the NEXT K-step to be issued
    auto issue = [&](Stage& S) {
        const int tap = itap, cstep = icstep;
        if (++icstep == a.kpt) { icstep = 0; ++itap; }
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        const int c = cstep * KC + chunk * EPC;
        const bool cok = c < a.Cin;
        S.cs = cok ? c : 0;
        S.okmask = 0u;
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            int ih = rih[i] + sgn * kh, iw = riw[i] + sgn * kw;
            bool ok = cok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            S.okmask |= ok ? (1u << i) : 0u;
            size_t off = ok ? ((size_t)(rbase[i] + ih * a.W + iw) * a.ldx + S.cs) : (size_t)0;
            S.a[i] = *(const u32x4*)(xg + off);
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            int p = tid + i * NT;
            int brow = p / CPR, bch = p % CPR;
            int cb = cstep * KC + bch * EPC;
            bool ok = (B_ITERS * NT == BN * CPR || p < BN * CPR) && n0 + brow < a.Cout && cb < a.Cin;
            S.okmask |= ok ? (1u << (16 + i)) : 0u;
            size_t off = ok ? (((size_t)(n0 + brow) * taps + tap) * a.Cin + cb) : (size_t)0;
            S.b[i] = *(const u32x4*)(wg + off);
        }
    };
    auto commit = [&](Stage& S, int buf) {
        unsigned char* sa = smem + buf * STAGE;
        unsigned char* sb = sa + BM * CPR * 16;
        const u32x4 z = {0u, 0u, 0u, 0u};
        if (has_pro) {
            float sc[EPC], sh[EPC];
#pragma unroll
            for (int j = 0; j < EPC; j += 4) {
                f32x4 s4 = *(const f32x4*)(s_pro + S.cs + j), t4 = *(const f32x4*)(s_pro + cpad + S.cs + j);
#pragma unroll
                for (int q = 0; q < 4; ++q) { sc[j + q] = s4[q]; sh[j + q] = t4[q]; }
            }
#pragma unroll
            for (int i = 0; i < A_PER_T; ++i) {
                float f[EPC];
                Vec16<T>::unpack(S.a[i], f);
#pragma unroll
                for (int j = 0; j < EPC; ++j) f[j] = fmaxf(fmaf(f[j], sc[j], sh[j]), relu_lo);
                S.a[i] = Vec16<T>::pack(f);
            }
        }
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            int row = tid / CPR + i * (NT / CPR);
            *(u32x4*)(sa + lds_off<CPR>(row, chunk)) = (S.okmask >> i) & 1u ? S.a[i] : z;
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            int p = tid + i * NT;
            if (B_ITERS * NT == BN * CPR || p < BN * CPR)
                *(u32x4*)(sb + lds_off<CPR>(p / CPR, p % CPR)) = (S.okmask >> (16 + i)) & 1u ? S.b[i] : z;
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) {
        const unsigned char* sa = smem + buf * STAGE;
        const unsigned char* sb = sa + BM * CPR * 16;
        const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int s = 0; s < CPR / 2; ++s) {
            u32x4 af[TI], bfr[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *(const u32x4*)(sa + lds_off<CPR>(wm0 + i * 32 + lr, 2 * s + lh));
#pragma unroll
            for (int j = 0; j < TJ; ++j) bfr[j] = *(const u32x4*)(sb + lds_off<CPR>(wn0 + j * 32 + lr, 2 * s + lh));
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) Mma<T>::run(af[i], bfr[j], acc[i][j]);
        }
    };

    Stage S0, S1;
    TSTAMP_INIT();
    TSTAMP(60);
    issue(S0);
    // the prologue vectors are fetched BEHIND the first operand loads: their round trip (two dependent ones to the statistic replicas with a
    // consumer-side BatchNorm finalize) overlaps the operands' instead of preceding it
    if (a.bnp.gamma != nullptr) {
        bn_prologue_fill<NT>(a.bnp, a.Cin, cpad, s_pro, blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0);
        __syncthreads();
    } else if (has_pro) {
        for (int i = tid; i < cpad; i += NT) { s_pro[i] = i < a.Cin ? a.pro_scale[i] : 0.f; s_pro[cpad + i] = i < a.Cin ? a.pro_shift[i] : 0.f; }
        __syncthreads();
    }
    commit(S0, 0);
    __syncthreads();
    if (nk > 1) issue(S0);                       // step 1 in flight
    TSTAMP(61);
    for (int ks = 0; ks < nk; ks += 2) {
        if (ks + 2 < nk) issue(S1);              // step ks+2
        TSTAMP(62);
        compute(0);
        TSTAMP(63);
        if (ks + 1 < nk) commit(S0, 1);
        TSTAMP(64);
        __syncthreads();
        TSTAMP(65);
        if (ks + 1 >= nk) break;
        if (ks + 3 < nk) issue(S0);              // step ks+3
        TSTAMP(66);
        compute(1);
        TSTAMP(67);
        if (ks + 2 < nk) commit(S1, 0);
        TSTAMP(68);
        __syncthreads();
        TSTAMP(69);
    }

    // ---- epilogue: statistics, bias, transpose through LDS, coalesced stores
    // per-channel partial sums of the waves that share a column range: one SLOT per row-wave, added in a fixed order below (float atomics on
    // LDS made the sum depend on the waves' arrival order -- and cost 12 cycles per active lane)
    constexpr int RW = BM / WM;
    float* s_sum = (float*)(smem + BM * BN * sizeof(T));          // [RW][2][BN]
    const bool do_stats = a.stat_sum != nullptr;
    __syncthreads();
    T* so = (T*)smem;
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int col = wn0 + j * 32 + lr;
        const float bv = (a.bias != nullptr && n0 + col < a.Cout) ? a.bias[n0 + col] : 0.f;
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r];
                s += v; ss += v * v;
                int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                Elem<T>::store(so + row * BN + col, a.act_relu ? fmaxf(v + bv, 0.f) : v + bv);
            }
        if (do_stats) {
            s += __shfl_xor(s, 32, 64); ss += __shfl_xor(ss, 32, 64);
            if (lh == 0) { float* slot = s_sum + (wave / (BN / WN)) * 2 * BN; slot[col] = s; slot[BN + col] = ss; }
        }
    }
    TSTAMP(71);
    __syncthreads();
    TSTAMP(72);
    if (do_stats && tid < BN && n0 + tid < a.Cout) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < RW; ++w) { t1 += s_sum[w * 2 * BN + tid]; t2 += s_sum[w * 2 * BN + BN + tid]; }
        const size_t ro = (size_t)(blockIdx.x % a.stat_replicas) * a.stat_rstride;
        atomicAdd(&a.stat_sum[ro + n0 + tid], (double)t1);
        atomicAdd(&a.stat_sumsq[ro + n0 + tid], (double)t2);
    }
    TSTAMP(73);
    constexpr int CH = BN / EPC;  // 16-byte chunks per output row
    T* __restrict__ yg = (T*)a.y;
    constexpr bool bnb = BNEPI;
    float e1[EPC], e2[EPC], esc[EPC], esh[EPC], emu[EPC], eis[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) e1[j] = e2[j] = 0.f;
    if (bnb) {   // NT % CH == 0: the channel chunk of a thread is loop-invariant -> per-channel constants in registers
        const int colf = n0 + (tid % CH) * EPC;
        const int cs = colf < a.Cout ? colf : 0;
#pragma unroll
        for (int j = 0; j < EPC; j += 4) {
            f32x4 v0 = *(const f32x4*)(a.epi.scale + cs + j), v1 = *(const f32x4*)(a.epi.shift + cs + j);
            f32x4 v2 = *(const f32x4*)(a.epi.mean + cs + j), v3 = *(const f32x4*)(a.epi.invstd + cs + j);
#pragma unroll
            for (int q = 0; q < 4; ++q) { esc[j + q] = v0[q]; esh[j + q] = v1[q]; emu[j + q] = v2[q]; eis[j + q] = v3[q]; }
        }
    }
    constexpr int S_ITERS = (BM * CH) / NT;
    static_assert((BM * CH) % NT == 0, "store loop must divide evenly");
    size_t opixv[S_ITERS]; bool okv[S_ITERS]; u32x4 xr[S_ITERS];
    const int colv = n0 + (tid % CH) * EPC;
#pragma unroll
    for (int i = 0; i < S_ITERS; ++i) {
        int p = tid + i * NT;
        int row = p / CH;
        int m = m0 + row;
        okv[i] = m < a.M && colv < a.Cout;
        size_t opix = m;
        if (a.transposed) {
            int n = m / (a.Mh * a.Mw), rem = m - n * (a.Mh * a.Mw);
            int q = rem / a.Mw, r = rem - q * a.Mw;
            opix = ((size_t)n * a.Ho + 2 * q + ph) * a.Wo + 2 * r + pw;
        }
        opixv[i] = okv[i] ? opix : 0;
        if (bnb) xr[i] = *(const u32x4*)((const T*)a.epi.bn_x + (okv[i] ? opixv[i] * a.epi.ld_bn_x + colv : (size_t)0));   // all loads in flight together
    }
#pragma unroll
    for (int i = 0; i < S_ITERS; ++i) {
        int p = tid + i * NT;
        int row = p / CH, ch = p - row * CH;
        u32x4 v = *(const u32x4*)(so + row * BN + ch * EPC);
        if (bnb) {   // fused BatchNorm-backward reduction: g = v*[relu mask], sums += g, g*xhat
            float g[EPC], xv[EPC];
            Vec16<T>::unpack(v, g);
            Vec16<T>::unpack(xr[i], xv);
            unsigned bits = 0xffu;          // ReLU decisions handed in as bits (a residual block's output): one byte per 8-channel chunk
            if constexpr (EPC == 8) { if (a.epi.relu_mask && okv[i]) bits = a.epi.relu_mask[opixv[i] * (size_t)(a.Cout >> 3) + (colv >> 3)]; }
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                if (EPC == 8 && a.epi.relu_mask) { if (!((bits >> j) & 1u)) g[j] = 0.f; }
                else if (a.epi.relu && !(fmaf(xv[j], esc[j], esh[j]) > 0.f)) g[j] = 0.f;
                if (!okv[i]) g[j] = 0.f;
                e1[j] += g[j];
                e2[j] = fmaf(g[j], (xv[j] - emu[j]) * eis[j], e2[j]);
            }
            if (a.epi.accumulate == 1) {   // y += scale[c] * g  (linear BN backward: corrections are applied per chunk later)
                float o[EPC];
                Vec16<T>::unpack(*(const u32x4*)(yg + opixv[i] * a.ldy + colv), o);
#pragma unroll
                for (int j = 0; j < EPC; ++j) g[j] = fmaf(esc[j], g[j], o[j]);
            } else if (a.epi.accumulate == 2) {   // y = scale[c] * g  (the first consumer's term of the linear form: nothing to read)
#pragma unroll
                for (int j = 0; j < EPC; ++j) g[j] *= esc[j];
            }
            v = Vec16<T>::pack(g);
        }
        if (okv[i]) *(u32x4*)(yg + opixv[i] * a.ldy + colv) = v;
    }
    TSTAMP(70);
    if (bnb) {   // NT % CH == 0: a thread always owns the same EPC channels -> its partial sums go to slot tid / CH of the (now idle) tile area
        __syncthreads();
        constexpr int SL = NT / CH;
        float* s_e = (float*)smem;                                // [SL][2][BN]
        const int ch = tid % CH, sl = tid / CH;
#pragma unroll
        for (int j = 0; j < EPC; ++j) { s_e[(sl * 2) * BN + ch * EPC + j] = e1[j]; s_e[(sl * 2 + 1) * BN + ch * EPC + j] = e2[j]; }
        __syncthreads();
        if (tid < BN && n0 + tid < a.Cout) {
            float t1 = 0.f, t2 = 0.f;
            for (int w = 0; w < SL; ++w) { t1 += s_e[(w * 2) * BN + tid]; t2 += s_e[(w * 2 + 1) * BN + tid]; }
            const size_t ro = (size_t)(blockIdx.x % a.epi.sums_replicas) * a.epi.sums_rstride;
            atomicAdd(&a.epi.sums[ro + n0 + tid], (double)t1);
            atomicAdd(&a.epi.sums[ro + a.Cout + n0 + tid], (double)t2);
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, int CPR, bool BNEPI>
static int launch_fwd_i(const IgemmArgs& a, int phases, hipStream_t st)
{
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    constexpr int STAGE = (BM + BN) * CPR * 16;
    constexpr int EPC_ = 16 / (int)sizeof(T);
    constexpr int EPI_TILE = BM * BN * (int)sizeof(T) + (BM / WM) * 2 * BN * 4;                 // output tile + one statistics slot per row-wave
    constexpr int EPI_BN = (NT / (BN / EPC_)) * 2 * BN * 4;                                        // BN-backward sums: one slot per thread group
    constexpr int EPI = EPI_TILE > EPI_BN ? EPI_TILE : EPI_BN;
    const int pro_bytes = (a.pro_scale || a.bnp.gamma) ? 2 * a.kpt * CPR * EPC_ * 4 : 0;      // prologue scale/shift vectors behind the two stages
    const int LDS = (2 * STAGE + pro_bytes > EPI) ? 2 * STAGE + pro_bytes : EPI;
    auto kern = conv_igemm_fwd_kernel<T, BM, BN, WM, WN, CPR, BNEPI>;
    static DeviceMaxLds attr;
    if (attr.raise(LDS)) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    dim3 grid(cdiv(a.M, BM), cdiv(a.Cout, BN), phases);
    hipLaunchKernelGGL(kern, grid, dim3(NT), LDS, st, a);
    static const KName kn("conv_igemm_fwd_kernel", type_name<T>(), BM, BN, WM, WN, CPR, BNEPI);
    SAUNET_CHECK_LAUNCH(kn.s);
    return SAUNET_OK;
}

template <typename T, int BM, int BN, int WM, int WN, int CPR>
static int launch_fwd(const IgemmArgs& a, int phases, hipStream_t st)
{
    return a.epi.bn_x ? launch_fwd_i<T, BM, BN, WM, WN, CPR, true>(a, phases, st) : launch_fwd_i<T, BM, BN, WM, WN, CPR, false>(a, phases, st);
}

template <typename T> static int dispatch_fwd(const IgemmArgs& a, int phases, hipStream_t st)
{
    constexpr int EPC = 16 / sizeof(T);
    // narrow K rows (<= 4 chunks of channels) use the 64-byte-row variant: half the zero padding
    const bool narrow = a.kpt == cdiv(a.Cin, 4 * EPC);       // the caller sized the K steps for 64-byte rows
    if (a.Cout <= 32) {
        return narrow ? launch_fwd<T, 256, 32, 64, 32, 4>(a, phases, st) : launch_fwd<T, 256, 32, 64, 32, 8>(a, phases, st);
    } else if (a.Cout <= 64) {
        return narrow ? launch_fwd<T, 128, 64, 64, 32, 4>(a, phases, st) : launch_fwd<T, 128, 64, 64, 32, 8>(a, phases, st);
    } else {
        // small problems (low-resolution maps): 64x64 tiles so that the grid still covers the 256 CUs
        const long blocks128 = (long)cdiv(a.M, 128) * cdiv(a.Cout, 128) * phases;
        // (measured slower and removed from the library in round 4: 32-row tiles of two waves for 16 x 16 maps, +0.1 ms per step; 64 pixels x 128
        // output channels per workgroup on the small maps)
        if (blocks128 < 384)
            return narrow ? launch_fwd<T, 64, 64, 32, 32, 4>(a, phases, st) : launch_fwd<T, 64, 64, 32, 32, 8>(a, phases, st);
        return narrow ? launch_fwd<T, 128, 128, 64, 64, 4>(a, phases, st) : launch_fwd<T, 128, 128, 64, 64, 8>(a, phases, st);
    }
}

// requirements of the MFMA path; everything else goes to conv_direct.hip
bool igemm_supported(const saunet_conv_desc* d)
{
    const int epc = d->dtype == SAUNET_BF16 ? 8 : 4;
    return d->Cin % epc == 0 && d->Cout % epc == 0 && d->ldx % epc == 0 && d->ldy % epc == 0 && d->Cin >= epc &&
           d->Cout >= 8;
}

int igemm_forward(const saunet_conv_desc* d, const void* x, const void* w, const float* bias, const float* ps,
                  const float* psh, void* y, double* ssum, double* ssq, const saunet_bn_epilogue* epi, hipStream_t st, const saunet_bn_prologue* bnp)
{
    IgemmArgs a;
    if (bnp) a.bnp = *bnp; else a.bnp.gamma = nullptr;
    if (epi) { a.epi = *epi; if (a.epi.sums_replicas < 1) a.epi.sums_replicas = 1; } else a.epi.bn_x = nullptr;
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.pro_scale = ps; a.pro_shift = psh; a.stat_sum = ssum; a.stat_sumsq = ssq;
    a.stat_replicas = d->stat_replicas > 1 ? d->stat_replicas : 1; a.stat_rstride = d->stat_rstride;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
    a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.ldy = d->ldy;
    a.stride = d->stride; a.pad = d->pad; a.transposed = d->transposed; a.pro_relu = d->pro_relu; a.act_relu = d->epi_relu;
    int phases = 1;
    if (d->transposed) {
        if (d->KH != 4 || d->KW != 4 || d->stride != 2 || d->pad != 1 || d->Ho != 2 * d->H || d->Wo != 2 * d->W)
            return set_error(SAUNET_UNSUPPORTED, "conv_transpose: only k=4 s=2 p=1");
        a.KH = 2; a.KW = 2; a.Mh = d->H; a.Mw = d->W; a.M = d->N * d->H * d->W; phases = 4;
    } else {
        a.KH = d->KH; a.KW = d->KW; a.Mh = d->Ho; a.Mw = d->Wo; a.M = d->N * d->Ho * d->Wo;
    }
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) return set_error(SAUNET_BAD_ALIGN, "conv: pointers must be 16-byte aligned");
    if (d->dtype == SAUNET_BF16) {
        a.kpt = cdiv(d->Cin, (d->Cin <= 32) ? 32 : 64);
        // pointwise convs on large maps are memory bound with a short K loop: 32-channel K steps halve the LDS footprint, so twice as many
        // workgroups are resident per CU to overlap each other's load / epilogue latencies (33.78 -> 33.69 ms/step)
        constexpr long narrow_minpix = 100000;
        if (d->KH == 1 && d->KW == 1 && !d->transposed && a.M >= narrow_minpix && d->Cin > 64) a.kpt = cdiv(d->Cin, 32);
        return dispatch_fwd<u16>(a, phases, st);
    } else if (d->dtype == SAUNET_F32) {
        a.kpt = cdiv(d->Cin, (d->Cin <= 16) ? 16 : 32);
        if (f32_split_wanted((long)a.M)) return dispatch_fwd<f32s>(a, phases, st);
        return dispatch_fwd<float>(a, phases, st);
    }
    return set_error(SAUNET_BAD_DTYPE, "conv: dtype %d", d->dtype);
}

// =====================================================================================================
// wgrad:  dW[m = out-grad channel][tap][n = input channel] += sum_pixels dy[p][m] * a[p (+) tap][n]
// Both operands are pixel-major in memory (channels contiguous), i.e. K-major for the MFMA, so the tiles
// are staged [pixel][channel] and the fragments are gathered with transposing LDS reads
// (f32: one ds_read_b32 per k; bf16: 8 ds_read_u16 per fragment).  grid = (m-tiles*n-tiles, splits, taps);
// split-K partial sums are added atomically into the float32 gradient (parameter layout via sM/sN).
struct WgradArgs {
    const void* x; const void* dy; float* dw;
    const float* pro_scale; const float* pro_shift;
    int N, H, W, Cin, ldx;       // gathered operand ("conv input")
    int Ho, Wo, Cout, lddy;      // streamed operand ("conv output gradient")
    int KH, KW, stride, pad, pro_relu;
    int P;                       // output pixels N*Ho*Wo
    int pix_per_split;
    int ntn;                     // n tiles
    long sM, sN;                 // dw element strides for m (dy channel) and n (x channel); tap stride 1
    FastDiv dHoWo, dWo;
};

template <typename T, int KP> struct WFrag;
// f32: tile rows = pixels (KP = 16), row-major [KP][C]; 32x32x2: A[i=l&31][k=l>>5]
// bf16: KP = 32; 32x32x16: lane (r=l&31, h=l>>5) holds k = 8h..8h+7 of each 16-pixel substep

template <typename T, int BMc, int BNc>
__global__ __launch_bounds__(256) void conv_igemm_wgrad_kernel(WgradArgs a)
{
    constexpr int EPC = 16 / sizeof(T);
    constexpr int KP = 32;                       // pixels per K-step
    constexpr int WMW = BMc / 32, WNW = BNc / 32;  // 32x32 wave tiles
    static_assert(WMW * WNW == 4 || WMW * WNW == 8 || WMW * WNW == 2 || WMW * WNW == 1, "tile");
    constexpr int TPW = (WMW * WNW) / 4 > 0 ? (WMW * WNW) / 4 : 1;  // tiles per wave (4 waves)
    constexpr int ROWB_M = BMc * sizeof(T), ROWB_N = BNc * sizeof(T);
    constexpr int ST_M = KP * ROWB_M, ST_N = KP * ROWB_N, STAGE = ST_M + ST_N;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, tm = tile / a.ntn, tn = tile - tm * a.ntn;
    const int mch0 = tm * BMc, nch0 = tn * BNc;
    const int tap = blockIdx.z, kh = tap / a.KW, kw = tap - kh * a.KW;
    const int p_begin = blockIdx.y * a.pix_per_split;
    const int p_end = min(p_begin + a.pix_per_split, a.P);
    const T* __restrict__ xg = (const T*)a.x;
    const T* __restrict__ dyg = (const T*)a.dy;
    const bool has_pro = a.pro_scale != nullptr;

    constexpr int CM = BMc / EPC, CN = BNc / EPC;     // chunks per row
    constexpr int PM = (KP * CM + 255) / 256, PN = (KP * CN + 255) / 256;
    u32x4 mreg[PM], nreg[PN];

    const float relu_lo = a.pro_relu ? 0.f : -__builtin_inff();
    auto load_tile = [&](int p0) {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < PM; ++i) {
            int q = tid + i * 256, row = q / CM, ch = q - row * CM;
            int p = p0 + row, c = mch0 + ch * EPC;
            bool ok = (PM * 256 == KP * CM || q < KP * CM) && p < p_end && c < a.Cout;
            u32x4 v = *(const u32x4*)(dyg + (ok ? ((size_t)p * a.lddy + c) : (size_t)0));
            mreg[i] = ok ? v : z;
        }
        bool okn[PN]; int cn[PN];
#pragma unroll
        for (int i = 0; i < PN; ++i) {
            int q = tid + i * 256, row = q / CN, ch = q - row * CN;
            int p = p0 + row, c = nch0 + ch * EPC;
            bool ok = (PN * 256 == KP * CN || q < KP * CN) && p < p_end && c < a.Cin;
            unsigned int pp = ok ? (unsigned)p : 0u;
            unsigned int n = a.dHoWo.div(pp), rem = pp - n * (a.Ho * a.Wo);
            unsigned int oh = a.dWo.div(rem), ow = rem - oh * a.Wo;
            int ih = (int)oh * a.stride - a.pad + kh, iw = (int)ow * a.stride - a.pad + kw;
            ok = ok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            okn[i] = ok; cn[i] = ok ? c : 0;
            nreg[i] = *(const u32x4*)(xg + (ok ? (((size_t)(n * a.H + ih) * a.W + iw) * a.ldx + c) : (size_t)0));
        }
        if (has_pro) {
#pragma unroll
            for (int i = 0; i < PN; ++i) {
                float f[EPC];
                Vec16<T>::unpack(nreg[i], f);
#pragma unroll
                for (int j = 0; j < EPC; j += 4) {
                    f32x4 s4 = *(const f32x4*)(a.pro_scale + cn[i] + j), t4 = *(const f32x4*)(a.pro_shift + cn[i] + j);
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) f[j + qq] = fmaxf(fmaf(f[j + qq], s4[qq], t4[qq]), relu_lo);
                }
                nreg[i] = Vec16<T>::pack(f);
            }
        }
#pragma unroll
        for (int i = 0; i < PN; ++i) nreg[i] = okn[i] ? nreg[i] : z;
    };
    auto store_tile = [&](int buf) {
        unsigned char* sm = smem + buf * STAGE;
        unsigned char* sn = sm + ST_M;
#pragma unroll
        for (int i = 0; i < PM; ++i) {
            int q = tid + i * 256;
            if (q < KP * CM) *(u32x4*)(sm + q * 16) = mreg[i];
        }
#pragma unroll
        for (int i = 0; i < PN; ++i) {
            int q = tid + i * 256;
            if (q < KP * CN) *(u32x4*)(sn + q * 16) = nreg[i];
        }
    };

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    const int nsteps = (p_end - p_begin + KP - 1) / KP;
    if (nsteps > 0) {
        load_tile(p_begin);
        store_tile(0);
    }
    __syncthreads();
    for (int ks = 0; ks < nsteps; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nsteps) load_tile(p_begin + (ks + 1) * KP);
        const unsigned char* sm = smem + buf * STAGE;
        const unsigned char* sn = sm + ST_M;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int wt = wave * TPW + t;
            if (wt < WMW * WNW) {
                const int wmo = (wt / WNW) * 32, wno = (wt % WNW) * 32;
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int k = 0; k < KP; k += 2) {
                        float av = *(const float*)(sm + (k + lh) * ROWB_M + (wmo + lr) * 4);
                        float bv = *(const float*)(sn + (k + lh) * ROWB_N + (wno + lr) * 4);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < KP; k += 16) {
                        u32x4 av, bv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            int k0 = k + 8 * lh + 2 * e;
                            unsigned int a0 = *(const u16*)(sm + k0 * ROWB_M + (wmo + lr) * 2);
                            unsigned int a1 = *(const u16*)(sm + (k0 + 1) * ROWB_M + (wmo + lr) * 2);
                            unsigned int b0 = *(const u16*)(sn + k0 * ROWB_N + (wno + lr) * 2);
                            unsigned int b1 = *(const u16*)(sn + (k0 + 1) * ROWB_N + (wno + lr) * 2);
                            av[e] = a0 | (a1 << 16); bv[e] = b0 | (b1 << 16);
                        }
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av),
                                                                         __builtin_bit_cast(bf16x8_t, bv), acc[t], 0, 0, 0);
                    }
                }
            }
        }
        if (ks + 1 < nsteps) store_tile(buf ^ 1);
        __syncthreads();
    }
    if (nsteps <= 0) return;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int wt = wave * TPW + t;
        if (wt < WMW * WNW) {
            const int wmo = (wt / WNW) * 32, wno = (wt % WNW) * 32;
            const int n = nch0 + wno + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = mch0 + wmo + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < a.Cout && n < a.Cin) atomicAdd(a.dw + (size_t)m * a.sM + (size_t)n * a.sN + tap, acc[t][r]);
            }
        }
    }
}

template <typename T, int BMc, int BNc> static int launch_wgrad(WgradArgs& a, hipStream_t st)
{
    const int ntm = cdiv(a.Cout, BMc);
    a.ntn = cdiv(a.Cin, BNc);
    const int taps = a.KH * a.KW;
    const long tiles = (long)ntm * a.ntn * taps;
    // aim for ~4 blocks per CU overall; keep each split a multiple of the 32-pixel K-step
    long want = (1024 + tiles - 1) / tiles;
    long max_splits = (a.P + 255) / 256;
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    int pps = (int)(((a.P + want - 1) / want + 31) / 32 * 32);
    a.pix_per_split = pps;
    int splits = cdiv(a.P, pps);
    dim3 grid(ntm * a.ntn, splits, taps);
    hipLaunchKernelGGL((conv_igemm_wgrad_kernel<T, BMc, BNc>), grid, dim3(256), 0, st, a);
    SAUNET_CHECK_LAUNCH("conv_igemm_wgrad");
    return SAUNET_OK;
}

template <typename T> static int dispatch_wgrad(WgradArgs& a, hipStream_t st)
{
    if (a.Cout <= 32) return launch_wgrad<T, 32, 128>(a, st);
    if (a.Cin <= 32) return launch_wgrad<T, 128, 32>(a, st);
    return launch_wgrad<T, 64, 64>(a, st);
}

int igemm_wgrad(const saunet_conv_desc* d, const void* x, const void* dy, const float* ps, const float* psh, float* dw,
                hipStream_t st)
{
    WgradArgs a;
    a.pro_scale = ps; a.pro_shift = psh; a.pro_relu = d->pro_relu; a.dw = dw;
    if (!d->transposed) {
        // dw[co][ci][kh][kw]: m = co (dy channel), n = ci (x channel)
        a.x = x; a.dy = dy;
        a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
        a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.lddy = d->ldy;
        a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
        a.sM = (long)d->Cin * d->KH * d->KW; a.sN = (long)d->KH * d->KW;
    } else {
        // ConvTranspose2d(k4,s2,p1): dw[ci][co][kh][kw] = sum x[n,ih,iw,ci] * dy[n,2ih-1+kh,2iw-1+kw,co]
        // = wgrad of a stride-2 pad-1 conv whose "input" is dy (2x res) and whose "output gradient" is x.
        if (ps != nullptr) return set_error(SAUNET_UNSUPPORTED, "transposed wgrad has no prologue on dy");
        a.x = dy; a.dy = x;
        a.N = d->N; a.H = d->Ho; a.W = d->Wo; a.Cin = d->Cout; a.ldx = d->ldy;
        a.Ho = d->H; a.Wo = d->W; a.Cout = d->Cin; a.lddy = d->ldx;
        a.KH = 4; a.KW = 4; a.stride = 2; a.pad = 1;
        a.sM = (long)d->Cout * 16; a.sN = 16;
    }
    a.P = a.N * a.Ho * a.Wo;
    a.dHoWo = FastDiv::make(a.Ho * a.Wo); a.dWo = FastDiv::make(a.Wo);
    if (((uintptr_t)x | (uintptr_t)dy) & 15) return set_error(SAUNET_BAD_ALIGN, "wgrad: pointers must be 16-byte aligned");
    if (d->dtype == SAUNET_BF16) return dispatch_wgrad<u16>(a, st);
    if (d->dtype == SAUNET_F32) return dispatch_wgrad<float>(a, st);
    return set_error(SAUNET_BAD_DTYPE, "wgrad: dtype %d", d->dtype);
}

}  // namespace saunet

SAUNET_TIMING_READER(conv_igemm)
