// DenseNet conv1 (1x1, Cin -> 128) forward with the consumer-side BatchNorm + ReLU prologue, for the LOW-RESOLUTION blocks (3 and 4: 40 of the
// 58 layers; 32 x 32 and 16 x 16 maps at 256 x 256 input).  torchvision _DenseLayer.norm1 / relu1 / conv1 as used at
// /root/reference/models/models.py:306-313.
//
// Why a second kernel next to conv_igemm_fwd_kernel (round 5, phase stamps in profiles/r05_phase_timing_raw.txt): on these maps the generic
// implicit GEMM runs 64 x 64 tiles of four waves whose K loop costs ~2000 cycles per 64-channel step for 4 MFMAs per wave (issue -> register
// stage -> BN+ReLU -> LDS -> barrier, two tiles per CU), reads the activation tile twice (once per 64-channel half of the 128 outputs) and
// stages the weights through registers: 17.8 us (block 4) / 23.8 us (block 3) per launch for 2-5 us of work.  Here
//   * one 8-wave workgroup owns BM pixels x ALL 128 output channels (activations are read once);
//   * BOTH operands reach the LDS by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction, no registers, no address VALU in the loop)
//     into rings of four 64-channel stages; the 16-byte XOR swizzle of the fragment reads is applied to the per-lane SOURCE address;
//   * the BatchNorm + ReLU prologue is applied IN PLACE in the LDS by the wave that requested the piece (a lane reads back exactly the 16 bytes
//     its own request delivered: a counted vmcnt suffices, no barrier) -- one ds_read_b128 / 8 FMA+max / ds_write_b128 per 16 bytes;
//   * one s_barrier per stage; requests run three stages ahead of the MFMAs.
// Epilogue as conv_igemm_fwd_kernel: per-channel sum / sum of squares of the accumulator (the next BatchNorm's statistics, float64 atomics
// into replicated accumulators), tile transposed through LDS, 16-byte row stores.
#include "common.h"

namespace saunet {

struct DenseFwdArgs {
    const u16* x; int ldx; const u16* w; u16* y; int ldy;
    double* stat_sum; double* stat_sumsq; int stat_replicas, stat_rstride;
    int P, Cin, nk;               // nk = ceil(Cin / 64)
    saunet_bn_prologue bnp;
};

constexpr int DF_KC = 64;          // channels per stage
constexpr int DF_RING = 4;            // requests run three stages ahead of the MFMAs (two were not enough: ~450 cycles of vmcnt wait per stage)
constexpr int DF_BN = 128;         // output channels (bn_size * growth)

template <int BM> struct DfLayout {
    static constexpr int A_STAGE = BM * DF_KC * 2, B_STAGE = DF_BN * DF_KC * 2;
    static constexpr int A_PIECES = A_STAGE / 1024 / 8, B_PIECES = B_STAGE / 1024 / 8;          // DMA instructions per wave and stage
    static constexpr int OFF_B = DF_RING * A_STAGE;
    static constexpr int OFF_PRO = OFF_B + DF_RING * B_STAGE;                                   // float[2][cpad]
};

// byte offset of 16-byte chunk `c` (0..7) of row `r` in a [rows][8] chunk image (128-byte rows): same swizzle as lds_off<8> of conv_igemm.hip
__device__ __forceinline__ int df_off(int r, int c) { return (r * 8 + (c ^ ((r >> 1) & 7))) * 16; }

template <int BM>
__global__ __launch_bounds__(512) void dense_conv1_fwd_kernel(DenseFwdArgs a)
{
    using LY = DfLayout<BM>;
    constexpr int NT = 512;
    constexpr int TI = BM / 64;                 // 32-row MFMA tiles per wave: waves are (BM / (32 * TI)) = 2 row groups x 4 column groups
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int m0 = blockIdx.x * BM;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    float* s_pro = (float*)(smem + LY::OFF_PRO);
    const int cpad = a.nk * DF_KC;

    // ---- per-lane DMA sources.  A piece covers 8 rows x 128 B: lane l delivers LDS slot (row = 8 * piece + (l >> 3), slot = l & 7), which holds
    // logical chunk  slot ^ ((row >> 1) & 7)  of that row.  Rows past P re-read the last pixel (their outputs are never stored and are masked
    // out of the statistics); the last stage of a Cin % 64 == 32 layer reads 32 channels past Cin -- activations of the next concat slice /
    // the next weight row, finite values that the zero prologue coefficients (scale = shift = 0 past Cin) turn into exact zeros of A.
    const unsigned char* asrc[LY::A_PIECES];
    int achunk[LY::A_PIECES];
#pragma unroll
    for (int j = 0; j < LY::A_PIECES; ++j) {
        const int piece = wave * LY::A_PIECES + j, row = piece * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        const int m = min(m0 + row, a.P - 1);
        asrc[j] = (const unsigned char*)(a.x + (size_t)m * a.ldx + c * 8);
        achunk[j] = c;
    }
    const unsigned char* bsrc[LY::B_PIECES];
    const unsigned char* const wend = (const unsigned char*)(a.w + (size_t)DF_BN * a.Cin) - 16;
#pragma unroll
    for (int j = 0; j < LY::B_PIECES; ++j) {
        const int piece = wave * LY::B_PIECES + j, row = piece * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        bsrc[j] = (const unsigned char*)(a.w + (size_t)row * a.Cin + c * 8);
    }
    auto issue = [&](int k) {
        const int st = k % DF_RING;
#pragma unroll
        for (int j = 0; j < LY::A_PIECES; ++j) mm_dma16(asrc[j] + (size_t)k * (DF_KC * 2), lds0 + st * LY::A_STAGE + (wave * LY::A_PIECES + j) * 1024);
#pragma unroll
        for (int j = 0; j < LY::B_PIECES; ++j) {
            const unsigned char* s = bsrc[j] + (size_t)k * (DF_KC * 2);
            mm_dma16(s < wend ? s : wend, lds0 + LY::OFF_B + st * LY::B_STAGE + (wave * LY::B_PIECES + j) * 1024);
        }
    };
    constexpr int PER_STAGE = LY::A_PIECES + LY::B_PIECES;
    // BatchNorm + ReLU on this wave's own pieces of stage k, in place (the requests of stage k have landed: counted wait by the caller)
    auto transform = [&](int k) {
        unsigned char* sa = smem + (k % DF_RING) * LY::A_STAGE;
#pragma unroll
        for (int j = 0; j < LY::A_PIECES; ++j) {
            unsigned char* q = sa + (wave * LY::A_PIECES + j) * 1024 + lane * 16;
            const int c0 = k * DF_KC + achunk[j] * 8;
            float f[8];
            Vec16<u16>::unpack(*(const u32x4*)q, f);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 sc = *(const f32x4*)(s_pro + c0 + 4 * h), sh = *(const f32x4*)(s_pro + cpad + c0 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) f[4 * h + e] = fmaxf(fmaf(f[4 * h + e], sc[e], sh[e]), 0.f);
            }
            *(u32x4*)q = Vec16<u16>::pack(f);
        }
    };

    f32x16 acc[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    auto compute = [&](int k) {
        const unsigned char* sa = smem + (k % DF_RING) * LY::A_STAGE;
        const unsigned char* sb = smem + LY::OFF_B + (k % DF_RING) * LY::B_STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const u32x4 bf = *(const u32x4*)(sb + df_off(wn * 32 + lr, 2 * s + lh));
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const u32x4 af = *(const u32x4*)(sa + df_off(wm * (32 * TI) + i * 32 + lr, 2 * s + lh));
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bf), acc[i], 0, 0, 0);
            }
        }
    };

    // ---- prologue: the first two stages are requested before the BatchNorm coefficients are derived (their round trips overlap)
    TSTAMP_INIT();
    TSTAMP(80);
    issue(0);
    if (a.nk > 1) issue(1);
    if (a.nk > 2) issue(2);
    bn_prologue_fill<NT>(a.bnp, a.Cin, cpad, s_pro, blockIdx.x == 0);
    __syncthreads();
    TSTAMP(81);
    if (a.nk > 2) mm_wait_vm<2 * PER_STAGE>(); else if (a.nk > 1) mm_wait_vm<PER_STAGE>(); else mm_wait_vm<0>();
    transform(0);
    mm_barrier();
    TSTAMP(82);
    for (int k = 0; k < a.nk; ++k) {
        if (k + 3 < a.nk) issue(k + 3);
        TSTAMP(83);
        if (k + 1 < a.nk) {
            // stage k+1 was requested three stages ago; the requests of stages k+2 and k+3 may still be in flight
            const int ahead = min(a.nk - 1, k + 3) - (k + 1);         // stages requested after k+1 (wave-uniform)
            if (ahead >= 2) mm_wait_vm<2 * PER_STAGE>(); else if (ahead == 1) mm_wait_vm<PER_STAGE>(); else mm_wait_vm<0>();
            TSTAMP(84);
            transform(k + 1);
        }
        TSTAMP(85);
        compute(k);
        TSTAMP(86);
        mm_barrier();
        TSTAMP(87);
    }

    // ---- epilogue: statistics, transpose through LDS, coalesced stores (the rings are idle now)
    u16* so = (u16*)smem;                                   // [BM][128]
    float* s_sum = (float*)(smem + BM * DF_BN * 2);         // [BM / 32 row tiles][2][128]
    const bool do_stats = a.stat_sum != nullptr;
    {
        const int col = wn * 32 + lr;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            float sv = 0.f, ssv = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (32 * TI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float v = (m0 + row < a.P) ? acc[i][r] : 0.f;
                sv += v; ssv += v * v;
                Elem<u16>::store(so + row * DF_BN + col, v);
            }
            if (do_stats) {
                sv += __shfl_xor(sv, 32, 64); ssv += __shfl_xor(ssv, 32, 64);
                if (lh == 0) { float* slot = s_sum + (wm * TI + i) * 2 * DF_BN; slot[col] = sv; slot[DF_BN + col] = ssv; }
            }
        }
    }
    TSTAMP(88);
    __syncthreads();
    if (do_stats && tid < DF_BN) {
        // float partial sums cover 64 rows (two 32-row tiles) exactly like the generic kernel's 64 x 64 tiles; 64-row groups are combined in double
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int g = 0; g < BM / 64; ++g) {
            t1 += (double)(s_sum[(2 * g) * 2 * DF_BN + tid] + s_sum[(2 * g + 1) * 2 * DF_BN + tid]);
            t2 += (double)(s_sum[(2 * g) * 2 * DF_BN + DF_BN + tid] + s_sum[(2 * g + 1) * 2 * DF_BN + DF_BN + tid]);
        }
        const size_t ro = (size_t)(blockIdx.x % a.stat_replicas) * a.stat_rstride;
        atomicAdd(&a.stat_sum[ro + tid], t1);
        atomicAdd(&a.stat_sumsq[ro + tid], t2);
    }
    constexpr int CH = DF_BN / 8;                           // 16 chunks per output row
#pragma unroll
    for (int i = 0; i < BM * CH / NT; ++i) {
        const int p = tid + i * NT, row = p / CH, ch = p - row * CH;
        if (m0 + row < a.P) *(u32x4*)(a.y + (size_t)(m0 + row) * a.ldy + ch * 8) = *(const u32x4*)(so + row * DF_BN + ch * 8);
    }
    TSTAMP(89);
}

// the low-resolution DenseNet conv1 geometry: bf16, 1x1, 128 outputs, long K, few pixels (fewer than 384 tiles of 128 x 128: where the generic
// implicit GEMM falls back to 64 x 64 tiles)
bool dense_conv1_small_supported(const saunet_conv_desc* d, const void* x, const void* w, const void* y, const float* bias)
{
    static const bool on = ab_env_on("SAUNET_DENSE_CONV1_SMALL");       // A/B switch (variant builds only)
    const long P = (long)d->N * d->H * d->W;
    return on && d->dtype == SAUNET_BF16 && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && !d->transposed && d->Cout == DF_BN &&
           d->Cin % 32 == 0 && d->Cin >= 128 && d->Cin <= 4096 && d->ldx % 8 == 0 && d->ldy % 8 == 0 && d->ldx >= ((d->Cin + 63) & ~63) &&
           bias == nullptr && !d->epi_relu && (P + 127) / 128 < 384 && P >= 64 && P < (1L << 30) &&
           !(((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15);
}

int dense_conv1_small_forward(const saunet_conv_desc* d, const void* x, const void* w, void* y, double* ssum, double* ssq, const saunet_bn_prologue* bnp,
                              hipStream_t st)
{
    DenseFwdArgs a;
    a.x = (const u16*)x; a.ldx = d->ldx; a.w = (const u16*)w; a.y = (u16*)y; a.ldy = d->ldy;
    a.stat_sum = ssum; a.stat_sumsq = ssq; a.stat_replicas = d->stat_replicas > 1 ? d->stat_replicas : 1; a.stat_rstride = d->stat_rstride;
    a.P = d->N * d->H * d->W; a.Cin = d->Cin; a.nk = (d->Cin + DF_KC - 1) / DF_KC;
    a.bnp = *bnp;
    // 128-pixel tiles while they still give every CU a workgroup, 64-pixel tiles below that
    static const int force_bm = ab_env_int("SAUNET_DENSE_CONV1_BM", 0);       // A/B switch (variant builds only)
    const bool big = force_bm ? force_bm == 128 : a.P >= 128 * 256;
    const int pro_bytes = 2 * a.nk * DF_KC * 4;
    if (big) {
        const int lds = DfLayout<128>::OFF_PRO + pro_bytes;
        static DeviceOnce attr;
        if (attr.first()) (void)hipFuncSetAttribute((const void*)dense_conv1_fwd_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(dense_conv1_fwd_kernel<128>, dim3((a.P + 127) / 128), dim3(512), lds, st, a);
        SAUNET_CHECK_LAUNCH("dense_conv1_fwd_kernel<128>");
    } else {
        const int lds = DfLayout<64>::OFF_PRO + pro_bytes;
        static DeviceOnce attr;
        if (attr.first()) (void)hipFuncSetAttribute((const void*)dense_conv1_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(dense_conv1_fwd_kernel<64>, dim3((a.P + 63) / 64), dim3(512), lds, st, a);
        SAUNET_CHECK_LAUNCH("dense_conv1_fwd_kernel<64>");
    }
    return SAUNET_OK;
}

}  // namespace saunet

SAUNET_TIMING_READER(dense_fwd)
