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
#ifndef SAUNET_DF_RING
#define SAUNET_DF_RING 4
#endif
constexpr int DF_RING = SAUNET_DF_RING;  // requests run DF_RING - 1 stages ahead of the MFMAs (two were not enough with ONE workgroup per CU: ~450 cycles of vmcnt wait per stage)
constexpr int DF_AHEAD = DF_RING - 1;
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
    if (DF_AHEAD > 2 && a.nk > 2) issue(2);
    bn_prologue_fill<NT>(a.bnp, a.Cin, cpad, s_pro, blockIdx.x == 0);
    __syncthreads();
    TSTAMP(81);
    if (DF_AHEAD > 2 && a.nk > 2) mm_wait_vm<2 * PER_STAGE>(); else if (a.nk > 1) mm_wait_vm<PER_STAGE>(); else mm_wait_vm<0>();
    transform(0);
    mm_barrier();
    TSTAMP(82);
    for (int k = 0; k < a.nk; ++k) {
        if (k + DF_AHEAD < a.nk) issue(k + DF_AHEAD);
        TSTAMP(83);
        if (k + 1 < a.nk) {
            // stage k+1 was requested DF_AHEAD stages ago; the requests of the stages behind it may still be in flight
            const int ahead = min(a.nk - 1, k + DF_AHEAD) - (k + 1);         // stages requested after k+1 (wave-uniform)
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
    static const long maxtiles = ab_env_int("SAUNET_DENSE_CONV1_MAXTILES", 384);
    const long P = (long)d->N * d->H * d->W;
    return on && d->dtype == SAUNET_BF16 && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && !d->transposed && d->Cout == DF_BN &&
           d->Cin % 32 == 0 && d->Cin >= 128 && d->Cin <= 4096 && d->ldx % 8 == 0 && d->ldy % 8 == 0 && d->ldx >= ((d->Cin + 63) & ~63) &&
           bias == nullptr && !d->epi_relu && (P + 127) / 128 < maxtiles && P >= 64 && P < (1L << 30) &&
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


// =====================================================================================================================================
// DenseNet conv2 (3x3, 128 -> 32, pad 1) forward with the BatchNorm + ReLU prologue of norm2 on the LOW-RESOLUTION blocks (torchvision
// _DenseLayer.norm2 / relu2 / conv2, /root/reference/models/models.py:306-313).  The resident 3x3 kernel (conv_tile.hip) gives a 16 x 16 map ONE
// workgroup per image (32 of 256 CUs at block 4) and spends 7k of its 18k cycles copying the 72 KB of weights through registers before the first
// halo arrives.  Here a workgroup owns an 8 x 8 pixel tile (128 workgroups at block 4, 512 at block 3) and requests EVERYTHING it will ever read
// at kernel start by LDS-DMA -- the 10 x 10 x 128-channel halo (25 KB) and all 32 x 9 x 128 weights (72 KB) -- derives the BatchNorm coefficients
// while the requests are in flight, applies BN + ReLU (and the zero padding, which belongs to the ACTIVATED tensor) in place to the pieces it
// requested itself, and then runs the whole K = 1152 product in one go: wave (m, q) = pixel half m x quarter q of the 72 k-steps, 18 MFMAs each,
// partial accumulators summed through the LDS.  Two barriers per workgroup.
struct DenseConv2Args {
    const u16* z; int ldz; const u16* w; u16* y; int ldy;
    double* stat_sum; double* stat_sumsq; int stat_replicas, stat_rstride;
    int N, H, W, tiles_x, tiles_y;
    saunet_bn_prologue bnp;
};
constexpr int C2_HP = 10;                                       // halo pixels per row of an 8-wide tile
constexpr int C2_W_PIECES = 32 * 144 / 64;                      // 72 requests: 32 rows x 144 chunks of 16 B
template <int TH> struct C2Layout {                             // TH = tile rows (8 or 16), 8 columns
    static constexpr int PIX = TH * 8, MT = TH / 4, KQ = 8 / MT, KSTEPS = 72 / KQ;      // 32-pixel MFMA tiles, K parts per tile, k-steps per wave
    static constexpr int HALO = (TH + 2) * C2_HP, HALO_PIECES = (HALO * 16 + 63) / 64, HPW = (HALO_PIECES + 7) / 8;
    static constexpr int OFF_W = HALO_PIECES * 1024;
    static constexpr int OFF_PRO = OFF_W + C2_W_PIECES * 1024;  // float[2][128]
    static constexpr int OFF_RED = OFF_PRO + 1024;              // float[KQ][PIX][32] = 32 KB
    static constexpr int OFF_SUM = OFF_RED + KQ * PIX * 32 * 4; // float[8 waves][2][32]
    static constexpr int LDS = OFF_SUM + 8 * 2 * 32 * 4;
};
static __device__ u32x4 g_c2_zeros[4];

template <int TH>
__global__ __launch_bounds__(512) void dense_conv2_fwd_kernel(DenseConv2Args a)
{
    using LY = C2Layout<TH>;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    float* s_pro = (float*)(smem + LY::OFF_PRO);
    const int t = blockIdx.x, txi = t % a.tiles_x, r1 = t / a.tiles_x, tyi = r1 % a.tiles_y, n = r1 / a.tiles_y;
    const int y0 = tyi * TH - 1, x0 = txi * 8 - 1;
    const u16* img = a.z + (size_t)n * a.H * a.W * a.ldz;

    // ---- requests.  Halo piece p covers halo pixels 4p .. 4p+3; lane l delivers slot (pixel 4p + (l >> 4), slot l & 15), which holds logical
    // chunk  slot ^ key(pixel),  key = (hx & 3) | (hy & 3) << 2: the 16 lanes of a ds_read_b128 group read 4 consecutive pixels of 4 consecutive
    // rows (tile rows are 8 wide) and so cover all 16 bank groups.  Out-of-image pixels source a zero page (and are zeroed again after the prologue).
    int hchunk[LY::HPW]; bool hin[LY::HPW];
#pragma unroll
    for (int j = 0; j < LY::HPW; ++j) {
        const int piece = wave + 8 * j;
        const int hp = piece * 4 + (lane >> 4), hy = hp / C2_HP, hx = hp - hy * C2_HP;
        const int c = (lane & 15) ^ ((hx & 3) | ((hy & 3) << 2));
        const int iy = y0 + hy, ix = x0 + hx;
        hin[j] = piece < LY::HALO_PIECES && hp < LY::HALO && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        hchunk[j] = c;
        if (piece < LY::HALO_PIECES)
            mm_dma16(hin[j] ? (const void*)(img + ((size_t)iy * a.W + ix) * a.ldz + c * 8) : (const void*)&g_c2_zeros[lane & 3], lds0 + piece * 1024);
    }
    // weights: row co = 1152 contiguous elements ([tap][ci]) = 144 chunks; LDS image [32][144] chunks, lane-linear per request; slot s of row r holds
    // logical chunk  s ^ (r & 15)  (XOR inside aligned groups of 16 chunks: rows are 9 x 256 B apart and would all hit the same bank group)
#pragma unroll
    for (int j = 0; j < C2_W_PIECES / 8; ++j) {
        const int q = (wave + 8 * j) * 64 + lane, row = q / 144, sl = q - row * 144;
        const int c = (sl & ~15) | ((sl & 15) ^ (row & 15));
        mm_dma16(a.w + (size_t)row * 1152 + c * 8, lds0 + LY::OFF_W + (wave + 8 * j) * 1024);
    }
    bn_prologue_fill<512>(a.bnp, 128, 128, s_pro, blockIdx.x == 0);
    __syncthreads();
    // ---- BN + ReLU in place on this wave's own halo pieces (the weights may still be in flight: C2_W_PIECES / 8 = 9 requests behind them)
    mm_wait_vm<C2_W_PIECES / 8>();
#pragma unroll
    for (int j = 0; j < LY::HPW; ++j) {
        const int piece = wave + 8 * j;
        if (piece < LY::HALO_PIECES) {
            unsigned char* q = smem + piece * 1024 + lane * 16;
            float f[8];
            Vec16<u16>::unpack(*(const u32x4*)q, f);
            const int c0 = hchunk[j] * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 sc = *(const f32x4*)(s_pro + c0 + 4 * h), sh = *(const f32x4*)(s_pro + 128 + c0 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) f[4 * h + e] = hin[j] ? fmaxf(fmaf(f[4 * h + e], sc[e], sh[e]), 0.f) : 0.f;
            }
            *(u32x4*)q = Vec16<u16>::pack(f);
        }
    }
    mm_wait_vm<0>();
    mm_barrier();
    // ---- the product: wave (m, q): pixels 32m .. 32m+31 of the tile (4 rows of 8), k-steps KSTEPS*q .. of the 72 (k-step = tap * 8 + 16-channel group)
    const int wm = wave % LY::MT, wq = wave / LY::MT;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        const int py = wm * 4 + (lr >> 3), px = lr & 7;
        const unsigned char* sw = smem + LY::OFF_W + lr * (144 * 16);
#pragma unroll
        for (int i = 0; i < LY::KSTEPS; ++i) {
            const int ks = wq * LY::KSTEPS + i, tap = ks >> 3, cg = ks & 7, kh = tap / 3, kw = tap - kh * 3;
            const int hy = py + kh, hx = px + kw, hp = hy * C2_HP + hx;
            const u32x4 af = *(const u32x4*)(smem + hp * 256 + (((2 * cg + lh) ^ ((hx & 3) | ((hy & 3) << 2))) << 4));
            const int bc = ks * 2 + lh;
            const u32x4 bf = *(const u32x4*)(sw + (((bc & ~15) | ((bc & 15) ^ (lr & 15))) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bf), acc, 0, 0, 0);
        }
    }
    // ---- sum the K parts through the LDS: partial [q][pixel][channel] float
    float* s_red = (float*)(smem + LY::OFF_RED);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        s_red[(wq * LY::PIX + row) * 32 + lr] = acc[r];
    }
    __syncthreads();
    // thread -> (pixel, 4 consecutive channels): PIX x 8 items, PIX / 64 per thread; a wave's 64 threads cover 8 pixels x 8 channel groups
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int c4 = (tid & 7) * 4;
#pragma unroll
    for (int it = 0; it < LY::PIX / 64; ++it) {
        const int prow = (tid >> 3) + it * 64;
        f32x4 v = *(const f32x4*)(s_red + prow * 32 + c4);
#pragma unroll
        for (int qq = 1; qq < LY::KQ; ++qq) { const f32x4 u = *(const f32x4*)(s_red + (qq * LY::PIX + prow) * 32 + c4); v += u; }
        const int oy = tyi * TH + (prow >> 3), ox = txi * 8 + (prow & 7);
        u16* yo = a.y + ((size_t)(n * a.H + oy) * a.W + ox) * a.ldy + c4;
        *(uint2*)yo = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
    }
    if (a.stat_sum != nullptr) {
        // per-channel sum / sum of squares over the tile: lanes of equal (tid & 7) hold the same channels -> xor-shuffle over bits 3..5, one slot
        // per wave, fixed-order fold
#pragma unroll
        for (int o = 8; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
        float* s_sum = (float*)(smem + LY::OFF_SUM);
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { s_sum[(wave * 2) * 32 + c4 + e] = s1[e]; s_sum[(wave * 2 + 1) * 32 + c4 + e] = s2[e]; }
        }
        __syncthreads();
        if (tid < 32) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) { t1 += s_sum[(w8 * 2) * 32 + tid]; t2 += s_sum[(w8 * 2 + 1) * 32 + tid]; }
            const size_t ro = (size_t)(blockIdx.x % a.stat_replicas) * a.stat_rstride;
            atomicAdd(&a.stat_sum[ro + tid], (double)t1);
            atomicAdd(&a.stat_sumsq[ro + tid], (double)t2);
        }
    }
}

bool dense_conv2_small_supported(const saunet_conv_desc* d, const void* x, const void* w, const void* y, const float* bias)
{
    static const bool on = ab_env_on("SAUNET_DENSE_CONV2_SMALL");       // A/B switch (variant builds only)
    const long tiles = (long)d->N * (d->H / 8) * (d->W / 8);
    return on && d->dtype == SAUNET_BF16 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && !d->transposed && d->Cin == 128 && d->Cout == 32 &&
           d->ldx == 128 && d->ldy % 4 == 0 && d->H % 8 == 0 && d->W % 8 == 0 && d->Ho == d->H && d->Wo == d->W && bias == nullptr && !d->epi_relu &&
           d->pro_relu && tiles >= 1 && tiles <= 1024 && !(((uintptr_t)x | (uintptr_t)w) & 15) && !((uintptr_t)y & 7) && (long)d->N * d->H * d->W < (1L << 30);
}

int dense_conv2_small_forward(const saunet_conv_desc* d, const void* x, const void* w, void* y, double* ssum, double* ssq, const saunet_bn_prologue* bnp,
                              hipStream_t st)
{
    DenseConv2Args a;
    a.z = (const u16*)x; a.ldz = d->ldx; a.w = (const u16*)w; a.y = (u16*)y; a.ldy = d->ldy;
    a.stat_sum = ssum; a.stat_sumsq = ssq; a.stat_replicas = d->stat_replicas > 1 ? d->stat_replicas : 1; a.stat_rstride = d->stat_rstride;
    a.N = d->N; a.H = d->H; a.W = d->W; a.tiles_x = d->W / 8; a.tiles_y = d->H / 8;
    a.bnp = *bnp;
    // 16 x 8 tiles while they still give every CU a workgroup (block 3 at B = 32: 256), 8 x 8 tiles below that
    static const int force_th = ab_env_int("SAUNET_DENSE_CONV2_TH", 0);       // A/B switch (variant builds only)
    const bool tall = force_th ? force_th == 16 && d->H % 16 == 0 : (d->H % 16 == 0 && (long)d->N * (d->H / 16) * (d->W / 8) >= 256);
    static DeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)dense_conv2_fwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_conv2_fwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (tall) {
        a.tiles_y = d->H / 16;
        hipLaunchKernelGGL(dense_conv2_fwd_kernel<16>, dim3(a.N * a.tiles_x * a.tiles_y), dim3(512), C2Layout<16>::LDS, st, a);
        SAUNET_CHECK_LAUNCH("dense_conv2_fwd_kernel<16>");
    } else {
        hipLaunchKernelGGL(dense_conv2_fwd_kernel<8>, dim3(a.N * a.tiles_x * a.tiles_y), dim3(512), C2Layout<8>::LDS, st, a);
        SAUNET_CHECK_LAUNCH("dense_conv2_fwd_kernel<8>");
    }
    return SAUNET_OK;
}

}  // namespace saunet

SAUNET_TIMING_READER(dense_fwd)
