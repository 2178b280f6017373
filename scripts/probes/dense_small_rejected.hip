// REJECTED (round 4, kept as a probe; not built into the library).  Measured at the bench geometry: block 3 conv1 forward 48.9 us against 23.6 us of
// conv_igemm_fwd, block 4 18.0 against 18.7; at blocks 1 / 2 2.3x slower.  Why: a fragment-shaped request (32 rows x 32 bytes per wave
// instruction) occupies the CU's texture-address unit for 32 cache lines whatever it uses of them; with BOTH operands fetched that way the
// kernel is bound by request processing, not by latency.  Full-line staging through the LDS is the right shape for the x operand.
// DenseNet layers on the LOW-RESOLUTION maps (dense blocks 3 and 4: 32 x 32 and 16 x 16 at the bench geometry; torchvision _DenseLayer as used
// at /root/reference/models/models.py:306-313).  A launch there has 8 192 - 32 768 pixels: too few 64 x 64 tiles to hide the serial chain
// global -> LDS -> barrier -> MFMA of the tiled kernels (conv1 forward 17-23 us against 2-5 us of data movement; phase stamps: a workgroup
// of conv_igemm_fwd lives 29k cycles of which 5.3k are the BatchNorm prologue fill and the rest five dependent load -> commit -> MFMA rounds).
//
// dense_conv1_small_kernel -- z1 = conv1x1(relu(bn1(x)), W1) -> 128 channels, with the statistics of z1 for norm2:
//   * tile 64 pixels x 64 output channels, FOUR waves that split K (the input channels): every wave owns a quarter of the K steps and the
//     whole tile, so its operand requests are independent of each other -- it issues up to eight K steps (32 16-byte loads per lane) before
//     it uses the first, ONE memory round trip instead of one per step;
//   * no LDS staging and no barrier in the K loop: x pieces (a lane = one pixel row, 8 consecutive channels) and weight pieces (a lane = one
//     output channel, 8 consecutive input channels) are MFMA fragments exactly as they lie in memory; BN+ReLU is applied to the x pieces in
//     registers with coefficients from an LDS copy that is built (consumer-side BatchNorm finalize, common.h bn_prologue_fill) BEHIND the
//     first batch of requests;
//   * the four partial tiles meet in LDS once, all 256 threads fold them in a fixed order, take the per-channel sums (per-wave slots, fixed
//     order, one float64 atomic per channel and workgroup) and store 16-byte row pieces.
#include "common.h"
#include <stdlib.h>

namespace saunet {

struct DsArgs {
    const u16* x; const u16* w; u16* y;
    const float* pro_scale; const float* pro_shift;
    double* stat_sum; double* stat_sumsq; int stat_replicas, stat_rstride;
    long P; int Cin, ldx, Cout, ldy, pro_relu;
    saunet_bn_prologue bnp;      // bnp.gamma != nullptr: derive the coefficients here
};

constexpr int DS_PITCH = 68;     // floats per row of a partial tile in LDS (64 + 4: rows shift by 16 B over the banks)
constexpr int DS_NB = 8;         // K steps (of 16 channels) requested per batch

__global__ __launch_bounds__(256, 2) void dense_conv1_small_kernel(DsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int cpad = (a.Cin + 15) & ~15;
    float* s_pro = (float*)smem;                               // [2][cpad]
    float* s_part = (float*)(smem + 2 * cpad * 4);             // [4][64][DS_PITCH]
    float* s_st = s_part + 4 * 64 * DS_PITCH;                  // [4 waves][2][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const long p0 = (long)blockIdx.x * 64;
    const int co0 = blockIdx.y * 64;
    const int nks = a.Cin >> 4;                                // K steps of 16 channels
    const int k_lo = (nks * wave) >> 2, k_hi = (nks * (wave + 1)) >> 2;
    const float relu_lo = a.pro_relu ? 0.f : -__builtin_inff();

    // lane bases: x rows p0 + lr (+32), weight rows co0 + lr (+32); channel offset 8 * lh inside a K step
    const u16* xa = a.x + (p0 + lr) * a.ldx + 8 * lh;
    const u16* xb = xa + 32L * a.ldx;
    int r0 = co0 + lr, r1 = co0 + 32 + lr;
    if (r0 >= a.Cout) r0 = a.Cout - 1;
    if (r1 >= a.Cout) r1 = a.Cout - 1;
    const u16* wa = a.w + (long)r0 * a.Cin + 8 * lh;
    const u16* wb = a.w + (long)r1 * a.Cin + 8 * lh;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 fa[DS_NB], fb[DS_NB], ga[DS_NB], gb[DS_NB];
    auto request = [&](int kb) {                               // K steps kb .. kb + 7 (clamped: every load is unconditional)
#pragma unroll
        for (int j = 0; j < DS_NB; ++j) {
            int ks = kb + j; if (ks >= k_hi) ks = k_hi - 1;
            const int c = ks << 4;
            fa[j] = *(const u32x4*)(xa + c); fb[j] = *(const u32x4*)(xb + c);
            ga[j] = *(const u32x4*)(wa + c); gb[j] = *(const u32x4*)(wb + c);
        }
    };
    request(k_lo);
    // coefficients behind the first requests
    if (a.bnp.gamma != nullptr) bn_prologue_fill<256>(a.bnp, a.Cin, cpad, s_pro, blockIdx.x == 0 && blockIdx.y == 0);
    else for (int i = tid; i < cpad; i += 256) { s_pro[i] = i < a.Cin ? a.pro_scale[i] : 0.f; s_pro[cpad + i] = i < a.Cin ? a.pro_shift[i] : 0.f; }
    __syncthreads();

    for (int kb = k_lo; kb < k_hi; kb += DS_NB) {
#pragma unroll
        for (int j = 0; j < DS_NB; ++j) {
            const int ks = kb + j;
            if (ks < k_hi) {                                   // wave-uniform
                const int c = (ks << 4) + 8 * lh;
                const f32x4 s0 = *(const f32x4*)(s_pro + c), s1 = *(const f32x4*)(s_pro + c + 4);
                const f32x4 t0 = *(const f32x4*)(s_pro + cpad + c), t1 = *(const f32x4*)(s_pro + cpad + c + 4);
                float f[8], g[8];
                Vec16<u16>::unpack(fa[j], f); Vec16<u16>::unpack(fb[j], g);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f[q] = fmaxf(fmaf(f[q], s0[q], t0[q]), relu_lo); f[q + 4] = fmaxf(fmaf(f[q + 4], s1[q], t1[q]), relu_lo);
                    g[q] = fmaxf(fmaf(g[q], s0[q], t0[q]), relu_lo); g[q + 4] = fmaxf(fmaf(g[q + 4], s1[q], t1[q]), relu_lo);
                }
                const u32x4 A0 = Vec16<u16>::pack(f), A1 = Vec16<u16>::pack(g);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A0), __builtin_bit_cast(bf16x8_t, ga[j]), acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A0), __builtin_bit_cast(bf16x8_t, gb[j]), acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A1), __builtin_bit_cast(bf16x8_t, ga[j]), acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A1), __builtin_bit_cast(bf16x8_t, gb[j]), acc[1][1], 0, 0, 0);
            }
        }
        if (kb + DS_NB < k_hi) request(kb + DS_NB);
    }

    // ---- the four K partials meet in LDS: [wave][pixel row][channel]
    float* mine = s_part + wave * 64 * DS_PITCH;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mine[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * DS_PITCH + j * 32 + lr] = acc[i][j][r];
    __syncthreads();
    // thread = (row pair, 8-channel group): rows rw and rw + 32, channels 8 * cg .. + 7
    const int cg = tid & 7, rw = tid >> 3;
    const bool do_stats = a.stat_sum != nullptr;
    float cs[8], cq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) cs[q] = cq[q] = 0.f;
    const bool cok = co0 + cg * 8 < a.Cout;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = rw + 32 * h;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float* src = s_part + (w * 64 + row) * DS_PITCH + cg * 8;
            const f32x4 u0 = *(const f32x4*)src, u1 = *(const f32x4*)(src + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] += u0[q]; v[q + 4] += u1[q]; }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { cs[q] += v[q]; cq[q] = fmaf(v[q], v[q], cq[q]); }
        if (cok) *(u32x4*)(a.y + (p0 + row) * a.ldy + co0 + cg * 8) = Vec16<u16>::pack(v);
    }
    if (do_stats) {
        // lanes of a wave = 8 rows x 8 channel groups: fold the rows with a fixed xor tree, one slot per wave, the waves in order
#pragma unroll
        for (int off = 8; off < 64; off <<= 1)
#pragma unroll
            for (int q = 0; q < 8; ++q) { cs[q] += __shfl_xor(cs[q], off, 64); cq[q] += __shfl_xor(cq[q], off, 64); }
        if (lane < 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { s_st[(wave * 2) * 64 + cg * 8 + q] = cs[q]; s_st[(wave * 2 + 1) * 64 + cg * 8 + q] = cq[q]; }
        }
        __syncthreads();
        if (tid < 64 && co0 + tid < a.Cout) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { t1 += s_st[(w * 2) * 64 + tid]; t2 += s_st[(w * 2 + 1) * 64 + tid]; }
            const size_t ro = (size_t)(blockIdx.x % a.stat_replicas) * a.stat_rstride;
            atomicAdd(&a.stat_sum[ro + co0 + tid], (double)t1);
            atomicAdd(&a.stat_sumsq[ro + co0 + tid], (double)t2);
        }
    }
}

bool dense_conv1_small_supported(const saunet_conv_desc* d, const void* x, const void* w, const void* y, const float* bias, const float* ps)
{
    static const bool on = !(getenv("SAUNET_DENSE_SMALL") && getenv("SAUNET_DENSE_SMALL")[0] == '0');                // A/B switch
    static const long maxpix = getenv("SAUNET_DENSE_SMALL_MAXPIX") ? atol(getenv("SAUNET_DENSE_SMALL_MAXPIX")) : 32768;
    const long P = (long)d->N * d->H * d->W;
    return on && d->dtype == SAUNET_BF16 && !d->transposed && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && bias == nullptr &&
           d->epi_relu == 0 && P % 64 == 0 && P <= maxpix && d->Cin % 32 == 0 && d->Cin >= 64 && d->Cin <= 2048 && d->Cout % 8 == 0 && d->Cout >= 64 &&
           d->Cout <= 128 && d->ldx % 8 == 0 && d->ldy % 8 == 0 && !(((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15);
}

int dense_conv1_small_forward(const saunet_conv_desc* d, const void* x, const void* w, const float* ps, const float* psh, void* y, double* ssum, double* ssq,
                              const saunet_bn_prologue* bnp, hipStream_t st)
{
    if (!bnp && !ps) return set_error(SAUNET_BAD_SHAPE, "dense_conv1_small: needs a prologue");
    DsArgs a;
    a.x = (const u16*)x; a.w = (const u16*)w; a.y = (u16*)y; a.pro_scale = ps; a.pro_shift = psh; a.stat_sum = ssum; a.stat_sumsq = ssq;
    a.stat_replicas = d->stat_replicas > 1 ? d->stat_replicas : 1; a.stat_rstride = d->stat_rstride;
    a.P = (long)d->N * d->H * d->W; a.Cin = d->Cin; a.ldx = d->ldx; a.Cout = d->Cout; a.ldy = d->ldy; a.pro_relu = d->pro_relu;
    if (bnp) a.bnp = *bnp; else a.bnp.gamma = nullptr;
    const int cpad = (d->Cin + 15) & ~15;
    const int lds = 2 * cpad * 4 + 4 * 64 * DS_PITCH * 4 + 4 * 2 * 64 * 4;
    static int attr_lds = 0;
    if (lds > attr_lds) { (void)hipFuncSetAttribute((const void*)dense_conv1_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_lds = lds; }
    hipLaunchKernelGGL(dense_conv1_small_kernel, dim3((unsigned)(a.P / 64), (d->Cout + 63) / 64), dim3(256), lds, st, a);
    SAUNET_CHECK_LAUNCH("dense_conv1_small");
    return SAUNET_OK;
}

}  // namespace saunet
