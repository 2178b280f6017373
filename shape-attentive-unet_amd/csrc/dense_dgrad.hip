// DenseNet conv1 data gradient with the fused "linear" BatchNorm-backward epilogue -- the single largest kernel of the
// training step (58 launches: every dense layer sends its 128-channel bottleneck gradient back to ALL earlier channels).
//
//     G[p][c]   = sum_k g[p][k] * W[c][k]                    K = 128 (bn_size * growth), c < Cin_i (64 .. 1024)
//     G         = G * [x[p][c]*scale[c] + shift[c] > 0]      ReLU mask of the layer's norm1 (recomputed from the concat buffer)
//     sums[c]  += G ,  sums[Cin+c] += G * xhat               the two BN-backward sums (float64, replicated)
//     y[p][c]  += scale[c] * G                               accumulate into the block's gradient buffer ("linear" BN backward)
//
// Traffic per launch: g once, x once, y read + write  ->  (128 + 3*Cin) * 2 B per pixel; 20 flop/B, i.e. HBM bound.
// The generic implicit-GEMM kernel (conv_igemm.hip) runs this with a 128x128 block tile: operands staged through LDS,
// the accumulator tile transposed through LDS again, and the x / y tiles requested only after the K loop -- two blocks
// per CU, every block serialising  load -> MFMA -> load -> store  with nothing else to hide the latencies (~1.9 TB/s).
// Here every WAVE is independent and there is no LDS staging and no barrier at all:
//   * the product is computed TRANSPOSED (rows = channels, columns = pixels): both MFMA operands are then 16-byte
//     K-contiguous pieces of g rows / packed weight rows and go straight from global memory into registers;
//   * a wave keeps the g fragments of its 32 pixels (32 VGPRs) for all channel tiles;
//   * in the accumulator layout a lane owns ONE pixel and 4 runs of 4 consecutive channels, so x / y are read and y is
//     written as 8-byte pieces per lane (the two half-waves complete 16 B per pixel row per instruction) -- requested for
//     a whole 64-channel group (one 128-byte line per row) before the MFMAs, with the next tile's weights in flight;
//   * per-channel sums over the 32 pixels of a tile are five DPP adds per value (VALU, not the LDS crossbar), then one
//     LDS atomic per channel per wave tile and one float64 atomic per channel per block.
#include "common.h"
#include <type_traits>

namespace saunet {

struct DenseDgradArgs {
    const u16* g; int ldg; const u16* w; const u16* x; int ldx; u16* y; int ldy;
    const float* scale; const float* shift; const float* mean; const float* invstd;
    double* sums; int reps, rstride;
    unsigned P; int Cin, relu, accumulate, group;
    // APPLY variant (round 5): the operand is NOT a ready gradient but the masked conv2 data gradient G of the same layer; the BatchNorm backward
    // of norm2,  dz1 = scale2 * (G - mean(G) - xhat(z1) * mean(G * xhat)),  is applied while the rows are loaded (the separate bn_bwd_apply pass
    // read G and z1 and wrote dz1; this kernel then re-read dz1).  dz1 is also written out (by the blocks of channel group 0) for the deferred
    // weight gradient of conv1.
    const u16* z; int ldz; u16* dz; int lddz;
    const double* sums2; int reps2, rstride2;      // [R][2][128]: sum G, sum G * xhat (dense_dgrad3 epilogue)
    const float* p2;                                // [4][128] scale, shift, mean, invstd of norm2
    double count;
    float* dgamma2; float* dbeta2;                  // [128] out (written by block (0, 0))
    // running coefficient sums of the block's "linear" BN1 backward: ab[r][0][c] += scale[c] * sum G, ab[r][1][c] += scale[c] * sum G * xhat
    double* ab; int ab_reps, ab_rstride, ab_half;
    int c_begin;                                    // LDS-staged kernel only: channels below c_begin are left alone (the layer pair's second launch covers them)
};

constexpr int DG_GROUP = 256;          // channels per block (weights of one group live in LDS: 256 rows x 272 B)
constexpr int DG_WPITCH = 136;         // u16 per LDS weight row: 128 + 8 pad -> 16 consecutive rows hit 16 different 16-byte bank groups
constexpr int DG_WAVES = 4;

// grid = (pixel-tile workers, channel groups of 256).  The block's weight rows are copied to LDS once; every WAVE then streams
// over 32-pixel tiles on its own (no barrier after the prologue): g fragments of the tile in registers, and for every 64-channel
// step (= one 128-byte line of the x / y rows) two 32x32 MFMA tiles whose A fragments come from LDS.
// Accumulator layout -> memory layout: after the MFMA a lane owns pixel (lane & 31) and the 4-channel runs 8j + 4*(lane >> 5);
// one v_permlane32_swap per value pair trades runs with the partner lane (same pixel, other half-wave) so that a lane owns
// 8 CONSECUTIVE channels twice per tile: x / y are then read and written as 16-byte pieces (measured 5.4 TB/s for this
// row-piece pattern against 3.8 TB/s with 8-byte pieces; scripts/probes/rowpiece_probe.hip).  The four output pieces of a
// step are stored together so the L2 merges them into whole lines.
// Round 3 (s_memtime stamps + scripts/probes/valu_probe.hip): a 64-channel step cost 9.3-9.7k cycles whether its operands came from HBM
// (block 1) or the L2 (block 4) -- the kernel was bound by its own instruction stream, not by bandwidth:
//   * every load sat under an exec-mask branch (`ok ? load : 0`), so the compiler could not count the vmcnt queue and waited vmcnt(0) in
//     front of the first MFMA of every step, i.e. for the y / next-x requests just issued: all loads are unconditional now (clamped
//     addresses, results discarded by the epilogue's `ok` tests and the guarded stores);
//   * `cur = nxt` register-set copies waited vmcnt(0) as well: the step sequence is written out per step (NS = steps per group is a
//     template parameter), each step a fixed set of registers, the hand-over copies sit where everything they wait for is old;
//   * the two BN-backward sums took 320 DPP adds (8 cycles each per wave) + 128 lane-atomics on LDS (12 cycles per lane) per step:
//     replaced by the transposing row reduction above (60 DPP adds) into REGISTER accumulators that live for the wave's lifetime.
template <int NS, bool APPLY = false> __global__ __launch_bounds__(DG_WAVES * 64, 2) void dense_dgrad_kernel(DenseDgradArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char d_smem[];
    TSTAMP_INIT();
    TSTAMP(40);
    const int g0 = blockIdx.y * a.group;
    const int GC = min(a.group, a.Cin - g0);           // channels of this group (multiple of 8)
    const int GCP = (GC + 31) & ~31;
    u16* s_w = (u16*)d_smem;                             // [GCP][DG_WPITCH]
    float* s_par = (float*)(d_smem + (size_t)GCP * DG_WPITCH * 2);   // [4][GCP] scale, shift, invstd, -mean*invstd
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
    const unsigned ntp = (a.P + 31) / 32;
    const unsigned stride = gridDim.x * DG_WAVES;

    // x / y pieces of one 64-channel step: piece i (tile t = i >> 1, run r = i & 1) = channels step*64 + 32t + 16r + 8*lh .. +8.
    // Pieces past the group's last channel read the group's last piece, rows past the last pixel the last pixel.
    struct XP { u32x4 x[4]; };
    auto request_x = [&](size_t pp, int step, XP& o, int lh8) {
        const u16* xr = a.x + pp * a.ldx + g0;
#pragma unroll
        for (int i = 0; i < 4; ++i) o.x[i] = *(const u32x4*)(xr + min(step * 64 + 16 * i + lh8, GC - 8));
    };
    u32x4 gf[8];
    u32x4 zr[APPLY ? 8 : 1];
    auto request_g = [&](size_t pp) {
        const u16* grow = a.g + pp * a.ldg + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) gf[ks] = *(const u32x4*)(grow + ks * 16);
        if constexpr (APPLY) {
            const u16* zrow = a.z + pp * a.ldz + lh * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) zr[ks] = *(const u32x4*)(zrow + ks * 16);
        }
    };
    float* s_cf = s_par + 4 * GCP;                     // APPLY: [3][128] a, b, c of  dz1 = a*G + b*z1 + c
    XP xa, xb;                        // x (needed first, for the mask) is requested one step ahead; y at the start of its own step
    // the two BN-backward sums stay in registers for the wave's whole lifetime: red[step][2t + r] = the lane's transposed partial sum
    // (row_transpose_sum) of the 8 channels of MFMA tile t, run pair r
    float red[NS][4];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[i][k] = 0.f;
    const bool sd0 = lane & 8, sd1 = lane & 4, sd2 = lane & 1, sd3 = lane & 2;
    {       // the first tile's operands are requested BEFORE the weight copy: both are one memory latency, now overlapped (they were sequential)
        const unsigned tp0 = min(blockIdx.x * DG_WAVES + wave, ntp - 1);
        const size_t pp0 = min(tp0 * 32u + lr, a.P - 1);
        request_g(pp0);
        request_x(pp0, 0, xa, 8 * lh);
    }
    constexpr int NT = DG_WAVES * 64;
    for (int i0 = threadIdx.x; i0 < GCP * 16; i0 += NT * 8) {     // 8 loads in flight per thread: the copy costs ~2 memory latencies
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + u * NT, GCP * 16 - 1), r = i >> 4, ch = i & 15;
            v[u] = *(const u32x4*)(a.w + (size_t)min(g0 + r, a.Cin - 1) * 128 + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * NT, r = i >> 4, ch = i & 15;
            if (i < GCP * 16) *(u32x4*)(s_w + r * DG_WPITCH + ch * 8) = v[u];
        }
    }
    for (int i = threadIdx.x; i < GCP; i += NT) {
        const bool ok = i < GC;
        const float is = ok ? a.invstd[g0 + i] : 0.f;
        s_par[i] = ok ? a.scale[g0 + i] : 0.f; s_par[GCP + i] = ok ? a.shift[g0 + i] : 0.f;
        s_par[2 * GCP + i] = is; s_par[3 * GCP + i] = ok ? -a.mean[g0 + i] * is : 0.f;
    }
    if constexpr (APPLY) {
        // the coefficients of norm2's backward from the replicated sums of the conv2 data gradient's epilogue (what bn_bwd_apply_kernel did once
        // per block): dz1 = s*(G - m1 - (z1 - mu)*is*m2) = s*G - s*is*m2 * z1 - s*(m1 - mu*is*m2)
        for (int k = threadIdx.x; k < 128; k += NT) {
            double S1, S2;
            rep_sum2(a.sums2, a.sums2 + 128, a.reps2, a.rstride2, k, S1, S2);
            const float sc = a.p2[k], mu = a.p2[256 + k], is = a.p2[384 + k];
            const float m1 = (float)(S1 / a.count), m2 = (float)(S2 / a.count);
            s_cf[k] = sc; s_cf[128 + k] = -sc * is * m2; s_cf[256 + k] = -sc * (m1 - mu * is * m2);
            if (blockIdx.x == 0 && blockIdx.y == 0 && a.dgamma2) { a.dbeta2[k] = (float)S1; a.dgamma2[k] = (float)S2; }
        }
    }
    __syncthreads();
    TSTAMP(41);
    const bool dz_writer = APPLY && blockIdx.y == 0;
    bool first_tile = true;
    for (unsigned tp = blockIdx.x * DG_WAVES + wave; tp < ntp; tp += stride) {
        const unsigned p = tp * 32u + lr;
        const bool live = p < a.P;
        const size_t pp = live ? p : a.P - 1;
        const size_t ppn = tp + stride < ntp ? min((tp + stride) * 32u + lr, a.P - 1) : pp;      // the wave's next tile (this one again when there is none: cache hits)
        TSTAMP(42);
        if constexpr (APPLY) {
            // (no cross-tile prefetch of the raw rows here: G and z1 together are 64 registers, the NS = 4 body has 19 to spare)
            if (!first_tile) request_g(pp);
            first_tile = false;
            int cfl = 8 * lh;
            asm volatile("" : "+v"(cfl));            // opaque: keeps the 48 coefficient vectors of a tile from being hoisted out of the loop
            u16* dzrow = a.dz + pp * a.lddz + 8 * lh;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                float G[8], Z[8], d[8];
                Vec16<u16>::unpack(gf[ks], G); Vec16<u16>::unpack(zr[ks], Z);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 ca = *(const f32x4*)(s_cf + ks * 16 + cfl + 4 * h), cb = *(const f32x4*)(s_cf + 128 + ks * 16 + cfl + 4 * h);
                    const f32x4 cc = *(const f32x4*)(s_cf + 256 + ks * 16 + cfl + 4 * h);
#pragma unroll
                    for (int q = 0; q < 4; ++q) d[4 * h + q] = fmaf(ca[q], G[4 * h + q], fmaf(cb[q], Z[4 * h + q], cc[q]));
                }
                gf[ks] = Vec16<u16>::pack(d);
                if (dz_writer && live) *(u32x4*)(dzrow + ks * 16) = gf[ks];
            }
        }
        u16* yrow = a.y + pp * a.ldy + g0 + 8 * lh;
        const u16* yrow0 = a.y + pp * a.ldy + g0;
        // the weight fragments are re-read from LDS for every tile ON PURPOSE: with the steps written out their addresses are tile-invariant
        // and the compiler would hoist all 16 x NS fragments out of the tile loop (64 registers per step: spills); this makes the base opaque
        int wlane = (lr * DG_WPITCH + lh * 8) * 2;
        asm volatile("" : "+v"(wlane));
        int plane = 8 * lh;                  // (the same for the per-channel parameter rows)
        asm volatile("" : "+v"(plane));
        // One 64-channel step.  On the tile's LAST step the x request is the next tile's first one and the next tile's g fragments are
        // requested as soon as the last MFMA has read the current ones (both MFMA tiles first, then the two epilogues cover the request).
        auto do_step = [&](auto step_c, XP& cur, XP& nxt) {
            constexpr int step = decltype(step_c)::value;
            constexpr bool LAST = step + 1 == NS;
            u32x4 yv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) yv[i] = u32x4{0u, 0u, 0u, 0u};
            if (a.accumulate == 1) {        // wave-uniform (2 = scaled store: y = scale * G, nothing to read)
#pragma unroll
                for (int i = 0; i < 4; ++i) yv[i] = *(const u32x4*)(yrow0 + min(step * 64 + 16 * i + plane, GC - 8));
            }
            request_x(LAST ? ppn : pp, LAST ? 0 : step + 1, nxt, plane);
            TSTAMP(43);
            u32x4 outv[4];          // the step's four 16-byte output pieces leave together: the L2 sees whole 128-byte lines
            auto mma = [&](auto t_c) -> f32x16 {
                constexpr int t = decltype(t_c)::value;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const u16* wrow = (const u16*)((const unsigned char*)s_w + wlane) + min(step * 64 + 32 * t, GCP - 32) * DG_WPITCH;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const u32x4 wf = *(const u32x4*)(wrow + ks * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, gf[ks]), acc, 0, 0, 0);
                }
                return acc;
            };
            auto epilogue = [&](auto t_c, const f32x16& acc) {
                constexpr int t = decltype(t_c)::value;
                constexpr int ct = step * 64 + 32 * t;       // first channel of this MFMA tile inside the group
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    // runs 2r (A) and 2r+1 (B) -> this lane's 8 consecutive channels cl .. cl+7
                    float G[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * r + q]), __float_as_uint(acc[8 * r + 4 + q]), false, false);
                        G[q] = __uint_as_float(sw[0]); G[4 + q] = __uint_as_float(sw[1]);
                    }
                    const int cl = ct + 16 * r + 8 * lh;
                    const bool ok = live && cl < GC;
                    const int cp = min(ct + 16 * r + plane, GCP - 8);             // parameter rows of channels past the group: any valid address
                    float xf[8], yf[8], o[8], e1[8], e2[8];
                    Vec16<u16>::unpack(cur.x[2 * t + r], xf); Vec16<u16>::unpack(yv[2 * t + r], yf);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 sc = *(const f32x4*)(s_par + cp + 4 * h), sh = *(const f32x4*)(s_par + GCP + cp + 4 * h);
                        const f32x4 a1 = *(const f32x4*)(s_par + 2 * GCP + cp + 4 * h), a0 = *(const f32x4*)(s_par + 3 * GCP + cp + 4 * h);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int e = 4 * h + q;
                            const bool keep = ok && (!a.relu || fmaf(xf[e], sc[q], sh[q]) > 0.f);
                            const float Gv = keep ? G[e] : 0.f;
                            e1[e] = Gv; e2[e] = Gv * fmaf(xf[e], a1[q], a0[q]);
                            o[e] = a.accumulate ? fmaf(sc[q], Gv, yf[e]) : Gv;        // (scaled store: yf = 0)
                        }
                    }
                    red[step][2 * t + r] += row_transpose_sum(e1, e2, sd0, sd1, sd2, sd3);
                    outv[2 * t + r] = Vec16<u16>::pack(o);
                }
            };
            constexpr std::integral_constant<int, 0> T0{};
            constexpr std::integral_constant<int, 1> T1{};
            const bool two = step * 64 + 32 < GCP;       // block-uniform: the step's second 32-channel tile exists
#pragma unroll
            for (int i = 2; i < 4; ++i) outv[i] = u32x4{0u, 0u, 0u, 0u};
            if constexpr (LAST) {
                const f32x16 acc0 = mma(T0);
                f32x16 acc1 = acc0;
                if (two) acc1 = mma(T1);
                if constexpr (!APPLY) request_g(ppn);                  // gf is free: the next tile's fragments travel behind this step's two epilogues
                epilogue(T0, acc0);
                if (two) epilogue(T1, acc1);
                if constexpr ((NS & 1) == 1) cur = nxt;     // odd number of steps: the next tile's step 0 reads the set this step read
            } else {
                const f32x16 acc0 = mma(T0);
                epilogue(T0, acc0);
                if (two) { const f32x16 acc1 = mma(T1); epilogue(T1, acc1); }
            }
            TSTAMP(44);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (live && step * 64 + 16 * i + 8 * lh < GC) *(u32x4*)(yrow + step * 64 + 16 * i) = outv[i];
            TSTAMP(45);
        };
        do_step(std::integral_constant<int, 0>{}, xa, xb);
        if constexpr (NS > 1) do_step(std::integral_constant<int, 1>{}, xb, xa);
        if constexpr (NS > 2) do_step(std::integral_constant<int, 2>{}, xa, xb);
        if constexpr (NS > 3) do_step(std::integral_constant<int, 3>{}, xb, xa);
    }
    TSTAMP(46);
    // block-level fold of the register accumulators: every wave parks its partial sums in the (now idle) weight area, then one thread per
    // channel adds the 4 waves x 2 DPP rows that hold it and sends the pair of sums to the replicated float64 accumulators
    __syncthreads();
    float* s_red = (float*)d_smem;                       // [wave][step * 4 + k][64 lanes]
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) s_red[(wave * (NS * 4) + i * 4 + k) * 64 + lane] = red[i][k];
    __syncthreads();
    TSTAMP(47);
    const size_t ro = (size_t)(blockIdx.x % a.reps) * a.rstride;
    for (int c = threadIdx.x; c < GC; c += NT) {
        // channel c of the group = step (c >> 6), register (c >> 4) & 3, lane bits: lh = bit 3 of c, then 4*b1 + 2*b0 + b2 = c & 7
        const int reg = (c >> 6) * 4 + ((c >> 4) & 3);
        const int l0 = 32 * ((c >> 3) & 1) + 4 * (c & 1) + 2 * ((c >> 2) & 1) + ((c >> 1) & 1);
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < DG_WAVES; ++w)
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const float* q = s_red + (w * (NS * 4) + reg) * 64 + l0 + 16 * row;
                t1 += q[0]; t2 += q[8];
            }
        atomicAdd(&a.sums[ro + g0 + c], (double)t1);
        atomicAdd(&a.sums[ro + a.Cin + g0 + c], (double)t2);
        if (a.ab) {       // the layer's share of the block's linear BN1-backward coefficients (replaces the per-layer coefficient launch)
            const float sc = a.scale[g0 + c];
            const size_t ra = (size_t)(blockIdx.x % a.ab_reps) * a.ab_rstride;
            atomicAdd(&a.ab[ra + g0 + c], (double)(sc * t1));
            atomicAdd(&a.ab[ra + a.ab_half + g0 + c], (double)(sc * t2));
        }
    }
    TSTAMP(48);
}

// ---------------------------------------------------------------------------------------------------------------------
// DenseNet conv2 (3x3, 128 -> 32) data gradient with the BatchNorm-backward reduction epilogue of norm2: the same
// wave-independent transposed-MFMA structure.  As a forward-view conv: 32 input channels (the layer's gradient chunk of the
// block buffer), 9 taps, 128 output channels:
//     G[p][n] = sum_{tap,k} W[n][tap][k] * g[p + tap][k]          K = 9 * 32 = 288
//     G       = G * [z1[p][n]*scale[n] + shift[n] > 0] ;  sums[n] += G ,  sums[128+n] += G * xhat ;  out[p][n] = G
// The B fragments are 16-byte pieces of the g rows of the 9 shifted pixels, loaded straight from global memory (64-byte rows,
// L1/L2 hits for the 9-fold reuse; out-of-image taps are zero); the 128 x 288 weight rows live in LDS (296-element pitch);
// a wave keeps its 18 g fragments (72 VGPRs) for the four 32-channel MFMA tiles; epilogue exactly as above (permlane32 swap,
// 16-byte row pieces of z1 / out, one 128-byte line per 64-channel step stored together).
// Traffic: (32 + 128 + 128) * 2 B per pixel; 128 flop/B -> HBM bound.
struct DenseDgrad3Args {
    const u16* g; int ldg; const u16* w; const u16* z; int ldz; u16* y; int ldy;
    const float* scale; const float* shift; const float* mean; const float* invstd;
    double* sums; int reps, rstride;
    int N, H, W; unsigned P; int relu;
    FastDiv dW, dHW;
    // CORR variant (round 5, per-wave kernel only): g is the block's gradient buffer chunk BEFORE the deferred correction of the linear BN1
    // backward; g' = g - (A[c] + B[c] * xhat(x)[c]) is applied to every fragment as it is loaded (zero padding stays zero) and the corrected
    // centre pixel goes to gc for the deferred weight gradient.  A, B = the running coefficient sums `ab` / count (dense_dgrad_kernel epilogue).
    const u16* xc; int ldxc;                       // the chunk's activations (block buffer slice)
    const double* ab; int ab_reps, ab_rstride, ab_half; double count;
    const float* xs; const float* xt;              // xhat rows of the chunk's channels
    u16* gc; int ldgc;
};
constexpr int D3_WPITCH = 296;
// HALO variant (maps whose H and W are multiples of 16): the workgroup (8 waves) owns a 16 x 16 pixel tile, wave w its rows 2w, 2w + 1;
// the 18 x 18 x 64 B gradient halo is staged by LDS-DMA (21 requests per tile instead of 18 fragment-shaped global loads per wave:
// those were the texture-addresser bound of the kernel), double-buffered across the tiles of the persistent loop, one barrier per tile.
// LDS pixel hp = hy * 18 + hx holds its four 16-byte channel chunks at slot chunk ^ ((hp >> 2) & 3): 16 consecutive pixels read by a
// 16-lane group of ds_read_b128 then cover all 16 bank groups.  Padding pixels are DMA'd from a zero page.
// Measured (block 1, 32 x 128 x 128): 132 -> 115 us with the halo, -> 100 us with z1 requested one step ahead (both variants); PMC: VALU
// 35 %, MFMA 17 % busy, 1.6 of 2 waves per SIMD resident, 25 % of wave cycles waiting on memory.  Skewing the two waves of a SIMD by
// s_sleep made no difference.  SAUNET_DGRAD3_HALO=0 selects the per-wave variant everywhere (A/B).
constexpr int D3_HALO_PIECES = 21, D3_HALO_BYTES = D3_HALO_PIECES * 1024;
static __device__ u32x4 g_dg_zeros[4];

template <bool HALO, bool CORR = false>
__global__ __launch_bounds__((HALO ? 8 : DG_WAVES) * 64, HALO ? 1 : 2) void dense_dgrad3_kernel(DenseDgrad3Args a)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char d_smem[];
    TSTAMP_INIT();
    TSTAMP(50);
    constexpr int WAVES = HALO ? 8 : DG_WAVES;
    // HALO && CORR (round 6): the deferred BN1 correction g' = g - (A + B * xhat(x)) is applied IN PLACE in the landed halo buffer by the wave that
    // requested the piece (its lane always holds the same 8 channels: the coefficients live in 16 registers), with the chunk's activations
    // staged by a second DMA stream into ONE extra buffer (it is only read between the landing and the barrier of its tile); the corrected
    // centre pixels go to gc.  The separate bn_bwd_correct_ab pass (read chunk + x, write dz2) and the re-read of dz2 disappear.
    constexpr int OFF_X = 2 * D3_HALO_BYTES;
    constexpr int OFF_W = HALO ? (CORR ? 3 : 2) * D3_HALO_BYTES : 0;  // the halo buffers first: their DMA destinations are 1 KB aligned
    u16* s_w = (u16*)(d_smem + OFF_W);                               // [128][D3_WPITCH]
    float* s_par = (float*)(d_smem + OFF_W + (size_t)128 * D3_WPITCH * 2);  // [4][128]
    constexpr int NT = WAVES * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lr = lane & 31, lh = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)d_smem;
    const unsigned tilesX = (unsigned)a.W >> 4, tilesY = (unsigned)a.H >> 4, ntile = HALO ? (unsigned)a.N * tilesX * tilesY : 0u;
    auto issue = [&](unsigned t, int buf) {
        const unsigned txi = t % tilesX, r1 = t / tilesX, tyi = r1 % tilesY, n = r1 / tilesY;
        const unsigned char* img = (const unsigned char*)(a.g + (size_t)n * a.H * a.W * a.ldg);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int piece = wave + 8 * j;
            if (piece < D3_HALO_PIECES) {
                const int hp = piece * 16 + (lane >> 2), sl = lane & 3;
                const int hy = hp / 18, hx = hp - hy * 18;
                const int iy = (int)tyi * 16 + hy - 1, ix = (int)txi * 16 + hx - 1;
                const bool ok = hp < 324 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const int ch = sl ^ ((hp >> 2) & 3);
                const unsigned char* src = ok ? img + ((size_t)(iy * a.W + ix) * a.ldg + ch * 8) * 2 : (const unsigned char*)g_dg_zeros + sl * 16;
                mm_dma16(src, lds0 + buf * D3_HALO_BYTES + piece * 1024);
                if constexpr (HALO && CORR) {
                    const unsigned char* ximg = (const unsigned char*)(a.xc + (size_t)n * a.H * a.W * a.ldxc);
                    const unsigned char* sx = ok ? ximg + ((size_t)(iy * a.W + ix) * a.ldxc + ch * 8) * 2 : (const unsigned char*)g_dg_zeros + sl * 16;
                    mm_dma16(sx, lds0 + OFF_X + piece * 1024);
                }
            }
        }
    };
    if constexpr (HALO) { if (blockIdx.x < ntile) issue(blockIdx.x, 0); }
    for (int i0 = threadIdx.x; i0 < 128 * 36; i0 += NT * 9) {     // 9 loads in flight per thread
        u32x4 v[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int i = min(i0 + u * NT, 128 * 36 - 1);
            v[u] = *(const u32x4*)(a.w + (size_t)i * 8);      // rows are contiguous in the packed weights
        }
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int i = i0 + u * NT, r = i / 36, ch = i - r * 36;
            if (i < 128 * 36) *(u32x4*)(s_w + r * D3_WPITCH + ch * 8) = v[u];
        }
    }
    for (int i = threadIdx.x; i < 128; i += NT) {
        const float is = a.invstd[i];
        s_par[i] = a.scale[i]; s_par[128 + i] = a.shift[i]; s_par[256 + i] = is; s_par[384 + i] = -a.mean[i] * is;
    }
    float* s_cc = s_par + 512;                     // CORR: [2][32]  g' = g - (cA + cB * x)
    if constexpr (CORR) {
        if (threadIdx.x < 32) {
            const int c = threadIdx.x;
            double A, B;
            rep_sum2(a.ab, a.ab + a.ab_half, a.ab_reps, a.ab_rstride, c, A, B);
            const float Af = (float)(A / a.count), Bf = (float)(B / a.count);
            s_cc[c] = fmaf(Bf, a.xt[c], Af); s_cc[32 + c] = Bf * a.xs[c];
        }
    }
    __syncthreads();
    float cA[CORR ? 16 : 1], cB[CORR ? 16 : 1];     // this lane's channels 16*h + 8*lh + j
    if constexpr (CORR && !HALO) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) { cA[8 * h + j] = s_cc[16 * h + 8 * lh + j]; cB[8 * h + j] = s_cc[32 + 16 * h + 8 * lh + j]; }
    }
    const unsigned ntp = (a.P + 31) / 32;
    // the two BN-backward sums in registers for the wave's lifetime (see dense_dgrad_kernel): red[step][2t + r]
    float red[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[i][k] = 0.f;
    const bool sd0 = lane & 8, sd1 = lane & 4, sd2 = lane & 1, sd3 = lane & 2;
    // the unit of the persistent loop: HALO -> 16 x 16 tile of the workgroup; else a 32-pixel run of the wave
    const unsigned u_first = HALO ? blockIdx.x : blockIdx.x * WAVES + wave, u_step = HALO ? gridDim.x : gridDim.x * WAVES, u_end = HALO ? ntile : ntp;
    int kbuf = 0;
    // clamped pixel of this lane in unit tp (the z1 / out row it owns)
    auto unit_pixel = [&](unsigned tp) -> unsigned {
        if constexpr (HALO) {
            const unsigned txi = tp % tilesX, r1 = tp / tilesX, tyi = r1 % tilesY, n = r1 / tilesY;
            return ((n * a.H + tyi * 16 + 2 * wave + (lr >> 4)) * a.W) + txi * 16 + (lr & 15);
        } else {
            return min(tp * 32u + lr, a.P - 1);
        }
    };
    // z1 runs one 64-channel step ahead of its use (two waves per SIMD do not hide an HBM round trip behind 18 MFMAs)
    u32x4 znext[4];
    if (u_first < u_end) {
        const u16* z0 = a.z + (size_t)unit_pixel(u_first) * a.ldz + 8 * lh;
#pragma unroll
        for (int i = 0; i < 4; ++i) znext[i] = *(const u32x4*)(z0 + 16 * i);
    }
    for (unsigned tp = u_first; tp < u_end; tp += u_step) {
        unsigned p; bool live;
        u32x4 gf[18];       // B fragments: [tap][k half]  (k = 16*h + 8*lh .. +8 of the 32 gradient channels)
        if constexpr (HALO) {
            p = unit_pixel(tp);
            live = true;
            mm_wait_vm<0>();
            if constexpr (CORR) {
                const unsigned txi = tp % tilesX, r1 = tp / tilesX, tyi = r1 % tilesY, n = r1 / tilesY;
                // the 8 channels of the lane's 16-byte slot are the same in every piece it requests: chunk = sl ^ ((hp >> 2) & 3) and
                // (hp >> 2) & 3 == (lane >> 4) & 3.  Their coefficients are re-read from the LDS per tile (the kernel sits at the register limit)
                const int chl = (lane & 3) ^ ((lane >> 4) & 3);
                float hA[8], hB[8];
                {
                    const f32x4 a0 = *(const f32x4*)(s_cc + chl * 8), a1 = *(const f32x4*)(s_cc + chl * 8 + 4);
                    const f32x4 b0 = *(const f32x4*)(s_cc + 32 + chl * 8), b1 = *(const f32x4*)(s_cc + 32 + chl * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hA[e] = a0[e]; hA[4 + e] = a1[e]; hB[e] = b0[e]; hB[4 + e] = b1[e]; }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int piece = wave + 8 * j;
                    if (piece < D3_HALO_PIECES) {
                        const int hp = piece * 16 + (lane >> 2);
                        const int hy = hp / 18, hx = hp - hy * 18;
                        const int iy = (int)tyi * 16 + hy - 1, ix = (int)txi * 16 + hx - 1;
                        const bool ok = hp < 324 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                        unsigned char* gp = d_smem + kbuf * D3_HALO_BYTES + piece * 1024 + lane * 16;
                        const u32x4 g4 = *(const u32x4*)gp, x4 = *(const u32x4*)(d_smem + OFF_X + piece * 1024 + lane * 16);
                        float gv[8], xv[8];
                        Vec16<u16>::unpack(g4, gv); Vec16<u16>::unpack(x4, xv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) gv[e] -= fmaf(hB[e], xv[e], hA[e]);
                        const u32x4 c4 = Vec16<u16>::pack(gv);
                        if (ok) {                       // (padding pixels came from the zero page and stay zero: the CORRECTED gradient is what is padded)
                            *(u32x4*)gp = c4;
                            if (hy >= 1 && hy <= 16 && hx >= 1 && hx <= 16)       // the tile's own pixels: what the conv2 weight gradient reads later
                                *(u32x4*)(a.gc + ((size_t)(n * a.H + iy) * a.W + ix) * a.ldgc + chl * 8) = c4;
                        }
                    }
                }
            }
            mm_barrier();
            TSTAMP(51);
            const unsigned char* hb = d_smem + kbuf * D3_HALO_BYTES;
            const int hp0 = (2 * wave + (lr >> 4)) * 18 + (lr & 15);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int hp = hp0 + (tap / 3) * 18 + tap % 3, key = (hp >> 2) & 3;
                gf[2 * tap] = *(const u32x4*)(hb + hp * 64 + ((lh ^ key) << 4));
                gf[2 * tap + 1] = *(const u32x4*)(hb + hp * 64 + (((2 + lh) ^ key) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (tp + u_step < u_end) issue(tp + u_step, kbuf ^ 1);
            kbuf ^= 1;
        } else {
            p = tp * 32u + lr;
            live = p < a.P;
        }
        const unsigned pc = live ? p : a.P - 1;
        if constexpr (!HALO) {
            const unsigned n = a.dHW.div(pc), rem = pc - n * (unsigned)(a.H * a.W);
            const int py = (int)a.dW.div(rem), px = (int)(rem - (unsigned)py * a.W);
            TSTAMP(51);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
                const bool ok = live && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
                const size_t pix = (size_t)n * a.H * a.W + (size_t)(ok ? yy : py) * a.W + (ok ? xx : px);
                const u16* row = a.g + pix * a.ldg + lh * 8;
                u32x4 v0 = *(const u32x4*)row, v1 = *(const u32x4*)(row + 16);
                if constexpr (CORR) {
                    const u16* xrow = a.xc + pix * a.ldxc + lh * 8;
                    const u32x4 x0 = *(const u32x4*)xrow, x1 = *(const u32x4*)(xrow + 16);
                    float gv[8], xv[8];
                    Vec16<u16>::unpack(v0, gv); Vec16<u16>::unpack(x0, xv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) gv[j] -= fmaf(cB[j], xv[j], cA[j]);
                    v0 = Vec16<u16>::pack(gv);
                    Vec16<u16>::unpack(v1, gv); Vec16<u16>::unpack(x1, xv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) gv[j] -= fmaf(cB[8 + j], xv[j], cA[8 + j]);
                    v1 = Vec16<u16>::pack(gv);
                    if (tap == 4 && live) {          // the corrected gradient of this lane's own pixel: what the conv2 weight gradient reads later
                        u16* gcrow = a.gc + (size_t)pc * a.ldgc + lh * 8;
                        *(u32x4*)gcrow = v0; *(u32x4*)(gcrow + 16) = v1;
                    }
                }
                gf[2 * tap] = ok ? v0 : u32x4{0u, 0u, 0u, 0u};
                gf[2 * tap + 1] = ok ? v1 : u32x4{0u, 0u, 0u, 0u};
            }
        }
        TSTAMP(52);
        const u16* zrow = a.z + (size_t)pc * a.ldz + 8 * lh;
        const u16* zrow_next = a.z + (size_t)unit_pixel(min(tp + u_step, u_end - 1)) * a.ldz + 8 * lh;
        u16* yrow = a.y + (size_t)pc * a.ldy + 8 * lh;
        // opaque per-tile bases: with the two steps written out the weight / parameter addresses are tile-invariant and the compiler would
        // hoist all 72 weight fragments + 32 parameter vectors out of the tile loop (spills)
        int wlane = (lr * D3_WPITCH + lh * 8) * 2, plane = 8 * lh;
        asm volatile("" : "+v"(wlane), "+v"(plane));
        auto do_step = [&](auto step_c) {
            constexpr int step = decltype(step_c)::value;
            u32x4 zv[4], outv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) zv[i] = znext[i];
            {
                const u16* nz = step == 0 ? zrow + 64 : zrow_next;
#pragma unroll
                for (int i = 0; i < 4; ++i) znext[i] = *(const u32x4*)(nz + 16 * i);
            }
            TSTAMP(53);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ct = step * 64 + 32 * t;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const u16* wrow = (const u16*)((const unsigned char*)s_w + wlane) + ct * D3_WPITCH;
#pragma unroll
                for (int ks = 0; ks < 18; ++ks) {
                    const u32x4 wf = *(const u32x4*)(wrow + ks * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, gf[ks]), acc, 0, 0, 0);
                }
                TSTAMP(54);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    float G[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * r + q]), __float_as_uint(acc[8 * r + 4 + q]), false, false);
                        G[q] = __uint_as_float(sw[0]); G[4 + q] = __uint_as_float(sw[1]);
                    }
                    const int cp = ct + 16 * r + plane;
                    float zf[8], o[8], e1[8], e2[8];
                    Vec16<u16>::unpack(zv[2 * t + r], zf);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 sc = *(const f32x4*)(s_par + cp + 4 * h), sh = *(const f32x4*)(s_par + 128 + cp + 4 * h);
                        const f32x4 a1 = *(const f32x4*)(s_par + 256 + cp + 4 * h), a0 = *(const f32x4*)(s_par + 384 + cp + 4 * h);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int e = 4 * h + q;
                            const bool keep = live && (!a.relu || fmaf(zf[e], sc[q], sh[q]) > 0.f);
                            const float Gv = keep ? G[e] : 0.f;
                            e1[e] = Gv; e2[e] = Gv * fmaf(zf[e], a1[q], a0[q]);
                            o[e] = Gv;
                        }
                    }
                    red[step][2 * t + r] += row_transpose_sum(e1, e2, sd0, sd1, sd2, sd3);
                    outv[2 * t + r] = Vec16<u16>::pack(o);
                }
                TSTAMP(55);
            }
            if (live) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *(u32x4*)(yrow + step * 64 + 16 * i) = outv[i];
            }
            TSTAMP(56);
        };
        do_step(std::integral_constant<int, 0>{});
        do_step(std::integral_constant<int, 1>{});
    }
    TSTAMP(57);
    // block-level fold of the register accumulators through the (now idle) weight area: see dense_dgrad_kernel
    __syncthreads();
    float* s_red = (float*)d_smem;                       // [wave][step * 4 + k][64 lanes]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) s_red[(wave * 8 + i * 4 + k) * 64 + lane] = red[i][k];
    __syncthreads();
    const size_t ro = (size_t)(blockIdx.x % a.reps) * a.rstride;
    for (int c = threadIdx.x; c < 128; c += NT) {
        const int reg = (c >> 6) * 4 + ((c >> 4) & 3);
        const int l0 = 32 * ((c >> 3) & 1) + 4 * (c & 1) + 2 * ((c >> 2) & 1) + ((c >> 1) & 1);
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w)
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const float* q = s_red + (w * 8 + reg) * 64 + l0 + 16 * row;
                t1 += q[0]; t2 += q[8];
            }
        atomicAdd(&a.sums[ro + c], (double)t1);
        atomicAdd(&a.sums[ro + 128 + c], (double)t2);
    }
}

// ---- small maps (blocks 3 / 4: 1024 / 256 runs of 32 pixels): the per-wave kernel above gives every wave ALL 128 output channels of one run --
// 72 dependent MFMAs and four transposed epilogues in a row behind a 75 KB weight copy into LDS that is used once (phase stamps, 16 x 16 x 32
// images: copy + barrier 5.8k cycles, fragment loads 1.8k, four x (MFMA 0.95k + epilogue 1.6k), fold -- 22k cycles for 2.4 GFLOP).  Here the
// four waves of a workgroup are the four 32-channel tiles of ONE run:
//   * a wave's 32 x 288 weight rows go straight from global memory into its registers (18 fragments; no LDS copy, nothing to wait for but the load);
//   * the 18 gradient fragments of the run are loaded -- and, CORR, corrected -- once, taps dealt over the waves, and exchanged through LDS (double
//     buffered: one barrier per run);
//   * the product is NOT transposed (rows = pixels, columns = channels; the same fragments with the MFMA operands swapped), so a lane owns one
//     channel: scale / shift / xhat coefficients are four registers and the two BatchNorm sums a per-lane add + one cross-half shuffle, carried
//     in registers to ONE pair of atomics per channel and wave -- the transposed epilogue (16 channels of one pixel per lane: coefficient vectors
//     from LDS per element, transposing DPP reductions) measured 2.4-2.9k cycles per 32 x 32 tile, this one ~0.8k; z1 and the output pass through
//     wave-private LDS tiles as 16-byte row pieces (80-byte pitch: the two lane halves, four rows apart, fall on different banks).
constexpr int D3CW_PITCH = 80;
template <bool CORR>
__global__ __launch_bounds__(256, 2) void dense_dgrad3_cw_kernel(DenseDgrad3Args a)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_frag[2][18 * 1024];    // [buffer][tap * 2 + k half][lane] 16-byte fragments
    __shared__ __attribute__((aligned(16))) unsigned char s_zt[4][32 * D3CW_PITCH];   // [wave]: z1 of the wave's 32 channels, 32 pixel rows
    __shared__ __attribute__((aligned(16))) unsigned char s_ot[4][32 * D3CW_PITCH];   // [wave]: the output tile
    __shared__ float s_cc[64];
    TSTAMP_INIT();
    TSTAMP(50);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lr = lane & 31, lh = lane >> 5;
    const int ct = wave * 32;
    u32x4 wf[18];
    {
        const u16* wrow = a.w + (size_t)(ct + lr) * 288 + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) wf[ks] = *(const u32x4*)(wrow + ks * 16);
    }
    // this lane's channel: the BN2 mask and xhat coefficients
    const float sc = a.scale[ct + lr], sh = a.shift[ct + lr], a1 = a.invstd[ct + lr], a0 = -a.mean[ct + lr] * a1;
    if constexpr (CORR) {
        if (threadIdx.x < 32) {
            const int c = threadIdx.x;
            double A, B;
            rep_sum2(a.ab, a.ab + a.ab_half, a.ab_reps, a.ab_rstride, c, A, B);
            const float Af = (float)(A / a.count), Bf = (float)(B / a.count);
            s_cc[c] = fmaf(Bf, a.xt[c], Af); s_cc[32 + c] = Bf * a.xs[c];
        }
    }
    const unsigned ntp = (a.P + 31) / 32;
    float red1 = 0.f, red2 = 0.f;
    unsigned char* zt = s_zt[wave];
    unsigned char* ot = s_ot[wave];
    int it = 0;
    for (unsigned tp = blockIdx.x; tp < ntp; tp += gridDim.x, ++it) {
        const unsigned p = tp * 32u + lr;
        const bool live = p < a.P;
        const unsigned pc = live ? p : a.P - 1;
        const unsigned n = a.dHW.div(pc), rem = pc - n * (unsigned)(a.H * a.W);
        const int py = (int)a.dW.div(rem), px = (int)(rem - (unsigned)py * a.W);
        // this wave's taps of the run: wave, wave + 4, wave + 8
        u32x4 gv0[3], gv1[3], xv0[3], xv1[3];
        bool okt[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int tap = wave + 4 * q;
            if (tap < 9) {
                const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
                okt[q] = live && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
                const size_t pix = (size_t)n * a.H * a.W + (size_t)(okt[q] ? yy : py) * a.W + (okt[q] ? xx : px);
                const u16* row = a.g + pix * a.ldg + lh * 8;
                gv0[q] = *(const u32x4*)row; gv1[q] = *(const u32x4*)(row + 16);
                if constexpr (CORR) {
                    const u16* xrow = a.xc + pix * a.ldxc + lh * 8;
                    xv0[q] = *(const u32x4*)xrow; xv1[q] = *(const u32x4*)(xrow + 16);
                }
            }
        }
        // z1 of the wave's 32 channels: 32 rows x 4 pieces of 16 bytes, two per lane
        u32x4 zv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = lane + 64 * i, row = q >> 2, piece = q & 3;
            zv[i] = *(const u32x4*)(a.z + (size_t)min(tp * 32u + row, a.P - 1) * a.ldz + ct + piece * 8);
        }
        if (CORR && it == 0) __syncthreads();          // the correction coefficients are in LDS
        TSTAMP(51);
        unsigned char* fb = s_frag[it & 1];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int tap = wave + 4 * q;
            if (tap < 9) {
                u32x4 v0 = gv0[q], v1 = gv1[q];
                if constexpr (CORR) {
                    float gv[8], xv[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        Vec16<u16>::unpack(h ? v1 : v0, gv); Vec16<u16>::unpack(h ? xv1[q] : xv0[q], xv);
                        const f32x4 ca0 = *(const f32x4*)(s_cc + 16 * h + 8 * lh), ca1 = *(const f32x4*)(s_cc + 16 * h + 8 * lh + 4);
                        const f32x4 cb0 = *(const f32x4*)(s_cc + 32 + 16 * h + 8 * lh), cb1 = *(const f32x4*)(s_cc + 32 + 16 * h + 8 * lh + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { gv[j] -= fmaf(cb0[j], xv[j], ca0[j]); gv[4 + j] -= fmaf(cb1[j], xv[4 + j], ca1[j]); }
                        if (h) v1 = Vec16<u16>::pack(gv); else v0 = Vec16<u16>::pack(gv);
                    }
                    if (tap == 4 && live) {            // the corrected gradient of this lane's own pixel: what the conv2 weight gradient reads later
                        u16* gcrow = a.gc + (size_t)pc * a.ldgc + lh * 8;
                        *(u32x4*)gcrow = v0; *(u32x4*)(gcrow + 16) = v1;
                    }
                }
                *(u32x4*)(fb + (2 * tap) * 1024 + lane * 16) = okt[q] ? v0 : u32x4{0u, 0u, 0u, 0u};
                *(u32x4*)(fb + (2 * tap + 1) * 1024 + lane * 16) = okt[q] ? v1 : u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = lane + 64 * i, row = q >> 2, piece = q & 3;
            *(u32x4*)(zt + row * D3CW_PITCH + piece * 16) = zv[i];
        }
        TSTAMP(52);
        __syncthreads();
        TSTAMP(53);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            const u32x4 gf = *(const u32x4*)(fb + ks * 1024 + lane * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, gf), __builtin_bit_cast(bf16x8_t, wf[ks]), acc, 0, 0, 0);
        }
        u16 zs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) zs[r] = *(const u16*)(zt + ((r & 3) + 8 * (r >> 2) + 4 * lh) * D3CW_PITCH + lr * 2);
        TSTAMP(54);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float zf = __uint_as_float((unsigned)zs[r] << 16);
            const bool keep = (int)(tp * 32u + row < a.P) & (int)(!a.relu | (fmaf(zf, sc, sh) > 0.f));
            const float Gv = keep ? acc[r] : 0.f;
            s1 += Gv; s2 = fmaf(Gv, fmaf(zf, a1, a0), s2);
            *(u16*)(ot + row * D3CW_PITCH + lr * 2) = __builtin_bit_cast(u16, (__bf16)Gv);
        }
        red1 += s1 + __shfl_xor(s1, 32, 64); red2 += s2 + __shfl_xor(s2, 32, 64);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = lane + 64 * i, row = q >> 2, piece = q & 3;
            if (tp * 32u + row < a.P) *(u32x4*)(a.y + (size_t)(tp * 32u + row) * a.ldy + ct + piece * 8) = *(const u32x4*)(ot + row * D3CW_PITCH + piece * 16);
        }
        TSTAMP(55);
    }
    if (lane < 32) {
        const size_t ro = (size_t)(blockIdx.x % a.reps) * a.rstride;
        atomicAdd(&a.sums[ro + ct + lane], (double)red1);
        atomicAdd(&a.sums[ro + 128 + ct + lane], (double)red2);
    }
    TSTAMP(57);
}

// maps on which the conv2 data gradient runs the LDS-DMA staged (HALO) kernel: multiples of 16 with at least one 16 x 16 tile per CU
static bool dense_dgrad3_halo(int N, int H, int W)
{
    static const bool halo_env = ab_env_on("SAUNET_DGRAD3_HALO");     // A/B switch (variant builds only)
    return halo_env && H % 16 == 0 && W % 16 == 0 && (long)N * (H >> 4) * (W >> 4) >= 256;
}

bool dense_dgrad3_supported(const saunet_conv_desc* d, const float* bias, const float* ps, const saunet_bn_epilogue* epi)
{
    return epi != nullptr && epi->bn_x != nullptr && !epi->accumulate && d->dtype == SAUNET_BF16 && d->KH == 3 && d->KW == 3 && d->stride == 1 &&
           d->pad == 1 && !d->transposed && d->Cin == 32 && d->Cout == 128 && d->ldx % 8 == 0 && d->ldy % 8 == 0 && epi->ld_bn_x % 8 == 0 &&
           bias == nullptr && ps == nullptr && (long)d->N * d->H * d->W < (1L << 31);
}

static int launch_dense_dgrad3(DenseDgrad3Args& a, bool corr, hipStream_t st)
{
    const size_t lds = (size_t)128 * D3_WPITCH * 2 + sizeof(float) * (4 * 128 + 64);
    static DeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)dense_dgrad3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_dgrad3_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_dgrad3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 124 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_dgrad3_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 148 * 1024);
    }
    static const bool cw_on = ab_env_on("SAUNET_DGRAD3_CW");     // A/B (variant builds only)
    const long runs = ((long)a.P + 31) / 32;
    if (cw_on && runs <= 2048 && !dense_dgrad3_halo(a.N, a.H, a.W)) {
        // low-resolution maps: one 32-pixel run per workgroup at a time, its waves = the four 32-channel tiles
        static const long cw_blocks = ab_env_int("SAUNET_DGRAD3_CW_BLOCKS", 512);
        const long bx = runs < cw_blocks ? runs : cw_blocks;
        if (corr) hipLaunchKernelGGL(dense_dgrad3_cw_kernel<true>, dim3((unsigned)bx), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(dense_dgrad3_cw_kernel<false>, dim3((unsigned)bx), dim3(256), 0, st, a);
        SAUNET_CHECK_LAUNCH(corr ? "dense_dgrad3_cw_kernel<true>" : "dense_dgrad3_cw_kernel<false>");
        return SAUNET_OK;
    }
    if (dense_dgrad3_halo(a.N, a.H, a.W)) {
        // one 8-wave workgroup per CU, 16 x 16 tiles: only for maps with at least one tile per CU
        const long ntile = (long)a.N * (a.H >> 4) * (a.W >> 4);
        const long bx = ntile < 256 ? ntile : 256;
        if (corr) hipLaunchKernelGGL((dense_dgrad3_kernel<true, true>), dim3((unsigned)bx), dim3(512), lds + 3 * D3_HALO_BYTES, st, a);
        else hipLaunchKernelGGL(dense_dgrad3_kernel<true>, dim3((unsigned)bx), dim3(512), lds + 2 * D3_HALO_BYTES, st, a);
        SAUNET_CHECK_LAUNCH(corr ? "dense_dgrad3_kernel<true, true>" : "dense_dgrad3_kernel<true, false>");
        return SAUNET_OK;
    } else {
        long bx = 512; const long maxbx = ((long)a.P + 32 * DG_WAVES - 1) / (32 * DG_WAVES);
        if (bx > maxbx) bx = maxbx; if (bx < 1) bx = 1;
        if (corr) hipLaunchKernelGGL((dense_dgrad3_kernel<false, true>), dim3((unsigned)bx), dim3(DG_WAVES * 64), lds, st, a);
        else hipLaunchKernelGGL(dense_dgrad3_kernel<false>, dim3((unsigned)bx), dim3(DG_WAVES * 64), lds, st, a);
    }
    SAUNET_CHECK_LAUNCH(corr ? "dense_dgrad3_kernel<false, true>" : "dense_dgrad3_kernel<false, false>");
    return SAUNET_OK;
}

int dense_dgrad3_forward(const saunet_conv_desc* d, const void* x, const void* w, void* y, const saunet_bn_epilogue* epi, hipStream_t st)
{
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)epi->bn_x) & 15)
        return set_error(SAUNET_BAD_ALIGN, "dense 3x3 dgrad: operands must be 16-byte aligned");
    DenseDgrad3Args a{};
    a.g = (const u16*)x; a.ldg = d->ldx; a.w = (const u16*)w; a.z = (const u16*)epi->bn_x; a.ldz = epi->ld_bn_x; a.y = (u16*)y; a.ldy = d->ldy;
    a.scale = epi->scale; a.shift = epi->shift; a.mean = epi->mean; a.invstd = epi->invstd;
    a.sums = epi->sums; a.reps = epi->sums_replicas > 1 ? epi->sums_replicas : 1; a.rstride = epi->sums_rstride;
    a.N = d->N; a.H = d->H; a.W = d->W; a.P = (unsigned)((long)d->N * d->H * d->W); a.relu = epi->relu;
    a.dW = FastDiv::make((unsigned)d->W); a.dHW = FastDiv::make((unsigned)(d->H * d->W));
    return launch_dense_dgrad3(a, false, st);
}

bool dense_dgrad_supported(const saunet_conv_desc* d, const float* bias, const float* ps, const saunet_bn_epilogue* epi)
{
    return epi != nullptr && epi->bn_x != nullptr && d->dtype == SAUNET_BF16 && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 &&
           !d->transposed && d->Cin == 128 && d->Cout % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0 && epi->ld_bn_x % 8 == 0 &&
           bias == nullptr && ps == nullptr && d->Cout <= 2048 && (long)d->N * d->H * d->W < (1L << 31);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the conv1 data gradient of the fused layer backward on the LOW-RESOLUTION blocks, NON-transposed and LDS-staged.
// dense_dgrad_kernel above computes the product transposed so that x / dbuf travel as 16-byte row pieces without LDS -- the right trade on the
// large maps (HBM-bound, long persistent loops), but its epilogue then holds 16 CHANNELS of one pixel per lane: per-channel coefficients come from
// LDS for every element and the two BatchNorm sums need transposing DPP reductions -- ~490 VALU instructions per 64-channel step and wave for 16
// MFMAs, and with two waves per SIMD that VALU stream IS the step on blocks 3 / 4 (PMC: VALU 24 %, 38 % waiting, 4.7k cycles per step).
// Here rows = pixels, columns = channels:
//   * a lane owns ONE channel and 16 pixels of a 32 x 32 MFMA tile: scale / shift / xhat coefficients are four registers, the two sums are a
//     per-lane add + one cross-half shuffle -- ~8 VALU per element;
//   * every operand arrives by LDS-DMA: the G and z1 tiles (128 pixels x 128), BN2-backward applied in place by the requesting wave (dz1, written
//     out once per pixel tile: no channel groups re-transforming the same rows), then the A fragments stay in registers and per 64-channel step
//     the weight rows, the x tile (mask, xhat) and the dbuf tile (read-modify-write IN the LDS, stored back as whole 16-byte row pieces);
//   * software pipeline of one stage: the product of step j (weights W(j)) runs interleaved with the epilogue of step j - 1 (x / dbuf tile
//     (j - 1)); W is requested one stage before its product and x / dbuf one stage before their epilogue, so both rings are two slots
//     (2 x 16 KB + 2 x 32 KB) and a stage has ONE barrier; the partial BatchNorm sums of up to 8 steps wait in LDS for one fold + atomics;
//   * the steps of a pixel tile may be split over blockIdx.y when the map has fewer than 256 tiles (block 4).
// Measured (MI355X, scripts/dense_chain_micro.py): block 3 backward 82.6 -> 76.8 us / layer, block 4 53.7 -> 46.4.  On block 3 the kernel now
// moves its ~150 MB (x read, dbuf read + write, G / z1 / dz1 once) in 35 us: ~4.2 TB/s, the rate every HBM-bound kernel of this step reaches
// (profiles/r05_step_pmc_summary.txt), so neither the pipelining (same time as the unpipelined loop) nor BM = 64 with two independent 4-wave
// blocks per CU (slower: a third more DMA requests per pixel) changes it -- what is left there is byte count, not schedule.
template <int BM_> struct DgLdsLayout {
    static constexpr int BM = BM_, BN = 64, NW = BM / 16, THREADS = NW * 64;     // waves: BM / 32 pixel groups x 2 channel groups
    static constexpr int A_BYTES = BM * 256, Z_BYTES = BM * 256;                 // G / dz1 tile, z1 tile (128 channels = 256 B rows)
    static constexpr int W_BYTES = BN * 256, X_BYTES = BM * BN * 2, Y_BYTES = BM * BN * 2, XY_BYTES = X_BYTES + Y_BYTES;
    static constexpr int OFF_XY = 0;                                             // two x / dbuf slots = the A / z1 space once the fragments are in registers
    static constexpr int OFF_W = 2 * XY_BYTES;                                   // two weight slots
    static_assert(A_BYTES + Z_BYTES <= OFF_W, "prologue tiles overlay the x / dbuf ring only");
    static constexpr int MAX_STEPS = 8;                                          // 64-channel steps whose partial sums the ring holds before a fold
    static constexpr int OFF_PART = OFF_W + 2 * W_BYTES;                         // float[MAX_STEPS][BM / 32 row waves][2][64] partial sums; before the
    static constexpr int PART_STEP = (BM / 32) * 128;                            // loop its first 1.5 KB hold the float[3][128] BN2-backward coefficients
    static constexpr int LDS = OFF_PART + MAX_STEPS * PART_STEP * 4;
    static_assert(MAX_STEPS * PART_STEP * 4 >= 3 * 128 * 4, "coefficient overlay");
};
// byte offset of 16-byte chunk c (0..15) of row r in a [rows][16] chunk image (256-byte rows)
__device__ __forceinline__ int dgl_off(int r, int c) { return (r * 16 + (c ^ (r & 15))) * 16; }

#ifndef SAUNET_DGL_VALU
#define SAUNET_DGL_VALU 20
#endif
template <int BM>
__global__ __launch_bounds__(DgLdsLayout<BM>::THREADS) void dense_conv1_dgrad_lds_kernel(DenseDgradArgs a)
{
    using LY = DgLdsLayout<BM>;
    constexpr int NT = LY::THREADS, WPIECES = 16 / LY::NW;        // weight pieces per wave and step (x / dbuf: always two, G / z1: always four)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                     // BM / 32 pixel groups of 32 x 2 channel groups of 32
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int m0 = blockIdx.x * LY::BM;
    const int nsteps = (a.Cin + LY::BN - 1) / LY::BN, jb = a.c_begin / LY::BN;
    const int spg = (nsteps - jb + gridDim.y - 1) / gridDim.y;   // steps per channel group
    const int j0 = jb + blockIdx.y * spg, j1 = min(j0 + spg, nsteps);
    if (j0 >= j1) return;
    float* s_cf = (float*)(smem + LY::OFF_PART);
    const int P = (int)a.P;

    // ---- requests of one 64-channel step: 16 weight pieces of 1 KB (one step ahead of the product) and BM / 8 pieces each of x and dbuf
    // (one step ahead of the epilogue, which runs one step behind the product)
    auto issue_w = [&](int j) {
        const int c0 = j * LY::BN;
        const unsigned dst = lds0 + LY::OFF_W + ((j - j0) & 1) * LY::W_BYTES;
#pragma unroll
        for (int q = 0; q < WPIECES; ++q) {
            // weight rows c0 + 4 piece .. +3 (row = input channel of conv1, 128 K-contiguous elements); rows past Cin repeat the last one
            const int piece = wave * WPIECES + q, row = piece * 4 + (lane >> 4), c = (lane & 15) ^ (row & 15);
            mm_dma16(a.w + (size_t)min(c0 + row, a.Cin - 1) * 128 + c * 8, dst + piece * 1024);
        }
    };
    auto issue_xy = [&](int j) {
        const int c0 = j * LY::BN;
        const unsigned dst = lds0 + LY::OFF_XY + ((j - j0) & 1) * LY::XY_BYTES;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // pixel rows of 128 B (64 channels), 8 pixels per piece; channels past the buffer row end are clamped (never used)
            const int piece = wave * 2 + q, px = piece * 8 + (lane >> 3), ch = lane & 7;
            const size_t m = (size_t)min(m0 + px, P - 1);
            const int cc = min(c0 + ch * 8, a.Cin - 8);
            mm_dma16(a.x + m * a.ldx + cc, dst + piece * 1024);
            mm_dma16(a.y + m * a.ldy + cc, dst + LY::X_BYTES + piece * 1024);
        }
    };
    // ---- prologue: G and z1 tiles (four pieces each per wave), the first step's weights, the BN2-backward coefficients meanwhile
    TSTAMP_INIT();
    TSTAMP(90);
    int arow[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = wave * 4 + q, row = piece * 4 + (lane >> 4), c = (lane & 15) ^ (row & 15);
        const size_t m = (size_t)min(m0 + row, P - 1);
        arow[q] = row | (c << 8);
        mm_dma16(a.g + m * a.ldg + c * 8, lds0 + piece * 1024);
        mm_dma16(a.z + m * a.ldz + c * 8, lds0 + LY::A_BYTES + piece * 1024);
    }
    issue_w(j0);
    for (int k = tid; k < 128; k += NT) {
        double S1, S2;
        rep_sum2(a.sums2, a.sums2 + 128, a.reps2, a.rstride2, k, S1, S2);
        const float sc = a.p2[k], mu = a.p2[256 + k], is = a.p2[384 + k];
        const float m1 = (float)(S1 / a.count), m2 = (float)(S2 / a.count);
        s_cf[k] = sc; s_cf[128 + k] = -sc * is * m2; s_cf[256 + k] = -sc * (m1 - mu * is * m2);
        if (blockIdx.x == 0 && blockIdx.y == 0 && a.dgamma2) { a.dbeta2[k] = (float)S1; a.dgamma2[k] = (float)S2; }
    }
    __syncthreads();
    TSTAMP(91);
    mm_wait_vm<WPIECES>();                                        // the G / z1 pieces have landed (the weights of the first step may not have)
    // dz1 = a*G + b*z1 + c in place, on the pieces this wave requested; written out once per pixel tile (by channel group 0)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = wave * 4 + q, row = arow[q] & 0xff, c = arow[q] >> 8;
        unsigned char* pg = smem + piece * 1024 + lane * 16;
        float G[8], Z[8], d[8];
        Vec16<u16>::unpack(*(const u32x4*)pg, G);
        Vec16<u16>::unpack(*(const u32x4*)(pg + LY::A_BYTES), Z);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 ca = *(const f32x4*)(s_cf + c * 8 + 4 * h), cb = *(const f32x4*)(s_cf + 128 + c * 8 + 4 * h), cc = *(const f32x4*)(s_cf + 256 + c * 8 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[4 * h + e] = fmaf(ca[e], G[4 * h + e], fmaf(cb[e], Z[4 * h + e], cc[e]));
        }
        const u32x4 v = Vec16<u16>::pack(d);
        *(u32x4*)pg = v;
        if (blockIdx.y == 0 && m0 + row < P) *(u32x4*)(a.dz + (size_t)(m0 + row) * a.lddz + c * 8) = v;
    }
    __syncthreads();
    // the wave's A fragments (its 32 pixels x K = 128) stay in registers for every step; after this barrier the A / z1 space is stage slot 0
    u32x4 af[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) af[ks] = *(const u32x4*)(smem + dgl_off(wm * 32 + lr, 2 * ks + lh));
    __syncthreads();

    TSTAMP(92);
    float* s_part = (float*)(smem + LY::OFF_PART);                // [step][row waves][2][64]: folded once after the loop
    // the lane's per-channel coefficients, fetched two steps ahead of their use (a global load at the epilogue would expose its whole latency)
    auto coeffs = [&](int j, float& sc, float& sh, float& a1, float& a0) {
        const int chs = min(j * LY::BN + wn * 32 + lr, a.Cin - 1);
        sc = a.scale[chs]; sh = a.shift[chs]; a1 = a.invstd[chs]; a0 = a.mean[chs];
    };
    const int col = wn * 32 + lr;
    float nsc, nsh, na1, na0, sc = 0.f, sh = 0.f, a1 = 0.f, a0 = 0.f;
    coeffs(j0, nsc, nsh, na1, na0);
    f32x16 accp;                                                  // the product of the previous step, waiting for its epilogue
#pragma unroll
    for (int r = 0; r < 16; ++r) accp[r] = 0.f;
    // One pipeline stage: the product of step j (matrix pipe, weights from LDS) interleaved with the epilogue of step j - 1 (vector pipe, its
    // x / dbuf tile from LDS): the two waves of a SIMD are always in the same stage (one barrier per step), so the overlap has to be inside the
    // instruction stream.  MM / EP switch the halves off for the first and the last stage.
    auto stage = [&](int j, auto MM, auto EP) {
        constexpr bool mm = decltype(MM)::value, ep = decltype(EP)::value;
        TSTAMP(93);
        mm_wait_vm<0>();                                          // W(j) and x / dbuf (j - 1), both requested one stage ago; the last stage's stores
        TSTAMP(94);
        mm_barrier();                                             // the only barrier of a stage: every wave is done with the slots refilled below
        TSTAMP(95);
        if (mm) {
            if (j + 1 < j1) issue_w(j + 1);
            issue_xy(j);
        }
        TSTAMP(96);
        const unsigned char* sw = smem + LY::OFF_W + ((j - j0) & 1) * LY::W_BYTES;
        unsigned char* sx = smem + LY::OFF_XY + ((j - 1 - j0) & 1) * LY::XY_BYTES;
        unsigned char* sy = sx + LY::X_BYTES;
        u16 xs[16], ys[16], out[16];
        if (ep) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                xs[r] = *(const u16*)(sx + row * 128 + col * 2);
                ys[r] = *(const u16*)(sy + row * 128 + col * 2);
            }
        }
        u32x4 bf[8];
        if (mm) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) bf[ks] = *(const u32x4*)(sw + dgl_off(wn * 32 + lr, 2 * ks + lh));
        }
        __builtin_amdgcn_sched_barrier(0);                        // every LDS read of the stage is in flight before the first MFMA / VALU instruction
        const bool cok = (j - 1) * LY::BN + col < a.Cin && (j - 1) * LY::BN + col >= a.c_begin;
        float s1 = 0.f, s2 = 0.f;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (mm) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[ks]), __builtin_bit_cast(bf16x8_t, bf[ks]), acc, 0, 0, 0);
            if (ep) {
#pragma unroll
                for (int r = 2 * ks; r < 2 * ks + 2; ++r) {
                    const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float xf = __uint_as_float((unsigned)xs[r] << 16), yf = __uint_as_float((unsigned)ys[r] << 16);
                    const bool keep = (int)cok & (int)(m0 + row < P) & (int)(fmaf(xf, sc, sh) > 0.f);
                    const float Dm = keep ? accp[r] : 0.f;
                    s1 += Dm; s2 = fmaf(Dm, fmaf(xf, a1, a0), s2);
                    out[r] = __builtin_bit_cast(u16, (__bf16)fmaf(sc, Dm, yf));
                }
            }
        }
        if (mm && ep) {
            // pin the interleave: one MFMA (64 matrix-pipe cycles), then the vector work of two pixels
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, SAUNET_DGL_VALU, 0);
            }
        }
        TSTAMP(97);
        if (ep) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                *(u16*)(sy + row * 128 + col * 2) = out[r];
            }
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lh == 0) { float* sp = s_part + ((j - 1 - j0) % LY::MAX_STEPS) * LY::PART_STEP + wm * 128; sp[col] = s1; sp[64 + col] = s2; }
            TSTAMP(98);
            // the wave's own 32 pixels x 32 channels of the dbuf tile back as 16-byte row pieces (32 rows x 4 pieces: two per lane): no other
            // wave's writes are read, and a wave's LDS operations execute in program order
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = lane + i * 64, px = wm * 32 + (q >> 2), cq = wn * 4 + (q & 3);
                if (m0 + px < P && (j - 1) * LY::BN + cq * 8 < a.Cin && (j - 1) * LY::BN + cq * 8 >= a.c_begin)
                    *(u32x4*)(a.y + (size_t)(m0 + px) * a.ldy + (j - 1) * LY::BN + cq * 8) = *(const u32x4*)(sy + px * 128 + cq * 16);
            }
        }
        if (mm) {
            accp = acc;
            sc = nsc; sh = nsh; a1 = na1; a0 = -na0 * na1;
            if (j + 1 < j1) coeffs(j + 1, nsc, nsh, na1, na0);
        }
        TSTAMP(99);
    };
    // fold the row waves' partial sums of steps [first, first + n) (fixed order) and add them to the replicas: (step, channel) pairs over the block
    auto fold = [&](int first, int n) {
        __syncthreads();
        for (int q = tid; q < n * 64; q += NT) {
            const int jj = q >> 6, cl = q & 63, c = (first + jj) * LY::BN + cl;
            if (c >= a.Cin || c < a.c_begin) continue;
            const float* sp = s_part + ((first + jj - j0) % LY::MAX_STEPS) * LY::PART_STEP + cl;
            float t1, t2;
            if (BM == 128) { t1 = ((sp[0] + sp[128]) + (sp[256] + sp[384])); t2 = ((sp[64] + sp[192]) + (sp[320] + sp[448])); }
            else           { t1 = sp[0] + sp[128]; t2 = sp[64] + sp[192]; }
            const size_t ro = (size_t)(blockIdx.x % a.reps) * a.rstride;
            atomicAdd(&a.sums[ro + c], (double)t1);
            atomicAdd(&a.sums[ro + a.Cin + c], (double)t2);
            if (a.ab) {
                const float scc = a.scale[c];
                const size_t ra = (size_t)(blockIdx.x % a.ab_reps) * a.ab_rstride;
                atomicAdd(&a.ab[ra + c], (double)(scc * t1));
                atomicAdd(&a.ab[ra + a.ab_half + c], (double)(scc * t2));
            }
        }
    };
    stage(j0, std::true_type{}, std::false_type{});
    int folded = j0;                                              // steps below this one are in the replicas
    for (int j = j0 + 1; j < j1; ++j) {
        stage(j, std::true_type{}, std::true_type{});             // epilogue of step j - 1
        if (j - folded == LY::MAX_STEPS) { fold(folded, LY::MAX_STEPS); folded = j; }   // the ring is full; the next stage's barrier orders its re-use
    }
    stage(j1, std::false_type{}, std::true_type{});
    fold(folded, j1 - folded);
}

// ---- two layers in one pass (round 5).  The conv1 data gradient is HBM-bound on x (read) and dbuf (read + write), 6 * Cin bytes per pixel and
// layer; layer l - 1 reads and writes the same rows again, one 32-channel chunk shorter.  The only dependency between the two is that chunk: the
// conv2 data gradient of layer l - 1 needs dbuf[:, Cin_l - 32 : Cin_l] with layer l's contribution.  So layer l first runs the kernel above on
// that chunk alone (c_begin = Cin_l - 32: one 64-channel step), then -- after layer l - 1's conv2 data gradient -- this kernel adds BOTH layers'
// contributions to the channels below it in one read-modify-write:  hi = layer l (its dz1 is final: A fragments straight from global memory),
// lo = layer l - 1 (dz1 from G and z1 as above).  Per 64-channel step two products (weights of both layers by DMA), one x / dbuf tile, one
// epilogue with two masks and four sums.  Bytes per pixel of the pair at Cin = 624:  5.5 KB instead of 8.8 KB.
struct DenseDgradPairArgs { DenseDgradArgs lo, hi; };
struct DgPairLayout {
    static constexpr int BM = 128, BN = 64, NW = 8, THREADS = 512;
    static constexpr int A_BYTES = BM * 256, Z_BYTES = BM * 256;
    static constexpr int W_BYTES = BN * 256, X_BYTES = BM * BN * 2, Y_BYTES = BM * BN * 2, STAGE = 2 * W_BYTES + X_BYTES + Y_BYTES;   // 64 KB
    static constexpr int OFF_S0 = 0, OFF_S1 = STAGE;             // slot 0 is the G / z1 space of the prologue
    static_assert(A_BYTES + Z_BYTES <= STAGE, "prologue tiles overlay slot 0 only");
    static constexpr int MAX_STEPS = 4, PART_STEP = 2 * 4 * 128;  // float[MAX_STEPS][layer][4 row waves][2][64]
    static constexpr int OFF_PART = 2 * STAGE;
    static constexpr int LDS = OFF_PART + MAX_STEPS * PART_STEP * 4;   // 144 KB
};

__global__ __launch_bounds__(512) void dense_conv1_dgrad_pair_kernel(DenseDgradPairArgs pa)
{
    using LY = DgPairLayout;
    const DenseDgradArgs& a = pa.lo;
    const DenseDgradArgs& b = pa.hi;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int m0 = blockIdx.x * LY::BM;
    const int nsteps = (a.Cin + LY::BN - 1) / LY::BN;            // the channels of the SHORTER layer: a.Cin = b.Cin - 32 = b.c_begin
    float* s_cf = (float*)(smem + LY::OFF_PART);
    float* s_part = (float*)(smem + LY::OFF_PART);
    const int P = (int)a.P;
    TSTAMP_INIT();
    TSTAMP(90);
    // layer l's A fragments (its dz1 rows, final) straight from global memory: 8 x 16 bytes per lane, consumed at the first product
    u32x4 af2[8];
    {
        const u16* row = b.dz + (size_t)min(m0 + wm * 32 + lr, P - 1) * b.lddz + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) af2[ks] = *(const u32x4*)(row + ks * 16);
    }
    auto issue = [&](int j, int slot_off) {
        const int c0 = j * LY::BN;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int piece = wave * 2 + q;
            {   // weight rows c0 + 4 piece .. + 3 of both layers; rows past Cin repeat the last one
                const int row = piece * 4 + (lane >> 4), c = (lane & 15) ^ (row & 15);
                const size_t off = (size_t)min(c0 + row, a.Cin - 1) * 128 + c * 8;
                mm_dma16(a.w + off, lds0 + slot_off + piece * 1024);
                mm_dma16(b.w + off, lds0 + slot_off + LY::W_BYTES + piece * 1024);
            }
            {
                const int px = piece * 8 + (lane >> 3), ch = lane & 7;
                const size_t m = (size_t)min(m0 + px, P - 1);
                const int cc = min(c0 + ch * 8, a.Cin - 8);
                mm_dma16(a.x + m * a.ldx + cc, lds0 + slot_off + 2 * LY::W_BYTES + piece * 1024);
                mm_dma16(a.y + m * a.ldy + cc, lds0 + slot_off + 2 * LY::W_BYTES + LY::X_BYTES + piece * 1024);
            }
        }
    };
    // layer l - 1's operand: G and z1 tiles by LDS-DMA (fragment-shaped global loads of the same rows measured 6k cycles slower: 64 separate
    // 16-byte segments per instruction), BN2 backward applied in place by the requesting wave, dz1 written out for the deferred weight gradient
    int arow[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = wave * 4 + q, row = piece * 4 + (lane >> 4), c = (lane & 15) ^ (row & 15);
        const size_t m = (size_t)min(m0 + row, P - 1);
        arow[q] = row | (c << 8);
        mm_dma16(a.g + m * a.ldg + c * 8, lds0 + piece * 1024);
        mm_dma16(a.z + m * a.ldz + c * 8, lds0 + LY::A_BYTES + piece * 1024);
    }
    issue(0, LY::OFF_S1);
    for (int k = tid; k < 128; k += LY::THREADS) {
        double S1, S2;
        rep_sum2(a.sums2, a.sums2 + 128, a.reps2, a.rstride2, k, S1, S2);
        const float sc = a.p2[k], mu = a.p2[256 + k], is = a.p2[384 + k];
        const float m1 = (float)(S1 / a.count), m2 = (float)(S2 / a.count);
        s_cf[k] = sc; s_cf[128 + k] = -sc * is * m2; s_cf[256 + k] = -sc * (m1 - mu * is * m2);
        if (blockIdx.x == 0 && a.dgamma2) { a.dbeta2[k] = (float)S1; a.dgamma2[k] = (float)S2; }
    }
    __syncthreads();
    mm_wait_vm<8>();                                              // the G / z1 pieces have landed (the eight requests of step 0 may not have)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = wave * 4 + q, row = arow[q] & 0xff, c = arow[q] >> 8;
        unsigned char* pg = smem + piece * 1024 + lane * 16;
        float G[8], Z[8], d[8];
        Vec16<u16>::unpack(*(const u32x4*)pg, G);
        Vec16<u16>::unpack(*(const u32x4*)(pg + LY::A_BYTES), Z);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 ca = *(const f32x4*)(s_cf + c * 8 + 4 * h), cb = *(const f32x4*)(s_cf + 128 + c * 8 + 4 * h), cc = *(const f32x4*)(s_cf + 256 + c * 8 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[4 * h + e] = fmaf(ca[e], G[4 * h + e], fmaf(cb[e], Z[4 * h + e], cc[e]));
        }
        const u32x4 v = Vec16<u16>::pack(d);
        *(u32x4*)pg = v;
        if (m0 + row < P) *(u32x4*)(a.dz + (size_t)(m0 + row) * a.lddz + c * 8) = v;
    }
    __syncthreads();
    u32x4 af1[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) af1[ks] = *(const u32x4*)(smem + dgl_off(wm * 32 + lr, 2 * ks + lh));
    __syncthreads();

    const int col = wn * 32 + lr;
    auto coeffs = [&](const DenseDgradArgs& L, int j, float& sc, float& sh, float& a1, float& a0) {
        const int chs = min(j * LY::BN + col, a.Cin - 1);
        sc = L.scale[chs]; sh = L.shift[chs]; a1 = L.invstd[chs]; a0 = L.mean[chs];
    };
    float n1[4], n2[4];
    coeffs(a, 0, n1[0], n1[1], n1[2], n1[3]);
    coeffs(b, 0, n2[0], n2[1], n2[2], n2[3]);
    // The partial sums of MAX_STEPS steps are folded (fixed order) into ONE (sum, sum * xhat) pair per thread -- 4 steps x 2 layers x 64 channels =
    // 512 entries -- and kept in registers; the float64 atomics all go out after the last step.  (Issued inside the loop they cost a stage: the
    // next stage's vmcnt(0) waited for their acknowledgement, 12k cycles per fold with 256 workgroups adding to the same 16 replicas.)
    constexpr int MAX_FOLDS = 4;                                  // 16 steps: Cin <= 1024
    float ft1[MAX_FOLDS], ft2[MAX_FOLDS];
    auto fold = [&](int k, int first, int n) {
        __syncthreads();
        const int jj = tid >> 7, layer = (tid >> 6) & 1, cl = tid & 63;
        float t1 = 0.f, t2 = 0.f;
        if (jj < n) {
            const float* sp = s_part + ((first + jj) % LY::MAX_STEPS) * LY::PART_STEP + layer * 512 + cl;
            t1 = ((sp[0] + sp[128]) + (sp[256] + sp[384])); t2 = ((sp[64] + sp[192]) + (sp[320] + sp[448]));
        }
#pragma unroll
        for (int i = 0; i < MAX_FOLDS; ++i) if (i == k) { ft1[i] = t1; ft2[i] = t2; }
    };
    auto publish = [&](int k, int first, int n) {
        const int jj = tid >> 7, layer = (tid >> 6) & 1, cl = tid & 63, c = (first + jj) * LY::BN + cl;
        if (jj >= n || c >= a.Cin) return;
        const DenseDgradArgs& L = layer ? b : a;
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_FOLDS; ++i) if (i == k) { t1 = ft1[i]; t2 = ft2[i]; }
        const size_t ro = (size_t)(blockIdx.x % L.reps) * L.rstride;
        atomicAdd(&L.sums[ro + c], (double)t1);
        atomicAdd(&L.sums[ro + L.Cin + c], (double)t2);
        if (L.ab) {
            const float scc = L.scale[c];
            const size_t ra = (size_t)(blockIdx.x % L.ab_reps) * L.ab_rstride;
            atomicAdd(&L.ab[ra + c], (double)(scc * t1));
            atomicAdd(&L.ab[ra + L.ab_half + c], (double)(scc * t2));
        }
    };
    int folded = 0;
    TSTAMP(92);
    for (int j = 0; j < nsteps; ++j) {
        const int slot_off = (j & 1) ? LY::OFF_S0 : LY::OFF_S1;
        TSTAMP(93);
        mm_wait_vm<0>();
        TSTAMP(94);
        mm_barrier();                                             // the only barrier of a step
        TSTAMP(95);
        if (j + 1 < nsteps) issue(j + 1, (j & 1) ? LY::OFF_S1 : LY::OFF_S0);
        TSTAMP(96);
        const float sc1 = n1[0], sh1 = n1[1], x1 = n1[2], o1 = -n1[3] * n1[2];
        const float sc2 = n2[0], sh2 = n2[1], x2 = n2[2], o2 = -n2[3] * n2[2];
        if (j + 1 < nsteps) { coeffs(a, j + 1, n1[0], n1[1], n1[2], n1[3]); coeffs(b, j + 1, n2[0], n2[1], n2[2], n2[3]); }
        const unsigned char* sw1 = smem + slot_off;
        const unsigned char* sw2 = sw1 + LY::W_BYTES;
        unsigned char* sx = smem + slot_off + 2 * LY::W_BYTES;
        unsigned char* sy = sx + LY::X_BYTES;
        u16 xs[16], ys[16], out[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            xs[r] = *(const u16*)(sx + row * 128 + col * 2);
            ys[r] = *(const u16*)(sy + row * 128 + col * 2);
        }
        f32x16 acc1, acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const u32x4 bf1 = *(const u32x4*)(sw1 + dgl_off(wn * 32 + lr, 2 * ks + lh));
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af1[ks]), __builtin_bit_cast(bf16x8_t, bf1), acc1, 0, 0, 0);
        }
        TSTAMP(97);
        // layer l's product (matrix pipe) runs under layer l - 1's half of the epilogue (vector pipe): the two waves of a SIMD are in the same
        // phase (one barrier per step), so the overlap has to be in the instruction stream
        // Two pixels per instruction: v_pk_fma_f32 / v_pk_add_f32 on register pairs (the accumulator rows r, r + 1 are consecutive pixels of the
        // lane's channel) -- 23 vector instructions per pixel pair instead of 34; with two waves per SIMD in the same phase the epilogue's
        // issue time is the stage's longest part.
        typedef float f2 __attribute__((ext_vector_type(2)));
        const bool cok = j * LY::BN + col < a.Cin;
        f2 s1a = {0.f, 0.f}, s2a = {0.f, 0.f}, s1b = {0.f, 0.f}, s2b = {0.f, 0.f};
        f2 D1[8], xf[8];
        const f2 vsc1 = {sc1, sc1}, vsh1 = {sh1, sh1}, vx1 = {x1, x1}, vo1 = {o1, o1}, vsc2 = {sc2, sc2}, vsh2 = {sh2, sh2}, vx2 = {x2, x2}, vo2 = {o2, o2};
        u32x4 bf2[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bf2[ks] = *(const u32x4*)(sw2 + dgl_off(wn * 32 + lr, 2 * ks + lh));
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af2[ks]), __builtin_bit_cast(bf16x8_t, bf2[ks]), acc2, 0, 0, 0);
            const int r = 2 * ks, row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;        // rows r and r + 1 are pixels row, row + 1
            xf[ks] = f2{__uint_as_float((unsigned)xs[r] << 16), __uint_as_float((unsigned)xs[r + 1] << 16)};
            const f2 t = __builtin_elementwise_fma(xf[ks], vsc1, vsh1);
            const int ok0 = (int)cok & (int)(m0 + row < P), ok1 = (int)cok & (int)(m0 + row + 1 < P);
            D1[ks] = f2{(ok0 & (int)(t[0] > 0.f)) ? acc1[r] : 0.f, (ok1 & (int)(t[1] > 0.f)) ? acc1[r + 1] : 0.f};
            s1a += D1[ks]; s2a = __builtin_elementwise_fma(D1[ks], __builtin_elementwise_fma(xf[ks], vx1, vo1), s2a);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int r = 2 * ks, row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const f2 yf = {__uint_as_float((unsigned)ys[r] << 16), __uint_as_float((unsigned)ys[r + 1] << 16)};
            const f2 t = __builtin_elementwise_fma(xf[ks], vsc2, vsh2);
            const int ok0 = (int)cok & (int)(m0 + row < P), ok1 = (int)cok & (int)(m0 + row + 1 < P);
            const f2 D2 = {(ok0 & (int)(t[0] > 0.f)) ? acc2[r] : 0.f, (ok1 & (int)(t[1] > 0.f)) ? acc2[r + 1] : 0.f};
            s1b += D2; s2b = __builtin_elementwise_fma(D2, __builtin_elementwise_fma(xf[ks], vx2, vo2), s2b);
            // the later layer first: the order its own launch would have added them in
            const f2 o = __builtin_elementwise_fma(vsc1, D1[ks], __builtin_elementwise_fma(vsc2, D2, yf));
            const unsigned pk = pack_bf16x2(o[0], o[1]);
            out[r] = (u16)pk; out[r + 1] = (u16)(pk >> 16);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            *(u16*)(sy + row * 128 + col * 2) = out[r];
        }
        float t1a = s1a[0] + s1a[1], t2a = s2a[0] + s2a[1], t1b = s1b[0] + s1b[1], t2b = s2b[0] + s2b[1];
        t1a += __shfl_xor(t1a, 32, 64); t2a += __shfl_xor(t2a, 32, 64); t1b += __shfl_xor(t1b, 32, 64); t2b += __shfl_xor(t2b, 32, 64);
        if (lh == 0) {
            float* sp = s_part + (j % LY::MAX_STEPS) * LY::PART_STEP + wm * 128;
            sp[col] = t1a; sp[64 + col] = t2a; sp[512 + col] = t1b; sp[512 + 64 + col] = t2b;
        }
        TSTAMP(98);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = lane + i * 64, px = wm * 32 + (q >> 2), cq = wn * 4 + (q & 3);
            if (m0 + px < P && j * LY::BN + cq * 8 < a.Cin)
                *(u32x4*)(a.y + (size_t)(m0 + px) * a.ldy + j * LY::BN + cq * 8) = *(const u32x4*)(sy + px * 128 + cq * 16);
        }
        TSTAMP(99);
        if (j + 1 - folded == LY::MAX_STEPS) { fold(folded / LY::MAX_STEPS, folded, LY::MAX_STEPS); folded = j + 1; }
    }
    if (folded < nsteps) fold(folded / LY::MAX_STEPS, folded, nsteps - folded);
    for (int k = 0; k * LY::MAX_STEPS < nsteps; ++k) publish(k, k * LY::MAX_STEPS, min(LY::MAX_STEPS, nsteps - k * LY::MAX_STEPS));
}

// maps on which the fused layer backward's conv1 data gradient runs the LDS-staged kernel (dense_conv1_dgrad_lds_kernel): up to 4096 runs of 32
// pixels -- block 2 (64 x 64 x 32 images) included: 175.9 -> 165.5 us / layer backward; block 1 is slower with it (424 -> 436)
static bool dense_lds_conv1_range(long P, int Cin)
{
    static const bool lds_on = ab_env_on("SAUNET_DG_LDS");       // A/B (variant builds only)
    static const long lds_maxtp = ab_env_int("SAUNET_DG_LDS_MAXTP", 4097);
    return lds_on && (P + 31) / 32 < lds_maxtp && Cin % 8 == 0 && Cin >= 64;
}

static int launch_dense_dgrad(DenseDgradArgs& a, bool apply, hipStream_t st)
{
    // channels per block: 256 (weights copied to LDS once per block) when every wave gets many pixel tiles, fewer on the
    // low-resolution blocks where the copy would dominate and more blocks are needed to fill the chip
    const long ntp = ((long)a.P + 31) / 32;
    a.group = ntp >= 4096 ? DG_GROUP : (ntp >= 1024 ? 128 : 64);
    static const int force_group = ab_env_int("SAUNET_DG_GROUP", 0), force_small = ab_env_int("SAUNET_DG_GROUP_SMALL", 0);     // A/B (variant builds only)
    if (force_group > 0) a.group = force_group;
    if (force_small > 0 && ntp < 4096) a.group = force_small;
    {   // equal groups: every group runs the same number of 64-channel steps (the step sequence is compiled per step count), so
        // Cin = 320 with at most 256 channels per group is 2 x 160, not 256 + 64
        const int ng = (a.Cin + a.group - 1) / a.group;
        a.group = ((a.Cin + ng - 1) / ng + 31) & ~31;
    }
    static const int lds_bm = ab_env_int("SAUNET_DG_LDS_BM", 128);   // 64: two 4-wave blocks per CU -- measured slower (a third more DMA requests per pixel)

    if (apply && dense_lds_conv1_range((long)a.P, a.Cin) && a.ldg == 128 && a.ldz == 128 && a.lddz == 128 && a.accumulate == 1 && a.relu) {
        // low-resolution maps: the LDS-staged kernel; the 64-channel steps split over blockIdx.y until the chip is full
        const int bm = lds_bm == 128 ? 128 : 64, slots = bm == 128 ? 256 : 512;
        const int tiles = (int)((a.P + bm - 1) / bm), nsteps = (a.Cin + 63) / 64 - a.c_begin / 64;
        int ng = slots / tiles; if (ng < 1) ng = 1; if (ng > nsteps) ng = nsteps;
        const int spg = (nsteps + ng - 1) / ng; ng = (nsteps + spg - 1) / spg;
        {
            static DeviceOnce attr0;
            if (attr0.first()) {
                (void)hipFuncSetAttribute((const void*)dense_conv1_dgrad_lds_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, DgLdsLayout<128>::LDS);
                (void)hipFuncSetAttribute((const void*)dense_conv1_dgrad_lds_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, DgLdsLayout<64>::LDS);
            }
            if (bm == 128) hipLaunchKernelGGL(dense_conv1_dgrad_lds_kernel<128>, dim3(tiles, ng), dim3(512), DgLdsLayout<128>::LDS, st, a);
            else           hipLaunchKernelGGL(dense_conv1_dgrad_lds_kernel<64>, dim3(tiles, ng), dim3(256), DgLdsLayout<64>::LDS, st, a);
            SAUNET_CHECK_LAUNCH(bm == 128 ? "dense_conv1_dgrad_lds_kernel<128>" : "dense_conv1_dgrad_lds_kernel<64>");
            return SAUNET_OK;
        }
    }
    if (a.c_begin) return set_error(SAUNET_BAD_SHAPE, "dense conv1 data gradient: a channel window needs the LDS-staged kernel (saunet_dense_layer_backward_pair_supported)");
    const int groups = (a.Cin + a.group - 1) / a.group;
    const int gcp = ((a.Cin < a.group ? a.Cin : a.group) + 31) & ~31;
    const int ns = (gcp + 63) / 64;                      // 64-channel steps of a full group (a shorter last group masks its surplus steps)
    size_t lds = (size_t)gcp * DG_WPITCH * 2 + sizeof(float) * 4 * gcp + (apply ? sizeof(float) * 3 * 128 : 0);
    const size_t red_bytes = (size_t)DG_WAVES * ns * 4 * 64 * sizeof(float);       // the end-of-kernel fold re-uses the weight area
    if (lds < red_bytes) lds = red_bytes;
    // two resident blocks per CU (registers): the grid must not EXCEED that capacity -- a handful of surplus blocks start only when the first
    // ones retire and double the kernel time (515 blocks for Cin = 640 at 32 x 32 ran 60 us, 512 blocks for Cin = 992 61 us); at least one
    // pixel tile per wave
    long bx = 512 / groups; const long maxbx = ((long)a.P + 32 * DG_WAVES - 1) / (32 * DG_WAVES);
    static const int small_blocks = ab_env_int("SAUNET_DG_BLOCKS_SMALL", 0);       // A/B (variant builds only): total blocks on the small maps
    if (small_blocks > 0 && ntp < 4096) bx = small_blocks / groups;
    if (bx > maxbx) bx = maxbx; if (bx < 1) bx = 1;
    static DeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)dense_dgrad_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_dgrad_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_dgrad_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_dgrad_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_dgrad_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_dgrad_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_dgrad_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_dgrad_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    }
    const dim3 grid((unsigned)bx, groups), block(DG_WAVES * 64);
    if (apply) {
        if (ns == 1) hipLaunchKernelGGL((dense_dgrad_kernel<1, true>), grid, block, lds, st, a);
        else if (ns == 2) hipLaunchKernelGGL((dense_dgrad_kernel<2, true>), grid, block, lds, st, a);
        else if (ns == 3) hipLaunchKernelGGL((dense_dgrad_kernel<3, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((dense_dgrad_kernel<4, true>), grid, block, lds, st, a);
    } else {
        if (ns == 1) hipLaunchKernelGGL(dense_dgrad_kernel<1>, grid, block, lds, st, a);
        else if (ns == 2) hipLaunchKernelGGL(dense_dgrad_kernel<2>, grid, block, lds, st, a);
        else if (ns == 3) hipLaunchKernelGGL(dense_dgrad_kernel<3>, grid, block, lds, st, a);
        else hipLaunchKernelGGL(dense_dgrad_kernel<4>, grid, block, lds, st, a);
    }
    const KName kn("dense_dgrad_kernel", ns > 4 ? 4 : ns, apply);
    SAUNET_CHECK_LAUNCH(kn.s);
    return SAUNET_OK;
}

int dense_dgrad_forward(const saunet_conv_desc* d, const void* x, const void* w, void* y, const saunet_bn_epilogue* epi, hipStream_t st)
{
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)epi->bn_x) & 15)
        return set_error(SAUNET_BAD_ALIGN, "dense dgrad: operands must be 16-byte aligned");
    DenseDgradArgs a{};
    a.g = (const u16*)x; a.ldg = d->ldx; a.w = (const u16*)w; a.x = (const u16*)epi->bn_x; a.ldx = epi->ld_bn_x; a.y = (u16*)y; a.ldy = d->ldy;
    a.scale = epi->scale; a.shift = epi->shift; a.mean = epi->mean; a.invstd = epi->invstd;
    a.sums = epi->sums; a.reps = epi->sums_replicas > 1 ? epi->sums_replicas : 1; a.rstride = epi->sums_rstride;
    a.P = (unsigned)((long)d->N * d->H * d->W); a.Cin = d->Cout; a.relu = epi->relu; a.accumulate = epi->accumulate;
    return launch_dense_dgrad(a, false, st);
}

int bn_backward_correct_ab(int dtype, const void* d, int ldd, const void* x, int ldx, void* y, int ldy, const double* ab, int ab_replicas,
                           int ab_rstride, int ab_half, double count, const float* xs, const float* xt, int64_t pixels, int C, hipStream_t st);

}  // namespace saunet

using namespace saunet;

static int dense_layer_check(const saunet_dense_layer_bwd* l, const char* who)
{
    if (!l || l->N < 1 || l->H < 1 || l->W < 1 || l->Cin < 8 || l->Cin % 8 || l->Ctot % 8 || l->Cin + 32 > l->Ctot || l->Cin > 2048 ||
        (long)l->N * l->H * l->W >= (1L << 31) || l->count < 1.0)
        return set_error(SAUNET_BAD_SHAPE, "%s: bad geometry", who);
    if (!l->buf || !l->dbuf || !l->xhat || !l->ab || !l->z1 || !l->g || !l->p2 || !l->sums2 || l->ab_replicas < 1 || l->sums2_replicas < 1 || l->ld_xhat < l->Ctot)
        return set_error(SAUNET_BAD_SHAPE, "%s: incomplete descriptor", who);
    if (((uintptr_t)l->buf | (uintptr_t)l->dbuf | (uintptr_t)l->z1 | (uintptr_t)l->g | (uintptr_t)l->dz1 | (uintptr_t)l->dz2 |
         (uintptr_t)l->w1_dgrad | (uintptr_t)l->w2_dgrad) & 15)
        return set_error(SAUNET_BAD_ALIGN, "%s: operands must be 16-byte aligned", who);
    return SAUNET_OK;
}

extern "C" {

int saunet_dense_layer_backward_conv2(const saunet_dense_layer_bwd* l, void* stream)
{
    if (int rc = dense_layer_check(l, "dense_layer_backward_conv2")) return rc;
    if (!l->w2_dgrad) return set_error(SAUNET_BAD_SHAPE, "dense_layer_backward_conv2: no packed weights");
    hipStream_t st = (hipStream_t)stream;
    const long P = (long)l->N * l->H * l->W;
    const u16* chunk = (const u16*)l->dbuf + l->Cin;
    DenseDgrad3Args a{};
    a.w = (const u16*)l->w2_dgrad; a.z = (const u16*)l->z1; a.ldz = 128; a.y = (u16*)l->g; a.ldy = 128;
    a.scale = l->p2; a.shift = l->p2 + 128; a.mean = l->p2 + 256; a.invstd = l->p2 + 384;
    a.sums = l->sums2; a.reps = l->sums2_replicas; a.rstride = l->sums2_rstride;
    a.N = l->N; a.H = l->H; a.W = l->W; a.P = (unsigned)P; a.relu = 1;
    a.dW = FastDiv::make((unsigned)l->W); a.dHW = FastDiv::make((unsigned)(l->H * l->W));
    if (l->dz2 == nullptr) {                       // nothing to correct: read the chunk in place
        a.g = chunk; a.ldg = l->Ctot;
        return launch_dense_dgrad3(a, false, st);
    }
    const double* ab = l->ab + l->Cin;
    static const bool halo_corr = ab_env_on("SAUNET_DGRAD3_HALO_CORR");     // A/B (variant builds only): 0 = the separate correction pass of round 5
    if (!halo_corr && dense_dgrad3_halo(l->N, l->H, l->W)) {     // large maps: a streaming correction pass into dz2, then the LDS-DMA staged kernel over dz2
        // (the correction folded into the halo buffer of a non-transposed form of that kernel was built and measured slower: registers --
        // scripts/probes/dense_dgrad3_halo2_rejected.hip)
        if (int rc = bn_backward_correct_ab(SAUNET_BF16, chunk, l->Ctot, (const u16*)l->buf + l->Cin, l->Ctot, l->dz2, 32, ab, l->ab_replicas, l->ab_rstride,
                                            l->Ctot, l->count, l->xhat + l->Cin, l->xhat + l->ld_xhat + l->Cin, P, 32, st)) return rc;
        a.g = (const u16*)l->dz2; a.ldg = 32;
        return launch_dense_dgrad3(a, false, st);
    }
    a.g = chunk; a.ldg = l->Ctot;
    a.xc = (const u16*)l->buf + l->Cin; a.ldxc = l->Ctot;
    a.ab = ab; a.ab_reps = l->ab_replicas; a.ab_rstride = l->ab_rstride; a.ab_half = l->Ctot; a.count = l->count;
    a.xs = l->xhat + l->Cin; a.xt = l->xhat + l->ld_xhat + l->Cin;
    a.gc = (u16*)l->dz2; a.ldgc = 32;
    return launch_dense_dgrad3(a, true, st);
}

int saunet_dense_layer_backward_conv1(const saunet_dense_layer_bwd* l, void* stream)
{
    if (int rc = dense_layer_check(l, "dense_layer_backward_conv1")) return rc;
    if (!l->w1_dgrad || !l->p1 || !l->sums1 || !l->dz1 || l->sums1_replicas < 1) return set_error(SAUNET_BAD_SHAPE, "dense_layer_backward_conv1: incomplete descriptor");
    DenseDgradArgs a{};
    a.g = (const u16*)l->g; a.ldg = 128; a.w = (const u16*)l->w1_dgrad; a.x = (const u16*)l->buf; a.ldx = l->Ctot; a.y = (u16*)l->dbuf; a.ldy = l->Ctot;
    a.scale = l->p1; a.shift = l->p1 + l->Cin; a.mean = l->p1 + 2 * l->Cin; a.invstd = l->p1 + 3 * l->Cin;
    a.sums = l->sums1; a.reps = l->sums1_replicas; a.rstride = l->sums1_rstride;
    a.P = (unsigned)((long)l->N * l->H * l->W); a.Cin = l->Cin; a.relu = 1; a.accumulate = 1;
    a.z = (const u16*)l->z1; a.ldz = 128; a.dz = (u16*)l->dz1; a.lddz = 128;
    a.sums2 = l->sums2; a.reps2 = l->sums2_replicas; a.rstride2 = l->sums2_rstride; a.p2 = l->p2; a.count = l->count;
    a.dgamma2 = l->dgamma2; a.dbeta2 = l->dbeta2;
    a.ab = l->ab; a.ab_reps = l->ab_replicas; a.ab_rstride = l->ab_rstride; a.ab_half = l->Ctot;
    a.c_begin = l->c_begin;
    if (l->c_begin < 0 || l->c_begin >= l->Cin || l->c_begin % 32) return set_error(SAUNET_BAD_SHAPE, "dense_layer_backward_conv1: bad channel window");
    return launch_dense_dgrad(a, true, (hipStream_t)stream);
}

int saunet_dense_layer_backward_pair_supported(const saunet_dense_layer_bwd* hi)
{
    if (!hi) return 0;
    const long P = (long)hi->N * hi->H * hi->W;
    // the LDS-staged kernel's range (launch_dense_dgrad), one 128-pixel tile per CU at least, and a shorter layer of at least one step
    return dense_lds_conv1_range(P, hi->Cin) && (P + 127) / 128 >= 192 && hi->Cin - 32 >= 64 && hi->Cin % 32 == 0;
}

int saunet_dense_layer_backward_conv1_pair(const saunet_dense_layer_bwd* hi, const saunet_dense_layer_bwd* lo, void* stream)
{
    if (int rc = dense_layer_check(hi, "dense_layer_backward_conv1_pair")) return rc;
    if (int rc = dense_layer_check(lo, "dense_layer_backward_conv1_pair")) return rc;
    if (!saunet_dense_layer_backward_pair_supported(hi) || lo->Cin != hi->Cin - 32 || lo->N != hi->N || lo->H != hi->H || lo->W != hi->W || lo->Ctot != hi->Ctot ||
        lo->buf != hi->buf || lo->dbuf != hi->dbuf || lo->ab != hi->ab)
        return set_error(SAUNET_BAD_SHAPE, "dense_layer_backward_conv1_pair: not a supported pair of consecutive layers of one block");
    if (!hi->w1_dgrad || !hi->p1 || !hi->sums1 || !hi->dz1 || !lo->w1_dgrad || !lo->p1 || !lo->sums1 || !lo->dz1 || hi->sums1_replicas < 1 || lo->sums1_replicas < 1)
        return set_error(SAUNET_BAD_SHAPE, "dense_layer_backward_conv1_pair: incomplete descriptor");
    auto fill = [](const saunet_dense_layer_bwd* l, DenseDgradArgs& a) {
        a.g = (const u16*)l->g; a.ldg = 128; a.w = (const u16*)l->w1_dgrad; a.x = (const u16*)l->buf; a.ldx = l->Ctot; a.y = (u16*)l->dbuf; a.ldy = l->Ctot;
        a.scale = l->p1; a.shift = l->p1 + l->Cin; a.mean = l->p1 + 2 * l->Cin; a.invstd = l->p1 + 3 * l->Cin;
        a.sums = l->sums1; a.reps = l->sums1_replicas; a.rstride = l->sums1_rstride;
        a.P = (unsigned)((long)l->N * l->H * l->W); a.Cin = l->Cin; a.relu = 1; a.accumulate = 1;
        a.z = (const u16*)l->z1; a.ldz = 128; a.dz = (u16*)l->dz1; a.lddz = 128;
        a.sums2 = l->sums2; a.reps2 = l->sums2_replicas; a.rstride2 = l->sums2_rstride; a.p2 = l->p2; a.count = l->count;
        a.dgamma2 = l->dgamma2; a.dbeta2 = l->dbeta2;
        a.ab = l->ab; a.ab_reps = l->ab_replicas; a.ab_rstride = l->ab_rstride; a.ab_half = l->Ctot;
    };
    DenseDgradPairArgs pa{};
    fill(lo, pa.lo); fill(hi, pa.hi);
    const int tiles = (int)((pa.lo.P + 127) / 128);
    static DeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute((const void*)dense_conv1_dgrad_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DgPairLayout::LDS);
    hipLaunchKernelGGL(dense_conv1_dgrad_pair_kernel, dim3(tiles), dim3(512), DgPairLayout::LDS, (hipStream_t)stream, pa);
    SAUNET_CHECK_LAUNCH("dense_conv1_dgrad_pair_kernel");
    return SAUNET_OK;
}

}  // extern "C"

SAUNET_TIMING_READER(dense_dgrad)
