// REJECTED (round 4) -- kept as a probe, not part of the library.  Correct (tests/test_hip_ops.py + test_hip_dense.py pass with it hooked in
// front of conv3x3_res_fwd_kernel in tile_forward) but SLOWER than the LDS-resident kernel on every layer it targets:
//     DenseNet conv2 forward 128 -> 32, block 1: 91 us (107 before the commit path was made branch-free) against 72 us; block 2 36 / 24; blocks 3-4 25 / 18
//     res1 64 -> 64 @256^2: 284-314 us against 233;  res2 103 / 89;  dec0 188 / 130
// Why: ONE wave per SIMD (the price of 288 weight registers) leaves nothing to cover a stall.  hipcc keeps about two ds_read_b128 in flight
// in the unrolled tap loop, so every `s_waitcnt lgkmcnt` in front of an MFMA pair exposes the full LDS latency (~130 cycles against 64 cycles
// of MFMA issue), and the interleaved commit pieces (ds_write + 8-wide transform) sit in the same in-order LDS queue.  The LDS-resident kernel
// reads twice the bytes per MFMA but has two waves per SIMD to overlap them.  Hand-scheduled fragment prefetch (sched_barrier groups as in
// conv_mm.hip) might recover it; not attempted.
//
// conv3x3_rw_kernel: 3x3 stride-1 convolution with REGISTER-RESIDENT weights for the small-K x small-N layers of the step
// (DenseNet conv2 128 -> 32, torchvision _DenseLayer as used at /root/reference/models/models.py:306-313; the shape stream's
// ResBlock convolutions 64 -> 64 / 32 -> 32, /root/reference/models/resnet.py:30-59 via models/models.py:316,322), bf16.
//
// Why: the LDS-resident kernel (conv3x3_res_fwd_kernel) reads TWO fresh 1 KB fragments from LDS per MFMA on these layers (one pixel
// fragment, one weight fragment; its 32 x 32 wave tile re-uses neither), which is the LDS read rate for as long as the MFMAs take, and it
// runs commit / barrier / MFMA / epilogue as separate phases because weights + one halo fill the LDS (s_memtime stamps: matrix cores busy
// 29 % of a unit on conv2 forward; scripts/probes/res_fwd_double_buffer_rejected.patch documents the double-buffered attempt that failed
// on the same LDS port).  K x N is only 36 864 weights here = 72 MFMA A-fragments, and gfx950 gives ONE wave per SIMD 512 registers:
//   * 4 waves per workgroup, one workgroup per CU, every wave holds ALL weight fragments (<= 72 x 4 registers) for the kernel's lifetime
//     and owns 64 pixels (four rows) of each 16 x 16 tile: one LDS read (the pixel fragment) per MFMA instead of two;
//   * the LDS holds nothing but two halo buffers, so the next unit's commit (BN+ReLU prologue, LDS write) and the prefetch request behind
//     it are spread over the MFMA taps of the current unit -- the wave's own VALU / LDS-write work issues while its MFMAs are in the pipe;
//   * transposed product D[out channel][pixel] (weights = A operand): after one v_permlane32_swap per value pair a lane owns 8 consecutive
//     output channels of one pixel, so the epilogue stores 16-byte row pieces straight from registers -- no LDS staging, ONE barrier per unit;
//   * BatchNorm statistics / BN-backward sums: transposing row reduction into registers that live for the wave's lifetime (common.h).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace saunet {

struct RwArgs {
    const u16* x; const u16* w; u16* y;
    const float* bias; const float* pro_scale; const float* pro_shift;
    double* stat_sum; double* stat_sumsq; int stat_replicas, stat_rstride;
    int N, H, W, Cin, ldx, Cout, ldy;
    int pro_relu, act_relu;
    int tiles_x, tiles_y;
    saunet_bn_epilogue epi;
    saunet_bn_prologue bnp;
};

constexpr int RW_HP = 18, RW_NPIX = RW_HP * RW_HP, RW_NT = 256;
constexpr int rw_rowb(int cpru) { return ((RW_HP * (cpru * 16 + 16) + 255) / 256) * 256; }   // halo rows start on 256 B multiples (see conv_tile.hip)

// CINP: padded input channels (32 / 64 / 128); NTL: 32-channel output tiles (1 / 2); BNEPI: BatchNorm-backward reduction epilogue.
// WREG: weight fragments kept in registers; the 72-fragment layers do not leave the compiler enough of the 512 registers for everything
// else (60-176 spilled registers with all 72 resident), so their last fragments live in LDS as a lane-linear copy (1 KB, conflict-free
// ds_read_b128 each) -- those MFMAs read two fragments like the LDS-resident kernel, the others one.
template <int CINP, int NTL, bool BNEPI, int WREG>
__global__ __launch_bounds__(RW_NT, 1) void conv3x3_rw_kernel(RwArgs a)
{
    constexpr int CPRU = CINP >= 64 ? 8 : 4;           // 16-byte chunks per pixel and unit
    constexpr int KCU = CPRU * 8;                      // channels per unit
    constexpr int NU = CINP / KCU;                     // units per tile
    constexpr int SS = CPRU / 2;                       // 16-deep MFMA steps per tap and unit
    constexpr int NFRAG = NU * 9 * SS * NTL;
    static_assert(NFRAG <= 72, "all weight fragments must fit in registers");
    constexpr int NREG = WREG < NFRAG ? WREG : NFRAG, NLDS = NFRAG - NREG;
    constexpr int PITCH = CPRU * 16 + 16, ROWB = rw_rowb(CPRU), HALO_BYTES = RW_HP * ROWB;
    constexpr int H_ITERS = (RW_NPIX * CPRU + RW_NT - 1) / RW_NT;
    constexpr int SLOTS = 9 * SS;
    constexpr int C0 = SLOTS - 2 * H_ITERS - 2 > 0 ? SLOTS - 2 * H_ITERS - 2 : 0;      // first commit slot: as late as the pieces still fit (the loads were requested one unit ago)
    constexpr int CSTEP = (SLOTS - C0) / H_ITERS > 1 ? 2 : 1;
    static_assert(C0 + CSTEP * (H_ITERS - 1) < SLOTS, "commit pieces must fit behind the MFMA slots");
    constexpr int NOUT = NTL * 32;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    unsigned char* s_halo = smem;                                   // [2][HALO_BYTES]
    float* s_pro = (float*)(smem + 2 * HALO_BYTES);                 // [2][CINP] prologue scale / shift
    float* s_par = s_pro + 2 * CINP;                                // BNEPI: [4][NOUT] scale, shift, invstd, -mean*invstd ; else [NOUT] bias
    unsigned char* s_wl = (unsigned char*)(s_par + 4 * NOUT);       // [NLDS][64 lanes][16 B] weight fragments that did not fit the registers
    unsigned char* s_dummy = s_wl + NLDS * 1024;                    // [256][16 B] sink of the commit pieces beyond the halo

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int ntile = a.tiles_x * a.tiles_y * a.N;
    const int per = (ntile + gridDim.x - 1) / gridDim.x;
    const int it0 = blockIdx.x * per, it1 = min(it0 + per, ntile);
    if (it0 >= it1) return;
    const int nunits = (it1 - it0) * NU;
    const bool has_pro = a.pro_scale != nullptr || a.bnp.gamma != nullptr;
    const float relu_lo = has_pro && a.pro_relu ? 0.f : -__builtin_inff();
    const int chunk = tid % CPRU;

    // ---- halo staging: global -> registers (one unit ahead) -> BN+ReLU -> LDS.  Piece i of this thread = halo pixel (tid / CPRU + i * 256 / CPRU), chunk tid % CPRU
    u32x4 hreg[H_ITERS];
    int p_u = 0, p_txi, p_tyi, p_n;        // prefetch cursor: unit inside the tile, tile coordinates
    int m_u = 0, m_txi, m_tyi, m_n;        // commit cursor (the unit the registers hold)
    int c_u = 0, c_txi, c_tyi, c_n;        // compute cursor
    {
        int bt = it0;
        p_txi = bt % a.tiles_x; bt /= a.tiles_x;
        p_tyi = bt % a.tiles_y; p_n = bt / a.tiles_y;
        m_txi = c_txi = p_txi; m_tyi = c_tyi = p_tyi; m_n = c_n = p_n;
    }
    auto advance = [&](int& u, int& txi, int& tyi, int& n) {
        if (++u == NU) { u = 0; if (++txi == a.tiles_x) { txi = 0; if (++tyi == a.tiles_y) { tyi = 0; ++n; } } }
    };
    auto piece_xy = [&](int i, int& hy, int& hx) -> bool {
        const int pix = tid / CPRU + i * (RW_NT / CPRU);
        hy = pix / RW_HP; hx = pix - hy * RW_HP;
        return pix < RW_NPIX;
    };
    auto issue_halo = [&]() {
        const int c = p_u * KCU + chunk * 8;
        const bool cok = c < a.Cin;
        const int y0 = p_tyi * 16 - 1, x0 = p_txi * 16 - 1;
        const u16* img = a.x + (size_t)p_n * a.H * a.W * a.ldx + (cok ? c : 0);
#pragma unroll
        for (int i = 0; i < H_ITERS; ++i) {
            int hy, hx;
            const bool in = piece_xy(i, hy, hx);
            const int iy = y0 + hy, ix = x0 + hx;
            const bool ok = in & cok & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
            hreg[i] = *(const u32x4*)(ok ? img + ((size_t)iy * a.W + ix) * a.ldx : a.x);
        }
        advance(p_u, p_txi, p_tyi, p_n);
    };
    struct ProV { float sc[8], sh[8]; };
    auto pro_vectors = [&](int u, ProV& pv) {          // (identity coefficients without a prologue: branch-free commit, see below)
        const int c = u * KCU + chunk * 8;
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
            const f32x4 s4 = *(const f32x4*)(s_pro + c + j), t4 = *(const f32x4*)(s_pro + CINP + c + j);
#pragma unroll
            for (int q = 0; q < 4; ++q) { pv.sc[j + q] = s4[q]; pv.sh[j + q] = t4[q]; }
        }
    };
    auto commit_piece = [&](int i, const ProV& pv, unsigned char* dst) {       // of the unit at the commit cursor
        int hy, hx;
        const bool in = piece_xy(i, hy, hx);
        const int iy = m_tyi * 16 - 1 + hy, ix = m_txi * 16 - 1 + hx;
        const bool ok = ((m_u * KCU + chunk * 8) < a.Cin) & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
        float f[8];
        Vec16<u16>::unpack(hreg[i], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], pv.sc[j], pv.sh[j]), relu_lo);
        const u32x4 v = Vec16<u16>::pack(f);
        // no branch here: the pieces sit between the MFMA slots of a fully unrolled loop, and any control flow there cuts the loop into basic
        // blocks (the fragment reads are then no longer hoisted over the MFMAs in front of them).  Pieces beyond the halo go to a dummy slot.
        unsigned char* p = in ? dst + hy * ROWB + hx * PITCH + chunk * 16 : s_dummy + tid * 16;
        *(u32x4*)p = ok ? v : u32x4{0u, 0u, 0u, 0u};
    };

    issue_halo();                                    // unit 0 travels while the weights are fetched

    // ---- all weight fragments of the layer: A operand rows = output channel nt * 32 + lr, 8 input channels (2s + lh) * 8 of (unit, tap)
    auto load_w = [&](int f) -> u32x4 {
        const int nt = f % NTL, s = (f / NTL) % SS, tap = (f / (NTL * SS)) % 9, u = f / (NTL * SS * 9);
        const int row = nt * 32 + lr, c = u * KCU + (2 * s + lh) * 8;
        const bool ok = row < a.Cout && c < a.Cin;
        const u32x4 v = *(const u32x4*)(ok ? a.w + ((size_t)row * 9 + tap) * a.Cin + c : a.w);
        return ok ? v : u32x4{0u, 0u, 0u, 0u};
    };
    for (int f = NREG + wave; f < NFRAG; f += RW_NT / 64) *(u32x4*)(s_wl + (f - NREG) * 1024 + lane * 16) = load_w(f);
    u32x4 wf[NREG];
#pragma unroll
    for (int f = 0; f < NREG; ++f) wf[f] = load_w(f);
    if (a.bnp.gamma != nullptr) bn_prologue_fill<RW_NT>(a.bnp, a.Cin, CINP, s_pro, blockIdx.x == 0);
    else {
        for (int i = tid; i < CINP; i += RW_NT) {
            s_pro[i] = has_pro ? (i < a.Cin ? a.pro_scale[i] : 0.f) : 1.f;
            s_pro[CINP + i] = has_pro && i < a.Cin ? a.pro_shift[i] : 0.f;
        }
    }
    if constexpr (BNEPI) {
        for (int i = tid; i < NOUT; i += RW_NT) {
            const bool ok = i < a.Cout;
            const float is = ok ? a.epi.invstd[i] : 0.f;
            s_par[i] = ok ? a.epi.scale[i] : 0.f; s_par[NOUT + i] = ok ? a.epi.shift[i] : 0.f;
            s_par[2 * NOUT + i] = is; s_par[3 * NOUT + i] = ok ? -a.epi.mean[i] * is : 0.f;
        }
    } else {
        for (int i = tid; i < NOUT; i += RW_NT) s_par[i] = (a.bias != nullptr && i < a.Cout) ? a.bias[i] : 0.f;
    }
    __syncthreads();
    {
        ProV pv;
        pro_vectors(0, pv);
#pragma unroll
        for (int i = 0; i < H_ITERS; ++i) commit_piece(i, pv, s_halo);
        advance(m_u, m_txi, m_tyi, m_n);
        if (nunits > 1) issue_halo();
    }
    __syncthreads();

    f32x16 acc[2][NTL];
    // two BatchNorm sums per output channel, register-resident for the wave's lifetime: red[nt][r] = this lane's transposed partial (row_transpose_sum)
    float red[NTL][2];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) { red[nt][0] = 0.f; red[nt][1] = 0.f; }
    const bool sd0 = lane & 8, sd1 = lane & 4, sd2 = lane & 1, sd3 = lane & 2;
    // B fragment base of the wave's two 32-pixel sub-tiles: tile rows 4 * wave + 2 * sub + (lr >> 4), column lr & 15
    int bbase[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) bbase[sub] = (4 * wave + 2 * sub + (lr >> 4)) * ROWB + (lr & 15) * PITCH + lh * 16;
    const bool do_stats = !BNEPI && a.stat_sum != nullptr;

    auto unit_body = [&](auto u_c, int unit) {
        constexpr int U = decltype(u_c)::value;          // unit inside the tile (compile time: it selects the weight registers)
        const unsigned char* hb = s_halo + (unit & 1) * HALO_BYTES;
        unsigned char* hn = s_halo + ((unit & 1) ^ 1) * HALO_BYTES;
        if constexpr (U == 0) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[sub][nt][r] = 0.f;
        }
        const bool more = unit + 1 < nunits;
        ProV pv;
        pro_vectors(more ? m_u : 0, pv);
        // BNEPI: the bn_x pieces of the epilogue are requested at the top of the tile's last unit
        u32x4 bnx[BNEPI ? 2 * NTL * 2 : 1];
        const size_t opix0 = ((size_t)c_n * a.H + c_tyi * 16 + 4 * wave + (lr >> 4)) * a.W + c_txi * 16 + (lr & 15);     // sub-tile 0; sub-tile 1 is two rows down
        if constexpr (BNEPI && U == NU - 1) {
            const u16* bx = (const u16*)a.epi.bn_x;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int c = nt * 32 + 16 * r + 8 * lh;
                        bnx[(sub * NTL + nt) * 2 + r] = *(const u32x4*)(bx + (opix0 + (size_t)sub * 2 * a.W) * a.epi.ld_bn_x + (c < a.Cout ? c : 0));
                    }
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int aoff = (tap / 3) * ROWB + (tap % 3) * PITCH;
#pragma unroll
            for (int s = 0; s < SS; ++s) {
                u32x4 bf[2];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) bf[sub] = *(const u32x4*)(hb + bbase[sub] + aoff + s * 32);
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
                    const int f = ((U * 9 + tap) * SS + s) * NTL + nt;
                    u32x4 wv;
                    if (f < NREG) wv = wf[f < NREG ? f : 0]; else wv = *(const u32x4*)(s_wl + (f - NREG) * 1024 + lane * 16);
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
                        acc[sub][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wv), __builtin_bit_cast(bf16x8_t, bf[sub]), acc[sub][nt], 0, 0, 0);
                }
                const int slot = tap * SS + s;
                if (slot >= C0 && (slot - C0) % CSTEP == 0 && (slot - C0) / CSTEP < H_ITERS) commit_piece((slot - C0) / CSTEP, pv, hn);     // (after the last unit: stale registers into a buffer nobody reads)
            }
        }
        if (more) advance(m_u, m_txi, m_tyi, m_n);
        if (unit + 2 < nunits) issue_halo();
        if constexpr (U == NU - 1) {
            // ---- epilogue of the tile, straight from the accumulators
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                u16* yrow = a.y + (opix0 + (size_t)sub * 2 * a.W) * a.ldy + 8 * lh;
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        float G[8];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[sub][nt][8 * r + q]), __float_as_uint(acc[sub][nt][8 * r + 4 + q]), false, false);
                            G[q] = __uint_as_float(sw[0]); G[4 + q] = __uint_as_float(sw[1]);
                        }
                        const int cp = nt * 32 + 16 * r + 8 * lh;          // this lane's 8 consecutive output channels
                        const bool cok = cp < a.Cout;
                        float o[8], e1[8], e2[8];
                        if constexpr (BNEPI) {
                            float zf[8];
                            Vec16<u16>::unpack(bnx[(sub * NTL + nt) * 2 + r], zf);
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const f32x4 sc = *(const f32x4*)(s_par + cp + 4 * h), sh = *(const f32x4*)(s_par + NOUT + cp + 4 * h);
                                const f32x4 a1 = *(const f32x4*)(s_par + 2 * NOUT + cp + 4 * h), a0 = *(const f32x4*)(s_par + 3 * NOUT + cp + 4 * h);
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int e = 4 * h + q;
                                    const bool keep = cok && (!a.epi.relu || fmaf(zf[e], sc[q], sh[q]) > 0.f);
                                    const float Gv = keep ? G[e] : 0.f;
                                    e1[e] = Gv; e2[e] = Gv * fmaf(zf[e], a1[q], a0[q]);
                                    o[e] = Gv;
                                }
                            }
                            red[nt][r] += row_transpose_sum(e1, e2, sd0, sd1, sd2, sd3);
                        } else {
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const f32x4 bv = *(const f32x4*)(s_par + cp + 4 * h);
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int e = 4 * h + q;
                                    const float v = cok ? G[e] : 0.f;
                                    e1[e] = v; e2[e] = v * v;
                                    o[e] = a.act_relu ? fmaxf(v + bv[q], 0.f) : v + bv[q];
                                }
                            }
                            if (do_stats) red[nt][r] += row_transpose_sum(e1, e2, sd0, sd1, sd2, sd3);
                        }
                        if (cok) *(u32x4*)(yrow + nt * 32 + 16 * r) = Vec16<u16>::pack(o);
                    }
            }
        }
        advance(c_u, c_txi, c_tyi, c_n);
        __syncthreads();            // unit end: the other buffer is complete, this one is free
    };
    for (int unit = 0; unit < nunits; unit += NU) {
        unit_body(std::integral_constant<int, 0>{}, unit);
        if constexpr (NU > 1) unit_body(std::integral_constant<int, 1>{}, unit + 1);
    }

    // ---- block-level fold of the register sums through the (now idle) halo area, one float64 atomic per channel and statistic
    if (!BNEPI && !do_stats) return;
    float* s_red = (float*)smem;                         // [wave][nt * 2 + r][64 lanes]
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
        for (int r = 0; r < 2; ++r) s_red[(wave * (NTL * 2) + nt * 2 + r) * 64 + lane] = red[nt][r];
    __syncthreads();
    for (int c = tid; c < NOUT; c += RW_NT) {
        if (c >= a.Cout) continue;
        const int reg = (c >> 5) * 2 + ((c >> 4) & 1);
        const int l0 = 32 * ((c >> 3) & 1) + 4 * (c & 1) + 2 * ((c >> 2) & 1) + ((c >> 1) & 1);
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < RW_NT / 64; ++w)
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const float* q = s_red + (w * (NTL * 2) + reg) * 64 + l0 + 16 * row;
                t1 += q[0]; t2 += q[8];
            }
        if constexpr (BNEPI) {
            const size_t ro = (size_t)(blockIdx.x % a.epi.sums_replicas) * a.epi.sums_rstride;
            atomicAdd(&a.epi.sums[ro + c], (double)t1);
            atomicAdd(&a.epi.sums[ro + a.Cout + c], (double)t2);
        } else {
            const size_t ro = (size_t)(blockIdx.x % a.stat_replicas) * a.stat_rstride;
            atomicAdd(&a.stat_sum[ro + c], (double)t1);
            atomicAdd(&a.stat_sumsq[ro + c], (double)t2);
        }
    }
}

#ifndef RW_WREG_A
#define RW_WREG_A 56
#endif
#ifndef RW_WREG_B
#define RW_WREG_B 40
#endif
#ifndef RW_WREG_C
#define RW_WREG_C 28
#endif
template <int CINP, int NTL, bool BNEPI> static int launch_rw(const RwArgs& a, hipStream_t st)
{
    constexpr int CPRU = CINP >= 64 ? 8 : 4;
    constexpr int NFRAG = (CINP / 16) * 9 * NTL;
    constexpr int WREG = NFRAG < 72 ? NFRAG : (BNEPI ? (NTL == 2 ? RW_WREG_C : RW_WREG_B) : RW_WREG_A);      // zero spilled registers in every instantiation
    constexpr int lds = 2 * RW_HP * rw_rowb(CPRU) + 2 * CINP * 4 + 4 * NTL * 32 * 4 + (NFRAG - WREG) * 1024 + RW_NT * 16;
    auto kern = conv3x3_rw_kernel<CINP, NTL, BNEPI, WREG>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    const int ntile = a.tiles_x * a.tiles_y * a.N;
    const int blocks = ntile < 256 ? ntile : 256;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(RW_NT), lds, st, a);
    SAUNET_CHECK_LAUNCH("conv3x3_rw");
    return SAUNET_OK;
}

// bf16 3x3 stride-1 pad-1 convolutions on 16-tiled maps whose weights fit the registers of one wave: (Cin <= 128, Cout <= 32) or (Cin <= 64, Cout <= 64)
bool rw_fwd_supported(const saunet_conv_desc* d, const void* x, const void* w, const void* y, const saunet_bn_epilogue* epi)
{
    static const bool on = !(getenv("SAUNET_CONV_RW") && getenv("SAUNET_CONV_RW")[0] == '0');          // A/B switch
    static const int min_tiles = getenv("SAUNET_RW_MINTILES") ? atoi(getenv("SAUNET_RW_MINTILES")) : 32;
    if (!on || d->dtype != SAUNET_BF16 || d->transposed || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return false;
    if (d->H % 16 || d->W % 16 || d->Ho != d->H || d->Wo != d->W) return false;
    if (d->Cin % 8 || d->Cout % 8 || d->ldx % 8 || d->ldy % 8) return false;
    if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) != 0) return false;
    if (epi && epi->bn_x && (epi->accumulate || epi->ld_bn_x % 8 || ((uintptr_t)epi->bn_x & 15))) return false;
    const int cinp = d->Cin <= 32 ? 32 : (d->Cin <= 64 ? 64 : 128), ntl = (d->Cout + 31) / 32;
    if (d->Cin > 128 || ntl > 2 || (cinp / 16) * 9 * ntl > 72) return false;
    return (long)d->N * (d->H / 16) * (d->W / 16) >= min_tiles;
}

int rw_forward(const saunet_conv_desc* d, const void* x, const void* w, const float* bias, const float* ps, const float* psh, void* y, double* ssum,
               double* ssq, const saunet_bn_epilogue* epi, hipStream_t st, const saunet_bn_prologue* bnp)
{
    RwArgs a;
    if (bnp) a.bnp = *bnp; else a.bnp.gamma = nullptr;
    if (epi) { a.epi = *epi; if (a.epi.sums_replicas < 1) a.epi.sums_replicas = 1; } else a.epi.bn_x = nullptr;
    a.x = (const u16*)x; a.w = (const u16*)w; a.y = (u16*)y; a.bias = bias; a.pro_scale = ps; a.pro_shift = psh; a.stat_sum = ssum; a.stat_sumsq = ssq;
    a.stat_replicas = d->stat_replicas > 1 ? d->stat_replicas : 1; a.stat_rstride = d->stat_rstride;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx; a.Cout = d->Cout; a.ldy = d->ldy;
    a.pro_relu = d->pro_relu; a.act_relu = d->epi_relu; a.tiles_y = d->H / 16; a.tiles_x = d->W / 16;
    const int cinp = d->Cin <= 32 ? 32 : (d->Cin <= 64 ? 64 : 128), ntl = (d->Cout + 31) / 32;
    const bool be = a.epi.bn_x != nullptr;
#define RW(C_, N_) (be ? launch_rw<C_, N_, true>(a, st) : launch_rw<C_, N_, false>(a, st))
    if (cinp == 128) return RW(128, 1);
    if (cinp == 64) return ntl == 1 ? RW(64, 1) : RW(64, 2);
    return ntl == 1 ? RW(32, 1) : RW(32, 2);
#undef RW
}

}  // namespace saunet
