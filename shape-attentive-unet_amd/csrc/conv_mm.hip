// 3x3 (stride 1, pad 1) convolution for the MFMA-bound layers of the decoder -- `c3x3rb` of the four DualAttBlocks
// (/root/reference/models/attention_blocks.py:215-220: Cin 1536 / 1024 / 512 / 256), their data gradients (the same convolution over dy with
// flipped weights, Cin <-> Cout) and every other 3x3 whose input needs no BatchNorm prologue and has Cin % 64 == 0.
//
// Structure (one workgroup per CU: 8 waves, two per SIMD; ~150 KB of LDS):
//   * output tile 16x16 pixels x BN output channels, wave tile 64 pixels x (BN/2) channels (4 x 2 waves), mfma_f32_32x32x16_bf16;
//   * the K loop runs over stages  (channel block of 64) x (tap):  16 MFMAs per wave and stage (8 with BN = 64);
//   * ALL operands reach the LDS by LDS-DMA (global_load_lds_dwordx4): the input tile with its one-pixel halo (18 x 18 pixels x 128 B, two
//     buffers: the next channel block is staged while the nine taps of the current one run) and one [BN][64] weight tile per stage in a ring
//     of four.  Nothing passes through registers, no address or pack VALU work in the loop; zero padding = lanes that source a zero page;
//   * the DMA destination is lane-linear, so the 16-byte XOR swizzle that keeps the ds_read_b128 fragment reads conflict-free is applied to
//     the per-lane SOURCE address (and again on the read): image row r holds source chunk (slot ^ key);
//   * one s_barrier per stage; loads are waited for with COUNTED vmcnt (a stage's weights were requested three stages earlier, the halo of
//     the next channel block up to nine), and a stage's weights are certified one stage EARLY, so the first fragments of stage s+1 are read
//     before the barrier that ends stage s: the ds_read -> MFMA software pipeline (two named fragment sets) runs straight through barriers.
//
// bf16 storage only; float32 storage keeps conv3x3_tile_fwd_kernel (exact f32 MFMA).
//
// CONVT (round 6): ConvTranspose2d(k = 4, s = 2, p = 1) forward -- `mrf.up` of the DualAttBlocks (/root/reference/models/attention_blocks.py:179-186)
// and DecoderBlock's transposed convolution (models/models.py:208-213) -- on the SAME stage pipeline.  Output pixel (2i + ph, 2j + pw) is a
// 2 x 2-tap convolution over the input around (i, j): tap (th, tw) reads input pixel (i + ph - th, j + pw - tw) = halo pixel (1 + ph - th,
// 1 + pw - tw) of the 18 x 18 halo and kernel element (1 - ph + 2 th, 1 - pw + 2 tw), i.e. row (ph, pw, co, th, tw) of the SAUNET_PACK_CONVT_FWD
// packing.  A workgroup = (16 x 16 INPUT pixel tile, output parity, n tile): the four parities ride on the n-tile index, so they share the
// halo on one XCD; K loop = (channel block) x (4 live taps) -- the dead taps of the "one 3 x 3 convolution with 4 Cout channels" formulation
// (scripts/probes/convt_forward_as_3x3_rejected.patch: 2.25x the multiply-adds) are never stages; the next channel block's six halo pieces
// per wave are requested two per stage behind taps 0 - 2 and certified at the barrier of tap 3; the epilogue scatters the tile to its
// parity's pixels of the 2H x 2W map.
#include "common.h"
#include <type_traits>

namespace saunet {

struct MmArgs {
    const u16* x; const u16* w; u16* y; const float* bias;
    double* stat_sum; double* stat_sumsq; int stat_replicas, stat_rstride;
    int N, H, W, Cin, ldx, Cout, ldy, act_relu;
    int tiles_x, tiles_y, ntiles, nnt;
    int splits; float* ws;      // CELL only: split-K over channel blocks (partials in accumulator layout, conv3x3_mm_finish_kernel)
};

static __device__ u32x4 g_mm_zeros[256];    // 4 KB of zeros: the source of every padding lane (a padding lane walks through it with the channel block: Cin <= 2048)

// Halo geometry.  Plain: one 16 x 16 pixel tile of one image, 18 x 18 halo = 41 DMA instructions of 1 KB (8 pixels x 128 B), 6 request slots per
// wave (8 x 6 >= 41; surplus slots go to the dummy region).  CELL (8 x 8 maps: `center`): the 16 x 16 tile is a 2 x 2 cell of FOUR images, each
// quadrant with its own zero border -- a 20 x 20 halo (two 10-wide halos side by side, two stacked), 50 instructions, 7 slots per wave; an output
// pixel (py, px) reads halo pixel (py + kh + 2 * (py >> 3), px + kw + 2 * (px >> 3)): still lane base + per-tap constant.
template <bool CELL> struct MmHalo {
    static constexpr int HP = CELL ? 20 : 18, NPIX = HP * HP;
    static constexpr int INSTR = (NPIX + 7) / 8;
    static constexpr int BYTES = INSTR * 1024;
    static constexpr int PER_WAVE = (INSTR + 7) / 8;
};
constexpr int MM_RING = 4;

// bias / ReLU, tile through LDS to 16-byte row stores, per-channel sums of the un-biased accumulator (BatchNorm statistics)
template <int BN, bool CELL, bool CONVT = false>
__device__ __forceinline__ void mm_epilogue(const MmArgs& a, f32x16 (&acc)[2][BN / 64], unsigned char* smem, int n, int ty0, int tx0, int n0, int par = 0)
{
    constexpr int TJ = BN / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    u16* so = (u16*)smem;                                          // [256][BN]
    float* s_sum = (float*)(smem + 256 * BN * 2);                  // [4 row waves][2][BN]
    const bool do_stats = a.stat_sum != nullptr;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int col = wn * (BN / 2) + j * 32 + lr;
        const float bv = (a.bias != nullptr && n0 + col < a.Cout) ? a.bias[n0 + col] : 0.f;
        float sv = 0.f, ssv = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[i][j][r];
                sv += v; ssv += v * v;
                const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                Elem<u16>::store(so + row * BN + col, a.act_relu ? fmaxf(v + bv, 0.f) : v + bv);
            }
        if (do_stats) {
            sv += __shfl_xor(sv, 32, 64); ssv += __shfl_xor(ssv, 32, 64);
            if (lh == 0) { float* slot = s_sum + wm * 2 * BN; slot[col] = sv; slot[BN + col] = ssv; }
        }
    }
    __syncthreads();
    if (do_stats && tid < BN && n0 + tid < a.Cout) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { t1 += s_sum[w * 2 * BN + tid]; t2 += s_sum[w * 2 * BN + BN + tid]; }
        const size_t ro = (size_t)(blockIdx.x % a.stat_replicas) * a.stat_rstride;
        atomicAdd(&a.stat_sum[ro + n0 + tid], (double)t1);
        atomicAdd(&a.stat_sumsq[ro + n0 + tid], (double)t2);
    }
    constexpr int CH = BN / 8;                                     // 16-byte chunks per row
    u16* __restrict__ yg = a.y + (size_t)n * (CELL || CONVT ? 4 : 1) * a.H * a.W * a.ldy;      // CONVT: the output map is 2H x 2W
    constexpr int S_ITERS = 256 * CH / 512;
    const int colv = n0 + (tid % CH) * 8;
    if (colv < a.Cout) {
#pragma unroll
        for (int i = 0; i < S_ITERS; ++i) {
            const int p = tid + i * 512;
            const int row = p / CH, ch = p - row * CH;
            const int py = row >> 4, px = row & 15;
            const size_t opix = CONVT ? (size_t)(2 * (ty0 + py) + (par >> 1)) * (2 * a.W) + 2 * (tx0 + px) + (par & 1)
                              : CELL ? (size_t)(2 * (py >> 3) + (px >> 3)) * 64 + (py & 7) * 8 + (px & 7) : (size_t)(ty0 + py) * a.W + tx0 + px;
            *(u32x4*)(yg + opix * a.ldy + colv) = *(const u32x4*)(so + row * BN + ch * 8);
        }
    }
}

// second half of a split-K cell-mode launch: one workgroup per (cell, n tile) sums the partials in split order and finishes the tile
__global__ __launch_bounds__(512) void conv3x3_mm_finish_kernel(MmArgs a)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, item = blockIdx.x, t = item / a.nnt, nt = item - t * a.nnt;
    const float* wsb = a.ws + (size_t)item * a.splits * (32 * 512);
    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    for (int sp = 0; sp < a.splits; ++sp) {          // split order (deterministic); the 32 loads of one split are independent and in flight together
        const float* q = wsb + (size_t)sp * (32 * 512) + tid;
        float v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = q[k * 512];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc[k >> 4][0][k & 15] += v[k];
    }
    mm_epilogue<64, true>(a, acc, smem, t, 0, 0, nt * 64);
}

// Where a stage's DMA requests are issued matters (round-4 phase stamps, profiles/r04_mm_kernel_notes.txt): one request blocks ITS wave for
// ~110 cycles (the CU's memory front end accepts one about every 30 cycles from all waves together), so with all eight waves requesting right
// behind the barrier every matrix pipe idled 330-700 cycles per stage.  Each wave now issues one request behind each of the first three
// k-steps: the partner wave of the SIMD keeps the pipe busy meanwhile (MFMA-busy 60 -> 63 % at dec3).  Measured and rejected: two dedicated
// loader waves (640-thread workgroups; a single wave sustains only one request per ~118 cycles, 12 per stage = 1400 cycles: slower, 55 %),
// anti-phase halves (waves 0-3 behind the barrier, 4-7 at the end of the stage: 56 %), requests pinned a full k-step ahead and s_setprio
// around the MFMA groups (both -2 %).
template <int BN, bool CELL, bool CONVT = false>
__global__ __launch_bounds__(512, 2) void conv3x3_mm_kernel(MmArgs a)
{
    static_assert(!(CELL && CONVT), "the transposed convolution runs on plain tiles");
    constexpr int TAPS = CONVT ? 4 : 9;
    constexpr int MM_HP = MmHalo<CELL>::HP, MM_NPIX = MmHalo<CELL>::NPIX, MM_HALO_INSTR = MmHalo<CELL>::INSTR, MM_HALO_BYTES = MmHalo<CELL>::BYTES;
    constexpr int MM_HALO_PER_WAVE = MmHalo<CELL>::PER_WAVE;
    TSTAMP_INIT();
    TSTAMP(20);
    constexpr int TJ = BN / 64;                       // 32-channel MFMA tiles per wave in N
    constexpr int WSLOT = BN * 128;                   // one stage's weight tile: [BN][64] bf16
    constexpr int WINSTR = BN / 64;                   // weight DMA instructions per wave and stage
    constexpr int OFF_W = 2 * MM_HALO_BYTES;
    constexpr int OFF_DUMMY = OFF_W + MM_RING * WSLOT;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;          // 4 x 2 waves
    // ---- workgroup -> (pixel tile, n tile): the n tiles of one pixel tile share its halo, so they run on the SAME XCD (block b -> XCD b % 8)
    int t, nt;
    const int split = CELL ? (int)(blockIdx.x % (unsigned)a.splits) : 0;
    {
        const int b = CELL ? (int)(blockIdx.x / (unsigned)a.splits) : (int)blockIdx.x;
        if ((a.ntiles & 7) == 0) { const int k = b >> 3; t = (k / a.nnt) * 8 + (b & 7); nt = k % a.nnt; }
        else { t = b / a.nnt; nt = b % a.nnt; }
    }
    const int txi = t % a.tiles_x; const int r1 = t / a.tiles_x;
    const int tyi = r1 % a.tiles_y; const int n = r1 / a.tiles_y;
    // CONVT: nt = parity * (n tiles per parity) + n tile; (ph, pw) = output row / column parity of this workgroup
    const int nnt_c = CONVT ? a.nnt >> 2 : a.nnt;
    const int par = CONVT ? nt / nnt_c : 0, ph = par >> 1, pw = par & 1;
    if constexpr (CONVT) nt -= par * nnt_c;
    const int ty0 = tyi * 16, tx0 = txi * 16, n0 = nt * BN;
    // CELL: a.tiles_x = a.tiles_y = 1 and n counts cells of four 8 x 8 images
    const u16* __restrict__ xg = a.x + (size_t)n * (CELL ? 4 : 1) * a.H * a.W * a.ldx;
    // CELL: this workgroup reduces channel blocks [cb0, cb0 + ncb) of the layer's Cin / 64
    const int ncb = CELL ? (a.Cin >> 6) / a.splits : a.Cin >> 6;
    const int cb0 = split * ncb;
    const int nstage = ncb * TAPS;
    const unsigned char* zsrc = (const unsigned char*)g_mm_zeros;

    // ---- DMA source offsets of this lane (bytes; channel block / tap terms are added per issue)
    // halo slot j of this wave = DMA instruction hidx = j * 8 + wave: LDS pixels 8 * hidx .. + 7, lane = (pixel, 16-byte slot)
    // a padding lane's pointer stays inside the zero page
    const unsigned char* hsrc[MM_HALO_PER_WAVE];
#pragma unroll
    for (int j = 0; j < MM_HALO_PER_WAVE; ++j) {
        const int hidx = j * 8 + wave;
        const int hp = hidx * 8 + (lane >> 3), sl = lane & 7;
        const int hy = hp / MM_HP, hx = hp - hy * MM_HP;
        int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
        long img = 0;
        if constexpr (CELL) {            // quadrant (hy / 10, hx / 10) = image 2 * qy + qx of the cell, its own 10 x 10 halo
            const int qy = hy / 10, qx = hx / 10;
            iy = hy - 10 * qy - 1; ix = hx - 10 * qx - 1;
            img = (long)(2 * qy + qx) * a.H * a.W;
        }
        const bool ok = hidx < MM_HALO_INSTR && hp < MM_NPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const int ch = sl ^ (((CELL && hx >= 10 ? hx - 2 : hx) >> 1) & 7);       // CELL: key of the un-gapped column, so that 16 lanes still cover the 16 bank groups
        hsrc[j] = ok ? (const unsigned char*)xg + ((img + iy * a.W + ix) * a.ldx + ch * 8) * 2 : zsrc + (lane & 7) * 16;
    }
    unsigned woff[WINSTR];
#pragma unroll
    for (int j = 0; j < WINSTR; ++j) {
        const int q = j * 8 + wave;
        const int r = q * 8 + (lane >> 3), sl = lane & 7;
        int row = n0 + r; if (row >= a.Cout) row = a.Cout - 1;
        woff[j] = (unsigned)(((size_t)row * TAPS * a.Cin + (sl ^ ((r >> 1) & 7)) * 8) * 2);
    }
    // requests past the end of the K loop keep the per-wave request count uniform (the counted waits rely on it): they re-load the last
    // channel block / stage into a buffer nobody reads any more
    auto issue_halo = [&](int j, int cb) {           // slot j of channel block cb
        const int hidx = j * 8 + wave;
        const int cbc = cb < ncb ? cb : ncb - 1;
        const unsigned dst = hidx < MM_HALO_INSTR ? lds0 + (cb & 1) * MM_HALO_BYTES + hidx * 1024 : lds0 + OFF_DUMMY;
        mm_dma16(hsrc[j] + (long)(cbc + cb0) * 128, dst);
    };
    auto issue_w1 = [&](int s, int j) {              // rows j of the weight tile of stage s into ring slot s % 4
        const int sc = s < nstage ? s : nstage - 1;
        const int cb = sc / TAPS, tap = sc - cb * TAPS;
        const unsigned sbase = lds0 + OFF_W + (s & (MM_RING - 1)) * WSLOT;
        // CONVT: rows (ph, pw, co) of the phase packing, 4 taps x Cin per row
        const unsigned char* wsrc = (const unsigned char*)a.w + ((size_t)par * a.Cout * TAPS * a.Cin + (size_t)tap * a.Cin + (size_t)(cb + cb0) * 64) * 2;
        mm_dma16(wsrc + woff[j], sbase + (j * 8 + wave) * 1024);
    };
    auto issue_w = [&](int s) {
#pragma unroll
        for (int j = 0; j < WINSTR; ++j) issue_w1(s, j);
    };

    // ---- fragment addresses (bytes inside smem).  A: halo pixel (py + kh, px + kw), 16-byte slot (2 ks + lh) ^ key(px + kw)
    const int py0 = wm * 4 + (lr >> 4), px = lr & 15;
    const int hy0 = CELL ? py0 + 2 * (py0 >> 3) : py0, hx0 = CELL ? px + 2 * (px >> 3) : px;      // (the wave's four rows sit in one half of the cell)
    // (CONVT: index tw = 0, 1 -> halo column offset 1 + pw - tw; the row offset 1 + ph - th is workgroup-uniform scalar arithmetic)
    int ak[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int hxk = hx0 + (CONVT ? 1 + pw - kw : kw);
        const int key = ((CELL && hxk >= 10 ? hxk - 2 : hxk) >> 1) & 7;
        ak[kw] = (hy0 * MM_HP + hxk) * 128 + (((lh ^ key) & 1) << 4) + ((key & 6) << 4);
    }
    const int brow = wn * (BN / 2) + lr;
    const int bkey = (brow >> 1) & 7;
    const int bk = OFF_W + brow * 128 + (((lh ^ bkey) & 1) << 4) + ((bkey & 6) << 4);

    f32x16 acc[2][TJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: halo of channel block 0, weights of stages 0, 1, 2
#pragma unroll
    for (int j = 0; j < MM_HALO_PER_WAVE; ++j) issue_halo(j, 0);
    issue_w(0); issue_w(1); issue_w(2);
    TSTAMP(21);
    mm_wait_vm<2 * WINSTR>();                         // halo 0 + weights 0 landed (this wave's pieces); stages 1, 2 may be in flight
    mm_barrier();
    TSTAMP(22);

    u32x4 fa0[2], fb0[TJ], fa1[2], fb1[TJ];
    auto load_frags = [&](u32x4* fa, u32x4* fb, int hb_off, int ws_off, int tap, int ks) {
        const int kh = CONVT ? 1 + ph - (tap >> 1) : tap / 3, kw = CONVT ? (tap & 1) : tap - (tap / 3) * 3;      // kw indexes ak[]
        const int aaddr = (ak[kw] ^ (ks << 5)) + hb_off + kh * (MM_HP * 128);
        const int baddr = (bk ^ (ks << 5)) + ws_off;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *(const u32x4*)(smem + aaddr + i * (2 * MM_HP * 128));
#pragma unroll
        for (int j = 0; j < TJ; ++j) fb[j] = *(const u32x4*)(smem + baddr + j * (32 * 128));
        // pin the request in front of the MFMAs that follow in program order: left alone, hipcc sinks the reads behind the previous k-step's
        // MFMAs and waits for them right away (no fragment is then in flight while the matrix pipe works)
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mma = [&](const u32x4* fa, const u32x4* fb) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[i]), __builtin_bit_cast(bf16x8_t, fb[j]), acc[i][j], 0, 0, 0);
    };
    load_frags(fa0, fb0, 0, 0, 0, 0);

    int s = 0;
    for (int cb = 0; cb < ncb; ++cb) {
        const int hb_off = (cb & 1) * MM_HALO_BYTES, hb_next = ((cb + 1) & 1) * MM_HALO_BYTES;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap, ++s) {
            // ---- stage boundary.  Outstanding (per wave, oldest first): [halo piece of stage s-2] weights(s+1) [halo piece of s-1] weights(s+2);
            // certify weights(s+1) (and with them every older halo piece): allow what stage s-1 issued to stay in flight
            TSTAMP(23);
            if constexpr (CONVT) {
                // stage s - 1 issued two halo pieces (taps 0 - 2 of a channel block) in FRONT of weights(s + 2): behind taps 0 and 1 they may stay
                // in flight, at the barrier of tap 3 they are certified with everything older than weights(s + 2) -- the first fragments of the
                // next channel block are read at the end of this stage
                static_assert(MM_HALO_PER_WAVE <= 6, "two halo pieces behind each of taps 0 - 2");
                if (tap == 1 || tap == 2) mm_wait_vm<2 + WINSTR>(); else mm_wait_vm<WINSTR>();
            } else {
                if (tap >= 1 && tap <= MM_HALO_PER_WAVE) mm_wait_vm<1 + WINSTR>(); else mm_wait_vm<WINSTR>();
            }
            TSTAMP(29);
            mm_barrier();
            TSTAMP(24);
            // the stage's requests of this wave as pieces 0 .. 2: [halo piece (taps 0-5)] [weight rows] [weight rows (BN = 128)]
            auto piece = [&](int k) {
                if constexpr (CONVT) {     // [halo piece 2 tap] [halo piece 2 tap + 1] [weight rows (both instructions with BN = 128)]
                    if (k < 2) { if (tap < 3 && 2 * tap + k < MM_HALO_PER_WAVE) issue_halo(2 * tap + k, cb + 1); else if (tap < 3) issue_halo(MM_HALO_PER_WAVE - 1, cb + 1); }
                    else issue_w(s + 3);
                } else {
                    if (k == 0) { if (tap < MM_HALO_PER_WAVE) issue_halo(tap, cb + 1); }
                    else if (k - 1 < WINSTR) issue_w1(s + 3, k - 1);
                }
            };
            TSTAMP(25);
            const int ws_off = (s & (MM_RING - 1)) * WSLOT, ws_next = ((s + 1) & (MM_RING - 1)) * WSLOT;
            // 4 k-steps of 16 channels; fragments of the next k-step (or of the next stage's first) are requested before this one's MFMAs;
            // one DMA request behind each of the first three k-steps
            load_frags(fa1, fb1, hb_off, ws_off, tap, 1);
            mma(fa0, fb0);
            piece(0);
            load_frags(fa0, fb0, hb_off, ws_off, tap, 2);
            mma(fa1, fb1);
            piece(1);
            load_frags(fa1, fb1, hb_off, ws_off, tap, 3);
            mma(fa0, fb0);
            piece(2);
            if (tap < TAPS - 1) load_frags(fa0, fb0, hb_off, ws_next, tap + 1, 0);
            else load_frags(fa0, fb0, hb_next, ws_next, 0, 0);
            mma(fa1, fb1);
        }
    }
    TSTAMP(26);
    mm_wait_vm<0>();
    __syncthreads();
    TSTAMP(27);
    if constexpr (CELL) {
        // split-K: every workgroup stores its accumulators in ACCUMULATOR layout (value k of thread tid at [k][tid]: coalesced, and the same
        // thread of the finishing workgroup re-reads them); conv3x3_mm_finish_kernel adds the partials in split order and runs the epilogue.
        // (An in-kernel finish by the last arriver -- __threadfence + ticket -- cost 30-75 us: a device-scope fence writes back and
        // invalidates the XCD's whole L2 on this chip.  The kernel boundary is the cheap fence.)
        if (a.splits > 1) {
            float* mine = a.ws + ((size_t)(t * a.nnt + nt) * a.splits + split) * (32 * TJ * 512);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mine[((i * TJ + j) * 16 + r) * 512 + tid] = acc[i][j][r];
            return;
        }
    }
    mm_epilogue<BN, CELL, CONVT>(a, acc, smem, n, ty0, tx0, n0, par);
    TSTAMP(28);
}


template <int BN, bool CELL = false, bool CONVT = false> static int launch_mm(const MmArgs& a, hipStream_t st)
{
    constexpr int LDS = 2 * MmHalo<CELL>::BYTES + MM_RING * BN * 128 + 1024;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv3x3_mm_kernel<BN, CELL, CONVT>;
    static DeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL(kern, dim3(a.ntiles * a.nnt * (CELL ? a.splits : 1)), dim3(512), LDS, st, a);
    static const KName kn("conv3x3_mm_kernel", BN, CELL, CONVT);
    SAUNET_CHECK_LAUNCH(kn.s);
    return SAUNET_OK;
}

// which of the two output-channel tiles: 128 unless that leaves CUs idle (fewer than 256 workgroups) and 64 does better
static int mm_pick_bn(const saunet_conv_desc* d)
{
    static const int force = ab_env_int("SAUNET_MM_BN", 0);      // A/B switch (variant builds only)
    if (force == 64 || force == 128) return force;
    if (d->Cout <= 64) return 64;
    const long tiles = (long)d->N * (d->H / 16) * (d->W / 16);
    const long b128 = tiles * ((d->Cout + 127) / 128);
    return b128 < 256 ? 64 : 128;
}

static bool mm_cell_geometry(const saunet_conv_desc* d)
{
    static const bool cell_on = ab_env_on("SAUNET_MM_CELL");        // A/B switch (variant builds only)
    return cell_on && d->H == 8 && d->W == 8 && d->N % 4 == 0;          // 8 x 8 maps: 2 x 2 image cells
}

bool mm_fwd_supported(const saunet_conv_desc* d, const void* x, const void* w, const void* y, const float* ps, const saunet_bn_epilogue* epi)
{
    static const bool on = ab_env_on("SAUNET_CONV_MM");             // A/B switch (variant builds only)
    static const int min_cin = ab_env_int("SAUNET_MM_MINCIN", 128);
    return on && d->dtype == SAUNET_BF16 && !d->transposed && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 &&
           ((d->H % 16 == 0 && d->W % 16 == 0) || mm_cell_geometry(d)) && d->Ho == d->H && d->Wo == d->W && ps == nullptr && (epi == nullptr || epi->bn_x == nullptr) && d->Cin % 64 == 0 &&
           d->Cin >= min_cin && d->Cin <= 2048 && d->Cout % 8 == 0 && d->Cout >= 64 && d->ldx % 8 == 0 && d->ldy % 8 == 0 &&
           !(((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) && (long)d->N * d->H * d->W * d->ldx < (1L << 30);
}

// cell mode (8 x 8 maps): few pixels, long K -- the channel blocks are split over workgroups until the chip is full.  -> number of splits
static int mm_cell_splits(const saunet_conv_desc* d, int* items_out)
{
    static const int max_splits = ab_env_int("SAUNET_MM_SPLITS", 8);      // A/B switch (variant builds only)
    const int ncb = d->Cin / 64, items = (d->N / 4) * ((d->Cout + 63) / 64);
    int splits = 1;
    while (splits * 2 <= max_splits && ncb % (splits * 2) == 0 && ncb / (splits * 2) >= 2 && items * splits * 2 <= 256) splits *= 2;
    if (items_out) *items_out = items;
    return splits;
}

// bytes of caller-owned workspace the forward of `d` can use (saunet_conv2d_forward_workspace): the split-K partials of cell mode
int64_t mm_forward_workspace(const saunet_conv_desc* d)
{
    if (!mm_fwd_supported(d, nullptr, nullptr, nullptr, nullptr, nullptr) || !(d->H == 8 && d->W == 8)) return 0;
    int items = 0;
    const int splits = mm_cell_splits(d, &items);
    return splits > 1 ? (int64_t)items * splits * 32 * 512 * (int64_t)sizeof(float) : 0;
}

int mm_forward(const saunet_conv_desc* d, const void* x, const void* w, const float* bias, void* y, double* ssum, double* ssq, hipStream_t st)
{
    MmArgs a;
    a.x = (const u16*)x; a.w = (const u16*)w; a.y = (u16*)y; a.bias = bias; a.stat_sum = ssum; a.stat_sumsq = ssq;
    a.stat_replicas = d->stat_replicas > 1 ? d->stat_replicas : 1; a.stat_rstride = d->stat_rstride;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx; a.Cout = d->Cout; a.ldy = d->ldy; a.act_relu = d->epi_relu;
    a.splits = 1; a.ws = nullptr;
    if (d->H == 8 && d->W == 8) {          // cell mode: four images per 16 x 16 tile, 64-wide output tiles (the 20 x 20 halos leave no room for the 128-wide ring)
        a.tiles_y = a.tiles_x = 1; a.ntiles = d->N / 4; a.nnt = (d->Cout + 63) / 64;
        // The partials live in the CALLER's workspace (saunet_conv_desc.workspace, sized by saunet_conv2d_forward_workspace); a call
        // without one -- or with one that is too small or misaligned -- runs the layer unsplit (same result up to the summation order).
        int items = 0;
        const int splits = mm_cell_splits(d, &items);
        const int64_t need = (int64_t)items * splits * 32 * 512 * (int64_t)sizeof(float);
        if (splits > 1 && d->workspace != nullptr && d->workspace_bytes >= need && ((uintptr_t)d->workspace & 15) == 0) {
            a.splits = splits; a.ws = (float*)d->workspace;
        }
        if (int rc = launch_mm<64, true>(a, st)) return rc;
        if (a.splits > 1) {
            hipLaunchKernelGGL(conv3x3_mm_finish_kernel, dim3(items), dim3(512), 256 * 64 * 2 + 4 * 2 * 64 * 4 + 1024, st, a);
            SAUNET_CHECK_LAUNCH("conv3x3_mm_finish");
        }
        return SAUNET_OK;
    }
    a.tiles_y = d->H / 16; a.tiles_x = d->W / 16; a.ntiles = a.tiles_x * a.tiles_y * a.N;
    const int bn = mm_pick_bn(d);
    a.nnt = (d->Cout + bn - 1) / bn;
    return bn == 64 ? launch_mm<64>(a, st) : launch_mm<128>(a, st);
}

// ---- ConvTranspose2d(4, 2, 1) forward on the same pipeline (CONVT): d is the transposed-convolution descriptor (H, W = input map, Ho = 2H)
bool mm_convt_supported(const saunet_conv_desc* d, const void* x, const void* w, const void* y, const float* ps, const saunet_bn_epilogue* epi)
{
    static const bool on = ab_env_on("SAUNET_CONVT_MM");            // A/B switch (variant builds only)
    return on && d->dtype == SAUNET_BF16 && d->transposed && d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad == 1 && d->Ho == 2 * d->H && d->Wo == 2 * d->W &&
           d->H % 16 == 0 && d->W % 16 == 0 && ps == nullptr && epi == nullptr && d->Cin % 64 == 0 && d->Cin >= 128 && d->Cin <= 2048 &&
           d->Cout % 8 == 0 && d->Cout >= 64 && d->ldx % 8 == 0 && d->ldy % 8 == 0 &&
           !(((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) && (long)d->N * d->H * d->W * d->ldx < (1L << 30) && (long)d->N * d->Ho * d->Wo * d->ldy < (1L << 31);
}

int mm_convt_forward(const saunet_conv_desc* d, const void* x, const void* w, const float* bias, void* y, double* ssum, double* ssq, hipStream_t st)
{
    MmArgs a;
    a.x = (const u16*)x; a.w = (const u16*)w; a.y = (u16*)y; a.bias = bias; a.stat_sum = ssum; a.stat_sumsq = ssq;
    a.stat_replicas = d->stat_replicas > 1 ? d->stat_replicas : 1; a.stat_rstride = d->stat_rstride;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx; a.Cout = d->Cout; a.ldy = d->ldy; a.act_relu = d->epi_relu;
    a.splits = 1; a.ws = nullptr;
    a.tiles_y = d->H / 16; a.tiles_x = d->W / 16; a.ntiles = a.tiles_x * a.tiles_y * a.N;
    // 128-wide output tiles unless that leaves CUs idle (four parities per pixel tile already multiply the grid by four)
    const int bn = (d->Cout <= 64 || (long)a.ntiles * 4 * ((d->Cout + 127) / 128) < 256) ? 64 : 128;
    a.nnt = 4 * ((d->Cout + bn - 1) / bn);
    return bn == 64 ? launch_mm<64, false, true>(a, st) : launch_mm<128, false, true>(a, st);
}

}  // namespace saunet

SAUNET_TIMING_READER(conv_mm)
