// 3x3 (stride 1, pad 1) and 1x1 convolution kernels built around a 16x16-pixel output tile whose input
// (with its 1-pixel halo) lives in LDS for the whole K loop:
//
//  * conv3x3_tile_fwd_kernel   forward / dgrad:  every input element is read from HBM/L2, BatchNorm+ReLU
//    transformed and written to LDS ONCE per block and then feeds all 9 taps straight out of LDS (the
//    generic kernel in conv_igemm.hip gathers and transforms it 9 times).
//  * conv_tile_wgrad_kernel    weight gradient:  dy tile + x halo tile staged [pixel][channel] (their natural
//    NHWC layout, coalesced loads); the MFMA K dimension is the PIXEL index, so fragments are fetched with the
//    gfx950 transposing LDS read ds_read_b64_tr_b16 (4 consecutive pixels of 16 channels per 16-lane group);
//    all 9 taps are accumulated by the same block from one staged halo.
//
// MFMA: mfma_f32_32x32x16_bf16 (bf16 storage) / mfma_f32_32x32x2_f32 (float32 storage, exact f32).
#include "common.h"

namespace saunet {

struct TileArgs {
    const void* x; const void* w; void* y;
    const float* bias; const float* pro_scale; const float* pro_shift;
    double* stat_sum; double* stat_sumsq; int stat_replicas, stat_rstride;
    int N, H, W, Cin, ldx, Cout, ldy;
    int pro_relu, act_relu;
    int tiles_x, tiles_y;   // H/16, W/16
    saunet_bn_epilogue epi;
    int lds_acc_off;        // resident kernel: byte offset of the block-lifetime accumulators in LDS
    saunet_bn_prologue bnp; // bnp.gamma != nullptr: the prologue coefficients are derived in the kernel (resident kernel only)
};

template <typename T> struct MmaT;
template <> struct MmaT<u16> {
    __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c)
    {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct MmaT<float> {
    __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) { mma_f32_chunk_exact(a, b, c); }
};
template <> struct MmaT<f32s> {
    __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) { mma_f32_chunk_split(a, b, c); }
};

template <int CPR> __device__ __forceinline__ int swz_off(int r, int c)
{
    constexpr int RPB = 16 / CPR;
    return (r * CPR + (c ^ ((r / RPB) & (CPR - 1)))) * 16;
}

constexpr int TILE = 16, HPITCH = 18, NPIX = HPITCH * HPITCH;

// Halo tile of the tile kernel: [18 x 18 pixels][CPR chunks of 16 B], chunk index XOR-swizzled.  gfx950 serves a ds_read_b128 in four
// groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32 -- over 64 four-byte banks, i.e. the 16 lanes of a group must
// hit 16 different 16-byte bank groups.  An A-fragment group takes pixels x..x+3, x+12..x+15 of one halo row and x+4..x+11 of the NEXT row:
// with 128-byte pixels (CPR = 8) the bank group is 8*(x & 1) + swizzled chunk, so the swizzle key must be distinct over 8 consecutive
// same-parity columns REGARDLESS of the row: key = (x >> 1) & 7.  (Keyed by the linear pixel index, as the weight tiles still are, the row
// pitch of 18 shifted the second row's keys onto the first row's: 2 of 8 bank groups collided, SQ_LDS_BANK_CONFLICT = 25-31 % of the LDS
// cycles of the decoder convolutions.)
template <int CPR> __device__ __forceinline__ int swz_halo(int hy, int hx, int c)
{
    if constexpr (CPR == 8) return (((hy * HPITCH + hx) << 3) + (c ^ ((hx >> 1) & 7))) * 16;
    else return swz_off<CPR>(hy * HPITCH + hx, c);
}

template <typename T, int BN, int WM, int WN, int CPR, bool BNEPI>
__global__ __launch_bounds__((256 / WM) * (BN / WN) * 64, BNEPI ? 1 : 2) void conv3x3_tile_fwd_kernel(TileArgs a)
{
    constexpr int NT = (256 / WM) * (BN / WN) * 64;
    constexpr int EPC = 16 / sizeof(T);
    constexpr int KC = CPR * EPC;
    constexpr int HALO_BYTES = NPIX * CPR * 16;
    constexpr int WB_BYTES = BN * CPR * 16;
    constexpr int H_ITERS = (NPIX * CPR + NT - 1) / NT;
    constexpr int B_ITERS = (BN * CPR + NT - 1) / NT;
    constexpr int TI = WM / 32, TJ = WN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_halo = smem;
    unsigned char* s_w = smem + HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
    int bt = blockIdx.x;
    const int txi = bt % a.tiles_x; bt /= a.tiles_x;
    const int tyi = bt % a.tiles_y; const int n = bt / a.tiles_y;
    const int ty0 = tyi * TILE, tx0 = txi * TILE;
    const int n0 = blockIdx.y * BN;
    const T* __restrict__ xg = (const T*)a.x + (size_t)n * a.H * a.W * a.ldx;
    const T* __restrict__ wg = (const T*)a.w;
    const bool has_pro = a.pro_scale != nullptr;
    const float relu_lo = a.pro_relu ? 0.f : -__builtin_inff();
    const int ncb = (a.Cin + KC - 1) / KC;

    // A-fragment rows of this lane
    int prow[TI], pcol[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        int row = wm0 + i * 32 + lr;
        prow[i] = row >> 4; pcol[i] = row & 15;
    }

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 breg[B_ITERS];
    auto load_w = [&](int cb, int tap) {
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            int p = tid + i * NT;
            int brow = p / CPR, bch = p % CPR;
            int c = cb * KC + bch * EPC;
            bool ok = (B_ITERS * NT == BN * CPR || p < BN * CPR) && n0 + brow < a.Cout && c < a.Cin;
            size_t off = ok ? (((size_t)(n0 + brow) * 9 + tap) * a.Cin + c) : (size_t)0;
            u32x4 v = *(const u32x4*)(wg + off);
            const u32x4 z = {0u, 0u, 0u, 0u};
            breg[i] = ok ? v : z;
        }
    };
    auto store_w = [&](int buf) {
        unsigned char* sb = s_w + buf * WB_BYTES;
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            int p = tid + i * NT;
            if (B_ITERS * NT == BN * CPR || p < BN * CPR) *(u32x4*)(sb + swz_off<CPR>(p / CPR, p % CPR)) = breg[i];
        }
    };

    int wbuf = 0;
    TSTAMP_INIT();
    TSTAMP(10);
    load_w(0, 0);
    for (int cb = 0; cb < ncb; ++cb) {
        TSTAMP(11);
        // ---- stage the (transformed) halo of this channel block; previous readers are done (barrier at loop end).
        // Batches of HB pieces per thread in a NON-unrolled loop: keeps the index math out of long-lived registers.
        {
            constexpr int HB = 6, NB = (H_ITERS + HB - 1) / HB;
            const int c0 = cb * KC;
#pragma unroll 1
            for (int b = 0; b < NB; ++b) {
                u32x4 hreg[HB]; int hl[HB]; bool hk[HB]; int hc[HB];
#pragma unroll
                for (int i = 0; i < HB; ++i) {
                    const int q = tid + (b * HB + i) * NT;
                    const int pix = q / CPR, ch = q % CPR;
                    const int hy = pix / HPITCH, hx = pix - hy * HPITCH;
                    const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
                    const int c = c0 + ch * EPC;
                    const bool inr = q < NPIX * CPR;
                    const bool ok = inr && c < a.Cin && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    hk[i] = ok; hc[i] = ok ? c : 0;
                    hl[i] = inr ? swz_halo<CPR>(hy, hx, ch) : -1;
                    hreg[i] = *(const u32x4*)(xg + (ok ? (size_t)(iy * a.W + ix) * a.ldx + c : (size_t)0));
                }
                if (has_pro) {
#pragma unroll
                    for (int i = 0; i < HB; ++i) {
                        float f[EPC];
                        Vec16<T>::unpack(hreg[i], f);
#pragma unroll
                        for (int j = 0; j < EPC; j += 4) {
                            f32x4 s4 = *(const f32x4*)(a.pro_scale + hc[i] + j), t4 = *(const f32x4*)(a.pro_shift + hc[i] + j);
#pragma unroll
                            for (int q = 0; q < 4; ++q) f[j + q] = fmaxf(fmaf(f[j + q], s4[q], t4[q]), relu_lo);
                        }
                        hreg[i] = Vec16<T>::pack(f);
                    }
                }
                const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int i = 0; i < HB; ++i)
                    if (hl[i] >= 0) *(u32x4*)(s_halo + hl[i]) = hk[i] ? hreg[i] : z;
            }
        }
        TSTAMP(12);
        store_w(wbuf);
        __syncthreads();
        TSTAMP(13);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            // prefetch the next weight tile (next tap, or tap 0 of the next channel block)
            const bool more = (tap + 1 < 9) || (cb + 1 < ncb);
            if (more) load_w(tap + 1 < 9 ? cb : cb + 1, tap + 1 < 9 ? tap + 1 : 0);
            const int kh = tap / 3, kw = tap - kh * 3;
            const unsigned char* sb = s_w + wbuf * WB_BYTES;
            int ay[TI], ax[TI];
#pragma unroll
            for (int i = 0; i < TI; ++i) { ay[i] = prow[i] + kh; ax[i] = pcol[i] + kw; }
#pragma unroll
            for (int s = 0; s < CPR / 2; ++s) {
                u32x4 af[TI], bfr[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = *(const u32x4*)(s_halo + swz_halo<CPR>(ay[i], ax[i], 2 * s + lh));
#pragma unroll
                for (int j = 0; j < TJ; ++j) bfr[j] = *(const u32x4*)(sb + swz_off<CPR>(wn0 + j * 32 + lr, 2 * s + lh));
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) MmaT<T>::run(af[i], bfr[j], acc[i][j]);
            }
            TSTAMP(14);
            if (more && tap + 1 < 9) store_w(wbuf ^ 1);   // the next cb's tap-0 tile is stored after its halo
            __syncthreads();
            TSTAMP(15);
            if (tap + 1 < 9) wbuf ^= 1;
        }
        wbuf ^= 1;
    }
    TSTAMP(16);

    // ---- epilogue (same scheme as the generic kernel)
    // one statistics slot per row-wave, added in a fixed order (no float atomics on LDS: deterministic, and cheaper)
    constexpr int RW = 256 / WM;
    float* s_sum = (float*)(smem + 256 * BN * sizeof(T));          // [RW][2][BN]
    const bool do_stats = a.stat_sum != nullptr;
    __syncthreads();
    T* so = (T*)smem;
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
    __syncthreads();
    if (do_stats && tid < BN && n0 + tid < a.Cout) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < RW; ++w) { t1 += s_sum[w * 2 * BN + tid]; t2 += s_sum[w * 2 * BN + BN + tid]; }
        const size_t ro = (size_t)(blockIdx.x % a.stat_replicas) * a.stat_rstride;
        atomicAdd(&a.stat_sum[ro + n0 + tid], (double)t1);
        atomicAdd(&a.stat_sumsq[ro + n0 + tid], (double)t2);
    }
    constexpr int CH = BN / EPC;
    T* __restrict__ yg = (T*)a.y + (size_t)n * a.H * a.W * a.ldy;
    constexpr bool bnb = BNEPI;
    const T* __restrict__ bx = (const T*)a.epi.bn_x + (size_t)n * a.H * a.W * a.epi.ld_bn_x;
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
    constexpr int S_ITERS = (256 * CH) / NT;
    static_assert((256 * CH) % NT == 0, "store loop must divide evenly");
    const int colv = n0 + (tid % CH) * EPC;
    const bool cok = colv < a.Cout;
    u32x4 xr[S_ITERS];
    if (bnb) {
#pragma unroll
        for (int i = 0; i < S_ITERS; ++i) {
            int row = (tid + i * NT) / CH;
            size_t opix = (size_t)(ty0 + (row >> 4)) * a.W + tx0 + (row & 15);
            xr[i] = *(const u32x4*)(bx + (cok ? opix * a.epi.ld_bn_x + colv : (size_t)0));
        }
    }
#pragma unroll
    for (int i = 0; i < S_ITERS; ++i) {
        int p = tid + i * NT;
        int row = p / CH, ch = p - row * CH;
        size_t opix = (size_t)(ty0 + (row >> 4)) * a.W + tx0 + (row & 15);
        u32x4 v = *(const u32x4*)(so + row * BN + ch * EPC);
        if (bnb) {
            float g[EPC], xv[EPC];
            Vec16<T>::unpack(v, g);
            Vec16<T>::unpack(xr[i], xv);
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                if (a.epi.relu && !(fmaf(xv[j], esc[j], esh[j]) > 0.f)) g[j] = 0.f;
                if (!cok) g[j] = 0.f;
                e1[j] += g[j];
                e2[j] = fmaf(g[j], (xv[j] - emu[j]) * eis[j], e2[j]);
            }
            v = Vec16<T>::pack(g);
        }
        if (cok) *(u32x4*)(yg + opix * a.ldy + colv) = v;
    }
    if (bnb) {      // slot tid / CH of the (now idle) tile area per thread group, folded in a fixed order
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

template <typename T, int BN, int WM, int WN, int CPR, bool BNEPI> static int launch_tile_fwd_i(const TileArgs& a, hipStream_t st)
{
    constexpr int NT = (256 / WM) * (BN / WN) * 64;
    constexpr int MAIN = NPIX * CPR * 16 + 2 * BN * CPR * 16;
    constexpr int EPC_ = 16 / (int)sizeof(T);
    constexpr int EPI_TILE = 256 * BN * (int)sizeof(T) + (256 / WM) * 2 * BN * 4;
    constexpr int EPI_BN = (NT / (BN / EPC_)) * 2 * BN * 4;
    constexpr int EPI = EPI_TILE > EPI_BN ? EPI_TILE : EPI_BN;
    constexpr int LDS = MAIN > EPI ? MAIN : EPI;
    auto kern = conv3x3_tile_fwd_kernel<T, BN, WM, WN, CPR, BNEPI>;
    static DeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    dim3 grid(a.tiles_x * a.tiles_y * a.N, cdiv(a.Cout, BN));
    hipLaunchKernelGGL(kern, grid, dim3(NT), LDS, st, a);
    static const KName kn("conv3x3_tile_fwd_kernel", type_name<T>(), BN, WM, WN, CPR, BNEPI);
    SAUNET_CHECK_LAUNCH(kn.s);
    return SAUNET_OK;
}

template <typename T, int BN, int WM, int WN, int CPR> static int launch_tile_fwd(const TileArgs& a, hipStream_t st)
{
    return a.epi.bn_x ? launch_tile_fwd_i<T, BN, WM, WN, CPR, true>(a, st) : launch_tile_fwd_i<T, BN, WM, WN, CPR, false>(a, st);
}

template <typename T> static int dispatch_tile_fwd(const TileArgs& a, hipStream_t st)
{
    constexpr int EPC = 16 / sizeof(T);
    const bool narrow = a.Cin <= 4 * EPC;
    if (a.Cout <= 32) return narrow ? launch_tile_fwd<T, 32, 64, 32, 4>(a, st) : launch_tile_fwd<T, 32, 64, 32, 8>(a, st);
    // low-resolution maps with many channels (center, dec5: 32 pixel tiles): 64-channel output tiles double the number of
    // workgroups instead of leaving half of the CUs idle; the halo re-reads come from the L2
    const long blocks128 = (long)a.tiles_x * a.tiles_y * a.N * ((a.Cout + 127) / 128);
    // two resident workgroups per CU hide the fragment-read latency that one four-wave workgroup per CU exposes (MFMA pipe 19-28 % busy on
    // dec4 / dec5): below 512 workgroups the output-channel tile is halved (dec4 233 -> 196 us with 64-wide tiles, dec5 259 -> 233 us with 32-wide
    // ones; the extra halo re-reads come from the L2).
    constexpr long t128 = 512, t64 = 512;
    const long blocks64 = (long)a.tiles_x * a.tiles_y * a.N * ((a.Cout + 63) / 64);
    if (!a.epi.bn_x && blocks64 < t64) return narrow ? launch_tile_fwd<T, 32, 64, 32, 4>(a, st) : launch_tile_fwd<T, 32, 64, 32, 8>(a, st);
    if (a.Cout <= 64 || (blocks128 < t128 && !a.epi.bn_x))
        return narrow ? launch_tile_fwd<T, 64, 64, 64, 4>(a, st) : launch_tile_fwd<T, 64, 64, 64, 8>(a, st);
    return narrow ? launch_tile_fwd<T, 128, 128, 64, 4>(a, st) : launch_tile_fwd<T, 128, 128, 64, 8>(a, st);
}

// =====================================================================================================
// conv3x3 "resident" forward: PERSISTENT workgroups (one per CU) keep ALL packed weights of their cout tile in LDS for
// the whole launch and walk over 16x16 pixel tiles; while the MFMAs of unit u = (tile, channel block) run, the halo of
// unit u+1 is already in flight into registers (one ~1 us MFMA phase hides the L2/HBM latency), then it is BN+ReLU
// transformed and written to LDS.  LDS rows are padded (pitch = row bytes + 16) instead of XOR-swizzled, so every
// fragment address is  lane_base + compile-time immediate  (no address VALU in the 9-tap x 4-substep MFMA loop).
// Used when 9 * Cin_padded * BN weights fit next to one halo (DenseNet conv2 fwd/dgrad, the shape-stream ResBlocks).
constexpr int res_halo_rowb(int cpr) { return ((HPITCH * (cpr * 16 + 16) + 255) / 256) * 256; }

template <typename T, int BN, int WM, int WN, int CPR, bool BNEPI>
__global__ __launch_bounds__((256 / WM) * (BN / WN) * 64) void conv3x3_res_fwd_kernel(TileArgs a)
{
    constexpr int NT = (256 / WM) * (BN / WN) * 64;
    constexpr int EPC = 16 / sizeof(T);
    constexpr int KC = CPR * EPC;
    constexpr int PITCH = CPR * 16 + 16;
    // halo rows start on multiples of 256 B: with the odd chunk pitch (9 or 5 chunks per pixel) 16 consecutive columns then hit 16 different
    // 16-byte bank groups whichever of the two halo rows of an A-fragment lane group they sit in (see swz_halo above); packed rows (18 pixels =
    // 162 chunks = 2 mod 16) made columns x+12, x+13 of one row collide with x+4, x+5 of the next
    constexpr int ROWB = res_halo_rowb(CPR);
    constexpr int HALO_BYTES = HPITCH * ROWB;
    constexpr int WTAP = BN * PITCH;                 // one (cb, tap) weight tile
    constexpr int H_ITERS = (NPIX * CPR + NT - 1) / NT;
    constexpr int TI = WM / 32, TJ = WN / 32;
    static_assert(NT % CPR == 0, "chunk index must be thread-invariant");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // region 0: halo tile, re-used as the output staging tile of the epilogue (whichever is larger); then the weights
    constexpr int EPI_BYTES = 256 * BN * (int)sizeof(T);
    constexpr int R0_BYTES = HALO_BYTES > EPI_BYTES ? HALO_BYTES : EPI_BYTES;
    unsigned char* s_halo = smem;
    unsigned char* s_w = smem + R0_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
    const int ncb = (a.Cin + KC - 1) / KC;
    const int ntile = a.tiles_x * a.tiles_y * a.N;
    const int nnt = (a.Cout + BN - 1) / BN;
    const T* __restrict__ wg = (const T*)a.w;
    const bool has_pro = a.pro_scale != nullptr || a.bnp.gamma != nullptr;
    const float relu_lo = a.pro_relu ? 0.f : -__builtin_inff();
    const int chunk = tid % CPR;

    // work items: (n-tile, pixel tile); a workgroup keeps one n-tile as long as possible so its weights stay resident
    const int items = ntile * nnt;
    const int per = (items + gridDim.x - 1) / gridDim.x;
    const int it0 = blockIdx.x * per, it1 = min(it0 + per, items);
    if (it0 >= it1) return;

    // block-lifetime accumulators (statistics / BN-backward sums): flushed to global memory once per n-tile, not per tile
    // one slot per wave, written by a unique owner lane (plain read-modify-write) and folded in wave order at the flush: no float atomics on LDS
    constexpr int NW = NT / 64;
    float* s_acc = (float*)(smem + a.lds_acc_off);      // [NW][2][BN]
    for (int i = tid; i < NW * 2 * BN; i += NT) s_acc[i] = 0.f;
    // BatchNorm prologue vectors live in LDS for the block's lifetime.  They must NOT be fetched from global memory inside the unit
    // loop: vmcnt retires in order, so waiting for a scale/shift load issued after the halo prefetch drains the whole prefetch queue
    // (s_waitcnt vmcnt(0) right before the commit -- the prefetch then overlaps nothing).  LDS reads count on lgkmcnt instead.
    float* s_pro = s_acc + NW * 2 * BN;                  // [2][ncb * KC]
    // (filled further down, behind the weight copy and the first halo prefetches: the fill's own loads -- with a consumer-side BatchNorm finalize
    // two dependent round trips to the statistic replicas -- then overlap those instead of delaying them)
    auto flush_acc = [&](int nt) {
        __syncthreads();
        const int n0f = nt * BN;
        if (tid < BN && n0f + tid < a.Cout) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { t1 += s_acc[(w * 2) * BN + tid]; t2 += s_acc[(w * 2 + 1) * BN + tid]; }
            if constexpr (BNEPI) {
                const size_t ro = (size_t)(blockIdx.x % a.epi.sums_replicas) * a.epi.sums_rstride;
                atomicAdd(&a.epi.sums[ro + n0f + tid], (double)t1);
                atomicAdd(&a.epi.sums[ro + a.Cout + n0f + tid], (double)t2);
            } else if (a.stat_sum != nullptr) {
                const size_t ro = (size_t)(blockIdx.x % a.stat_replicas) * a.stat_rstride;
                atomicAdd(&a.stat_sum[ro + n0f + tid], (double)t1);
                atomicAdd(&a.stat_sumsq[ro + n0f + tid], (double)t2);
            }
        }
        __syncthreads();
        for (int i = tid; i < NW * 2 * BN; i += NT) s_acc[i] = 0.f;
    };

    int cur_nt = -1;
    // TWO register stages: the halo loads of units u+1 and u+2 are both in flight while the MFMAs of unit u run, so a load has two
    // MFMA phases plus a commit to land and the CU always has ~80 KB outstanding (one stage left HBM idle during every commit: the
    // DenseNet conv2 forward ran at a third of the per-CU load rate).  The accumulators are small here, the registers are free.
    u32x4 hregA[H_ITERS], hregB[H_ITERS];
    bool hokA[H_ITERS], hokB[H_ITERS];

    // Prefetch cursor: the (tile, channel block) unit whose halo is issued next.  Units are visited in order, so the tile coordinates advance
    // incrementally (no integer division in the loop: the five runtime div/mod of a from-scratch decode cost ~1000 cycles per issue), and
    // everything that depends only on the thread (its halo pixel and chunk per piece) is decoded once.
    int poff[H_ITERS], phyx[H_ITERS];
#pragma unroll
    for (int i = 0; i < H_ITERS; ++i) {
        const int q = tid + i * NT;
        const int pix = q / CPR;
        const int hy = pix / HPITCH, hx = pix - hy * HPITCH;
        poff[i] = (hy * a.W + hx) * a.ldx;
        phyx[i] = q < NPIX * CPR ? (hy | (hx << 8)) : (0x7f | (0x7f << 8));      // pieces beyond the halo never pass the bounds test
    }
    int p_cb = 0, p_txi, p_tyi, p_n;
    {
        int bt = it0 % ntile;
        p_txi = bt % a.tiles_x; bt /= a.tiles_x;
        p_tyi = bt % a.tiles_y; p_n = bt / a.tiles_y;
    }
    auto issue_halo = [&](u32x4 (&hreg)[H_ITERS], bool (&hok)[H_ITERS]) {   // global -> registers (no wait); advances the cursor by one unit
        const int c = p_cb * KC + chunk * EPC;
        const bool cok = c < a.Cin;
        const int y0 = p_tyi * TILE - 1, x0 = p_txi * TILE - 1;
        const T* xg = (const T*)a.x + ((size_t)p_n * a.H * a.W + (ptrdiff_t)y0 * a.W + x0) * a.ldx + (cok ? c : 0);
#pragma unroll
        for (int i = 0; i < H_ITERS; ++i) {
            const int iy = y0 + (phyx[i] & 0xff), ix = x0 + (phyx[i] >> 8);
            const bool ok = cok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            hok[i] = ok;
            hreg[i] = *(const u32x4*)(ok ? xg + poff[i] : (const T*)a.x);
        }
        if (++p_cb == ncb) {
            p_cb = 0;
            if (++p_txi == a.tiles_x) { p_txi = 0; if (++p_tyi == a.tiles_y) { p_tyi = 0; if (++p_n == a.N) p_n = 0; } }
        }
    };
    auto commit_halo = [&](int cb, u32x4 (&hreg)[H_ITERS], bool (&hok)[H_ITERS]) {            // transform + registers -> LDS
        const int c = cb * KC + chunk * EPC;
        const int cs = c < a.Cin ? c : 0;
        float sc[EPC], sh[EPC];
        if (has_pro) {
            const int cpad = ncb * KC;
#pragma unroll
            for (int j = 0; j < EPC; j += 4) {
                f32x4 s4 = *(const f32x4*)(s_pro + cs + j), t4 = *(const f32x4*)(s_pro + cpad + cs + j);
#pragma unroll
                for (int q = 0; q < 4; ++q) { sc[j + q] = s4[q]; sh[j + q] = t4[q]; }
            }
        }
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < H_ITERS; ++i) {
            const int q = tid + i * NT;
            u32x4 v = hreg[i];
            if (has_pro) {
                float f[EPC];
                Vec16<T>::unpack(v, f);
#pragma unroll
                for (int j = 0; j < EPC; ++j) f[j] = fmaxf(fmaf(f[j], sc[j], sh[j]), relu_lo);
                v = Vec16<T>::pack(f);
            }
            if (q < NPIX * CPR) *(u32x4*)(s_halo + (phyx[i] & 0xff) * ROWB + (phyx[i] >> 8) * PITCH + chunk * 16) = hok[i] ? v : z;
        }
    };

    f32x16 acc[TI][TJ];
    // lane base addresses (bytes) of the A / B fragments
    int abase[TI], bbase[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int row = wm0 + i * 32 + lr;
        abase[i] = (row >> 4) * ROWB + (row & 15) * PITCH + lh * 16;
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) bbase[j] = (wn0 + j * 32 + lr) * PITCH + lh * 16;

    const int nunits = (it1 - it0) * ncb;
    TSTAMP_INIT();
    TSTAMP(0);
    constexpr bool TWO = BN < 128;      // the 128-wide tile has no registers to spare for a second stage
    constexpr int AHEAD = TWO ? 2 : 1;
    // The block's first n-tile of weights goes to LDS here, before anything else is live in registers: 9 pieces per thread in flight, so the
    // copy costs about one memory latency per batch (piece by piece inside the unit loop it was one latency PER PIECE: 19k cycles for the 73 KB
    // of a 128 -> 32 layer, all of it exposed on the small maps -- s_memtime stamps).  The first halo prefetches are issued behind the last
    // batch's loads and land while the weights are stored.
    {
        constexpr int WB = 9;
        const int n0 = (it0 / ntile) * BN;
        const int pieces = ncb * 9 * BN * CPR;
        const int nb = (pieces + NT * WB - 1) / (NT * WB);
        for (int b = 0; b < nb; ++b) {
            u32x4 v[WB];
#pragma unroll
            for (int u = 0; u < WB; ++u) {
                const int q = tid + (b * WB + u) * NT;
                const int ch = q % CPR; int t = q / CPR;
                const int brow = t % BN; t /= BN;
                const int tap = t % 9, cb = t / 9;
                const int c = cb * KC + ch * EPC;
                v[u] = u32x4{0u, 0u, 0u, 0u};
                if (q < pieces && n0 + brow < a.Cout && c < a.Cin) v[u] = *(const u32x4*)(wg + ((size_t)(n0 + brow) * 9 + tap) * a.Cin + c);
            }
            if (b == nb - 1) {
                issue_halo(hregA, hokA);
                if (TWO && nunits > 1) issue_halo(hregB, hokB);
            }
#pragma unroll
            for (int u = 0; u < WB; ++u) {
                const int q = tid + (b * WB + u) * NT;
                const int ch = q % CPR; int t = q / CPR;
                const int brow = t % BN; t /= BN;
                const int tap = t % 9, cb = t / 9;
                if (q < pieces) *(u32x4*)(s_w + (cb * 9 + tap) * WTAP + brow * PITCH + ch * 16) = v[u];
            }
        }
        cur_nt = it0 / ntile;
    }
    if (a.bnp.gamma != nullptr) bn_prologue_fill<NT>(a.bnp, a.Cin, ncb * KC, s_pro, blockIdx.x == 0);
    else if (has_pro) {
        const int cpad = ncb * KC;
        for (int i = tid; i < cpad; i += NT) { s_pro[i] = i < a.Cin ? a.pro_scale[i] : 0.f; s_pro[cpad + i] = i < a.Cin ? a.pro_shift[i] : 0.f; }
    }

    // Compute cursor (the unit whose MFMAs run): advanced incrementally like the prefetch cursor.
    int c_cb = 0, c_nt = it0 / ntile, c_txi, c_tyi, c_n;
    {
        int bt = it0 % ntile;
        c_txi = bt % a.tiles_x; bt /= a.tiles_x;
        c_tyi = bt % a.tiles_y; c_n = bt / a.tiles_y;
    }
    // One unit = one (tile, channel block).  The body is instantiated once per register stage and the unit loop below is unrolled by two,
    // so each stage is a FIXED set of registers in straight-line code: selecting the stage with a runtime `unit & 1` made the compiler shuffle
    // the two register sets through v_mov copies behind an s_waitcnt vmcnt(0), i.e. it waited for the loads it had just issued (3-6k cycles on
    // every second unit, measured with s_memtime stamps).
    auto unit_body = [&](u32x4 (&hreg)[H_ITERS], bool (&hok)[H_ITERS], int unit) {
        const int n0 = c_nt * BN;
        if (c_cb == 0) {
            if (c_nt != cur_nt) {   // (re)load this n-tile's weights: [cb][tap][BN rows][PITCH]
                if (cur_nt >= 0) flush_acc(cur_nt);
                __syncthreads();
                const int pieces = ncb * 9 * BN * CPR;
                for (int q = tid; q < pieces; q += NT) {
                    const int ch = q % CPR; int t = q / CPR;
                    const int brow = t % BN; t /= BN;
                    const int tap = t % 9, cb = t / 9;
                    const int c = cb * KC + ch * EPC;
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (n0 + brow < a.Cout && c < a.Cin) v = *(const u32x4*)(wg + ((size_t)(n0 + brow) * 9 + tap) * a.Cin + c);
                    *(u32x4*)(s_w + (cb * 9 + tap) * WTAP + brow * PITCH + ch * 16) = v;
                }
                cur_nt = c_nt;
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        TSTAMP(1);
        __syncthreads();                 // previous unit's fragment reads (and the epilogue's use of the halo area) are done
        TSTAMP(2);
        commit_halo(c_cb, hreg, hok);
        TSTAMP(3);
        __syncthreads();
        TSTAMP(4);
        if (unit + AHEAD < nunits) issue_halo(hreg, hok);      // this stage is free again: refill it with unit u+2 (u+1 with one stage)
        TSTAMP(5);
        const unsigned char* wb = s_w + c_cb * 9 * WTAP;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int aoff = (tap / 3) * ROWB + (tap % 3) * PITCH;
#pragma unroll
            for (int s = 0; s < CPR / 2; ++s) {
                u32x4 af[TI], bfr[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = *(const u32x4*)(s_halo + abase[i] + aoff + s * 32);
#pragma unroll
                for (int j = 0; j < TJ; ++j) bfr[j] = *(const u32x4*)(wb + bbase[j] + tap * WTAP + s * 32);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) MmaT<T>::run(af[i], bfr[j], acc[i][j]);
            }
        }
        TSTAMP(6);
        if (++c_cb < ncb) return;
        c_cb = 0;
        // ---- epilogue for this tile (output staged in the halo area)
        __syncthreads();
        const int n = c_n, ty0 = c_tyi * TILE, tx0 = c_txi * TILE;
        float* s_sum = s_acc + (wave * 2) * BN;           // this wave's slot
        float* s_sq = s_sum + BN;
        const bool do_stats = !BNEPI && a.stat_sum != nullptr;
        T* so = (T*)s_halo;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = wn0 + j * 32 + lr;
            const float bv = (a.bias != nullptr && n0 + col < a.Cout) ? a.bias[n0 + col] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r];
                    s1 += v; s2 += v * v;
                    int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    Elem<T>::store(so + row * BN + col, a.act_relu ? fmaxf(v + bv, 0.f) : v + bv);
                }
            if (do_stats) {
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                if (lh == 0) { s_sum[col] += s1; s_sq[col] += s2; }
            }
        }
        __syncthreads();
        constexpr int CH = BN / EPC;
        constexpr int S_ITERS = (256 * CH) / NT;
        static_assert((256 * CH) % NT == 0, "store loop must divide evenly");
        T* __restrict__ yg = (T*)a.y + (size_t)n * a.H * a.W * a.ldy;
        const int colv = n0 + (tid % CH) * EPC;
        const bool cok = colv < a.Cout;
        if constexpr (BNEPI) {
            const T* __restrict__ bx = (const T*)a.epi.bn_x + (size_t)n * a.H * a.W * a.epi.ld_bn_x;
            float e1[EPC], e2[EPC], esc[EPC], esh[EPC], emu[EPC], eis[EPC];
            const int cs = cok ? colv : 0;
#pragma unroll
            for (int j = 0; j < EPC; j += 4) {
                f32x4 v0 = *(const f32x4*)(a.epi.scale + cs + j), v1 = *(const f32x4*)(a.epi.shift + cs + j);
                f32x4 v2 = *(const f32x4*)(a.epi.mean + cs + j), v3 = *(const f32x4*)(a.epi.invstd + cs + j);
#pragma unroll
                for (int q = 0; q < 4; ++q) { esc[j + q] = v0[q]; esh[j + q] = v1[q]; emu[j + q] = v2[q]; eis[j + q] = v3[q]; e1[j + q] = 0.f; e2[j + q] = 0.f; }
            }
            u32x4 xr[S_ITERS];
#pragma unroll
            for (int i = 0; i < S_ITERS; ++i) {
                int row = (tid + i * NT) / CH;
                size_t opix = (size_t)(ty0 + (row >> 4)) * a.W + tx0 + (row & 15);
                xr[i] = *(const u32x4*)(bx + (cok ? opix * a.epi.ld_bn_x + colv : (size_t)0));
            }
#pragma unroll
            for (int i = 0; i < S_ITERS; ++i) {
                int p = tid + i * NT;
                int row = p / CH, ch = p - row * CH;
                size_t opix = (size_t)(ty0 + (row >> 4)) * a.W + tx0 + (row & 15);
                float g[EPC], xv[EPC];
                Vec16<T>::unpack(*(const u32x4*)(so + row * BN + ch * EPC), g);
                Vec16<T>::unpack(xr[i], xv);
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    if (a.epi.relu && !(fmaf(xv[j], esc[j], esh[j]) > 0.f)) g[j] = 0.f;
                    if (!cok) g[j] = 0.f;
                    e1[j] += g[j];
                    e2[j] = fmaf(g[j], (xv[j] - emu[j]) * eis[j], e2[j]);
                }
                if (cok) *(u32x4*)(yg + opix * a.ldy + colv) = Vec16<T>::pack(g);
            }
            // the 64 / CH lanes of the wave that own the same channel chunk: fixed xor tree, then one owner lane per chunk adds into the wave's slot
            static_assert(64 % CH == 0, "channel chunks per row divide the wave");
            const int ch = lane % CH;
#pragma unroll
            for (int off = CH; off < 64; off <<= 1) {
#pragma unroll
                for (int j = 0; j < EPC; ++j) { e1[j] += __shfl_xor(e1[j], off, 64); e2[j] += __shfl_xor(e2[j], off, 64); }
            }
            if (lane < CH) {
#pragma unroll
                for (int j = 0; j < EPC; ++j) { s_sum[ch * EPC + j] += e1[j]; s_sq[ch * EPC + j] += e2[j]; }
            }
        } else {
#pragma unroll
            for (int i = 0; i < S_ITERS; ++i) {
                int p = tid + i * NT;
                int row = p / CH, ch = p - row * CH;
                size_t opix = (size_t)(ty0 + (row >> 4)) * a.W + tx0 + (row & 15);
                if (cok) {
                    u32x4 v = *(const u32x4*)(so + row * BN + ch * EPC);
                    if (a.epi.accumulate) {        // y += conv: the residual branch's gradient is already in y (block-uniform)
                        float f[EPC], g[EPC];
                        Vec16<T>::unpack(v, f); Vec16<T>::unpack(*(const u32x4*)(yg + opix * a.ldy + colv), g);
#pragma unroll
                        for (int j = 0; j < EPC; ++j) f[j] += g[j];
                        v = Vec16<T>::pack(f);
                    }
                    *(u32x4*)(yg + opix * a.ldy + colv) = v;
                }
            }
        }
        if (++c_txi == a.tiles_x) { c_txi = 0; if (++c_tyi == a.tiles_y) { c_tyi = 0; if (++c_n == a.N) { c_n = 0; ++c_nt; } } }
    };
    if constexpr (TWO) {
        for (int unit = 0; unit < nunits; unit += 2) {
            unit_body(hregA, hokA, unit);
            if (unit + 1 < nunits) unit_body(hregB, hokB, unit + 1);
        }
    } else {
        for (int unit = 0; unit < nunits; ++unit) unit_body(hregA, hokA, unit);
    }
    flush_acc(cur_nt);
}

template <typename T, int BN, int WM, int WN, int CPR> static constexpr int res_lds_bytes(int ncb)
{
    return HPITCH * res_halo_rowb(CPR) + ncb * 9 * BN * (CPR * 16 + 16);
}

template <typename T, int BN, int WM, int WN, int CPR, bool BNEPI> static int launch_res_fwd_i(const TileArgs& a_in, hipStream_t st)
{
    TileArgs a = a_in;
    constexpr int NT = (256 / WM) * (BN / WN) * 64;
    constexpr int EPC = 16 / sizeof(T);
    const int ncb = (a.Cin + CPR * EPC - 1) / (CPR * EPC);
    constexpr int HALO_B = HPITCH * res_halo_rowb(CPR), EPI_B = 256 * BN * (int)sizeof(T);
    int lds = (HALO_B > EPI_B ? HALO_B : EPI_B) + ncb * 9 * BN * (CPR * 16 + 16);
    a.lds_acc_off = lds; lds += (NT / 64) * 2 * BN * 4 + 2 * ncb * CPR * EPC * 4;      // accumulators (one slot per wave) + the prologue scale/shift vectors
    auto kern = conv3x3_res_fwd_kernel<T, BN, WM, WN, CPR, BNEPI>;
    static DeviceMaxLds attr;
    if (attr.raise(lds)) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int items = a.tiles_x * a.tiles_y * a.N * cdiv(a.Cout, BN);
    const int per_cu = (160 * 1024) / lds > 0 ? (160 * 1024) / lds : 1;
    int blocks = 256 * per_cu; if (blocks > items) blocks = items;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, st, a);
    static const KName kn("conv3x3_res_fwd_kernel", type_name<T>(), BN, WM, WN, CPR, BNEPI);
    SAUNET_CHECK_LAUNCH(kn.s);
    return SAUNET_OK;
}

// does the resident-weight kernel apply, and with which tile?  (weights of one n-tile + one halo must fit in 160 KB)
static bool res_fwd_applies(int elem_bytes, int Cin, int Cout, long tiles)
{
    const int EPC = 16 / elem_bytes;
    const bool narrow = Cin <= 4 * EPC;
    const int cpr = narrow ? 4 : 8;
    const int ncb = (Cin + cpr * EPC - 1) / (cpr * EPC);
    const int pitch = cpr * 16 + 16;
    const int bn = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128);
    const long halo_b = (long)HPITCH * res_halo_rowb(cpr), epi_b = 256L * bn * elem_bytes;
    long lds = (halo_b > epi_b ? halo_b : epi_b) + (long)ncb * 9 * bn * pitch + 8 * 2 * bn * 4 + 2L * ncb * cpr * EPC * 4;   // (at most 8 wave slots)
    return lds <= 154 * 1024 && tiles >= 32;   // even at one tile per block a single bulk weight load beats nine dependent per-tap loads
}

template <typename T> static int dispatch_res_fwd(const TileArgs& a, hipStream_t st, bool* handled)
{
    constexpr int EPC = 16 / sizeof(T);
    const bool narrow = a.Cin <= 4 * EPC;
    const int bn = a.Cout <= 32 ? 32 : (a.Cout <= 64 ? 64 : 128);
    *handled = res_fwd_applies((int)sizeof(T), a.Cin, a.Cout, (long)a.tiles_x * a.tiles_y * a.N);
    if (!*handled) return SAUNET_OK;
#define RES(BN_, WM_, WN_, CPR_) (a.epi.bn_x ? launch_res_fwd_i<T, BN_, WM_, WN_, CPR_, true>(a, st) : launch_res_fwd_i<T, BN_, WM_, WN_, CPR_, false>(a, st))
    // 8 waves per block (one resident block per CU: the weights + one halo fill the LDS): twice the waves to hide the halo latency
    if (bn == 32) return narrow ? RES(32, 64, 32, 4) : RES(32, 32, 32, 8);
    if (bn == 64) return narrow ? RES(64, 64, 64, 4) : RES(64, 64, 32, 8);
    return narrow ? RES(128, 128, 64, 4) : RES(128, 128, 64, 8);
#undef RES
}

// y += conv(x, w) (saunet_bn_epilogue with bn_x == NULL and accumulate == 1: a residual branch's gradient is already in y) is served by the
// resident-weight kernel's plain epilogue only
bool tile_fwd_accumulate_supported(const saunet_conv_desc* d)
{
    return !d->transposed && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->H % TILE == 0 && d->W % TILE == 0 && d->Ho == d->H &&
           d->Wo == d->W && (d->dtype == SAUNET_BF16 || d->dtype == SAUNET_F32) &&
           res_fwd_applies(d->dtype == SAUNET_BF16 ? 2 : 4, d->Cin, d->Cout, (long)d->N * (d->H / TILE) * (d->W / TILE));
}

bool tile_fwd_supported(const saunet_conv_desc* d)
{
    return !d->transposed && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->H % TILE == 0 && d->W % TILE == 0 &&
           d->Ho == d->H && d->Wo == d->W;
}

int bn_prologue_finalize(const saunet_bn_prologue* p, int Cin, hipStream_t st);

int tile_forward(const saunet_conv_desc* d, const void* x, const void* w, const float* bias, const float* ps, const float* psh,
                 void* y, double* ssum, double* ssq, const saunet_bn_epilogue* epi, hipStream_t st, const saunet_bn_prologue* bnp)
{
    TileArgs a;
    if (bnp) a.bnp = *bnp; else a.bnp.gamma = nullptr;
    if (epi) { a.epi = *epi; if (a.epi.sums_replicas < 1) a.epi.sums_replicas = 1; } else { a.epi.bn_x = nullptr; a.epi.accumulate = 0; }
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.pro_scale = ps; a.pro_shift = psh; a.stat_sum = ssum; a.stat_sumsq = ssq;
    a.stat_replicas = d->stat_replicas > 1 ? d->stat_replicas : 1; a.stat_rstride = d->stat_rstride;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx; a.Cout = d->Cout; a.ldy = d->ldy;
    a.pro_relu = d->pro_relu; a.act_relu = d->epi_relu; a.tiles_y = d->H / TILE; a.tiles_x = d->W / TILE;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) return set_error(SAUNET_BAD_ALIGN, "conv: pointers must be 16-byte aligned");
    bool handled = false;
    // only the resident kernel derives the BatchNorm coefficients itself: in front of the tile kernel they are finalised by their own launch
    auto unfuse = [&]() -> int {
        if (!bnp) return SAUNET_OK;
        if (int rc = bn_prologue_finalize(bnp, d->Cin, st)) return rc;
        a.pro_scale = bnp->params; a.pro_shift = bnp->params + d->Cin; a.bnp.gamma = nullptr;
        return SAUNET_OK;
    };
    if (d->dtype == SAUNET_BF16) { int rc = dispatch_res_fwd<u16>(a, st, &handled); if (handled) return rc; if ((rc = unfuse())) return rc; return dispatch_tile_fwd<u16>(a, st); }
    if (d->dtype == SAUNET_F32 && f32_split_wanted((long)d->N * d->H * d->W)) { int rc = dispatch_res_fwd<f32s>(a, st, &handled); if (handled) return rc; if ((rc = unfuse())) return rc; return dispatch_tile_fwd<f32s>(a, st); }
    if (d->dtype == SAUNET_F32) { int rc = dispatch_res_fwd<float>(a, st, &handled); if (handled) return rc; if ((rc = unfuse())) return rc; return dispatch_tile_fwd<float>(a, st); }
    return set_error(SAUNET_BAD_DTYPE, "conv: dtype %d", d->dtype);
}

// Register prefetch of the next pixel tile (compile-time switch, OFF): measured round 3 -- 1x1 kernels neutral (they sit at the HBM floor), 3x3
// kernels SLOWER (block-1 conv2 92 -> 123 us, res1 263 -> 409 us): their 144 accumulator registers leave no room, the prefetch registers
// spill (76-80 B / lane), and the 3x3 inner loop is bound by LDS operand bandwidth (one fresh 1 KB B fragment per MFMA), not by load latency.
#ifndef SAUNET_WGRAD_PREFETCH
#define SAUNET_WGRAD_PREFETCH 0
#endif
// =====================================================================================================
// wgrad on pixel tiles: dW[co][ci][tap] += sum_{pixels of the tile} dy[p][co] * a[p (+) tap][ci]
struct TileWgradArgs {
    const void* x; const void* dy; float* dw;
    const float* pro_scale; const float* pro_shift;
    int N, H, W, Cin, ldx, Cout, lddy, pro_relu;
    int tiles_x, tiles_y, ntiles, ncit;   // tiles per row / column, total, number of ci tiles
    // haloed operand: element strides of (image, row, pixel); 0 = dense NHWC (H*W*ldx, W*ldx, ldx).
    // KS == 2 (ConvTranspose2d k4 s2 p1, see tile_wgrad_convt): the haloed operand is one PARITY sub-image of dy, blockIdx.y also carries the parity
    long xs_n, xs_r, xs_p;
    int ncot, Wo;
    long sM, sN;
    float* ws; long wsize;                 // per-group partial gradients [groups][wsize] (plain stores, reduced afterwards)
    saunet_wgrad_pending* pend;            // non-null: do not launch the reduction, describe it here (saunet_wgrad_reduce_multi runs it later)
};

// one ds_read_b64_tr_b16: every 16-lane group reads a [4 rows][16 cols] block of 16-bit elements (each lane
// supplies the address of 4 consecutive elements of one row) and receives one COLUMN (4 consecutive rows).
// Two of them (rows r..r+3 and r+4..r+7) make one 8-deep MFMA operand fragment.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
__device__ __forceinline__ void tr_read2(const unsigned char* p0, int pitch4, u32x4& out)
{
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + pitch4));
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
    out[0] = a[0]; out[1] = a[1]; out[2] = b[0]; out[3] = b[1];
}

// 16-byte chunk of EPC channels starting at channel c of a pixel row; the scalar form serves channel counts /
// strides that are not multiples of the chunk (the shape stream's C = 33, 17, 9, 1 ...)
template <typename T, bool ALIGNED> __device__ __forceinline__ u32x4 load_chunk(const T* row, int c, int C)
{
    if constexpr (ALIGNED) return *(const u32x4*)(row + c);
    else {
        constexpr int EPC = 16 / sizeof(T);
        float f[EPC];
#pragma unroll
        for (int j = 0; j < EPC; ++j) f[j] = (c + j < C) ? Elem<T>::load(row + c + j) : 0.f;
        return Vec16<T>::pack(f);
    }
}

// body shared by the single-problem and the grouped launch: (gx, ngroups) = pixel-tile group of this block (tiles gx, gx + ngroups, ...; its
// partial gradient goes to ws[gx]), by = channel tile
template <typename T, int KS, int TR, int CO_T, int CI_T, int WM, int WN, int KSPLIT, bool ALIGNED>
__device__ __forceinline__ void tile_wgrad_body(const TileWgradArgs& a, const int gx, const int ngroups, const int by, unsigned char* smem)
{
    constexpr int EPC = 16 / sizeof(T);
    constexpr int PAD = KS / 2;
    constexpr int HR = TR + 2 * PAD, HC = TILE + 2 * PAD;          // halo rows / cols
    constexpr int NPX = HR * HC, NPY = TR * TILE;
    // row pitch: a multiple of 16 bytes that is == 64 (mod 128) so the 4 rows of a transposing read (and the two
    // 16-lane groups that issue together) fall on disjoint banks
    constexpr int PY_RAW = CO_T * (int)sizeof(T), PX_RAW = CI_T * (int)sizeof(T);
    constexpr int PY = sizeof(T) == 2 ? ((PY_RAW % 128 == 64) ? PY_RAW : PY_RAW + 64) : PY_RAW;
    constexpr int PX = sizeof(T) == 2 ? ((PX_RAW % 128 == 64) ? PX_RAW : PX_RAW + 64) : PX_RAW;
    constexpr int MI = WM / 32, NI = WN / 32, TAPS = KS * KS;
    constexpr int CHY = CO_T / EPC, CHX = CI_T / EPC;
    constexpr int YI = (NPY * CHY + 255) / 256, XI = (NPX * CHX + 255) / 256;
    constexpr int XB = XI > 6 ? (XI + 1) / 2 : XI;   // stage the halo in two batches when it is large (VGPR budget)
    unsigned char* s_y = smem;
    unsigned char* s_x = smem + NPY * PY;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // KSPLIT > 0: the waves beyond the channel sub-tiles split K (tile rows), each holds all taps and they are summed through LDS at the end.
    // KSPLIT < 0: "tap split" -- -KSPLIT waves share the taps (wave w takes taps w, w + 4, w + 8), each runs all tile rows: a third of the
    //             accumulator registers (48 instead of 144 for 3x3), which is what makes room for the register prefetch of the next tile.
    constexpr bool TS = KSPLIT < 0;
    constexpr int KSP = TS ? 1 : KSPLIT, NTW = TS ? -KSPLIT : 1;
    constexpr int TPW = (KS * KS + NTW - 1) / NTW;                 // taps held per wave
    static_assert((CO_T / WM) * (CI_T / WN) * KSP * NTW == 4, "4 waves per block");
    constexpr int CWAVES = (CO_T / WM) * (CI_T / WN);
    const int cwave = wave % CWAVES, kwave = TS ? 0 : wave / CWAVES, twave = TS ? wave / CWAVES : 0;   // channel sub-tile / K slice / tap slice
    const int wm0 = (cwave / (CI_T / WN)) * WM, wn0 = (cwave % (CI_T / WN)) * WN;
    // KS == 2: weight gradient of ConvTranspose2d(k=4, s=2, p=1) as four 2x2-tap problems, one per output parity (py, px):
    //   dW[ci][co][kh0 + 2 th][kw0 + 2 tw] = sum x[n, iy, ix, ci] * D[n, iy + th + kh0 - 1, ix + tw + kw0 - 1, co],   D[i][j] = dy[2 i + 1 - kh0][2 j + 1 - kw0]
    // (kh0 = 1 - py, kw0 = 1 - px).  The kernel's haloed operand "x" is D (a strided view of dy, its channels are the layer's OUTPUT channels), its
    // "dy" operand is the layer input; the staged halo is the 3x3 one and the taps read rows / columns (kh + kh0, kw + kw0) of it.
    int kh0 = 0, kw0 = 0, byc = by;
    if constexpr (KS == 2) { const int per = a.ncot * a.ncit, par = by / per; byc = by - par * per; kh0 = 1 - (par >> 1); kw0 = 1 - (par & 1); }
    const int cot = byc / a.ncit, cit = byc - cot * a.ncit;
    const int co0 = cot * CO_T, ci0 = cit * CI_T;
    const size_t xs_p = a.xs_p ? (size_t)a.xs_p : (size_t)a.ldx, xs_r = a.xs_r ? (size_t)a.xs_r : (size_t)a.W * a.ldx,
                 xs_n = a.xs_n ? (size_t)a.xs_n : (size_t)a.H * a.W * a.ldx;
    const T* __restrict__ xg = (const T*)a.x + (KS == 2 ? ((size_t)(1 - kh0) * a.Wo + (1 - kw0)) * a.ldx : (size_t)0);
    const T* __restrict__ dyg = (const T*)a.dy;
    const bool has_pro = a.pro_scale != nullptr;
    const float relu_lo = a.pro_relu ? 0.f : -__builtin_inff();

    f32x16 acc[MI][NI][TPW];
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][t][r] = 0.f;

    const int li = lane & 15, lg = lane >> 4, nhalf = lg & 1, khalf = lg >> 1;

    // 256 threads stage chunk q = tid + i*256 and 256 % CHX == 0: a thread handles the SAME channel chunk of every halo pixel, so its
    // prologue coefficients are loaded once per block, not once per chunk per tile
    static_assert(!ALIGNED || 256 % CHX == 0, "a thread keeps one channel chunk");
    float pro_s[EPC], pro_t[EPC];
    if constexpr (ALIGNED) {
        const int c = ci0 + (tid % CHX) * EPC;
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
            const bool ok = has_pro && c + j < a.Cin;
            pro_s[j] = ok ? a.pro_scale[c + j] : 1.f; pro_t[j] = ok ? a.pro_shift[c + j] : 0.f;
        }
    }

    TSTAMP_INIT();
    TSTAMP(20);
    // ---- staging of one pixel tile: dy tile + activation halo, global -> registers -> (prologue) -> LDS
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    auto tile_coords = [&](int tile, int& n, int& ty0, int& tx0) {
        int bt = tile;
        const int txi = bt % a.tiles_x; bt /= a.tiles_x;
        const int tyi = bt % a.tiles_y; n = bt / a.tiles_y;
        ty0 = tyi * TR; tx0 = txi * TILE;
    };
    auto load_y = [&](int n, int ty0, int tx0, u32x4* yreg) {
#pragma unroll
        for (int i = 0; i < YI; ++i) {
            int q = tid + i * 256, pix = q / CHY, ch = q - pix * CHY;
            int c = co0 + ch * EPC;
            bool ok = (YI * 256 == NPY * CHY || q < NPY * CHY) && c < a.Cout;
            if constexpr (ALIGNED) {
                // tile origin = block-uniform 64-bit scalar arithmetic; per lane only a 32-bit offset inside the tile (the 64-bit multiply chains of
                // the absolute form were ~20 VALU instructions per 16-byte load)
                const T* ytile = dyg + (((size_t)n * a.H + ty0) * a.W + tx0) * a.lddy;
                const int rel = (((pix / TILE) * a.W + (pix % TILE)) * a.lddy + c) & -(int)ok;      // masked, not selected: a select becomes an exec-mask branch around the load
                yreg[i] = *(const u32x4*)(ytile + rel);
            } else {
                size_t off = ok ? (((size_t)n * a.H + ty0 + pix / TILE) * a.W + tx0 + (pix % TILE)) * a.lddy : (size_t)0;
                yreg[i] = load_chunk<T, ALIGNED>(dyg + off, ok ? c : 0, a.Cout);
            }
        }
    };
    auto store_y = [&](const u32x4* yreg) {
#pragma unroll
        for (int i = 0; i < YI; ++i) {
            int q = tid + i * 256, pix = q / CHY, ch = q - pix * CHY;
            bool ok = (YI * 256 == NPY * CHY || q < NPY * CHY) && co0 + ch * EPC < a.Cout;
            if (YI * 256 == NPY * CHY || q < NPY * CHY) *(u32x4*)(s_y + pix * PY + ch * 16) = ok ? yreg[i] : zero4;
        }
    };
    // chunks b0 .. b0 + cnt - 1 of the halo
    auto x_ok = [&](int ty0, int tx0, int i, int& c) {
        int q = tid + i * 256, pix = q / CHX, ch = q - pix * CHX;
        int hy = pix / HC, hx = pix - hy * HC;
        int iy = ty0 + hy - PAD, ix = tx0 + hx - PAD;
        c = ci0 + ch * EPC;
        return (i < XI) && q < NPX * CHX && c < a.Cin && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    };
    auto load_x = [&](int n, int ty0, int tx0, u32x4* xreg, int b0, int cnt) {
#pragma unroll
        for (int i = 0; i < cnt; ++i) {
            int q = tid + (b0 + i) * 256, pix = q / CHX;
            int hy = pix / HC, hx = pix - hy * HC;
            int iy = ty0 + hy - PAD, ix = tx0 + hx - PAD;
            int c; const bool ok = x_ok(ty0, tx0, b0 + i, c);
            if constexpr (ALIGNED) {
                const T* xtile = xg + (size_t)n * xs_n + (size_t)ty0 * xs_r + (size_t)tx0 * xs_p;      // block-uniform
                const int rel = ((hy - PAD) * (int)xs_r + (hx - PAD) * (int)xs_p + c) & -(int)ok;      // |rel| < 2^31: a few rows of the map
                xreg[i] = *(const u32x4*)(xtile + rel);
            } else {
                xreg[i] = load_chunk<T, ALIGNED>(xg + (ok ? (size_t)n * xs_n + (size_t)iy * xs_r + (size_t)ix * xs_p : (size_t)0), ok ? c : 0, a.Cin);
            }
        }
    };
    auto store_x = [&](int ty0, int tx0, u32x4* xreg, int b0, int cnt) {
        if (has_pro) {
#pragma unroll
            for (int i = 0; i < cnt; ++i) {
                int c; const bool ok = x_ok(ty0, tx0, b0 + i, c);
                const int cx = ok ? c : 0;
                float f[EPC];
                Vec16<T>::unpack(xreg[i], f);
                if constexpr (ALIGNED) {
#pragma unroll
                    for (int j = 0; j < EPC; ++j) f[j] = fmaxf(fmaf(f[j], pro_s[j], pro_t[j]), relu_lo);
                } else {
#pragma unroll
                    for (int j = 0; j < EPC; ++j) {
                        const int cc = cx + j < a.Cin ? cx + j : 0;
                        f[j] = (cx + j < a.Cin) ? fmaxf(fmaf(f[j], a.pro_scale[cc], a.pro_shift[cc]), relu_lo) : 0.f;
                    }
                }
                xreg[i] = Vec16<T>::pack(f);
            }
        }
#pragma unroll
        for (int i = 0; i < cnt; ++i) {
            int q = tid + (b0 + i) * 256, pix = q / CHX, ch = q - pix * CHX;
            int c; const bool ok = x_ok(ty0, tx0, b0 + i, c);
            if ((b0 + i < XI) && q < NPX * CHX) *(u32x4*)(s_x + pix * PX + ch * 16) = ok ? xreg[i] : zero4;
        }
    };
    // register prefetch: the NEXT tile's global loads are issued right after this tile is in LDS and fly during its matrix-core loop
    // (without it every tile pays load latency -> LDS -> barrier -> MFMA in sequence; two resident blocks per CU hide only part of it)
    constexpr bool PF = (SAUNET_WGRAD_PREFETCH || (TS && WM == 32) || KS == 2) && ALIGNED && sizeof(T) == 2 && (YI + XI) <= 16;   // (the 64 x 64 tap-split tile has no registers to spare)
    u32x4 pyreg[PF ? YI : 1], pxreg[PF ? XI : 1];
    if constexpr (PF) {
        if (gx < a.ntiles) { int n, ty0, tx0; tile_coords(gx, n, ty0, tx0); load_y(n, ty0, tx0, pyreg); load_x(n, ty0, tx0, pxreg, 0, XI); }
    }
    for (int tile = gx; tile < a.ntiles; tile += ngroups) {
        int n, ty0, tx0; tile_coords(tile, n, ty0, tx0);
        TSTAMP(21);
        __syncthreads();   // previous tile fully consumed
        TSTAMP(22);
        if constexpr (PF) {
            store_y(pyreg);
            TSTAMP(23);
            store_x(ty0, tx0, pxreg, 0, XI);
            TSTAMP(24);
            __syncthreads();
            if (tile + ngroups < a.ntiles) {
                int n2, ty2, tx2; tile_coords(tile + ngroups, n2, ty2, tx2);
                load_y(n2, ty2, tx2, pyreg); load_x(n2, ty2, tx2, pxreg, 0, XI);
            }
        } else {
            {
                u32x4 yreg[YI];
                load_y(n, ty0, tx0, yreg);
                store_y(yreg);
            }
            TSTAMP(23);
#pragma unroll
            for (int b0 = 0; b0 < XI; b0 += XB) {
                u32x4 xreg[XB];
                load_x(n, ty0, tx0, xreg, b0, XB);
                store_x(ty0, tx0, xreg, b0, XB);
            }
            TSTAMP(24);
            __syncthreads();
        }
        TSTAMP(25);
        // ---- one MFMA K-step = one tile row (16 pixels)
        auto k_step = [&](int ty) {
            if constexpr (sizeof(T) == 2) {
                u32x4 af[MI];
#pragma unroll
                for (int m = 0; m < MI; ++m) {
                    const unsigned char* base = s_y + (ty * TILE + 8 * khalf + (li >> 2)) * PY + (wm0 + m * 32 + 16 * nhalf + 4 * (li & 3)) * 2;
                    tr_read2(base, 4 * PY, af[m]);
                }
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    const int t = TS ? twave + NTW * j : j;
                    if (TS && t >= TAPS) continue;             // wave-uniform
                    const int kh = t / KS, kw = t - kh * KS;
#pragma unroll
                    for (int nn = 0; nn < NI; ++nn) {
                        u32x4 bf;
                        const unsigned char* base = s_x + ((ty + kh + kh0) * HC + kw + kw0 + 8 * khalf + (li >> 2)) * PX + (wn0 + nn * 32 + 16 * nhalf + 4 * (li & 3)) * 2;
                        tr_read2(base, 4 * PX, bf);
#pragma unroll
                        for (int m = 0; m < MI; ++m)
                            acc[m][nn][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[m]), __builtin_bit_cast(bf16x8_t, bf), acc[m][nn][j], 0, 0, 0);
                    }
                }
            } else if constexpr (std::is_same<T, f32s>::value) {
                // float32 storage, 3 x bf16 split products (common.h): K = 8 pixels per step, a lane half holds 4 consecutive pixels of its channel;
                // the dy fragment is split once per step, every x fragment once per (tap, channel tile)
                const int lr = lane & 31, lh = lane >> 5;
                static_assert(!TS || sizeof(T) == 2, "tap split: bf16 only");
#pragma unroll
                for (int k = 0; k < TILE; k += 8) {
                    F32Split As[MI];
#pragma unroll
                    for (int m = 0; m < MI; ++m) {
                        u32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = *(const unsigned*)(s_y + (ty * TILE + k + 4 * lh + j) * PY + (wm0 + m * 32 + lr) * 4);
                        As[m] = f32_split3(v);
                    }
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) {
                        const int kh = t / KS, kw = t - kh * KS;
#pragma unroll
                        for (int nn = 0; nn < NI; ++nn) {
                            u32x4 v;
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = *(const unsigned*)(s_x + ((ty + kh + kh0) * HC + kw + kw0 + k + 4 * lh + j) * PX + (wn0 + nn * 32 + lr) * 4);
                            const F32Split Bs = f32_split3(v);
#pragma unroll
                            for (int m = 0; m < MI; ++m) mma_f32_split_pre(As[m], Bs, acc[m][nn][t]);
                        }
                    }
                }
            } else {
                const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
                for (int k = 0; k < TILE; k += 2) {
                    float af[MI];
#pragma unroll
                    for (int m = 0; m < MI; ++m) af[m] = *(const float*)(s_y + (ty * TILE + k + lh) * PY + (wm0 + m * 32 + lr) * 4);
                    static_assert(!TS || sizeof(T) == 2, "tap split: bf16 only");
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) {
                        const int kh = t / KS, kw = t - kh * KS;
#pragma unroll
                        for (int nn = 0; nn < NI; ++nn) {
                            float bv = *(const float*)(s_x + ((ty + kh + kh0) * HC + kw + kw0 + k + lh) * PX + (wn0 + nn * 32 + lr) * 4);
#pragma unroll
                            for (int m = 0; m < MI; ++m) acc[m][nn][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bv, acc[m][nn][t], 0, 0, 0);
                        }
                    }
                }
            }
        };
        // 3x3, bf16, one wave per 32 x 32 channel tile holding all nine taps (ROLL): the x fragments of a halo row serve THREE consecutive tile rows
        // (as kh = 2, 1, 0) and the kw = 1 fragment is the kw = 0 one shifted by a pixel -- lane (channel, k half) holds 8 consecutive pixels, so
        // columns 1..8 are elements 1..7 of the kw = 0 fragment plus element 6 of the kw = 2 fragment (columns 2..9) of the SAME lane: three
        // v_alignbit and one more, no LDS.  Per tile row the loop reads the dy fragment and TWO x fragments (kw = 0, 2 of halo row ty + 2) instead
        // of nine: the transposing reads (one fresh 1 KB fragment per MFMA, 75 B/clk per CU) capped this loop at half the matrix-core rate.
        constexpr int RPW = TR / KSP;                 // tile rows per K wave (a contiguous block of rows when ROLL)
        constexpr bool ROLL = KS == 3 && sizeof(T) == 2 && !TS && MI == 1 && NI == 1 && TR % KSP == 0 && RPW <= 8;
        if constexpr (ROLL) {
            const int lx = (8 * khalf + (li >> 2)) * PX + (wn0 + 16 * nhalf + 4 * (li & 3)) * 2;
            auto load_row = [&](int hy, u32x4* f) {          // f[0], f[2] from LDS, f[1] derived
                const unsigned char* base = s_x + hy * HC * PX + lx;
                tr_read2(base, 4 * PX, f[0]);
                tr_read2(base + 2 * PX, 4 * PX, f[2]);
                f[1][0] = __builtin_amdgcn_alignbit(f[0][1], f[0][0], 16); f[1][1] = __builtin_amdgcn_alignbit(f[0][2], f[0][1], 16);
                f[1][2] = __builtin_amdgcn_alignbit(f[0][3], f[0][2], 16); f[1][3] = __builtin_amdgcn_alignbit(f[2][3], f[0][3], 16);
            };
            u32x4 rows[3][3];
            const int r0 = kwave * RPW;
            load_row(r0, rows[0]); load_row(r0 + 1, rows[1]);
#pragma unroll
            for (int tr = 0; tr < RPW; ++tr) {
                const int ty = r0 + tr;
                load_row(ty + 2, rows[(tr + 2) % 3]);
                u32x4 af;
                tr_read2(s_y + (ty * TILE + 8 * khalf + (li >> 2)) * PY + (wm0 + 16 * nhalf + 4 * (li & 3)) * 2, 4 * PY, af);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        acc[0][0][kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, rows[(tr + kh) % 3][kw]),
                                                                                       acc[0][0][kh * 3 + kw], 0, 0, 0);
            }
        } else
        if constexpr (KS == 2) {       // two tile rows per iteration: the second row's fragment reads are in flight behind the first row's MFMAs
            static_assert(KS != 2 || (KSP == 1 && TR % 2 == 0), "conv-transpose variant: unsplit K, even tile rows");
            for (int ty = 0; ty < TR; ty += 2) { k_step(ty); k_step(ty + 1); }
        } else {
            for (int ty = kwave; ty < TR; ty += KSP) k_step(ty);
        }
    }
    TSTAMP(26);
    const int lr = lane & 31, lh = lane >> 5;
    float* s_red = (float*)smem;   // [channel wave][KSP-1][16][64 lanes] floats per (m, n, tap) round
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int nn = 0; nn < NI; ++nn) {
            const int ci = ci0 + wn0 + nn * 32 + lr;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const int t = TS ? twave + NTW * j : j;
                if constexpr (KSP > 1) {
                    __syncthreads();
                    if (kwave > 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) s_red[((cwave * (KSP - 1) + kwave - 1) * 16 + r) * 64 + lane] = acc[m][nn][j][r];
                    }
                    __syncthreads();
                    if (kwave == 0) {
#pragma unroll
                        for (int k = 0; k < KSP - 1; ++k)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[m][nn][j][r] += s_red[((cwave * (KSP - 1) + k) * 16 + r) * 64 + lane];
                    }
                }
                if (kwave == 0 && t < TAPS) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int co = co0 + wm0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int toff = KS == 2 ? (kh0 + 2 * (t >> 1)) * 4 + kw0 + 2 * (t & 1) : t;
                        if (co < a.Cout && ci < a.Cin) a.ws[(size_t)gx * a.wsize + (size_t)co * a.sM + (size_t)ci * a.sN + toff] = acc[m][nn][j][r];
                    }
                }
            }
        }
}

template <typename T, int KS, int TR, int CO_T, int CI_T, int WM, int WN, int KSPLIT, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void conv_tile_wgrad_kernel(TileWgradArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    tile_wgrad_body<T, KS, TR, CO_T, CI_T, WM, WN, KSPLIT, ALIGNED>(a, blockIdx.x, gridDim.x, blockIdx.y, smem);
}

// ---- grouped launch: the weight gradients of up to SAUNET_WGRAD_GROUP_MAX convolutions of ONE geometry (same map, same kernel size; per-problem
// channel counts, operands and prologue vectors) in one grid.  A DenseNet block's backward defers the two weight gradients of every layer to
// the end of the block: on the low-resolution blocks a single layer's problem has 32-128 pixel tiles and needed 32-128 pixel groups to fill
// the chip (each writing a full partial gradient: the partials outweighed the activations); all layers together fill it with 2-8 groups.
struct GWItem { const void* x; const void* dy; float* dw; const float* ps; const float* psh; float* ws; int Cin, ldx, Cout, lddy, ncit, blk0; };
struct GroupedWgradArgs {
    int N, H, W, pro_relu, tiles_x, tiles_y, ntiles, groups, count, taps;
    GWItem item[SAUNET_WGRAD_GROUP_MAX];
};

template <typename T, int KS, int TR, int CO_T, int CI_T, int WM, int WN, int KSPLIT>
__global__ __launch_bounds__(256, 2) void conv_tile_wgrad_grouped_kernel(GroupedWgradArgs g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int p = 0;
    while (p + 1 < g.count && (int)blockIdx.x >= g.item[p + 1].blk0) ++p;
    const GWItem& it = g.item[p];
    TileWgradArgs a;
    a.x = it.x; a.dy = it.dy; a.dw = it.dw; a.pro_scale = it.ps; a.pro_shift = it.psh;
    a.N = g.N; a.H = g.H; a.W = g.W; a.Cin = it.Cin; a.ldx = it.ldx; a.Cout = it.Cout; a.lddy = it.lddy; a.pro_relu = g.pro_relu;
    a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y; a.ntiles = g.ntiles; a.ncit = it.ncit;
    a.sM = (long)it.Cin * g.taps; a.sN = g.taps;
    a.ws = it.ws; a.wsize = (long)it.Cout * it.Cin * g.taps; a.pend = nullptr;
    a.xs_n = a.xs_r = a.xs_p = 0; a.ncot = 0; a.Wo = 0;
    const int local = (int)blockIdx.x - it.blk0;
    const int by = local / g.groups, gx = local - by * g.groups;
    tile_wgrad_body<T, KS, TR, CO_T, CI_T, WM, WN, KSPLIT, true>(a, gx, g.groups, by, smem);
}

// dw[i] += sum_g ws[g][i]; blockIdx.y takes a slice of the groups (atomics only between slices)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, long wsize, int groups, int gper, float* __restrict__ dw)
{
    const int g0 = blockIdx.y * gper, g1 = min(g0 + gper, groups);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < wsize; i += (long)gridDim.x * 256) {
        float s = 0.f;
#pragma unroll 4
        for (int g = g0; g < g1; ++g) s += ws[(size_t)g * wsize + i];
        if (gridDim.y == 1) dw[i] += s; else atomicAdd(dw + i, s);
    }
}

static int tile_wgrad_groups(int ntiles, int nchan_tiles)
{
    int groups = 512 / nchan_tiles; if (groups < 1) groups = 1;
    if (groups > ntiles) groups = ntiles;
    // every group writes a full partial gradient: on small maps (few pixel tiles) let a block take at least two tiles as long as
    // 256 blocks remain
    if (groups > ntiles / 2 && (ntiles / 2) * nchan_tiles >= 256) groups = ntiles / 2;
    return groups;
}

template <typename T, int KS, int TR, int CO_T, int CI_T, int WM, int WN, int KSPLIT, bool ALIGNED = true> static int launch_tile_wgrad(TileWgradArgs& a, size_t ws_bytes, size_t* need, hipStream_t st)
{
    constexpr int PAD = KS / 2;
    constexpr int NPX = (TR + 2 * PAD) * (TILE + 2 * PAD), NPY = TR * TILE;
    constexpr int PY_RAW = CO_T * (int)sizeof(T), PX_RAW = CI_T * (int)sizeof(T);
    constexpr int PY = sizeof(T) == 2 ? ((PY_RAW % 128 == 64) ? PY_RAW : PY_RAW + 64) : PY_RAW;
    constexpr int PX = sizeof(T) == 2 ? ((PX_RAW % 128 == 64) ? PX_RAW : PX_RAW + 64) : PX_RAW;
    constexpr int LDS_MAIN = NPY * PY + NPX * PX;
    constexpr int LDS_RED = KSPLIT > 1 ? (4 / KSPLIT) * (KSPLIT - 1) * 16 * 64 * 4 : 0;   // per channel wave; (tap split: no cross-wave reduction)
    constexpr int LDS = LDS_MAIN > LDS_RED ? LDS_MAIN : LDS_RED;
    static_assert(LDS <= 160 * 1024, "tile does not fit LDS");
    auto kern = conv_tile_wgrad_kernel<T, KS, TR, CO_T, CI_T, WM, WN, KSPLIT, ALIGNED>;
    static DeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    a.tiles_y = a.H / TR; a.tiles_x = a.W / TILE; a.ntiles = a.N * a.tiles_y * a.tiles_x;
    constexpr int PARS = KS == 2 ? 4 : 1;                      // conv-transpose: the four output parities ride on blockIdx.y
    const int ncot = cdiv(a.Cout, CO_T); a.ncit = cdiv(a.Cin, CI_T); a.ncot = ncot;
    const int groups = tile_wgrad_groups(a.ntiles, ncot * a.ncit * PARS);
    a.wsize = (long)a.Cout * a.Cin * (KS == 2 ? 16 : KS * KS);
    const size_t bytes = (size_t)groups * a.wsize * sizeof(float);
    if (need) { *need = bytes; return SAUNET_OK; }
    if (a.ws == nullptr || ws_bytes < bytes) return set_error(SAUNET_BAD_SHAPE, "wgrad: workspace %zu < %zu bytes", ws_bytes, bytes);
    dim3 grid(groups, ncot * a.ncit * PARS);
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, st, a);
    static const KName kn("conv_tile_wgrad_kernel", type_name<T>(), KS, TR, CO_T, CI_T, WM, WN, KSPLIT, ALIGNED);
    SAUNET_CHECK_LAUNCH(kn.s);
    if (a.pend) {
        a.pend->ws = a.ws; a.pend->dw = a.dw; a.pend->wsize = a.wsize; a.pend->groups = groups; a.pend->taps = 0;
        return SAUNET_OK;
    }
    long rb = (a.wsize + 255) / 256; if (rb > 2048) rb = 2048;
    int gsl = 1;                       // group slices: enough blocks to fill the chip even for small weight tensors
    while (rb * gsl < 512 && gsl * 8 < groups) gsl *= 2;
    const int gper = (groups + gsl - 1) / gsl; gsl = (groups + gper - 1) / gper;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rb, gsl), dim3(256), 0, st, a.ws, a.wsize, groups, gper, a.dw);
    SAUNET_CHECK_LAUNCH("wgrad_reduce");
    return SAUNET_OK;
}

template <typename T> static int dispatch_tile_wgrad(TileWgradArgs& a, int ks, bool aligned, size_t wsb, size_t* need, hipStream_t st)
{
    if (!aligned) {   // pointwise layers with odd channel counts: scalar staging, 64x64 channel tile, K split over the waves
        if (ks != 1) return set_error(SAUNET_UNSUPPORTED, "unaligned tile wgrad is 1x1 only");
        return launch_tile_wgrad<T, 1, 16, 64, 64, 64, 64, 4, false>(a, wsb, need, st);
    }
    // few channel tiles -> 32x32 channel tile with the 4 waves splitting the K (pixel-row) dimension: the per-block
    // partial gradient (what has to be reduced across blocks afterwards) is 4x smaller
    const bool small = (long)a.Cout * a.Cin <= 64 * 128;
    if constexpr (sizeof(T) == 2) {
        if (ks == 3) {
            if (small) return launch_tile_wgrad<T, 3, 16, 32, 32, 32, 32, 4>(a, wsb, need, st);
            // measured and rejected (round 4): a 64 x 64 wave tile with the taps split 3/2/2/2 over the four waves (0.72 instead of 1.11 fragments per
            // MFMA): dec4 267 -> 343 us, dec3 270 -> 333, dec2 275 -> 358 -- 192 accumulator registers spill the staging registers (29 VGPRs) and
            // the wave holding three taps paces the other three
            return launch_tile_wgrad<T, 3, 8, 64, 64, 32, 32, 1>(a, wsb, need, st);
        }
        if (small) return launch_tile_wgrad<T, 1, 16, 64, 64, 64, 64, 4>(a, wsb, need, st);
        return launch_tile_wgrad<T, 1, 8, 128, 128, 64, 64, 1>(a, wsb, need, st);
    } else {
        if (ks == 3) {
            if (small) return launch_tile_wgrad<T, 3, 16, 32, 32, 32, 32, 4>(a, wsb, need, st);
            return launch_tile_wgrad<T, 3, 8, 64, 64, 32, 32, 1>(a, wsb, need, st);
        }
        return launch_tile_wgrad<T, 1, 8, 64, 64, 32, 32, 1>(a, wsb, need, st);
    }
}


// =====================================================================================================
// conv3x3_wgrad_mm_kernel: weight gradient of the 3x3 convolutions whose input needs no prologue and has Cin % 128 == 0, Cout % 64 == 0 (the
// decoder's c3x3rb, models/attention_blocks.py:215-220) -- the tiled kernel above spent 57 % of a pixel tile's time staging it (global ->
// registers -> LDS, synchronously; profiles/r03_phase_timing_raw.txt dec3wgrad) and half of the rest waiting for transposing fragment reads.
//   * 8 waves = 2 (output channels) x 4 (input channels) tiles of 32 x 32 channels x 9 taps; channel tile 64 x 128 per workgroup;
//   * pixel tile 8 rows x 16 columns; dy tile (16 KB) and x halo (10 x 18 pixels, 45 KB) arrive by LDS-DMA into one of TWO buffers: the next
//     tile's 61 requests are issued (two per wave behind each of the first four tile rows) while the matrix cores work on this one;
//     ONE barrier per tile, vmcnt(0) in front of it (nothing else is ever outstanding);
//   * the 64-byte segments of a pixel row are XOR-swizzled by the pixel's column (source address + fragment address), which puts the four pixel
//     rows of a transposing read on disjoint banks without padding (the DMA destination is lane-linear: no pad possible);
//   * the K loop is the rolling-row loop of tile_wgrad_body: one dy fragment + two x fragments read per tile row, the other seven x fragments
//     are the previous rows' and a one-pixel register shift.
struct WgradMmArgs {
    const u16* x; const u16* dy; float* ws;
    int N, H, W, Cin, ldx, Cout, lddy, tiles_x, tiles_y, ntiles, ncit;
    long wsize;
};
static __device__ u32x4 g_wg_zeros[4];

constexpr int WM_DYB = 128 * 128, WM_XPIX = 10 * 18, WM_XB = WM_XPIX * 256, WM_BUF = WM_DYB + WM_XB;      // 16384 + 46080 = 62464
constexpr int WM_PIECES = 16 + WM_XPIX / 4;                                                               // 61

__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_mm_kernel(WgradMmArgs a)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4, nhalf = lg & 1, khalf = lg >> 1;
    const int wm0 = (wave & 1) * 32, wn0 = (wave >> 1) * 32;
    const int gx = blockIdx.x, ngroups = gridDim.x;
    const int cot = blockIdx.y / a.ncit, cit = blockIdx.y - cot * a.ncit;
    const int co0 = cot * 64, ci0 = cit * 128;
    const unsigned char* zsrc = (const unsigned char*)g_wg_zeros;

    // ---- DMA slots of this wave: piece id = j * 8 + wave; 0..15 = dy (8 pixels x 128 B), 16..60 = x halo (4 pixels x 256 B), 61..63 = nothing
    int prel[8];            // byte offset of this lane's 16 bytes relative to the tile origin of its operand
    int phy[8], phx[8];     // x pieces: halo coordinates (for the border test)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int id = j * 8 + wave;
        if (id < 16) {
            const int pix = id * 8 + (lane >> 3), sl = lane & 7, row = pix >> 4, col = pix & 15;
            const int ch = sl ^ (((col >> 1) & 1) << 2);
            prel[j] = ((row * a.W + col) * a.lddy + co0 + ch * 8) * 2; phy[j] = 1; phx[j] = 1;
        } else {
            const int hp = (id - 16) * 4 + (lane >> 4), sl = lane & 15;
            const int hy = hp / 18, hx = hp - hy * 18;
            const int ch = sl ^ ((hx & 3) << 2);
            prel[j] = (((hy - 1) * a.W + (hx - 1)) * a.ldx + ci0 + ch * 8) * 2; phy[j] = hy; phx[j] = hx;
        }
    }
    // per tile: origin pointers and the border tests are wave-uniform / cheap; the decode of the tile index (two runtime divisions) is done ONCE
    struct TileOrg { const unsigned char* xb; const unsigned char* yb; int ty0, tx0; };
    auto origin = [&](int tile) {
        int bt = tile;
        const int txi = bt % a.tiles_x; bt /= a.tiles_x;
        const int tyi = bt % a.tiles_y; const int n = bt / a.tiles_y;
        TileOrg o;
        o.ty0 = tyi * 8; o.tx0 = txi * 16;
        o.yb = (const unsigned char*)a.dy + (((size_t)n * a.H + o.ty0) * a.W + o.tx0) * a.lddy * 2;
        o.xb = (const unsigned char*)a.x + (((long)n * a.H + o.ty0) * a.W + o.tx0) * a.ldx * 2;
        return o;
    };
    auto issue = [&](int j, const TileOrg& o, int buf) {
        const int id = j * 8 + wave;
        if (id >= WM_PIECES) return;                                    // wave-uniform
        const unsigned char* src;
        if (id < 16) src = o.yb + prel[j];
        else {
            const bool ok = (unsigned)(o.ty0 + phy[j] - 1) < (unsigned)a.H && (unsigned)(o.tx0 + phx[j] - 1) < (unsigned)a.W;
            src = ok ? o.xb + prel[j] : zsrc + (lane & 3) * 16;
        }
        mm_dma16(src, lds0 + buf * WM_BUF + id * 1024);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // fragment addresses inside a buffer (lo part; the hi part is 4 pixels further: same swizzle key)
    const int prow = 8 * khalf + (li >> 2);                              // pixel (column) of the lo part
    const int ab = (wm0 + 16 * nhalf + 4 * (li & 3)) * 2, bb = (wn0 + 16 * nhalf + 4 * (li & 3)) * 2;
    const int a_off = prow * 128 + (ab ^ ((((prow >> 1) & 1)) << 6));                                   // + ty * 16 * 128
    const int b0_off = WM_DYB + prow * 256 + (bb ^ ((prow & 3) << 6));                                  // + hy * 18 * 256   (kw = 0)
    const int b2_off = WM_DYB + (prow + 2) * 256 + (bb ^ (((prow + 2) & 3) << 6));                      // (kw = 2)

    TSTAMP_INIT();
    TSTAMP(80);
    int tile = gx;
    if (tile < a.ntiles) {
        const TileOrg o = origin(tile);
#pragma unroll
        for (int j = 0; j < 8; ++j) issue(j, o, 0);
    }
    // the two waves of a SIMD (w, w + 4) request in ANTI-PHASE: one behind tile rows 0, 2, 4, the other behind rows 1, 3, 5 -- a request blocks its
    // wave ~100 cycles and the CU accepts one per ~40, so eight waves requesting at once stall every matrix pipe for the whole burst
    const int phase = wave >> 2;
    int buf = 0;
    for (; tile < a.ntiles; tile += ngroups, buf ^= 1) {
        TSTAMP(81);
        mm_wait_vm<0>();
        TSTAMP(82);
        mm_barrier();                  // this tile has landed (every wave waited for its own requests); everybody is done with the other buffer
        const unsigned char* sb = smem + buf * WM_BUF;
        const int nxt = tile + ngroups;
        TileOrg on;
        if (nxt < a.ntiles) on = origin(nxt);
        auto load_row = [&](int hy, u32x4* f) {
            tr_read2(sb + b0_off + hy * (18 * 256), 4 * 256, f[0]);
            tr_read2(sb + b2_off + hy * (18 * 256), 4 * 256, f[2]);
            f[1][0] = __builtin_amdgcn_alignbit(f[0][1], f[0][0], 16); f[1][1] = __builtin_amdgcn_alignbit(f[0][2], f[0][1], 16);
            f[1][2] = __builtin_amdgcn_alignbit(f[0][3], f[0][2], 16); f[1][3] = __builtin_amdgcn_alignbit(f[2][3], f[0][3], 16);
        };
        TSTAMP(83);
        u32x4 rows[3][3];
        load_row(0, rows[0]); load_row(1, rows[1]);
#pragma unroll
        for (int ty = 0; ty < 8; ++ty) {
            if (ty == 4) TSTAMP(84);
            load_row(ty + 2, rows[(ty + 2) % 3]);
            u32x4 af;
            tr_read2(sb + a_off + ty * (16 * 128), 4 * 128, af);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, rows[(ty + kh) % 3][kw]),
                                                                             acc[kh * 3 + kw], 0, 0, 0);
            if (ty < 6 && (ty & 1) == phase && nxt < a.ntiles) {
                constexpr int first[3] = {0, 3, 6};
                const int k = ty >> 1;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (first[k] + j < 8) issue(first[k] + j, on, buf ^ 1);
            }
        }
    }
    TSTAMP(85);
    // ---- partial gradient of this pixel group: ws[gx][tap][co][ci]
    const int lr = lane & 31, lh = lane >> 5;
    const int ci = ci0 + wn0 + lr;
    float* wsg = a.ws + (size_t)gx * a.wsize;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            wsg[((size_t)t * a.Cout + co) * a.Cin + ci] = acc[t][r];       // [tap][co][ci]: a wave row = 128 contiguous bytes (the reduce kernel permutes)
        }
    TSTAMP(86);
}

int wgrad_reduce_multi(const saunet_wgrad_reduce_list* l, hipStream_t st);
// dw[co][ci][tap] += sum_g ws[g][tap][co][ci] of ONE problem: a one-entry list for wgrad_reduce_multi_kernel (LDS-transposed, contiguous writes)
static int wgrad_reduce_tco_one(const float* ws, float* dw, long wsize, int groups, int taps, hipStream_t st)
{
    saunet_wgrad_reduce_list l; l.count = 1; l.reserved = 0;
    l.item[0].ws = ws; l.item[0].dw = dw; l.item[0].wsize = wsize; l.item[0].groups = groups; l.item[0].taps = taps;
    return wgrad_reduce_multi(&l, st);
}

bool wgrad_mm_supported(const TileWgradArgs& a, int ks, int dtype, bool aligned)
{
    static const bool on = ab_env_on("SAUNET_WGRAD_MM");                // A/B switch (variant builds only)
    return on && aligned && dtype == SAUNET_BF16 && ks == 3 && a.pro_scale == nullptr && a.Cout % 64 == 0 && a.Cin % 128 == 0 && a.H % 8 == 0 && a.W % 16 == 0 &&
           a.ldx % 8 == 0 && a.lddy % 8 == 0 && (long)a.N * a.H * a.W * (a.ldx > a.lddy ? a.ldx : a.lddy) < (1L << 30);
}

int launch_wgrad_mm(TileWgradArgs& t, size_t ws_bytes, size_t* need, hipStream_t st)
{
    WgradMmArgs a;
    a.x = (const u16*)t.x; a.dy = (const u16*)t.dy; a.ws = t.ws;
    a.N = t.N; a.H = t.H; a.W = t.W; a.Cin = t.Cin; a.ldx = t.ldx; a.Cout = t.Cout; a.lddy = t.lddy;
    a.tiles_y = t.H / 8; a.tiles_x = t.W / 16; a.ntiles = t.N * a.tiles_y * a.tiles_x; a.ncit = t.Cin / 128;
    const int chan_tiles = (t.Cout / 64) * a.ncit;
    int groups = 256 / chan_tiles; if (groups < 1) groups = 1;          // one 8-wave workgroup per CU
    if (groups > a.ntiles) groups = a.ntiles;
    a.wsize = (long)t.Cout * t.Cin * 9;
    const size_t bytes = (size_t)groups * a.wsize * sizeof(float);
    if (need) { *need = bytes; return SAUNET_OK; }
    if (a.ws == nullptr || ws_bytes < bytes) return set_error(SAUNET_BAD_SHAPE, "wgrad: workspace %zu < %zu bytes", ws_bytes, bytes);
    constexpr int LDS = 2 * WM_BUF;
    static DeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_mm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL(conv3x3_wgrad_mm_kernel, dim3(groups, chan_tiles), dim3(512), LDS, st, a);
    SAUNET_CHECK_LAUNCH("conv3x3_wgrad_mm");
    if (t.pend) {        // deferred: the permuting reduction ([tap][co][ci] partials -> parameter layout) runs in saunet_wgrad_reduce_multi
        t.pend->ws = t.ws; t.pend->dw = t.dw; t.pend->wsize = a.wsize; t.pend->groups = groups; t.pend->taps = 9;
        return SAUNET_OK;
    }
    return wgrad_reduce_tco_one(t.ws, t.dw, a.wsize, groups, 9, st);
}


// =====================================================================================================
// convt_wgrad_mm_kernel: weight gradient of ConvTranspose2d(k = 4, s = 2, p = 1) (the decoder's mrf.up, models/attention_blocks.py:179-186) with the
// machinery of conv3x3_wgrad_mm_kernel.  Per output parity (py, px) it is a 2 x 2-tap problem between x (un-haloed operand, K = its pixels) and the
// parity sub-image D[i][j] = dy[2 i + 1 - kh0][2 j + 1 - kw0] (kh0 = 1 - py, kw0 = 1 - px; see tile_wgrad_body KS == 2):
//     dW[ci][co][kh0 + 2 th][kw0 + 2 tw] = sum x[n, i, j, ci] * D[n, i + th + kh0 - 1, j + tw + kw0 - 1, co]
//   * channel tile 128 (Cin) x 128 (Cout), 8 waves = 4 x 2 tiles of 32 x 64 channels x 4 taps (8 accumulators);
//   * pixel tile 8 x 16 of x; LDS per buffer: x tile 32 KB + 9 x 18 pixels of D (read IN PLACE from dy with stride-2 pixel addresses) 41 KB, two
//     buffers; one D row serves two consecutive tile rows (th = 1, 0), the tw = 1 fragment is the register shift of the tw = 0 one;
//   * blockIdx.y = parity x channel tile; partial gradients [tap][ci][co] per pixel group, permuted by wgrad_reduce_multi_kernel.
struct ConvtWgradMmArgs {
    const u16* x; const u16* dy; float* ws;
    int N, H, W, Cin, ldx, Cout, lddy, tiles_x, tiles_y, ntiles, nbt, nct;
    long wsize;
};
constexpr int CW_AB = 128 * 256, CW_BPIX = 9 * 18, CW_BPIECES = (CW_BPIX + 3) / 4, CW_BB = CW_BPIECES * 1024, CW_BUF = CW_AB + CW_BB;      // 32768 + 41984
constexpr int CW_PIECES = 32 + CW_BPIECES;                                                                                                 // 73

__global__ __launch_bounds__(512, 2) void convt_wgrad_mm_kernel(ConvtWgradMmArgs a)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4, nhalf = lg & 1, khalf = lg >> 1;
    const int wa0 = (wave & 3) * 32, wb0 = (wave >> 2) * 64;
    const int gx = blockIdx.x, ngroups = gridDim.x;
    const int par = blockIdx.y / a.nct, ct = blockIdx.y - par * a.nct;
    const int kh0 = 1 - (par >> 1), kw0 = 1 - (par & 1);
    const int at = ct / a.nbt, bt = ct - at * a.nbt;
    const int a0 = at * 128, b0 = bt * 128;
    const int Wo = 2 * a.W;
    const unsigned char* zsrc = (const unsigned char*)g_wg_zeros;

    int prel[10], pr[10], pc[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const int id = j * 8 + wave;
        const int sl = lane & 15;
        if (id < 32) {
            const int pix = id * 4 + (lane >> 4), row = pix >> 4, col = pix & 15;
            prel[j] = ((row * a.W + col) * a.ldx + a0 + (sl ^ ((col & 3) << 2)) * 8) * 2; pr[j] = 0; pc[j] = 1;
        } else {
            const int hp = (id - 32) * 4 + (lane >> 4);
            const int r = hp / 18, c = hp - r * 18;
            prel[j] = (((2 * r + kh0 - 1) * Wo + 2 * c - 1 - kw0) * a.lddy + b0 + (sl ^ ((c & 3) << 2)) * 8) * 2;
            pr[j] = hp < CW_BPIX ? r + kh0 - 1 : -100000; pc[j] = c - 1;
        }
    }
    struct TileOrg { const unsigned char* xb; const unsigned char* yb; int ty0, tx0; };
    auto origin = [&](int tile) {
        int q = tile;
        const int txi = q % a.tiles_x; q /= a.tiles_x;
        const int tyi = q % a.tiles_y; const int n = q / a.tiles_y;
        TileOrg o;
        o.ty0 = tyi * 8; o.tx0 = txi * 16;
        o.xb = (const unsigned char*)a.x + (((size_t)n * a.H + o.ty0) * a.W + o.tx0) * a.ldx * 2;
        o.yb = (const unsigned char*)a.dy + (((long)n * 2 * a.H + 2 * o.ty0) * Wo + 2 * o.tx0) * a.lddy * 2;
        return o;
    };
    auto issue = [&](int j, const TileOrg& o, int buf) {
        const int id = j * 8 + wave;
        if (id >= CW_PIECES) return;                                    // wave-uniform
        const unsigned char* src;
        if (id < 32) src = o.xb + prel[j];
        else {
            const bool ok = (unsigned)(o.ty0 + pr[j]) < (unsigned)a.H && (unsigned)(o.tx0 + pc[j]) < (unsigned)a.W;
            src = ok ? o.yb + prel[j] : zsrc + (lane & 3) * 16;
        }
        mm_dma16(src, lds0 + buf * CW_BUF + id * 1024);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

    const int prow = 8 * khalf + (li >> 2);
    const int ab = (wa0 + 16 * nhalf + 4 * (li & 3)) * 2, bb = (wb0 + 16 * nhalf + 4 * (li & 3)) * 2;
    const int a_off = prow * 256 + (ab ^ ((prow & 3) << 6));                                            // + ty * 16 * 256
    const int ca = prow + kw0, cc = ca + 2;                                                             // D columns of the tw = 0 fragment and of the one two pixels on
    const int ba_off = CW_AB + ca * 256, bc_off = CW_AB + cc * 256;                                     // + r' * 18 * 256 + swizzled channel offset
    const int ka = (ca & 3) << 6, kc = (cc & 3) << 6;

    int tile = gx;
    if (tile < a.ntiles) {
        const TileOrg o = origin(tile);
#pragma unroll
        for (int j = 0; j < 10; ++j) issue(j, o, 0);
    }
    const int phase = wave >> 2;
    int buf = 0;
    for (; tile < a.ntiles; tile += ngroups, buf ^= 1) {
        mm_wait_vm<0>();
        mm_barrier();
        const unsigned char* sb = smem + buf * CW_BUF;
        const int nxt = tile + ngroups;
        TileOrg on;
        if (nxt < a.ntiles) on = origin(nxt);
        auto load_row = [&](int r, u32x4 (*f)[3]) {          // f[j][0] = tw 0, f[j][1] = tw 1 (derived), f[j][2] = helper (two pixels on)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int chb = bb + j * 64;
                tr_read2(sb + ba_off + r * (18 * 256) + (chb ^ ka), 4 * 256, f[j][0]);
                tr_read2(sb + bc_off + r * (18 * 256) + (chb ^ kc), 4 * 256, f[j][2]);
                f[j][1][0] = __builtin_amdgcn_alignbit(f[j][0][1], f[j][0][0], 16); f[j][1][1] = __builtin_amdgcn_alignbit(f[j][0][2], f[j][0][1], 16);
                f[j][1][2] = __builtin_amdgcn_alignbit(f[j][0][3], f[j][0][2], 16); f[j][1][3] = __builtin_amdgcn_alignbit(f[j][2][3], f[j][0][3], 16);
            }
        };
        u32x4 rows[2][2][3];
        load_row(0, rows[0]);
#pragma unroll
        for (int ty = 0; ty < 8; ++ty) {
            load_row(ty + 1, rows[(ty + 1) & 1]);
            u32x4 af;
            tr_read2(sb + a_off + ty * (16 * 256), 4 * 256, af);
#pragma unroll
            for (int th = 0; th < 2; ++th)
#pragma unroll
                for (int tw = 0; tw < 2; ++tw)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[th * 2 + tw][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, rows[(ty + th) & 1][j][tw]),
                                                                                    acc[th * 2 + tw][j], 0, 0, 0);
            if (ty < 6 && (ty & 1) == phase && nxt < a.ntiles) {
                constexpr int first[3] = {0, 4, 7};
                const int k = ty >> 1;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (first[k] + j < (k == 2 ? 10 : first[k + 1 > 2 ? 2 : k + 1])) issue(first[k] + j, on, buf ^ 1);
            }
        }
    }
    // ---- partial gradient: ws[gx][tap = kh * 4 + kw][ci][co]
    const int lr = lane & 31, lh = lane >> 5;
    float* wsg = a.ws + (size_t)gx * a.wsize;
#pragma unroll
    for (int th = 0; th < 2; ++th)
#pragma unroll
        for (int tw = 0; tw < 2; ++tw) {
            const int toff = (kh0 + 2 * th) * 4 + kw0 + 2 * tw;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = b0 + wb0 + j * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = a0 + wa0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    wsg[((size_t)toff * a.Cin + ci) * a.Cout + co] = acc[th * 2 + tw][j][r];
                }
            }
        }
}

bool convt_wgrad_mm_supported(const saunet_conv_desc* d)
{
    static const bool on = ab_env_on("SAUNET_WGRAD_MM");                // A/B switch (variant builds only)
    return on && d->Cin % 128 == 0 && d->Cout % 128 == 0 && d->H % 8 == 0 && d->W % 16 == 0 &&
           (long)d->N * d->Ho * d->Wo * d->ldy < (1L << 30) && (long)d->N * d->H * d->W * d->ldx < (1L << 30);
}

int launch_convt_wgrad_mm(const saunet_conv_desc* d, const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, size_t* need, hipStream_t st,
                          saunet_wgrad_pending* pend = nullptr)
{
    ConvtWgradMmArgs a;
    a.x = (const u16*)x; a.dy = (const u16*)dy; a.ws = (float*)ws;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx; a.Cout = d->Cout; a.lddy = d->ldy;
    a.tiles_y = d->H / 8; a.tiles_x = d->W / 16; a.ntiles = d->N * a.tiles_y * a.tiles_x;
    a.nbt = d->Cout / 128; a.nct = (d->Cin / 128) * a.nbt;
    int groups = 256 / (a.nct * 4); if (groups < 1) groups = 1;
    if (groups > a.ntiles) groups = a.ntiles;
    a.wsize = (long)d->Cin * d->Cout * 16;
    const size_t bytes = (size_t)groups * a.wsize * sizeof(float);
    if (need) { *need = bytes; return SAUNET_OK; }
    if (ws == nullptr || ws_bytes < bytes) return set_error(SAUNET_BAD_SHAPE, "conv-transpose wgrad: workspace %zu < %zu bytes", ws_bytes, bytes);
    constexpr int LDS = 2 * CW_BUF;
    static DeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute((const void*)convt_wgrad_mm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL(convt_wgrad_mm_kernel, dim3(groups, a.nct * 4), dim3(512), LDS, st, a);
    SAUNET_CHECK_LAUNCH("convt_wgrad_mm");
    if (pend) { pend->ws = (const float*)ws; pend->dw = dw; pend->wsize = a.wsize; pend->groups = groups; pend->taps = 16; return SAUNET_OK; }
    return wgrad_reduce_tco_one((const float*)ws, dw, a.wsize, groups, 16, st);
}

bool tile_wgrad_unaligned_supported(const saunet_conv_desc* d)
{
    return !d->transposed && d->KH == 1 && d->KW == 1 && d->pad == 0 && d->stride == 1 && d->H % TILE == 0 && d->W % TILE == 0 &&
           d->Ho == d->H && d->Wo == d->W;
}

bool tile_wgrad_supported(const saunet_conv_desc* d)
{
    const bool k3 = d->KH == 3 && d->KW == 3 && d->pad == 1, k1 = d->KH == 1 && d->KW == 1 && d->pad == 0;
    return !d->transposed && (k3 || k1) && d->stride == 1 && d->H % TILE == 0 && d->W % TILE == 0 && d->Ho == d->H && d->Wo == d->W;
}

// all pending reductions of a list in ONE launch: blockIdx.y = entry, dw[i] += sum over the groups of ws[g][i] (no atomics, fixed order:
// deterministic).  Entries with many groups and few weights (res3: 512 partial copies of 2304 values) would be a chain of dependent load batches
// on a handful of threads, so a block deals its 256 threads as (lanes x group slices): 16 slices from 128 groups, 4 from 32; slice sums are
// folded through the LDS in slice order.
// taps > 0: the partials are [tap][co * ci] (LDS-DMA kernels) and the parameter is [co * ci][tap].  Writing dw[r * taps + t] straight from the
// thread that summed (t, r) is a 4-byte read-modify-write at a 36 / 64-byte stride whose neighbours are handled by far-away blocks: a whole
// 64-byte sector in and out of HBM per value (rocprofv3, round 6: 15.7 M such values per step = 2 GB of traffic, 300 us).  So a block takes a
// run of `lanes` channel pairs through ALL taps, transposes through the LDS and writes lanes * taps consecutive floats.
static __host__ __device__ inline int wgrad_reduce_slices(int groups) { return groups >= 128 ? 16 : groups >= 32 ? 4 : 1; }

template <int TAPS> static __device__ __forceinline__ void wgrad_reduce_tco_entry(const saunet_wgrad_pending& e, float* s_part, float* s_out)
{
    constexpr int PITCH = TAPS | 1;
    const float* __restrict__ ws = e.ws;
    float* __restrict__ dw = e.dw;
    const long wsize = e.wsize, cc = wsize / TAPS;
    const int groups = e.groups;
    const int slices = wgrad_reduce_slices(groups), lanes = 256 / slices;
    const int tid = (int)threadIdx.x, sl = tid / lanes, ln = tid - sl * lanes;
    const int gper = (groups + slices - 1) / slices, g0 = sl * gper, g1 = min(g0 + gper, groups);
    for (long r0 = (long)blockIdx.x * lanes; r0 < cc; r0 += (long)gridDim.x * lanes) {       // (block-uniform trip count)
        const long r = r0 + ln;
        if (slices == 1) {
            float s[TAPS];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) s[t] = 0.f;
            if (r < cc)
                for (int g = 0; g < groups; ++g) {
                    const float* src = ws + (size_t)g * wsize + r;
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) s[t] += src[(size_t)t * cc];          // TAPS independent loads in flight
                }
#pragma unroll
            for (int t = 0; t < TAPS; ++t) s_out[ln * PITCH + t] = s[t];
        } else {
            for (int t = 0; t < TAPS; ++t) {
                float s = 0.f;
                if (r < cc) {
                    const float* src = ws + (size_t)t * cc + r;
                    int g = g0;
                    for (; g + 8 <= g1; g += 8) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(g + u) * wsize];
#pragma unroll
                        for (int u = 0; u < 8; ++u) s += v[u];
                    }
                    for (; g < g1; ++g) s += src[(size_t)g * wsize];
                }
                s_part[(t & 1) * 256 + tid] = s;
                __syncthreads();                                                           // (two slot sets: one barrier per tap)
                if (sl == 0) {
                    for (int q = 1; q < slices; ++q) s += s_part[(t & 1) * 256 + q * lanes + ln];
                    s_out[ln * PITCH + t] = s;
                }
            }
        }
        __syncthreads();
        const long rem = cc - r0;
        const int nv = (int)(rem < lanes ? rem : lanes) * TAPS;
        float* dst = dw + r0 * TAPS;
        for (int j = tid; j < nv; j += 256) { const int q = j / TAPS; dst[j] += s_out[q * PITCH + (j - q * TAPS)]; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(saunet_wgrad_reduce_list l)
{
    __shared__ float s_part[512];
    __shared__ float s_out[256 * 17];
    const saunet_wgrad_pending& e = l.item[blockIdx.y];
    if (e.taps == 9) { wgrad_reduce_tco_entry<9>(e, s_part, s_out); return; }
    if (e.taps == 16) { wgrad_reduce_tco_entry<16>(e, s_part, s_out); return; }
    const float* __restrict__ ws = e.ws;
    float* __restrict__ dw = e.dw;
    const long wsize = e.wsize;
    const int groups = e.groups;
    const int slices = wgrad_reduce_slices(groups), lanes = 256 / slices;
    const int sl = (int)threadIdx.x / lanes, ln = (int)threadIdx.x - sl * lanes;
    const int gper = (groups + slices - 1) / slices, g0 = sl * gper, g1 = min(g0 + gper, groups);
    for (long base = (long)blockIdx.x * lanes; base < wsize; base += (long)gridDim.x * lanes) {       // (block-uniform trip count)
        const long i = base + ln;
        float s = 0.f;
        if (i < wsize) {
            int g = g0;
            for (; g + 8 <= g1; g += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(g + u) * wsize + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; g < g1; ++g) s += ws[(size_t)g * wsize + i];
        }
        if (slices > 1) {
            s_part[threadIdx.x] = s;
            __syncthreads();
            if (sl == 0) for (int q = 1; q < slices; ++q) s += s_part[q * lanes + ln];
            __syncthreads();
        }
        if (sl == 0 && i < wsize) dw[i] += s;
    }
}

int wgrad_reduce_multi(const saunet_wgrad_reduce_list* l, hipStream_t st)
{
    long bx = 1;
    for (int e = 0; e < l->count; ++e) {
        const saunet_wgrad_pending& it = l->item[e];
        const long lanes = 256 / wgrad_reduce_slices(it.groups);
        const long nb = it.taps > 0 ? (it.wsize / it.taps + lanes - 1) / lanes             // one run of `lanes` channel pairs (all taps) per block
                                    : (it.wsize + 4 * lanes - 1) / (4 * lanes);             // about four chunks per block
        if (nb > bx) bx = nb;
    }
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)bx, l->count), dim3(256), 0, st, *l);
    SAUNET_CHECK_LAUNCH("wgrad_reduce_multi");
    return SAUNET_OK;
}

template <typename T, int KS, int TR, int CO_T, int CI_T, int WM, int WN, int KSPLIT>
static int launch_tile_wgrad_grouped(GroupedWgradArgs& g, const saunet_wgrad_group* src, void* ws, size_t ws_bytes, size_t* need, hipStream_t st)
{
    constexpr int PAD = KS / 2;
    constexpr int NPX = (TR + 2 * PAD) * (TILE + 2 * PAD), NPY = TR * TILE;
    constexpr int PY_RAW = CO_T * (int)sizeof(T), PX_RAW = CI_T * (int)sizeof(T);
    constexpr int PY = sizeof(T) == 2 ? ((PY_RAW % 128 == 64) ? PY_RAW : PY_RAW + 64) : PY_RAW;
    constexpr int PX = sizeof(T) == 2 ? ((PX_RAW % 128 == 64) ? PX_RAW : PX_RAW + 64) : PX_RAW;
    constexpr int LDS_MAIN = NPY * PY + NPX * PX;
    constexpr int LDS_RED = KSPLIT > 1 ? (4 / KSPLIT) * (KSPLIT - 1) * 16 * 64 * 4 : 0;   // per channel wave; (tap split: no cross-wave reduction)
    constexpr int LDS = LDS_MAIN > LDS_RED ? LDS_MAIN : LDS_RED;
    auto kern = conv_tile_wgrad_grouped_kernel<T, KS, TR, CO_T, CI_T, WM, WN, KSPLIT>;
    static DeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    g.tiles_y = g.H / TR; g.tiles_x = g.W / TILE; g.ntiles = g.N * g.tiles_y * g.tiles_x;
    long chan_tiles = 0, welems = 0;
    for (int i = 0; i < g.count; ++i) {
        g.item[i].ncit = cdiv(g.item[i].Cin, CI_T);
        chan_tiles += (long)cdiv(g.item[i].Cout, CO_T) * g.item[i].ncit;
        welems += (long)g.item[i].Cout * g.item[i].Cin * g.taps;
    }
    // pixel groups: ONE co-resident wave of workgroups (a grid of 1.1x the resident capacity runs as long as one of 2x), each with an equal
    // share of the pixel tiles
    static int capacity = 0;
    if (capacity == 0) {
        int per_cu = 0, dev = 0; hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 256, LDS) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        capacity = per_cu * prop.multiProcessorCount;
        (void)hipGetLastError();
    }
    const long target = capacity;
    long groups_l = target / chan_tiles;
    if (groups_l > g.ntiles) groups_l = g.ntiles;
    if (groups_l < 1) groups_l = 1;
    const int groups = (int)groups_l;
    g.groups = groups;
    const size_t bytes = groups > 1 ? (size_t)groups * welems * sizeof(float) : 0;
    if (need) { *need = bytes; return SAUNET_OK; }
    if (bytes && (ws == nullptr || ws_bytes < bytes)) return set_error(SAUNET_BAD_SHAPE, "grouped wgrad: workspace %zu < %zu bytes", ws_bytes, bytes);
    long blk = 0; float* wsp = (float*)ws;
    for (int i = 0; i < g.count; ++i) {
        g.item[i].blk0 = (int)blk;
        blk += (long)cdiv(g.item[i].Cout, CO_T) * g.item[i].ncit * groups;
        const long wsize = (long)g.item[i].Cout * g.item[i].Cin * g.taps;
        if (groups > 1) { g.item[i].ws = wsp; wsp += (size_t)groups * wsize; }
        else g.item[i].ws = g.item[i].dw;            // one group: the block owns the whole problem and stores the gradient itself
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blk), dim3(256), LDS, st, g);
    static const KName kn("conv_tile_wgrad_grouped_kernel", type_name<T>(), KS, TR, CO_T, CI_T, WM, WN, KSPLIT);
    SAUNET_CHECK_LAUNCH(kn.s);
    if (groups > 1) {
        for (int i0 = 0; i0 < g.count; i0 += SAUNET_WGRAD_REDUCE_MAX) {
            saunet_wgrad_reduce_list l; l.reserved = 0;
            l.count = g.count - i0 < SAUNET_WGRAD_REDUCE_MAX ? g.count - i0 : SAUNET_WGRAD_REDUCE_MAX;
            for (int i = 0; i < l.count; ++i) {
                const GWItem& it = g.item[i0 + i];
                l.item[i].ws = it.ws; l.item[i].dw = it.dw; l.item[i].wsize = (long)it.Cout * it.Cin * g.taps; l.item[i].groups = groups; l.item[i].taps = 0;
            }
            if (int rc = wgrad_reduce_multi(&l, st)) return rc;
        }
    }
    return SAUNET_OK;
}

// =====================================================================================================
// conv3x3_wgrad_sc_kernel (round 6): the weight gradients of ALL DenseNet conv2 layers of a block (3x3, 128 -> 32, BatchNorm + ReLU prologue on
// z1; torchvision _DenseLayer.conv2 as sliced at /root/reference/models/models.py:306-313) on the LDS-DMA staged, rolling-row design of
// conv3x3_wgrad_mm_kernel -- VERDICT r5 item 2 (i).  The tiled kernel it replaces staged every pixel tile synchronously through registers
// (global -> VGPR -> prologue -> LDS, two barriers per tile), read the dy tile once per 64 input channels and walked the tiles with a stride
// of the group count (neighbouring halos on different XCDs): 2.0 TB/s of algorithmic bytes over the four blocks.
//   * one 8-wave workgroup per CU = (problem, pixel group); channel tile = the whole layer (32 x 128): 4 input-channel waves x 2 K halves
//     (tile rows 0-3 / 4-7), each wave 9 taps x one 32 x 32 accumulator; the K halves are summed through the (idle) LDS at the end;
//   * pixel tile 8 rows x 16 columns; dy tile (8 KB) + x halo (10 rows x 20 pixel slots x 256 B: 18 real columns, the pitch of 20 makes the
//     swizzle key  hx & 3  a function of the LANE, so a lane always handles the same 8 channels) arrive by LDS-DMA into one of two buffers
//     while the matrix cores work on the other; one barrier per tile.  (Measured and rejected: 4-row tiles in a ring of four buffers with the
//     activation between the MFMAs of the tile before -- no DMA wait left, but 2.1k cycles per 18 MFMAs of fragment-read latency and a third
//     more halo: 430 us at block 1 against 384);
//   * the BatchNorm + ReLU prologue is applied IN PLACE in the LDS by the wave that requested the piece (its 16 coefficients live in registers;
//     pieces sourced from the zero page -- padding of the ACTIVATED tensor -- are left alone);
//   * a group owns a CONTIGUOUS run of tiles in column-strip order (tile row fastest), so the two halo rows it shares with the tile it has
//     just finished are L2 hits on its own XCD;
//   * partial gradients [group][tap][co][ci] (128-byte runs), permuted into the parameter layout by wgrad_reduce_multi_kernel.
struct ScItem { const u16* x; const u16* dy; float* ws; float* dw; const float* ps; const float* psh; int ldx, lddy, blk0, pad_; };
struct ScArgs { int N, H, W, tiles_x, tiles_y, ntiles, groups, count, pro_relu, pad_; ScItem item[SAUNET_WGRAD_GROUP_MAX]; };
static __device__ u32x4 g_sc_zeros[4];

constexpr int SC_HPW = 20;                                                   // halo pixel slots per row (18 used)
constexpr int SC_TR = 8;                                                     // tile rows
constexpr int SC_DYB = SC_TR * 16 * 64, SC_XB = (SC_TR + 2) * SC_HPW * 256, SC_BUF = SC_DYB + SC_XB;      // 8192 + 51200 = 59392
constexpr int SC_DY_PIECES = SC_DYB / 1024, SC_PIECES = SC_DY_PIECES + SC_XB / 1024, SC_SLOTS = (SC_PIECES + 7) / 8;     // 8 + 50 = 58 pieces, 8 slots per wave
constexpr int SC_NBUF = 2;
constexpr int SC_OFF_DUMMY = SC_NBUF * SC_BUF, SC_OFF_PRO = SC_OFF_DUMMY + 1024, SC_LDS = SC_OFF_PRO + 1024;

template <bool PRO>
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_sc_kernel(ScArgs a)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4, nhalf = lg & 1, khalf = lg >> 1;
    const int wn0 = (wave & 3) * 32, kh2 = wave >> 2;                         // input-channel tile of the wave; its K half (tile rows 2 kh2, 2 kh2 + 1)
    int pi = 0;
    while (pi + 1 < a.count && (int)blockIdx.x >= a.item[pi + 1].blk0) ++pi;
    const ScItem& it = a.item[pi];
    const int gx = (int)blockIdx.x - it.blk0;
    const int t0 = (int)((long)gx * a.ntiles / a.groups), t1 = (int)((long)(gx + 1) * a.ntiles / a.groups);
    const unsigned char* zsrc = (const unsigned char*)g_sc_zeros;

    // ---- DMA slots of this wave: piece id = j * 8 + wave; 0..3 = dy (16 pixels x 64 B), 4..33 = x halo (4 pixel slots x 256 B), 34..39 = dummies
    // (zero page -> scratch KB: every wave issues the same number of requests per tile, which is what the counted waits rely on)
    int prel[SC_SLOTS], phyx[SC_SLOTS];      // phyx = halo row << 16 | halo column (one register: the kernel sits at the 256-register limit)
#pragma unroll
    for (int j = 0; j < SC_SLOTS; ++j) {
        const int id = j * 8 + wave;
        if (id < SC_DY_PIECES) {
            const int pix = id * 16 + (lane >> 2), sl = lane & 3, row = pix >> 4, col = pix & 15;
            prel[j] = ((row * a.W + col) * it.lddy + sl * 8) * 2; phyx[j] = (1 << 16) | 1;
        } else {
            const int hp = (id - SC_DY_PIECES) * 4 + (lane >> 4), sl = lane & 15;
            const int hy = hp / SC_HPW, hx = hp - hy * SC_HPW;
            const int ch = sl ^ ((hx & 3) << 2);                              // (hx & 3 == lane >> 4: the lane's channel chunk never changes)
            prel[j] = (((hy - 1) * a.W + (hx - 1)) * it.ldx + ch * 8) * 2; phyx[j] = (hy << 16) | ((hx < 18 && id < SC_PIECES) ? hx : 0x7fff);     // slots 18, 19 and dummies: never inside
        }
    }
    // prologue coefficients of the lane's eight channels
    // (they reach the registers THROUGH THE LDS: the compiler counts its own global loads in vmcnt and, at the first use inside the tile loop,
    // waits vmcnt(0) for them on every iteration -- which also drains every LDS-DMA request of the ring, invisible to it: 5000 cycles per tile)
    float psc[8], psh[8];
    if constexpr (PRO) {
        float* s_pro = (float*)(smem + SC_OFF_PRO);
        if (tid < 128) { s_pro[tid] = it.ps[tid]; s_pro[128 + tid] = it.psh[tid]; }
        __syncthreads();
        const int c0 = ((lane & 15) ^ ((lane >> 4) << 2)) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { psc[e] = s_pro[c0 + e]; psh[e] = s_pro[128 + c0 + e]; }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // nothing of the compiler's own is in flight when the ring starts
    }
    const bool relu_on = a.pro_relu != 0;
    // tile cursors (column strips: tile row fastest).  Two of them walk the group's run: the REQUEST cursor runs one tile in front of the
    // MULTIPLY cursor; stepping is incremental (the two divisions of a from-scratch decode cost ~200 cycles per use)
    struct Cur { int tyi, txi, n; };
    auto decode = [&](int tile) { Cur c; c.tyi = tile % a.tiles_y; const int r = tile / a.tiles_y; c.txi = r % a.tiles_x; c.n = r / a.tiles_x; return c; };
    auto step = [&](Cur& c) { if (++c.tyi == a.tiles_y) { c.tyi = 0; if (++c.txi == a.tiles_x) { c.txi = 0; ++c.n; } } };
    auto inside = [&](int j, const Cur& c) { return (unsigned)(c.tyi * SC_TR + (phyx[j] >> 16) - 1) < (unsigned)a.H && (unsigned)(c.txi * 16 + (phyx[j] & 0xffff) - 1) < (unsigned)a.W; };
    auto issue_slot = [&](int j, const Cur& c, int buf) {
        const int id = j * 8 + wave;
        const size_t pix0 = ((size_t)c.n * a.H + c.tyi * SC_TR) * a.W + c.txi * 16;       // tile origin (block-uniform)
        const unsigned char* src;
        if (id < SC_DY_PIECES) src = (const unsigned char*)it.dy + pix0 * it.lddy * 2 + prel[j];
        else src = inside(j, c) ? (const unsigned char*)it.x + (long)pix0 * it.ldx * 2 + prel[j] : zsrc + (lane & 3) * 16;
        mm_dma16(src, id < SC_PIECES ? lds0 + buf * SC_BUF + id * 1024 : lds0 + SC_OFF_DUMMY);
    };
    // BatchNorm + ReLU on this wave's own x pieces of a landed tile, in three steps so that the VALU work can sit BETWEEN the MFMAs of the tile
    // before it (both waves of a SIMD transforming at once behind the barrier cost 1500 cycles per tile: 2 x 160 VALU instructions x 4 cycles):
    // fetch (LDS -> registers), activate (a lane outside the image keeps its zeros by a select, not by a branch), put (registers -> LDS)
    auto tf_fetch = [&](u32x4* v, int buf, int j0) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = j0 + jj;
            const int id = j * 8 + wave;
            if (id < SC_DY_PIECES || id >= SC_PIECES) continue;             // wave-uniform
            v[jj] = *(const u32x4*)(smem + buf * SC_BUF + id * 1024 + lane * 16);
        }
    };
    auto tf_activate = [&](u32x4& v, int j, const Cur& c) {
        const int id = j * 8 + wave;
        if (id < SC_DY_PIECES || id >= SC_PIECES) return;
        // per channel pair: unpack (2), v_pk_fma_f32 (1), v_cvt_pk_bf16_f32 (1), ReLU on the PACKED pair as v_pk_max_i16 against 0 (a negative
        // bf16 is a negative int16) (1), mask (1): 24 VALU instructions per 16 bytes instead of 40 -- the activation is what bounds this kernel
        const unsigned okm = inside(j, c) ? 0xffffffffu : 0u;
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        typedef __attribute__((ext_vector_type(2))) short s16x2_t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x2_t q = {bf16_lo(v[e]), bf16_hi(v[e])};
            const f32x2_t sc = {psc[2 * e], psc[2 * e + 1]}, sh = {psh[2 * e], psh[2 * e + 1]};
            q = __builtin_elementwise_fma(q, sc, sh);
            unsigned pk = pack_bf16x2(q[0], q[1]);
            if (relu_on) pk = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, pk), s16x2_t{0, 0}));
            v[e] = (pk & okm) | (v[e] & ~okm);
        }
    };
    auto tf_put = [&](const u32x4* v, int buf, int j0) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = j0 + jj;
            const int id = j * 8 + wave;
            if (id < SC_DY_PIECES || id >= SC_PIECES) continue;
            *(u32x4*)(smem + buf * SC_BUF + id * 1024 + lane * 16) = v[jj];
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // fragment addresses inside a buffer: dy rows of 64 B need no swizzle (four consecutive pixels x 32 B of both channel halves = 256
    // contiguous bytes); x rows of 256 B: 64-byte segments XOR-swizzled by the halo column
    const int prow = 8 * khalf + (li >> 2);
    const int a_off = prow * 64 + (16 * nhalf + 4 * (li & 3)) * 2;                                       // + ty * 16 * 64
    const int bb = (wn0 + 16 * nhalf + 4 * (li & 3)) * 2;
    const int b0_off = SC_DYB + prow * 256 + (bb ^ ((prow & 3) << 6));                                   // + hy * SC_HPW * 256   (kw = 0)
    const int b2_off = SC_DYB + (prow + 2) * 256 + (bb ^ (((prow + 2) & 3) << 6));                       // (kw = 2)

    TSTAMP_INIT();
    TSTAMP(90);
    Cur cm = decode(t0), cr = cm;                // multiply / request cursors
    if (t0 < t1) {
#pragma unroll
        for (int j = 0; j < SC_SLOTS; ++j) issue_slot(j, cr, 0);
        step(cr);
    }
    // the two waves of a SIMD (w, w + 4) request in ANTI-PHASE: one behind tile rows 0, 2 of its K half, the other behind rows 1, 3 -- a request
    // blocks its wave ~100 cycles and the CU accepts one per ~27, so eight waves requesting at once stall every matrix pipe for the whole burst
    const int phase = wave >> 2;
    int buf = 0;
    for (int tile = t0; tile < t1; ++tile, buf ^= 1) {
        TSTAMP(91);
        mm_wait_vm<0>();
        TSTAMP(92);
        if constexpr (PRO) {
            static_assert(SC_SLOTS % 4 == 0, "four pieces per batch");
#pragma unroll
            for (int j0 = 0; j0 < SC_SLOTS; j0 += 4) {
                u32x4 v[4];
                tf_fetch(v, buf, j0);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) tf_activate(v[jj], j0 + jj, cm);
                tf_put(v, buf, j0);
            }
        }
        TSTAMP(93);
        mm_barrier();                  // this tile has landed and is activated; everybody is done with the other buffer
        TSTAMP(94);
        const bool more = tile + 1 < t1;
        const unsigned char* sb = smem + buf * SC_BUF;
        auto load_row = [&](int hy, u32x4* f) {
            tr_read2(sb + b0_off + hy * (SC_HPW * 256), 4 * 256, f[0]);
            tr_read2(sb + b2_off + hy * (SC_HPW * 256), 4 * 256, f[2]);
            f[1][0] = __builtin_amdgcn_alignbit(f[0][1], f[0][0], 16); f[1][1] = __builtin_amdgcn_alignbit(f[0][2], f[0][1], 16);
            f[1][2] = __builtin_amdgcn_alignbit(f[0][3], f[0][2], 16); f[1][3] = __builtin_amdgcn_alignbit(f[2][3], f[0][3], 16);
        };
        u32x4 rows[3][3];
        const int r0 = 4 * kh2;
        load_row(r0, rows[0]); load_row(r0 + 1, rows[1]);
#pragma unroll
        for (int tr = 0; tr < 4; ++tr) {
            load_row(r0 + tr + 2, rows[(tr + 2) % 3]);
            u32x4 af;
            tr_read2(sb + a_off + (r0 + tr) * (16 * 64), 4 * 64, af);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, rows[(tr + kh) % 3][kw]),
                                                                             acc[kh * 3 + kw], 0, 0, 0);
            if ((tr & 1) == phase && more) {
                const int k = tr >> 1;                                       // slots 0-3 behind the first row of this wave's phase, 4-7 behind the second
#pragma unroll
                for (int j = 0; j < 4; ++j) issue_slot(4 * k + j, cr, buf ^ 1);
            }
        }
        if (more) step(cr);
        step(cm);
        TSTAMP(96);
    }
    TSTAMP(97);
    mm_wait_vm<0>();
    __syncthreads();
    // ---- the two K halves are summed through the LDS (three taps per round: 4 waves x 3 x 16 x 64 floats = 48 KB), then the partial gradient of
    // this pixel group goes out as ws[gx][tap][co][ci]
    const int lr = lane & 31, lh = lane >> 5;
    float* red = (float*)smem;
    float* wsg = it.ws + (size_t)gx * (9 * 32 * 128);
#pragma unroll
    for (int round = 0; round < 3; ++round) {
        if (kh2 == 1) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((wave & 3) * 3 + t) * 16 + r) * 64 + lane] = acc[round * 3 + t][r];
        }
        __syncthreads();
        if (kh2 == 0) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[round * 3 + t][r] + red[(((wave & 3) * 3 + t) * 16 + r) * 64 + lane];
                    const int co = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    wsg[((size_t)(round * 3 + t) * 32 + co) * 128 + wn0 + lr] = v;       // a wave row = 128 contiguous bytes
                }
        }
        __syncthreads();
    }
    TSTAMP(98);
}

// the grouped DenseNet conv2 geometry: bf16, 3x3 pad 1, every problem exactly 128 -> 32 with a prologue, maps tiling into 8 x 16 pixel tiles
static bool wgrad_sc_supported(const saunet_wgrad_group* s)
{
    static const bool on = ab_env_on("SAUNET_WGRAD_SC");                // A/B switch (variant builds only)
    if (!on || s->dtype != SAUNET_BF16 || s->KH != 3 || s->pad != 1 || s->H % SC_TR || s->W % 16 || s->count < 1) return false;
    if ((long)s->N * s->H * s->W >= (1L << 22) * 8) return false;
    for (int i = 0; i < s->count; ++i) {
        const saunet_wgrad_group_item& it = s->item[i];
        if (it.Cin != 128 || it.Cout != 32 || it.ldx % 8 || it.lddy % 8 || !it.pro_scale || !it.pro_shift) return false;
        if ((long)s->N * s->H * s->W * (it.ldx > it.lddy ? it.ldx : it.lddy) >= (1L << 30)) return false;
    }
    return true;
}

static int launch_wgrad_sc(const saunet_wgrad_group* s, void* ws, size_t ws_bytes, size_t* need, hipStream_t st)
{
    ScArgs a;
    a.N = s->N; a.H = s->H; a.W = s->W; a.tiles_y = s->H / SC_TR; a.tiles_x = s->W / 16; a.ntiles = s->N * a.tiles_y * a.tiles_x;
    a.count = s->count; a.pro_relu = s->pro_relu; a.pad_ = 0;
    // pixel groups: one workgroup per CU over all problems; at least four tiles per group (a group's prologue + epilogue cost about two)
    static int cus = 0;
    if (cus == 0) {
        int dev = 0; hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        (void)hipGetLastError();
    }
    int groups = cus / s->count; if (groups < 1) groups = 1;
    if (groups > a.ntiles / 4) groups = a.ntiles / 4;
    if (groups < 1) groups = 1;
    a.groups = groups;
    const long wsize = 9L * 32 * 128;
    const size_t bytes = (size_t)s->count * groups * wsize * sizeof(float);
    if (need) { *need = bytes; return SAUNET_OK; }
    if (ws == nullptr || ws_bytes < bytes) return set_error(SAUNET_BAD_SHAPE, "grouped wgrad: workspace %zu < %zu bytes", ws_bytes, bytes);
    for (int i = 0; i < s->count; ++i) {
        const saunet_wgrad_group_item& it = s->item[i];
        a.item[i] = ScItem{(const u16*)it.x, (const u16*)it.dy, (float*)ws + (size_t)i * groups * wsize, it.dw, it.pro_scale, it.pro_shift, it.ldx, it.lddy, i * groups, 0};
    }
    for (int i = s->count; i < SAUNET_WGRAD_GROUP_MAX; ++i) a.item[i] = ScItem{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    static DeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_sc_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SC_LDS);
    hipLaunchKernelGGL(conv3x3_wgrad_sc_kernel<true>, dim3(s->count * groups), dim3(512), SC_LDS, st, a);
    SAUNET_CHECK_LAUNCH("conv3x3_wgrad_sc");
    // dw[co][ci][tap] += sum_g ws[g][tap][co][ci] for every problem of the launch: one multi-entry reduction
    static_assert(SAUNET_WGRAD_GROUP_MAX <= SAUNET_WGRAD_REDUCE_MAX, "one reduce list per grouped launch");
    saunet_wgrad_reduce_list l; l.count = s->count; l.reserved = 0;
    for (int i = 0; i < s->count; ++i) { l.item[i].ws = a.item[i].ws; l.item[i].dw = a.item[i].dw; l.item[i].wsize = wsize; l.item[i].groups = groups; l.item[i].taps = 9; }
    return wgrad_reduce_multi(&l, st);
}

bool tile_wgrad_grouped_supported(const saunet_wgrad_group* s)
{
    if (s->count < 1 || s->count > SAUNET_WGRAD_GROUP_MAX) return false;
    if (s->dtype != SAUNET_BF16 && s->dtype != SAUNET_F32) return false;
    const int epc = s->dtype == SAUNET_BF16 ? 8 : 4;
    const long P = (long)s->N * s->H * s->W;
    if (s->KH == 3) { if (s->pad != 1 || s->H % TILE || s->W % TILE) return false; }
    else if (s->KH == 1) { if (s->pad != 0 || P % 256) return false; }
    else return false;
    for (int i = 0; i < s->count; ++i) {
        const saunet_wgrad_group_item& it = s->item[i];
        if (it.Cin % epc || it.Cout % epc || it.ldx % epc || it.lddy % epc || it.Cin < epc || it.Cout < 8) return false;
        if (((uintptr_t)it.x | (uintptr_t)it.dy) & 15) return false;
        if (!it.x || !it.dy || !it.dw) return false;
    }
    return true;
}

// need != nullptr: only compute the workspace size
int tile_wgrad_grouped(const saunet_wgrad_group* s, void* ws, size_t ws_bytes, size_t* need, hipStream_t st)
{
    if (!tile_wgrad_grouped_supported(s)) return set_error(SAUNET_UNSUPPORTED, "grouped wgrad: geometry not on the tiled kernels");
    if (wgrad_sc_supported(s)) return launch_wgrad_sc(s, ws, ws_bytes, need, st);
    GroupedWgradArgs g;
    g.N = s->N; g.H = s->H; g.W = s->W; g.pro_relu = s->pro_relu; g.count = s->count; g.taps = s->KH * s->KH;
    if (s->KH == 1) { g.N = (int)((long)s->N * s->H * s->W / 256); g.H = g.W = 16; }       // pixels are just rows for a 1x1 conv
    bool small = true;
    for (int i = 0; i < s->count; ++i) {
        const saunet_wgrad_group_item& it = s->item[i];
        g.item[i].x = it.x; g.item[i].dy = it.dy; g.item[i].dw = it.dw; g.item[i].ps = it.pro_scale; g.item[i].psh = it.pro_shift; g.item[i].ws = nullptr;
        g.item[i].Cin = it.Cin; g.item[i].ldx = it.ldx; g.item[i].Cout = it.Cout; g.item[i].lddy = it.lddy; g.item[i].ncit = 0; g.item[i].blk0 = 0;
        if ((long)it.Cout * it.Cin > 64 * 128) small = false;
    }
    for (int i = s->count; i < SAUNET_WGRAD_GROUP_MAX; ++i) g.item[i] = GWItem{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0};
    if (s->dtype == SAUNET_BF16) {
        if (s->KH == 3) {
            // measured slower and removed from the library in round 4 (scripts/build_variant.sh + git history reach them): all 128 input channels per
            // workgroup (+0.15 ms per step: two 66 KB workgroups per CU overlap less than four 37 KB ones); tap split + register prefetch
            // (block 1 90.6 -> 114.5 us per layer, blocks 2-4 5-15 % faster, step +0.25 ms)
            bool all_128_32 = true;
            for (int i = 0; i < s->count; ++i) if (s->item[i].Cout > 32 || s->item[i].Cin % 128) all_128_32 = false;
            // DenseNet conv2 (128 -> 32): 64 input channels per workgroup (two channel waves x two K waves) -- the dy tile is re-read per 64 instead of per
            // 32 input channels (the block-1 launch is HBM-bound at 1.47x its algorithmic bytes): 27.71 -> 27.50 ms per step
            if (small && all_128_32) return launch_tile_wgrad_grouped<u16, 3, 16, 32, 64, 32, 32, 2>(g, s, ws, ws_bytes, need, st);
            if (small) return launch_tile_wgrad_grouped<u16, 3, 16, 32, 32, 32, 32, 4>(g, s, ws, ws_bytes, need, st);
            return launch_tile_wgrad_grouped<u16, 3, 8, 64, 64, 32, 32, 1>(g, s, ws, ws_bytes, need, st);
        }
        if (small) return launch_tile_wgrad_grouped<u16, 1, 16, 64, 64, 64, 64, 4>(g, s, ws, ws_bytes, need, st);
        return launch_tile_wgrad_grouped<u16, 1, 8, 128, 128, 64, 64, 1>(g, s, ws, ws_bytes, need, st);
    }
    if (f32_split_wanted((long)s->N * s->H * s->W)) {
        if (s->KH == 3) {
            if (small) return launch_tile_wgrad_grouped<f32s, 3, 16, 32, 32, 32, 32, 4>(g, s, ws, ws_bytes, need, st);
            return launch_tile_wgrad_grouped<f32s, 3, 8, 64, 64, 32, 32, 1>(g, s, ws, ws_bytes, need, st);
        }
        return launch_tile_wgrad_grouped<f32s, 1, 8, 64, 64, 32, 32, 1>(g, s, ws, ws_bytes, need, st);
    }
    if (s->KH == 3) {
        if (small) return launch_tile_wgrad_grouped<float, 3, 16, 32, 32, 32, 32, 4>(g, s, ws, ws_bytes, need, st);
        return launch_tile_wgrad_grouped<float, 3, 8, 64, 64, 32, 32, 1>(g, s, ws, ws_bytes, need, st);
    }
    return launch_tile_wgrad_grouped<float, 1, 8, 64, 64, 32, 32, 1>(g, s, ws, ws_bytes, need, st);
}

int tile_wgrad(const saunet_conv_desc* d, const void* x, const void* dy, const float* ps, const float* psh, float* dw,
               void* ws, size_t ws_bytes, size_t* need, bool aligned, hipStream_t st, saunet_wgrad_pending* pend)
{
    TileWgradArgs a;
    a.pend = pend;
    a.ws = (float*)ws;
    a.x = x; a.dy = dy; a.dw = dw; a.pro_scale = ps; a.pro_shift = psh; a.pro_relu = d->pro_relu;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx; a.Cout = d->Cout; a.lddy = d->ldy;
    a.sM = (long)d->Cin * d->KH * d->KW; a.sN = (long)d->KH * d->KW;
    a.xs_n = a.xs_r = a.xs_p = 0; a.ncot = 0; a.Wo = 0;
    if (aligned && !need && (((uintptr_t)x | (uintptr_t)dy) & 15)) return set_error(SAUNET_BAD_ALIGN, "wgrad: pointers must be 16-byte aligned");
    if (wgrad_mm_supported(a, d->KH, d->dtype, aligned)) {
        if (need) {      // the workspace query has no operands: it cannot know whether a prologue will come -- size for both kernels
            size_t n1 = 0, n2 = 0;
            if (int rc = launch_wgrad_mm(a, 0, &n1, st)) return rc;
            if (int rc = dispatch_tile_wgrad<u16>(a, d->KH, aligned, 0, &n2, st)) return rc;
            *need = n1 > n2 ? n1 : n2;
            return SAUNET_OK;
        }
        return launch_wgrad_mm(a, ws_bytes, need, st);
    }
    if (d->dtype == SAUNET_BF16) return dispatch_tile_wgrad<u16>(a, d->KH, aligned, ws_bytes, need, st);
    if (d->dtype == SAUNET_F32 && aligned && f32_split_wanted((long)d->N * d->H * d->W)) return dispatch_tile_wgrad<f32s>(a, d->KH, aligned, ws_bytes, need, st);
    if (d->dtype == SAUNET_F32) return dispatch_tile_wgrad<float>(a, d->KH, aligned, ws_bytes, need, st);
    return set_error(SAUNET_BAD_DTYPE, "wgrad: dtype %d", d->dtype);
}

// Weight gradient of ConvTranspose2d(k = 4, s = 2, p = 1) straight from the NHWC tensors (no im2col of dy: that pass wrote and the GEMM re-read four
// copies of dy): d describes the transposed convolution (x [N,H,W,Cin], dy [N,2H,2W,Cout], dw [Cin][Cout][4][4]).  See the KS == 2 notes in tile_wgrad_body.
bool tile_wgrad_convt_supported(const saunet_conv_desc* d)
{
    return d->transposed && d->dtype == SAUNET_BF16 && d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad == 1 && d->Ho == 2 * d->H && d->Wo == 2 * d->W &&
           d->H % TILE == 0 && d->W % TILE == 0 && d->Cin % 8 == 0 && d->Cout % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0;
}
int tile_wgrad_convt(const saunet_conv_desc* d, const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, size_t* need, hipStream_t st,
                     saunet_wgrad_pending* pend)
{
    TileWgradArgs a;
    a.pend = pend; a.ws = (float*)ws;
    a.x = dy; a.dy = x; a.dw = dw; a.pro_scale = nullptr; a.pro_shift = nullptr; a.pro_relu = 0;      // roles swapped: the haloed operand is dy's parity image
    a.N = d->N; a.H = d->H; a.W = d->W;
    a.Cin = d->Cout; a.ldx = d->ldy;          // "input channels" of the kernel = channels of the haloed operand = the layer's output channels
    a.Cout = d->Cin; a.lddy = d->ldx;         // "output channels" = channels of the un-haloed operand = the layer's input channels
    a.xs_p = 2L * d->ldy; a.xs_r = 2L * d->Wo * d->ldy; a.xs_n = (long)d->Ho * d->Wo * d->ldy; a.Wo = d->Wo; a.ncot = 0;
    a.sM = (long)d->Cout * 16; a.sN = 16;     // dw[ci][co][kh][kw]
    if (!need && (((uintptr_t)x | (uintptr_t)dy) & 15)) return set_error(SAUNET_BAD_ALIGN, "conv-transpose wgrad: pointers must be 16-byte aligned");
    if (convt_wgrad_mm_supported(d)) {
        if (need) {      // (the query sees no operands: size for both kernels)
            size_t n1 = 0, n2 = 0;
            if (int rc = launch_convt_wgrad_mm(d, x, dy, dw, ws, 0, &n1, st)) return rc;
            if (int rc = launch_tile_wgrad<u16, 2, 8, 64, 64, 32, 32, 1>(a, 0, &n2, st)) return rc;
            *need = n1 > n2 ? n1 : n2;
            return SAUNET_OK;
        }
        return launch_convt_wgrad_mm(d, x, dy, dw, ws, ws_bytes, nullptr, st, pend);
    }
    // (measured at dec4, 189 us: a 128 x 64 channel tile with 64 x 32 per wave -- each haloed-operand fragment feeding two MFMAs -- 264 us; the register
    // prefetch of the next tile, which fits here without spills, 194 us: the tile loop is bound by the transposing LDS fragment reads, not by load latency)
    return launch_tile_wgrad<u16, 2, 8, 64, 64, 32, 32, 1>(a, ws_bytes, need, st);
}

}  // namespace saunet

SAUNET_TIMING_READER(conv_tile)
