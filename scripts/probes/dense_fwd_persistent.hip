// Persistent forward of a whole DenseNet block on the low-resolution maps (dense blocks 3 / 4 of the 256x256 step: 32x32 and 16x16 maps,
// 24 + 16 layers).  As separate launches every layer costs  finalize + conv1 + finalize + conv2  = four launches of 5-25 us each whose work is
// 1-5 us: the step spent ~2.4 ms there for 12 % of the forward FLOPs.  Here ONE launch walks all layers of the block:
//
//     one workgroup per CU (<= number of 8x16 pixel tiles), every workgroup keeps its pixel tiles for the whole launch;
//     per layer:  conv1 (1x1, Cin -> 128, BN1+ReLU prologue, statistics of z1)      -> GRID BARRIER (z1 halo rows + its statistics are global)
//                 conv2 (3x3, 128 -> 32, BN2+ReLU prologue) into the concat slice    -> GRID BARRIER (statistics of the 32 new channels)
//     BatchNorm finalize is done redundantly by every workgroup from the replicated float64 accumulators (per-channel mean / var of the concat
//     channels are kept in LDS for the whole launch: a channel's batch statistics never change, only gamma / beta differ per layer); workgroup
//     l % grid also writes the layer's BNParams (what backward reads) and updates the running statistics.
//
// Both convolutions are computed TRANSPOSED (rows = output channels from LDS, columns = pixels): the activation operand of conv1 is a 16-byte
// K-contiguous piece of a concat row per lane, straight from global memory into registers (BN+ReLU applied there), the weights stream through
// LDS in 128-channel chunks (double buffered); conv2 stages its 10x18 halo of z1 (BN2+ReLU applied) and the whole 32x1152 weight tile in LDS.
// Accumulator -> memory as in dense_dgrad.hip: one v_permlane32_swap per value pair gives every lane 8 consecutive channels of its pixel
// (16-byte stores), per-channel statistics are five DPP adds per value, one LDS atomic per wave and one float64 atomic per workgroup.
//
// Grid barrier: arrive = agent-scope release fence + atomic increment, wait = polling an agent-scope load with a BOUNDED spin count; on
// expiry a sticky abort flag is raised and every later barrier falls through (wrong numbers, but never a hung GPU).  All workgroups are
// co-resident by construction (grid <= CUs, one workgroup per CU fits) as long as no OTHER persistent kernel competes for the CUs: the host
// side does not use this path when several processes may share the device.
//
// Replaces, for training-mode bf16 blocks on small maps, the per-layer launches of torchvision's _DenseLayer as used at
// /root/reference/models/models.py:271,306-313.
#include "common.h"
#include <stdlib.h>

namespace saunet {

namespace {

template <int CTRL> __device__ __forceinline__ float dpp_add_f(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(moved);
}
// sum over the 32 lanes of each wave half; valid in lanes 16..31 and 48..63 (see dense_dgrad.hip)
__device__ __forceinline__ float half_sum(float v)
{
    v = dpp_add_f<0xB1>(v);
    v = dpp_add_f<0x4E>(v);
    v = dpp_add_f<0x141>(v);
    v = dpp_add_f<0x140>(v);
    v = dpp_add_f<0x142>(v);
    return v;
}

constexpr int DF_THREADS = 256;
constexpr int DF_TR = 8, DF_TW = 16, DF_NPX = DF_TR * DF_TW;            // pixel tile 8 x 16
constexpr int DF_HR = DF_TR + 2, DF_HC = DF_TW + 2, DF_NH = DF_HR * DF_HC;
constexpr int DF_KCH = 128;                                               // conv1 weight chunk (channels)
constexpr int DF_W1P = DF_KCH + 8;                                        // u16 pitch of a weight-chunk row (272 B: rows 4 banks apart)
constexpr int DF_HP = 128 + 8;                                            // u16 pitch of a halo pixel (17 chunks of 16 B: odd)
constexpr int DF_HROW = ((DF_HC * DF_HP * 2 + 255) / 256) * 128;          // u16 pitch of a halo ROW: a multiple of 256 B, so the two halo rows an MFMA
                                                                          // fragment lane group touches do not collide (see swz_halo in conv_tile.hip)
constexpr int DF_W2P = 9 * 128 + 8;                                       // u16 pitch of a conv2 weight row (2320 B)
constexpr int DF_CMAX = 1024;                                             // concat channels kept in LDS (mean / var / scale / shift)
constexpr unsigned DF_SPIN_LIMIT = 1u << 22;

struct DenseFwdArgs {
    saunet_dense_fwd_desc d;
    unsigned* sync;       // [0] arrival counter, [1] abort flag (both zero at launch)
    int tiles_x, tiles_y, ntiles;
};

// Two-level barrier.  256 workgroups incrementing and polling ONE word serialise at its memory channel (~85 ns per arrival: 22 us per barrier,
// s_memtime stamps); here 16 groups count their members on 16 different 128-byte lines, the last member of a group counts at the root, the last
// group releases 16 per-group flags and every workgroup polls only its group's flag.  Counters are cumulative (generation gen = number of
// barriers so far); word 1 is the sticky abort flag.
constexpr int DF_BAR_GROUPS = 16;
constexpr int DF_BAR_LINE = 32;                                                    // words per 128-byte line
__device__ __forceinline__ void grid_barrier(unsigned* sync, unsigned gen, int nblk, int bid)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                              // EVERY wave: its write-through stores / atomics have completed
    __syncthreads();
    if (threadIdx.x == 0) {
        const int gsize = (nblk + DF_BAR_GROUPS - 1) / DF_BAR_GROUPS, ngroups = (nblk + gsize - 1) / gsize;
        const int g = bid / gsize, members = min(gsize, nblk - g * gsize);
        unsigned* gcount = sync + DF_BAR_LINE * (1 + g);
        unsigned* root = sync + DF_BAR_LINE * (1 + DF_BAR_GROUPS);
        unsigned* flags = sync + DF_BAR_LINE * (2 + DF_BAR_GROUPS);
        // No agent-scope FENCE here: a release / acquire fence writes back and invalidates the whole 4 MB L2 of the XCD, once per workgroup
        // (32 per XCD) and barrier -- 20 us per barrier in the stamps.  Everything another workgroup reads before the kernel ends is instead
        // written with agent-scope (write-through) stores or float64 atomics and read with agent-scope loads (st_agent / ld_agent below);
        // __syncthreads() above has waited for this workgroup's outstanding stores (vmcnt counts them on gfx9).
        if (__hip_atomic_fetch_add(gcount, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gen * (unsigned)members) {
            if (__hip_atomic_fetch_add(root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gen * (unsigned)ngroups) {
                for (int i = 0; i < ngroups; ++i) __hip_atomic_store(flags + DF_BAR_LINE * i, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        unsigned spins = 0;
        while (__hip_atomic_load(flags + DF_BAR_LINE * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
            if (__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (++spins > DF_SPIN_LIMIT) { __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
}

// 16 bytes to / from memory at agent scope: buffer stores / loads with the sc1 bit (write-through stores, L1-bypassing loads).  One 16-byte
// sc1 access costs what a plain one does; two 8-byte agent atomics (what __hip_atomic_store lowers to) are 2.7x slower per byte.
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int v4u32_t;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void st_agent(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, const u32x4& v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32_t, v), rsrc, (int)byte_off, 0, 16);
}
__device__ __forceinline__ u32x4 ld_agent(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off)
{
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 16));
}
__device__ __forceinline__ double ld_agent(const double* p)
{
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
constexpr size_t DF_SYNC_BYTES = (size_t)DF_BAR_LINE * 4 * (2 + 2 * DF_BAR_GROUPS);

// sum of the replicas of one accumulator: the loads of a batch of eight are independent (one memory round trip per batch, not per replica)
__device__ __forceinline__ double rep_sum_d(const double* p, int reps, int rstride, int c)
{
    double s = 0.0;
    for (int r0 = 0; r0 < reps; r0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = r0 + u < reps ? ld_agent(p + (size_t)(r0 + u) * rstride + c) : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    return s;
}

}  // namespace

__global__ __launch_bounds__(DF_THREADS, 1) void dense_block_fwd_kernel(DenseFwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const saunet_dense_fwd_desc& d = a.d;
    float* s_mean = (float*)smem;                  // [CMAX] batch mean of every concat channel
    float* s_var = s_mean + DF_CMAX;               // [CMAX] biased batch variance
    float* s_scale = s_var + DF_CMAX;              // [CMAX] current layer's prologue scale (norm1; first 128 entries re-used for norm2)
    float* s_shift = s_scale + DF_CMAX;            // [CMAX]
    float* s_sum = s_shift + DF_CMAX;              // [2][128] statistics partials of the workgroup
    unsigned char* s_big = (unsigned char*)(s_sum + 256);
    u16* s_w1 = (u16*)s_big;                       // conv1: [3][128][W1P]
    u16* s_halo = (u16*)s_big;                     // conv2: [NH][HP]
    u16* s_w2 = s_halo + DF_HR * DF_HROW;          // conv2: [32][W2P]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const double count = (double)d.N * d.H * d.W;
    const int ctot = d.c0 + 32 * d.nl;
    u16* const buf = (u16*)d.buf;
    unsigned bar = 0;                               // barriers passed so far

    // per-channel batch statistics of channels [lo, hi) of the concat buffer -> LDS
    auto load_stats = [&](int lo, int hi) {
        // thread -> (channel, which sum): both sums of a channel sit in neighbouring lanes, exchanged with one DPP-free shuffle
        const int npair = 2 * (hi - lo);
        for (int i = tid; i < ((npair + 63) & ~63); i += DF_THREADS) {
            const int c = lo + (i >> 1), which = i & 1;
            const double sv = i < npair ? rep_sum_d(d.stats + (which ? ctot : 0), d.stat_reps, d.stat_rstride, c) : 0.0;
            const double other = __shfl_xor(sv, 1, 64);
            if (i < npair && which == 0) {
                const double m = sv / count;
                double var = other / count - m * m;
                if (var < 0.0) var = 0.0;
                s_mean[c] = (float)m; s_var[c] = (float)var;
            }
        }
    };
    // scale / shift of a BatchNorm over channels [0, C) whose statistics are mean[] / var[] (LDS or computed), as bn_finalize_kernel does;
    // the designated workgroup also writes BNParams [4][C] and the running statistics
    auto finalize = [&](int C, const float* mean, const float* var, const float* gamma, const float* beta, float eps, float mom,
                        float* rmean, float* rvar, float* pout, bool writer) {
        for (int c = tid; c < C; c += DF_THREADS) {
            const float m = mean[c];
            const double v = (double)var[c];
            const float invstd = (float)(1.0 / sqrt(v + (double)eps));
            const float s = gamma[c] * invstd, t = beta[c] - m * s;
            s_scale[c] = s; s_shift[c] = t;
            if (writer) {
                pout[c] = s; pout[C + c] = t; pout[2 * C + c] = m; pout[3 * C + c] = invstd;
                if (rmean) {
                    const double unb = count > 1.0 ? v * count / (count - 1.0) : v;
                    rmean[c] = (1.f - mom) * rmean[c] + mom * m;
                    rvar[c] = (1.f - mom) * rvar[c] + mom * (float)unb;
                }
            }
        }
    };

    TSTAMP_INIT();
    load_stats(0, d.c0);
    __syncthreads();

    for (int l = 0; l < d.nl; ++l) {
        TSTAMP(80);
        const saunet_dense_fwd_layer& L = d.layer[l];
        const int cin = d.c0 + 32 * l;
        const bool writer = (l % nblk) == bid;
        const u16* __restrict__ w1 = (const u16*)L.w1;
        const u16* __restrict__ w2 = (const u16*)L.w2;
        u16* __restrict__ z1 = (u16*)L.z1;
        const __amdgpu_buffer_rsrc_t z1rs = make_rsrc(z1, (unsigned)((size_t)d.N * d.H * d.W * 128 * 2));

        // ================================================================ norm1 coefficients
        finalize(cin, s_mean, s_var, L.gamma1, L.beta1, L.eps, L.momentum, L.rmean1, L.rvar1, L.p1, writer);
        for (int i = tid; i < 256; i += DF_THREADS) s_sum[i] = 0.f;
        __syncthreads();
        TSTAMP(81);

        // ================================================================ conv1: z1[p][0:128] = W1 . relu(bn1(x[p][0:cin]))
        const int nchunks = (cin + DF_KCH - 1) / DF_KCH;
        for (int tile = bid; tile < a.ntiles; tile += nblk) {
            int bt = tile;
            const int txi = bt % a.tiles_x; bt /= a.tiles_x;
            const int tyi = bt % a.tiles_y; const int n = bt / a.tiles_y;
            const int prow = tyi * DF_TR + 2 * wave + (lr >> 4), pcol = txi * DF_TW + (lr & 15);
            const size_t pix = ((size_t)n * d.H + prow) * d.W + pcol;
            const u16* __restrict__ xrow = buf + pix * d.ldbuf + 8 * lh;

            f32x16 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

            // Three-deep software pipeline over the 128-channel chunks of K: while the MFMAs of chunk c run, the activation pieces and weight
            // pieces of chunks c+1 AND c+2 are in flight (128 KB per CU; with one chunk ahead the loads had a single ~1k-cycle MFMA phase to
            // land and conv1 ran at a third of the per-CU load rate).  Register sets and LDS buffers rotate through compile-time names (the
            // loop is unrolled by six = lcm of 3 activation sets and 2 weight sets), so the in-order vmcnt waits stay partial.
            auto w_issue = [&](int c, u32x4 (&wr)[8]) {       // weight chunk c -> registers (8 pieces of 16 B per thread)
                const int k0 = c * DF_KCH;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = tid + u * DF_THREADS, row = q >> 4, ch = q & 15;
                    const int k = k0 + ch * 8;
                    wr[u] = k < cin ? *(const u32x4*)(w1 + (size_t)row * cin + k) : u32x4{0u, 0u, 0u, 0u};
                }
            };
            auto w_commit = [&](int b, const u32x4 (&wr)[8]) {  // registers -> LDS buffer b
                u16* dst = s_w1 + b * 128 * DF_W1P;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = tid + u * DF_THREADS, row = q >> 4, ch = q & 15;
                    *(u32x4*)(dst + row * DF_W1P + ch * 8) = wr[u];
                }
            };
            auto x_issue = [&](int c, u32x4 (&x)[8]) {        // activation pieces of chunk c (8 k-steps of 16 channels) -> registers
                const int k0 = c * DF_KCH;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int k = k0 + ks * 16;
                    x[ks] = k < cin ? *(const u32x4*)(xrow + k) : u32x4{0u, 0u, 0u, 0u};
                }
            };
            auto x_transform = [&](int c, u32x4 (&x)[8]) {
                const int k0 = c * DF_KCH + 8 * lh;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int k = k0 + ks * 16;
                    if (k < cin) {                       // cin is a multiple of 32: the 8 channels of a piece are all valid or all beyond cin
                        float f[8];
                        Vec16<u16>::unpack(x[ks], f);
                        const f32x4 s0 = *(const f32x4*)(s_scale + k), s1 = *(const f32x4*)(s_scale + k + 4);
                        const f32x4 t0 = *(const f32x4*)(s_shift + k), t1 = *(const f32x4*)(s_shift + k + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { f[j] = fmaxf(fmaf(f[j], s0[j], t0[j]), 0.f); f[4 + j] = fmaxf(fmaf(f[4 + j], s1[j], t1[j]), 0.f); }
                        x[ks] = Vec16<u16>::pack(f);
                    }
                }
            };
            auto mma_chunk = [&](int c, int b, const u32x4 (&x)[8]) {
                const u16* wb = s_w1 + b * 128 * DF_W1P + lr * DF_W1P + lh * 8;
                const int ksn = min(8, (cin - c * DF_KCH) / 16);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks < ksn) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const u32x4 wf = *(const u32x4*)(wb + t * 32 * DF_W1P + ks * 16);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, x[ks]), __builtin_bit_cast(bf16x8_t, wf), acc[t], 0, 0, 0);
                        }
                    }
                }
            };
            u32x4 x0[8], x1[8], x2[8], wa[8], wb_[8];
            // iteration c: xc = pieces of chunk c, xn = free set (receives chunk c+2), wn = weights of chunk c+1 (in flight), wf = free set
            // (receives chunk c+2), bc / bn = LDS buffers of chunks c / c+1
            auto step = [&](int c, u32x4 (&xc)[8], u32x4 (&xn)[8], u32x4 (&wn)[8], u32x4 (&wf)[8], int bc, int bn) {
                TSTAMP(91);
                if (c + 2 < nchunks) { w_issue(c + 2, wf); x_issue(c + 2, xn); }
                TSTAMP(92);
                x_transform(c, xc);
                TSTAMP(93);
                mma_chunk(c, bc, xc);
                TSTAMP(94);
                if (c + 1 < nchunks) w_commit(bn, wn);
                TSTAMP(95);
                __syncthreads();
                TSTAMP(96);
            };

            __syncthreads();                               // the previous tile / phase no longer reads the big LDS region
            w_issue(0, wa); x_issue(0, x0);
            if (nchunks > 1) { w_issue(1, wb_); x_issue(1, x1); }
            w_commit(0, wa);
            __syncthreads();
            for (int c = 0; c < nchunks; c += 6) {
                step(c, x0, x2, wb_, wa, 0, 1);
                if (c + 1 >= nchunks) break;
                step(c + 1, x1, x0, wa, wb_, 1, 2);
                if (c + 2 >= nchunks) break;
                step(c + 2, x2, x1, wb_, wa, 2, 0);
                if (c + 3 >= nchunks) break;
                step(c + 3, x0, x2, wa, wb_, 0, 1);
                if (c + 4 >= nchunks) break;
                step(c + 4, x1, x0, wb_, wa, 1, 2);
                if (c + 5 >= nchunks) break;
                step(c + 5, x2, x1, wa, wb_, 2, 0);
            }

            // epilogue.  conv1 is computed with rows = pixels, columns = output channels: a lane owns ONE channel (32 t + lr) and 16 pixels, so the
            // statistics are 16 adds per lane, one cross-half shuffle and one LDS atomic per channel and wave (in the transposed form every value
            // needed a five-step DPP reduction over the pixels: 10k cycles per tile in the stamps).  The bf16 tile goes through LDS
            // ([pixel][channel], the weight buffers are free after the last barrier of the K loop) and leaves as 16-byte agent-scope stores.
            TSTAMP(97);
            u16* so = s_w1;                                  // [128 pixels][W1P]
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float sv = 0.f, sq = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[t][r];
                    sv += v; sq = fmaf(v, v, sq);
                    const int px = 32 * wave + 8 * (r >> 2) + 4 * lh + (r & 3);
                    so[px * DF_W1P + 32 * t + lr] = __builtin_bit_cast(u16, (__bf16)v);
                }
                sv += __shfl_xor(sv, 32, 64); sq += __shfl_xor(sq, 32, 64);
                if (lh == 0) { atomicAdd(&s_sum[32 * t + lr], sv); atomicAdd(&s_sum[128 + 32 * t + lr], sq); }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = tid + u * DF_THREADS, px = q >> 4, ch = q & 15;
                const size_t gp = ((size_t)n * d.H + tyi * DF_TR + (px >> 4)) * d.W + txi * DF_TW + (px & 15);
                st_agent(z1rs, (unsigned)((gp * 128 + ch * 8) * 2), *(const u32x4*)(so + px * DF_W1P + ch * 8));
            }
        }
        __syncthreads();
        TSTAMP(82);
        if (bid < a.ntiles) {
            double* st = L.st2 + (size_t)(bid % L.st2_reps) * L.st2_rstride;
            for (int i = tid; i < 128; i += DF_THREADS) { atomicAdd(&st[i], (double)s_sum[i]); atomicAdd(&st[128 + i], (double)s_sum[128 + i]); }
        }
        TSTAMP(83);
        grid_barrier(a.sync, ++bar, nblk, bid);
        TSTAMP(84);

        // ================================================================ norm2 coefficients (statistics of z1)
        {
            float* m2 = s_sum, * v2 = s_sum + 128;          // the partials are consumed: re-use them for mean / var of the 128 channels
            __syncthreads();
            {
                const int c = tid >> 1, which = tid & 1;                    // 256 threads = 128 channels x (sum, sum of squares)
                const double sv = rep_sum_d(L.st2 + (which ? 128 : 0), L.st2_reps, L.st2_rstride, c);
                const double other = __shfl_xor(sv, 1, 64);
                if (which == 0) {
                    const double m = sv / count;
                    double var = other / count - m * m;
                    if (var < 0.0) var = 0.0;
                    m2[c] = (float)m; v2[c] = (float)var;
                }
            }
            __syncthreads();
            finalize(128, m2, v2, L.gamma2, L.beta2, L.eps, L.momentum, L.rmean2, L.rvar2, L.p2, writer);
            __syncthreads();
            for (int i = tid; i < 256; i += DF_THREADS) s_sum[i] = 0.f;
        }

        // ================================================================ conv2: buf[p][cin:cin+32] = W2 (3x3) . relu(bn2(z1))
        // weights of the layer -> LDS (18 pieces per thread, two batches)
        __syncthreads();
        TSTAMP(85);
#pragma unroll 1
        for (int b = 0; b < 2; ++b) {
            u32x4 v[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                const int q = tid + (b * 9 + u) * DF_THREADS;        // 32 rows x 144 pieces
                v[u] = *(const u32x4*)(w2 + (size_t)q * 8);
            }
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                const int q = tid + (b * 9 + u) * DF_THREADS, row = q / 144, ch = q - row * 144;
                *(u32x4*)(s_w2 + row * DF_W2P + ch * 8) = v[u];
            }
        }
        TSTAMP(86);
        float sc2[8], sh2[8];
        {
            const int ch = (tid & 15) * 8;                            // 256 % 16 == 0: the thread's halo pieces all carry the same 8 channels
#pragma unroll
            for (int j = 0; j < 8; ++j) { sc2[j] = s_scale[ch + j]; sh2[j] = s_shift[ch + j]; }
        }
        for (int tile = bid; tile < a.ntiles; tile += nblk) {
            int bt = tile;
            const int txi = bt % a.tiles_x; bt /= a.tiles_x;
            const int tyi = bt % a.tiles_y; const int n = bt / a.tiles_y;
            const int ty0 = tyi * DF_TR, tx0 = txi * DF_TW;
            __syncthreads();                               // previous tile's fragment reads are done
            // halo of z1: 180 pixels x 16 pieces = 2880 pieces, 12 per thread in two batches of six
#pragma unroll 1
            for (int b = 0; b < 2; ++b) {
                u32x4 v[6]; bool ok[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int q = tid + (b * 6 + u) * DF_THREADS, hp = q >> 4, ch = q & 15;
                    const int hy = hp / DF_HC, hx = hp - hy * DF_HC;
                    const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
                    ok[u] = q < DF_NH * 16 && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                    v[u] = ok[u] ? ld_agent(z1rs, (unsigned)(((((size_t)n * d.H + iy) * d.W + ix) * 128 + ch * 8) * 2)) : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int q = tid + (b * 6 + u) * DF_THREADS, hp = q >> 4, ch = q & 15;
                    if (q < DF_NH * 16) {
                        u32x4 o = {0u, 0u, 0u, 0u};
                        if (ok[u]) {
                            float f[8];
                            Vec16<u16>::unpack(v[u], f);
#pragma unroll
                            for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], sc2[j], sh2[j]), 0.f);
                            o = Vec16<u16>::pack(f);
                        }
                        *(u32x4*)(s_halo + (hp / DF_HC) * DF_HROW + (hp % DF_HC) * DF_HP + ch * 8) = o;
                    }
                }
            }
            __syncthreads();
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int hbase = (2 * wave + (lr >> 4)) * DF_HROW + (lr & 15) * DF_HP;       // halo element offset of this lane's output pixel, tap (0,0)
            const u16* wrow = s_w2 + lr * DF_W2P + lh * 8;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const u16* hrow = s_halo + hbase + (tap / 3) * DF_HROW + (tap % 3) * DF_HP + lh * 8;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const u32x4 wf = *(const u32x4*)(wrow + tap * 128 + ks * 16);
                    const u32x4 xf = *(const u32x4*)(hrow + ks * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, xf), acc, 0, 0, 0);
                }
            }
            const int prow = ty0 + 2 * wave + (lr >> 4), pcol = tx0 + (lr & 15);
            u16* orow = buf + (((size_t)n * d.H + prow) * d.W + pcol) * d.ldbuf + cin + 8 * lh;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float G[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * r + q]), __float_as_uint(acc[8 * r + 4 + q]), false, false);
                    G[q] = __uint_as_float(sw[0]); G[4 + q] = __uint_as_float(sw[1]);
                }
                const int cl = 16 * r + 8 * lh;
                float e1[8], e2[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { e1[q] = half_sum(G[q]); e2[q] = half_sum(G[q] * G[q]); }
                if (lr == 31) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { atomicAdd(&s_sum[cl + q], e1[q]); atomicAdd(&s_sum[128 + cl + q], e2[q]); }
                }
                *(u32x4*)(orow + 16 * r) = Vec16<u16>::pack(G);
            }
        }
        __syncthreads();
        TSTAMP(87);
        if (bid < a.ntiles) {
            double* st = d.stats + (size_t)(bid % d.stat_reps) * d.stat_rstride;
            for (int i = tid; i < 32; i += DF_THREADS) { atomicAdd(&st[cin + i], (double)s_sum[i]); atomicAdd(&st[ctot + cin + i], (double)s_sum[128 + i]); }
        }
        TSTAMP(88);
        grid_barrier(a.sync, ++bar, nblk, bid);
        TSTAMP(89);
        load_stats(cin, cin + 32);
        __syncthreads();
    }
    // a barrier that expired (workgroups not co-resident: CU masking, a second process on the device) left wrong numbers behind; nobody may
    // train on them silently -- the run must fail loudly: poison the block's statistics, every later BatchNorm and the loss turn NaN
    if (bid == 0 && tid == 0 && __hip_atomic_load(&a.sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
        d.stats[0] = __builtin_nan("");
}

}  // namespace saunet

SAUNET_TIMING_READER(dense_fwd)

extern "C" {

int saunet_dense_block_forward(const saunet_dense_fwd_desc* d, void* sync_ws, void* stream)
{
    using namespace saunet;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype != SAUNET_BF16) return set_error(SAUNET_BAD_DTYPE, "dense_block_forward: bf16 storage only");
    if (d->nl < 1 || d->nl > SAUNET_DENSE_MAX_LAYERS) return set_error(SAUNET_BAD_SHAPE, "dense_block_forward: %d layers", d->nl);
    const int ctot = d->c0 + 32 * d->nl;
    if (d->N < 1 || d->H % DF_TR || d->W % DF_TW || d->c0 % 32 || d->c0 < 32 || ctot > DF_CMAX || d->ldbuf < ctot || d->ldbuf % 8)
        return set_error(SAUNET_BAD_SHAPE, "dense_block_forward: N=%d H=%d W=%d c0=%d layers=%d ld=%d", d->N, d->H, d->W, d->c0, d->nl, d->ldbuf);
    if ((long)d->N * d->H * d->W * 256 >= (1L << 32)) return set_error(SAUNET_BAD_SHAPE, "dense_block_forward: z1 must stay below 4 GiB (buffer addressing)");
    if (d->stat_reps < 1 || sync_ws == nullptr || d->buf == nullptr || d->stats == nullptr) return set_error(SAUNET_BAD_SHAPE, "dense_block_forward: null buffer");
    if (((uintptr_t)d->buf) & 15) return set_error(SAUNET_BAD_ALIGN, "dense_block_forward: concat buffer must be 16-byte aligned");
    for (int l = 0; l < d->nl; ++l) {
        const saunet_dense_fwd_layer& L = d->layer[l];
        if (!L.w1 || !L.w2 || !L.gamma1 || !L.beta1 || !L.gamma2 || !L.beta2 || !L.z1 || !L.p1 || !L.p2 || !L.st2 || L.st2_reps < 1)
            return set_error(SAUNET_BAD_SHAPE, "dense_block_forward: layer %d has a null pointer", l);
        if ((((uintptr_t)L.w1) | ((uintptr_t)L.w2) | ((uintptr_t)L.z1)) & 15) return set_error(SAUNET_BAD_ALIGN, "dense_block_forward: layer %d operands must be 16-byte aligned", l);
    }
    DenseFwdArgs a;
    a.d = *d; a.sync = (unsigned*)sync_ws;
    a.tiles_x = d->W / DF_TW; a.tiles_y = d->H / DF_TR; a.ntiles = d->N * a.tiles_x * a.tiles_y;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        return set_error(SAUNET_LAUNCH_FAILED, "dense_block_forward: cannot query the device");
    const int blocks = a.ntiles < cus ? a.ntiles : cus;            // one workgroup per CU: all of them are resident, the grid barrier cannot starve
    const size_t lds_small = sizeof(float) * (4 * DF_CMAX + 256);
    const size_t big1 = (size_t)3 * 128 * DF_W1P * 2, big2 = ((size_t)DF_HR * DF_HROW + (size_t)32 * DF_W2P) * 2;
    const size_t lds = lds_small + (big1 > big2 ? big1 : big2);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)dense_block_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return set_error(SAUNET_LAUNCH_FAILED, "dense_block_forward: %zu bytes of LDS refused", lds);
        attr_set = true;
    }
    if (hipMemsetAsync(sync_ws, 0, DF_SYNC_BYTES, st) != hipSuccess) return set_error(SAUNET_LAUNCH_FAILED, "dense_block_forward: memset");
    hipLaunchKernelGGL(dense_block_fwd_kernel, dim3(blocks), dim3(DF_THREADS), lds, st, a);
    SAUNET_CHECK_LAUNCH("dense_block_forward");
    return SAUNET_OK;
}

}  // extern "C"
