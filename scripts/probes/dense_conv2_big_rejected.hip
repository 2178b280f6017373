// REJECTED (round 6) -- kept with its measurement, not part of the library.  To rebuild: paste into csrc/dense_fwd.hip in front of the closing
// `}  // namespace saunet`, declare dense_conv2_big_supported / _forward in conv.hip and call them from saunet_conv2d_forward_bnpro.
//
// DenseNet conv2 (3x3, 128 -> 32) forward on the large maps as a persistent LDS-DMA kernel: resident weights (one DMA at kernel start), the
// 18 x 18 x 64-channel halo of a (tile, channel half) unit double-buffered by LDS-DMA, BN + ReLU in place in the LDS by the requesting wave,
// one barrier per unit, transposed product with a permlane32_swap epilogue (no staging tile), statistics in registers.  162 VGPRs, no scratch,
// 157 KB LDS.  Measured (scripts/dense_chain_micro.py, graph of the whole block, same box, SAUNET_DENSE_CONV2_BIG=0/1 in a variant build):
//     block 1 forward 168.3 -> 169.0 us per layer, block 2 68.8 -> 72.3.
// The register-staged resident kernel of conv_tile.hip (conv3x3_res_fwd_kernel<32, 32, 32, 8>) is therefore not improved by taking the staging off
// the waves (its phase stamps: profiles/r06_phase_timing_raw.txt conv2fwd).
// Parity was not established for this first form (the measurement came first).
//
// SECOND FORM (below the first): 32-channel units in a RING OF THREE buffers, the next unit's BN + ReLU interleaved between this unit's MFMAs (the
// conv3x3_wgrad_sc_kernel recipe), counted vmcnt waits, one barrier per unit, 128 VGPRs.  Parity-green
// (tests/test_hip_dense.py::test_small_map_conv2_forward_with_bn_prologue_matches_float64 at 4 x 128 x 128, 5 x 112 x 64, 3 x 144 x 160), and slower still:
//     block 1 forward 172.8 -> 184.9 us per layer, block 2 71.7 -> 76.2 (same box, same harness).
// Four units per tile re-request each 256-byte pixel row in 64-byte pieces a unit apart; the register-staged kernel's 128-byte pieces with two
// units in flight are what the memory system prefers here.

// =====================================================================================================================================
// dense_conv2_big_kernel (round 6): the same layer on the LARGE maps (blocks 1 / 2: at least one 16 x 16 pixel tile per CU).  The resident 3x3
// kernel of conv_tile.hip stages every (tile, 64-channel block) unit through registers (global -> VGPR -> BN + ReLU -> LDS, two barriers per
// unit, the transform serial in front of the MFMAs): 22k cycles per tile where HBM needs 11k and the LDS fragment reads 9k
// (profiles/r06_step_pmc_summary.txt: 2.7 TB/s of real traffic).  Here:
//   * one persistent 8-wave workgroup per CU; all 32 x 9 x 128 weights arrive ONCE by LDS-DMA (72 KB, XOR layout of the small-map kernel);
//   * unit = (tile, channel half): the 18 x 18 x 64-channel halo (41 pieces of 1 KB: 8 pixels x 128 B) lands by LDS-DMA in one of TWO buffers
//     while the matrix cores work on the other; slot s of pixel hp holds chunk  s ^ (hp & 7)  -- 8 consecutive pixels of a fragment read cover
//     the 8 bank groups, and the chunk a lane delivers is the same in every piece ((lane & 7) ^ (lane >> 3 & 7));
//   * BN + ReLU (and the zero padding of the ACTIVATED tensor) in place in the LDS by the wave that requested the piece, one barrier per unit;
//   * the product is taken TRANSPOSED (A = weights, B = pixels): a lane ends up with channels of ONE pixel, permlane32_swap makes them two
//     16-byte row pieces (dense_dgrad3_kernel's epilogue), no staging tile; the statistics of the 32 new channels stay in registers for the
//     workgroup's lifetime and are folded once.
constexpr int C2B_PIECES = (18 * 18 * 8 + 63) / 64;                       // 41
constexpr int C2B_BUF = C2B_PIECES * 1024;                                // 41984
constexpr int C2B_OFF_W = 2 * C2B_BUF;                                    // weights: [32][144] chunks, 73728 B
constexpr int C2B_OFF_PRO = C2B_OFF_W + C2_W_PIECES * 1024;               // float[2][128]
constexpr int C2B_OFF_SUM = C2B_OFF_PRO + 1024;                           // float[8 waves][2][32]
constexpr int C2B_LDS = C2B_OFF_SUM + 8 * 2 * 32 * 4;                     // 160768
static_assert(C2B_LDS <= 160 * 1024, "dense_conv2_big: LDS");

__global__ __launch_bounds__(512, 1) void dense_conv2_big_kernel(DenseConv2Args a)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    float* s_pro = (float*)(smem + C2B_OFF_PRO);
    const unsigned ntile = (unsigned)a.N * a.tiles_x * a.tiles_y;
    const int chl = (lane & 7) ^ ((lane >> 3) & 7);                       // the chunk (of the unit's 8) this lane delivers / transforms in every piece

    // halo requests of unit (tile t, half h) into buffer b; piece = wave + 8 j
    auto issue = [&](unsigned t, int h, int b) {
        const unsigned txi = t % a.tiles_x, r1 = t / a.tiles_x, tyi = r1 % a.tiles_y, n = r1 / a.tiles_y;
        const unsigned char* img = (const unsigned char*)(a.z + (size_t)n * a.H * a.W * a.ldz);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int piece = wave + 8 * j;
            if (piece < C2B_PIECES) {
                const int hp = piece * 8 + (lane >> 3);
                const int hy = hp / 18, hx = hp - hy * 18;
                const int iy = (int)tyi * 16 + hy - 1, ix = (int)txi * 16 + hx - 1;
                const bool ok = hp < 324 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned char* src = ok ? img + ((size_t)(iy * a.W + ix) * a.ldz + (h * 8 + chl) * 8) * 2 : (const unsigned char*)g_c2_zeros + (lane & 3) * 16;
                mm_dma16(src, lds0 + b * C2B_BUF + piece * 1024);
            }
        }
    };
    unsigned t_req = blockIdx.x; int h_req = 0;                           // request cursor (one unit ahead of the multiply cursor)
    if (t_req < ntile) issue(t_req, 0, 0);
    // weights (once): as in dense_conv2_fwd_kernel
#pragma unroll
    for (int j = 0; j < C2_W_PIECES / 8; ++j) {
        const int q = (wave + 8 * j) * 64 + lane, row = q / 144, sl = q - row * 144;
        const int c = (sl & ~15) | ((sl & 15) ^ (row & 15));
        mm_dma16(a.w + (size_t)row * 1152 + c * 8, lds0 + C2B_OFF_W + (wave + 8 * j) * 1024);
    }
    bn_prologue_fill<512>(a.bnp, 128, 128, s_pro, blockIdx.x == 0);
    __syncthreads();

    float st1[16], st2[16];                                               // statistics of this lane's 16 channels (16 r + 8 lh + j), all its pixels
#pragma unroll
    for (int i = 0; i < 16; ++i) st1[i] = st2[i] = 0.f;
    f32x16 acc;
    const int prow = 2 * wave + (lr >> 4), pcol = lr & 15;               // this lane's pixel inside the tile
    const unsigned char* sw = smem + C2B_OFF_W + lr * (144 * 16);         // weight row co = lr
    int buf = 0;
    for (unsigned t = blockIdx.x; t < ntile; t += gridDim.x) {
        const unsigned txi = t % a.tiles_x, r1 = t / a.tiles_x, tyi = r1 % a.tiles_y, n = r1 / a.tiles_y;
#pragma unroll
        for (int h = 0; h < 2; ++h, buf ^= 1) {
            mm_wait_vm<0>();
            // ---- BN + ReLU in place on this wave's own pieces of the landed unit
            {
                float sc[8], sh[8];
                const int c0 = (h * 8 + chl) * 8;
                const f32x4 a0 = *(const f32x4*)(s_pro + c0), a1 = *(const f32x4*)(s_pro + c0 + 4);
                const f32x4 b0 = *(const f32x4*)(s_pro + 128 + c0), b1 = *(const f32x4*)(s_pro + 128 + c0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[e] = a0[e]; sc[4 + e] = a1[e]; sh[e] = b0[e]; sh[4 + e] = b1[e]; }
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int piece = wave + 8 * j;
                    if (piece < C2B_PIECES) {
                        const int hp = piece * 8 + (lane >> 3);
                        const int hy = hp / 18, hx = hp - hy * 18;
                        const int iy = (int)tyi * 16 + hy - 1, ix = (int)txi * 16 + hx - 1;
                        const bool ok = hp < 324 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                        unsigned char* q = smem + buf * C2B_BUF + piece * 1024 + lane * 16;
                        float f[8];
                        Vec16<u16>::unpack(*(const u32x4*)q, f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f);
                        if (ok) *(u32x4*)q = Vec16<u16>::pack(f);           // (padding pixels came from the zero page and stay zero)
                    }
                }
            }
            mm_barrier();                                                 // the unit is activated; everybody is done with the other buffer
            // ---- request the next unit into the other buffer
            {
                unsigned tn = t; int hn = h + 1;
                if (hn == 2) { hn = 0; tn = t + gridDim.x; }
                if (tn < ntile) issue(tn, hn, buf ^ 1);
            }
            if (h == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
            // ---- 9 taps x 4 k-steps: A = weights (row co = lr, k half lh), B = pixels (this lane's pixel shifted by the tap, k half lh)
            const unsigned char* hb = smem + buf * C2B_BUF;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int hp = (prow + tap / 3) * 18 + pcol + tap % 3;
                const unsigned char* hpx = hb + hp * 128;
                const int key = hp & 7;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x4 pf = *(const u32x4*)(hpx + (((2 * ks + lh) ^ key) << 4));
                    const int bc = tap * 16 + h * 8 + 2 * ks + lh;
                    const u32x4 wf = *(const u32x4*)(sw + (((bc & ~15) | ((bc & 15) ^ (lr & 15))) << 4));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, pf), acc, 0, 0, 0);
                }
            }
        }
        // ---- epilogue of the tile: channels of this lane's pixel as two 16-byte row pieces (16 r + 8 lh .. + 8), statistics in registers
        u16* yrow = a.y + ((size_t)(n * a.H + tyi * 16 + prow) * a.W + txi * 16 + pcol) * a.ldy + 8 * lh;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float G[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const auto swp = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * r + q]), __float_as_uint(acc[8 * r + 4 + q]), false, false);
                G[q] = __uint_as_float(swp[0]); G[4 + q] = __uint_as_float(swp[1]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { st1[8 * r + e] += G[e]; st2[8 * r + e] = fmaf(G[e], G[e], st2[8 * r + e]); }
            *(u32x4*)(yrow + 16 * r) = Vec16<u16>::pack(G);
        }
    }
    if (a.stat_sum != nullptr) {
        // lanes of equal lh hold the same channels: xor tree over the 32 pixel lanes, one slot per wave, fixed-order fold over the waves
#pragma unroll
        for (int o = 1; o < 32; o <<= 1)
#pragma unroll
            for (int i = 0; i < 16; ++i) { st1[i] += __shfl_xor(st1[i], o, 64); st2[i] += __shfl_xor(st2[i], o, 64); }
        float* s_sum = (float*)(smem + C2B_OFF_SUM);
        if (lr == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s_sum[(wave * 2) * 32 + 16 * r + 8 * lh + e] = st1[8 * r + e];
                    s_sum[(wave * 2 + 1) * 32 + 16 * r + 8 * lh + e] = st2[8 * r + e];
                }
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

bool dense_conv2_big_supported(const saunet_conv_desc* d, const void* x, const void* w, const void* y, const float* bias)
{
    static const bool on = ab_env_on("SAUNET_DENSE_CONV2_BIG");         // A/B switch (variant builds only)
    const long tiles = (long)d->N * (d->H / 16) * (d->W / 16);
    return on && d->dtype == SAUNET_BF16 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && !d->transposed && d->Cin == 128 && d->Cout == 32 &&
           d->ldx % 8 == 0 && d->ldy % 8 == 0 && d->H % 16 == 0 && d->W % 16 == 0 && d->Ho == d->H && d->Wo == d->W && bias == nullptr && !d->epi_relu &&
           d->pro_relu && tiles >= 256 && !(((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) && (long)d->N * d->H * d->W * (d->ldx > d->ldy ? d->ldx : d->ldy) < (1L << 31);
}

int dense_conv2_big_forward(const saunet_conv_desc* d, const void* x, const void* w, void* y, double* ssum, double* ssq, const saunet_bn_prologue* bnp,
                            hipStream_t st)
{
    DenseConv2Args a;
    a.z = (const u16*)x; a.ldz = d->ldx; a.w = (const u16*)w; a.y = (u16*)y; a.ldy = d->ldy;
    a.stat_sum = ssum; a.stat_sumsq = ssq; a.stat_replicas = d->stat_replicas > 1 ? d->stat_replicas : 1; a.stat_rstride = d->stat_rstride;
    a.N = d->N; a.H = d->H; a.W = d->W; a.tiles_x = d->W / 16; a.tiles_y = d->H / 16;
    a.bnp = *bnp;
    static DeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute((const void*)dense_conv2_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C2B_LDS);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0; hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        (void)hipGetLastError();
    }
    const long ntile = (long)a.N * a.tiles_x * a.tiles_y;
    hipLaunchKernelGGL(dense_conv2_big_kernel, dim3((unsigned)(ntile < cus ? ntile : cus)), dim3(512), C2B_LDS, st, a);
    SAUNET_CHECK_LAUNCH("dense_conv2_big_kernel");
    return SAUNET_OK;
}



// ===================================================================== SECOND FORM =====================================================================
// =====================================================================================================================================
// dense_conv2_big_v2_kernel (round 6): DenseNet conv2 (3x3, 128 -> 32) forward on the LARGE maps (blocks 1 / 2: at least one 16 x 16 pixel tile per
// CU).  The resident 3x3 kernel of conv_tile.hip runs its phases one after the other under two barriers per (tile, 64-channel) unit --
// s_memtime stamps at block 1 (profiles/r06_phase_timing_raw.txt conv2fwd): transform + LDS store 1.7k, load issue 1.1k, barriers 0.75k, the
// 36 MFMAs 2.4k (pipe-bound: two waves per SIMD x 32 cycles) of 6.1k cycles per unit -- so the matrix pipe idles 60 % of the time.  Here the
// sc-kernel recipe (conv3x3_wgrad_sc_kernel): one persistent 8-wave workgroup per CU, weights resident (one DMA), unit = (tile, 32 channels),
// halo (18 x 18 x 64 B = 21 pieces) by LDS-DMA into a RING OF THREE buffers: while the matrix cores work on unit u, unit u + 1 (landed) gets
// its BN + ReLU in place by the wave that requested the piece, BETWEEN the MFMAs, and unit u + 2 is in flight; one barrier per unit.
// Transposed product (A = weights, B = pixels), permlane32_swap epilogue (two 16-byte row pieces per lane), statistics in registers.
constexpr int C2V_PIECES = 21, C2V_BUF = C2V_PIECES * 1024;               // 18 x 18 pixels x 64 B
constexpr int C2V_OFF_W = 3 * C2V_BUF;                                    // weights: [32][144] chunks, 73728 B
constexpr int C2V_OFF_PRO = C2V_OFF_W + C2_W_PIECES * 1024;               // float[2][128]
constexpr int C2V_OFF_SUM = C2V_OFF_PRO + 1024;                           // float[8 waves][2][32]
constexpr int C2V_OFF_DUMMY = C2V_OFF_SUM + 8 * 2 * 32 * 4;               // 1 KB: landing zone of the padding requests
constexpr int C2V_LDS = C2V_OFF_DUMMY + 1024;
static_assert(C2V_OFF_DUMMY % 1024 == 0 && C2V_LDS <= 160 * 1024, "dense_conv2_big: LDS");

__global__ __launch_bounds__(512, 1) void dense_conv2_big_v2_kernel(DenseConv2Args a)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    float* s_pro = (float*)(smem + C2V_OFF_PRO);
    const unsigned ntile = (unsigned)a.N * a.tiles_x * a.tiles_y;
    const unsigned mytiles = blockIdx.x < ntile ? (ntile - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int nunits = (int)mytiles * 4;
    const int chl = (lane & 3) ^ ((lane >> 4) & 3);      // the 16-byte chunk (of the unit's 4) this lane delivers / transforms in EVERY piece: slot ^ ((hp >> 2) & 3)

    struct Pos { int y0, x0, n; const unsigned char* img; };              // tile origin (minus the halo), image index and base
    auto pos_of = [&](int u) {
        const unsigned t = blockIdx.x + (unsigned)(u >> 2) * gridDim.x;
        const unsigned txi = t % a.tiles_x, r1 = t / a.tiles_x, tyi = r1 % a.tiles_y, n = r1 / a.tiles_y;
        Pos p; p.y0 = (int)tyi * 16 - 1; p.x0 = (int)txi * 16 - 1; p.n = (int)n; p.img = (const unsigned char*)(a.z + (size_t)n * a.H * a.W * a.ldz);
        return p;
    };
    // every wave issues THREE requests per unit (pieces wave, wave + 8, wave + 16; beyond 20: a padding request): the counted wait relies on it
    auto issue = [&](int u, int b) {
        const Pos p = pos_of(u);
        const int q = u & 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int piece = wave + 8 * j;
            const int hp = piece * 16 + (lane >> 2);
            const int hy = hp / 18, hx = hp - hy * 18;
            const int iy = p.y0 + hy, ix = p.x0 + hx;
            const bool ok = piece < C2V_PIECES && hp < 324 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned char* src = ok ? p.img + ((size_t)(iy * a.W + ix) * a.ldz + q * 32 + chl * 8) * 2 : (const unsigned char*)g_c2_zeros + (lane & 3) * 16;
            mm_dma16(src, piece < C2V_PIECES ? lds0 + b * C2V_BUF + piece * 1024 : lds0 + C2V_OFF_DUMMY);
        }
    };
    // BN + ReLU in place on piece j of unit u (this wave's own request) in buffer b; padding pixels came from the zero page and stay zero
    auto transform = [&](int u, int b, int j) {
        const int piece = wave + 8 * j;
        if (piece >= C2V_PIECES) return;                                   // (wave-uniform)
        const Pos p = pos_of(u);
        const int c0 = (u & 3) * 32 + chl * 8;
        const f32x4 a0 = *(const f32x4*)(s_pro + c0), a1 = *(const f32x4*)(s_pro + c0 + 4);
        const f32x4 b0 = *(const f32x4*)(s_pro + 128 + c0), b1 = *(const f32x4*)(s_pro + 128 + c0 + 4);
        const int hp = piece * 16 + (lane >> 2);
        const int hy = hp / 18, hx = hp - hy * 18;
        const bool ok = hp < 324 && (unsigned)(p.y0 + hy) < (unsigned)a.H && (unsigned)(p.x0 + hx) < (unsigned)a.W;
        unsigned char* qp = smem + b * C2V_BUF + piece * 1024 + lane * 16;
        float f[8];
        Vec16<u16>::unpack(*(const u32x4*)qp, f);
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[e] = fmaxf(fmaf(f[e], a0[e], b0[e]), 0.f); f[4 + e] = fmaxf(fmaf(f[4 + e], a1[e], b1[e]), 0.f); }
        if (ok) *(u32x4*)qp = Vec16<u16>::pack(f);
    };

    if (nunits > 0) issue(0, 0);
    if (nunits > 1) issue(1, 1);
    // weights (once): as in dense_conv2_fwd_kernel
#pragma unroll
    for (int j = 0; j < C2_W_PIECES / 8; ++j) {
        const int q = (wave + 8 * j) * 64 + lane, row = q / 144, sl = q - row * 144;
        const int c = (sl & ~15) | ((sl & 15) ^ (row & 15));
        mm_dma16(a.w + (size_t)row * 1152 + c * 8, lds0 + C2V_OFF_W + (wave + 8 * j) * 1024);
    }
    bn_prologue_fill<512>(a.bnp, 128, 128, s_pro, blockIdx.x == 0);
    __syncthreads();
    mm_wait_vm<0>();
    if (nunits > 0) { transform(0, 0, 0); transform(0, 0, 1); transform(0, 0, 2); }
    mm_barrier();

    float st1[16], st2[16];                                               // statistics of this lane's 16 channels (16 r + 8 lh + j), all its pixels
#pragma unroll
    for (int i = 0; i < 16; ++i) st1[i] = st2[i] = 0.f;
    f32x16 acc;
    const int prow = 2 * wave + (lr >> 4), pcol = lr & 15;               // this lane's pixel inside the tile
    const unsigned char* sw = smem + C2V_OFF_W + lr * (144 * 16);         // weight row co = lr
    int b0 = 0, b1 = 1, b2 = 2;                                           // buffers of units u, u + 1, u + 2
    for (int u = 0; u < nunits; ++u) {
        const int q = u & 3;
        if (u + 2 < nunits) { issue(u + 2, b2); mm_wait_vm<3>(); }        // unit u + 1 (requested one unit ago) has landed; u + 2 is in flight
        else mm_wait_vm<0>();
        if (q == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
        const unsigned char* hb = smem + b0 * C2V_BUF;
        const bool more = u + 1 < nunits;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int hp = (prow + tap / 3) * 18 + pcol + tap % 3;
            const unsigned char* hpx = hb + hp * 64;
            const int key = (hp >> 2) & 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u32x4 pf = *(const u32x4*)(hpx + (((2 * ks + lh) ^ key) << 4));
                const int bc = tap * 16 + q * 4 + 2 * ks + lh;
                const u32x4 wf = *(const u32x4*)(sw + (((bc & ~15) | ((bc & 15) ^ (lr & 15))) << 4));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, pf), acc, 0, 0, 0);
            }
            if (more && tap % 3 == 1) transform(u + 1, b1, tap / 3);      // the next unit's activation between this unit's MFMAs
        }
        if (q == 3) {
            // ---- epilogue of the tile: channels of this lane's pixel as two 16-byte row pieces (16 r + 8 lh .. + 8), statistics in registers
            const Pos p = pos_of(u);
            u16* yrow = a.y + (((size_t)p.n * a.H + p.y0 + 1 + prow) * a.W + p.x0 + 1 + pcol) * a.ldy + 8 * lh;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float G[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const auto swp = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * r + k]), __float_as_uint(acc[8 * r + 4 + k]), false, false);
                    G[k] = __uint_as_float(swp[0]); G[4 + k] = __uint_as_float(swp[1]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { st1[8 * r + e] += G[e]; st2[8 * r + e] = fmaf(G[e], G[e], st2[8 * r + e]); }
                *(u32x4*)(yrow + 16 * r) = Vec16<u16>::pack(G);
            }
        }
        mm_barrier();                                                     // unit u + 1 is activated; everybody is done with buffer b0
        const int tb = b0; b0 = b1; b1 = b2; b2 = tb;
    }
    if (a.stat_sum != nullptr) {
        // lanes of equal lh hold the same channels: xor tree over the 32 pixel lanes, one slot per wave, fixed-order fold over the waves
#pragma unroll
        for (int o = 1; o < 32; o <<= 1)
#pragma unroll
            for (int i = 0; i < 16; ++i) { st1[i] += __shfl_xor(st1[i], o, 64); st2[i] += __shfl_xor(st2[i], o, 64); }
        float* s_sum = (float*)(smem + C2V_OFF_SUM);
        if (lr == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s_sum[(wave * 2) * 32 + 16 * r + 8 * lh + e] = st1[8 * r + e];
                    s_sum[(wave * 2 + 1) * 32 + 16 * r + 8 * lh + e] = st2[8 * r + e];
                }
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

bool dense_conv2_big_v2_supported(const saunet_conv_desc* d, const void* x, const void* w, const void* y, const float* bias)
{
    static const bool on = ab_env_on("SAUNET_DENSE_CONV2_BIG");         // A/B switch (variant builds only)
    const long tiles = (long)d->N * (d->H / 16) * (d->W / 16);
    return on && d->dtype == SAUNET_BF16 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && !d->transposed && d->Cin == 128 && d->Cout == 32 &&
           d->ldx % 8 == 0 && d->ldy % 8 == 0 && d->H % 16 == 0 && d->W % 16 == 0 && d->Ho == d->H && d->Wo == d->W && bias == nullptr && !d->epi_relu &&
           d->pro_relu && tiles >= 256 && !(((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) && (long)d->N * d->H * d->W * (d->ldx > d->ldy ? d->ldx : d->ldy) < (1L << 31);
}

int dense_conv2_big_v2_forward(const saunet_conv_desc* d, const void* x, const void* w, void* y, double* ssum, double* ssq, const saunet_bn_prologue* bnp,
                            hipStream_t st)
{
    DenseConv2Args a;
    a.z = (const u16*)x; a.ldz = d->ldx; a.w = (const u16*)w; a.y = (u16*)y; a.ldy = d->ldy;
    a.stat_sum = ssum; a.stat_sumsq = ssq; a.stat_replicas = d->stat_replicas > 1 ? d->stat_replicas : 1; a.stat_rstride = d->stat_rstride;
    a.N = d->N; a.H = d->H; a.W = d->W; a.tiles_x = d->W / 16; a.tiles_y = d->H / 16;
    a.bnp = *bnp;
    static DeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute((const void*)dense_conv2_big_v2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C2V_LDS);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0; hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        (void)hipGetLastError();
    }
    const long ntile = (long)a.N * a.tiles_x * a.tiles_y;
    hipLaunchKernelGGL(dense_conv2_big_v2_kernel, dim3((unsigned)(ntile < cus ? ntile : cus)), dim3(512), C2V_LDS, st, a);
    SAUNET_CHECK_LAUNCH("dense_conv2_big_v2_kernel");
    return SAUNET_OK;
}

