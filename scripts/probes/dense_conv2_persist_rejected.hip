// REJECTED (round 5) -- kept as a record, not compiled into the library.
// VERDICT r4 item 5's open design for the DenseNet conv2 forward (3x3, 128 -> 32) on the large maps: persistent workgroups over 16 x 8 tiles,
// K split over wave PAIRS with the wave's 36 weight fragments (144 VGPRs) resident in registers, halo by double-buffered LDS-DMA with the
// BatchNorm + ReLU prologue in place.  Parity-green (tests/test_hip_dense.py incl. (20, 64, 64) and (5, 128, 48): 54 passed) but slower:
//   block 1 forward 169.6 -> 219.8 us / layer, block 2 69.0 -> 85.5 (scripts/dense_chain_micro.py, same box, variant switch).
// s_memtime stamps at 32 x 128 x 128 (scripts/phase_timing.py conv2fwdbn1), per 128-pixel tile 14.5k cycles against 7.5k of the resident kernel:
//   * matrix loop 4.0k for 36 MFMAs (1.15k of matrix-pipe time): 144 registers of weights + accumulators fill the 256-register budget (44 B of
//     scratch), so the A fragments are read from LDS just in time, one LDS latency per MFMA;
//   * in-place prologue 3.4k and halo requests 4.6k: per-piece tile decoding (three runtime divisions) on the vector ALU -- fixable (decode once
//     per tile on the scalar unit), but not enough: with both at their floor the tile is ~6k cycles, and
//   * the prologue is 25k cycles (36 fragment-shaped weight loads per lane: 64 separate 16-byte segments per instruction, + the BatchNorm
//     finalize): more than the whole kernel's budget at block 2 (4 tiles per workgroup).
// Lesson: register-resident weights need the fragments to be the only other long-lived registers AND a DMA-staged one-time weight load; the
// generic resident kernel (weights in LDS, two-stage register halo prefetch) stays.
// The code below is the kernel as measured (it needs dense_fwd.hip's DenseConv2Args, C2_HP, g_c2_zeros and helpers to compile).

// ---- the same convolution on the LARGE maps (blocks 1 / 2: 4096 / 1024 tiles of 16 x 8 pixels), persistent: VERDICT r4 item 5's open design.
// The resident generic kernel (conv3x3_res_fwd_kernel) keeps the weights in LDS, so every MFMA reads TWO fragments from LDS and -- with one
// 32-channel column tile -- uses each of them once: the LDS reads (1.15 MB per 256 pixels) are as long as the tile's HBM time, its halo goes through
// registers with a transform + commit phase of its own, and an 11 us weight / statistics prologue is half of the kernel at block 2.  Here:
//   * K is split over wave PAIRS (KQ = 2), so a wave's share of the weights is 36 fragments = 144 registers, loaded ONCE per workgroup
//     straight from global memory; the matrix loop reads only A fragments from LDS;
//   * the halo of the next tile arrives by LDS-DMA into the second buffer while this tile's MFMAs run; BatchNorm + ReLU in place by the
//     requesting wave; two barriers per tile; statistics stay in registers until the workgroup's last tile.
struct C2PLayout {
    static constexpr int TH = 16, PIX = 128, MT = 4, KQ = 2, KSTEPS = 36;
    static constexpr int HALO = (TH + 2) * C2_HP, HALO_PIECES = (HALO * 16 + 63) / 64, HPW = (HALO_PIECES + 7) / 8;   // 180 pixels, 45 pieces
    static constexpr int HALO_BYTES = HALO_PIECES * 1024;
    static constexpr int OFF_PRO = 2 * HALO_BYTES;              // float[2][128]
    static constexpr int OFF_RED = OFF_PRO + 1024;              // float[KQ][PIX][32] = 32 KB
    static constexpr int OFF_SUM = OFF_RED + KQ * PIX * 32 * 4; // float[8 waves][2][32]
    static constexpr int LDS = OFF_SUM + 8 * 2 * 32 * 4;
};

__global__ __launch_bounds__(512) void dense_conv2_persist_kernel(DenseConv2Args a)
{
    using LY = C2PLayout;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    float* s_pro = (float*)(smem + LY::OFF_PRO);
    const int ntile = a.N * a.tiles_x * a.tiles_y;
    const int wm = wave % LY::MT, wq = wave / LY::MT;
    TSTAMP_INIT();
    TSTAMP(70);
    // halo piece p covers halo pixels 4p .. 4p+3; lane l delivers (pixel 4p + (l >> 4), slot l & 15) = logical chunk slot ^ key(pixel) -- as above
    auto halo_slot = [&](int t, int j, int& c, const u16*& src) -> bool {
        const int txi = t % a.tiles_x, r1 = t / a.tiles_x, tyi = r1 % a.tiles_y, n = r1 / a.tiles_y;
        const int piece = wave + 8 * j;
        const int hp = piece * 4 + (lane >> 4), hy = hp / C2_HP, hx = hp - hy * C2_HP;
        c = (lane & 15) ^ ((hx & 3) | ((hy & 3) << 2));
        const int iy = tyi * LY::TH - 1 + hy, ix = txi * 8 - 1 + hx;
        const bool in = hp < LY::HALO && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        src = a.z + ((size_t)(n * a.H + (in ? iy : 0)) * a.W + (in ? ix : 0)) * a.ldz + c * 8;
        return in;
    };
    auto issue = [&](int t, int buf) {
#pragma unroll
        for (int j = 0; j < LY::HPW; ++j) {
            if (wave + 8 * j < LY::HALO_PIECES) {
                int c; const u16* src;
                const bool in = halo_slot(t, j, c, src);
                mm_dma16(in ? (const void*)src : (const void*)&g_c2_zeros[lane & 3], lds0 + buf * LY::HALO_BYTES + (wave + 8 * j) * 1024);
            }
        }
    };
    issue(blockIdx.x, 0);
    // this wave's half of the weights: B fragments of k-steps 36 wq .. (k-step = tap * 8 + 16-channel group), row co = lr
    u32x4 wr[LY::KSTEPS];
    {
        const u16* wrow = a.w + (size_t)lr * 1152 + lh * 8;
#pragma unroll
        for (int i = 0; i < LY::KSTEPS; ++i) wr[i] = *(const u32x4*)(wrow + (wq * LY::KSTEPS + i) * 16);
    }
    bn_prologue_fill<512>(a.bnp, 128, 128, s_pro, blockIdx.x == 0);
    __syncthreads();
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int c4 = (tid & 7) * 4;
    float* s_red = (float*)(smem + LY::OFF_RED);
    int it = 0;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x, ++it) {
        const int buf = it & 1;
        unsigned char* hb = smem + buf * LY::HALO_BYTES;
        TSTAMP(71);
        mm_wait_vm<0>();
        TSTAMP(72);
        // BN + ReLU in place on this wave's own halo pieces (padding pixels back to zero)
#pragma unroll
        for (int j = 0; j < LY::HPW; ++j) {
            if (wave + 8 * j < LY::HALO_PIECES) {
                int c; const u16* src;
                const bool in = halo_slot(t, j, c, src);
                unsigned char* q = hb + (wave + 8 * j) * 1024 + lane * 16;
                float f[8];
                Vec16<u16>::unpack(*(const u32x4*)q, f);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 sc = *(const f32x4*)(s_pro + c * 8 + 4 * h), sh = *(const f32x4*)(s_pro + 128 + c * 8 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[4 * h + e] = in ? fmaxf(fmaf(f[4 * h + e], sc[e], sh[e]), 0.f) : 0.f;
                }
                *(u32x4*)q = Vec16<u16>::pack(f);
            }
        }
        TSTAMP(73);
        mm_barrier();                                             // every wave is past its reads of the other buffer (the previous tile's product)
        TSTAMP(74);
        if (t + (int)gridDim.x < ntile) issue(t + gridDim.x, buf ^ 1);
        TSTAMP(75);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            int py = wm * 4 + (lr >> 3), px = lr & 7;
            asm volatile("" : "+v"(py), "+v"(px));               // opaque: 36 loop-invariant fragment offsets would otherwise be kept in registers (spills)
#pragma unroll
            for (int i = 0; i < LY::KSTEPS; ++i) {
                const int ks = wq * LY::KSTEPS + i, tap = ks >> 3, cg = ks & 7, kh = tap / 3, kw = tap - kh * 3;
                const int hy = py + kh, hx = px + kw, hp = hy * C2_HP + hx;
                const u32x4 af = *(const u32x4*)(hb + hp * 256 + (((2 * cg + lh) ^ ((hx & 3) | ((hy & 3) << 2))) << 4));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, wr[i]), acc, 0, 0, 0);
            }
        }
        TSTAMP(76);
        // the two K halves through the LDS: partial [q][pixel][channel] float
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            s_red[(wq * LY::PIX + row) * 32 + lr] = acc[r];
        }
        __syncthreads();
        TSTAMP(77);
        {
            const int txi = t % a.tiles_x, r1 = t / a.tiles_x, tyi = r1 % a.tiles_y, n = r1 / a.tiles_y;
#pragma unroll
            for (int k = 0; k < LY::PIX / 64; ++k) {
                const int prow = (tid >> 3) + k * 64;
                f32x4 v = *(const f32x4*)(s_red + prow * 32 + c4);
                const f32x4 u = *(const f32x4*)(s_red + (LY::PIX + prow) * 32 + c4);
                v += u;
                const int oy = tyi * LY::TH + (prow >> 3), ox = txi * 8 + (prow & 7);
                u16* yo = a.y + ((size_t)(n * a.H + oy) * a.W + ox) * a.ldy + c4;
                *(uint2*)yo = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
            }
        }
        TSTAMP(78);
    }
    if (a.stat_sum != nullptr) {
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

