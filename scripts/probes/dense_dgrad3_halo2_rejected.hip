// REJECTED (round 5) -- kept as a record, not compiled into the library.
// Second form of the LDS-DMA staged conv2 data gradient for the large maps (blocks 1 / 2): non-transposed product (lane = channel epilogue,
// z1 / output through a wave-private LDS tile) and the deferred BN1 correction applied IN the halo buffer (x halo by LDS-DMA into a third
// buffer), replacing the separate bn_bwd_correct_ab pass.  Parity-green (tests/test_hip_dense.py, 49 passed) but slower on MI355X:
//   block 1 backward 444 -> 487 us / layer, block 2 170.6 -> 183.2 (scripts/dense_chain_micro.py, same box, variant build switch).
// Why (s_memtime stamps, scripts/phase_timing.py k3corr1): the kernel needs the 18 gradient fragments of the tile (72 VGPRs) alive through four
// channel tiles next to accumulators, the lane's 16 coefficients and the z1 prefetch; at two waves per SIMD that is the whole 256-register
// budget: 156-284 bytes of scratch per lane, and every spilled prefetch register turns its load into an immediate s_waitcnt vmcnt(0) -- which,
// behind 42 in-order DMA requests, is a full tile of memory latency (x pieces in registers: 14.5k cycles per tile in the request phase; z1 one
// pixel tile ahead: 23k).  With x by DMA and z1 one channel tile ahead the tile costs 23k cycles against 19.4k for the transposed kernel
// plus its correction pass: the epilogue is cheaper (1.26k vs 1.71k cycles per 32-channel tile) but the z1 pieces wait behind the halo requests.
// What would make it win: the four channel tiles dealt over four wave groups (one tile's weights in registers, as dense_dgrad3_cw_kernel does
// on the small maps) so that the fragments are the only long-lived registers.
// The code below is the kernel as measured (it needs dense_dgrad.hip's DenseDgrad3Args, D3_* constants and helpers to compile).

// ---- large maps, second form of the HALO kernel (round 5): the same 16 x 16 tiles, double-buffered LDS-DMA halo and LDS-resident weights, but
//   * the product is NOT transposed (operands swapped: rows = the wave's 32 pixels, columns = a 32-channel tile), so a lane owns one channel:
//     its mask / xhat coefficients are registers for the kernel's lifetime and the two BatchNorm sums a per-lane add -- the transposed epilogue
//     (coefficient vectors from LDS per element, permlane swaps, transposing DPP reductions) was 35 % of the tile; z1 and the output pass
//     through one wave-private LDS tile (80-byte pitch) as 16-byte row pieces;
//   * CORR: the deferred correction of the linear BN1 backward, g' = g - (A + B * xhat(x)), is applied IN the halo buffer by the lane whose DMA
//     request put the 16-byte slot there (so no barrier of its own), x pieces loaded next to the DMA requests and held in registers; the
//     corrected interior goes to gc for the weight gradient.  Replaces the separate bn_bwd_correct_ab pass (a launch and 192 B per pixel).
constexpr int D3H_TPITCH = 80;
template <bool CORR>
__global__ __launch_bounds__(512, 1) void dense_dgrad3_halo2_kernel(DenseDgrad3Args a)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char d_smem[];
    constexpr int OFF_X = 2 * D3_HALO_BYTES, OFF_W = OFF_X + (CORR ? D3_HALO_BYTES : 0), OFF_CC = OFF_W + 128 * D3_WPITCH * 2, OFF_TILE = OFF_CC + 64 * 4;
    u16* s_w = (u16*)(d_smem + OFF_W);                               // [128][D3_WPITCH]
    float* s_cc = (float*)(d_smem + OFF_CC);                         // CORR: [2][32]  g' = g - (cA + cB * x)
    TSTAMP_INIT();
    TSTAMP(50);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lr = lane & 31, lh = lane >> 5;
    unsigned char* tl = d_smem + OFF_TILE + wave * (32 * D3H_TPITCH);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)d_smem;
    const unsigned tilesX = (unsigned)a.W >> 4, tilesY = (unsigned)a.H >> 4, ntile = (unsigned)a.N * tilesX * tilesY;
    auto issue = [&](unsigned t, int buf) {
        const unsigned txi = t % tilesX, r1 = t / tilesX, tyi = r1 % tilesY, n = r1 / tilesY;
        const size_t img = (size_t)n * a.H * a.W;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int piece = wave + 8 * j;
            if (piece < D3_HALO_PIECES) {
                const int hp = piece * 16 + (lane >> 2), sl = lane & 3;
                const int hy = hp / 18, hx = hp - hy * 18;
                const int iy = (int)tyi * 16 + hy - 1, ix = (int)txi * 16 + hx - 1;
                const bool ok = hp < 324 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const int ch = sl ^ ((hp >> 2) & 3);
                const size_t pix = img + (size_t)(ok ? iy * a.W + ix : 0);
                const void* src = ok ? (const void*)(a.g + pix * a.ldg + ch * 8) : (const void*)((const unsigned char*)g_dg_zeros + sl * 16);
                mm_dma16(src, lds0 + buf * D3_HALO_BYTES + piece * 1024);
                // CORR: the activations of the same slots into the (single) x buffer: free again once every wave has corrected its slots, i.e.
                // after the barrier in front of the fragment reads -- which is where the next tile is requested
                if constexpr (CORR) mm_dma16(ok ? (const void*)(a.xc + pix * a.ldxc + ch * 8) : (const void*)((const unsigned char*)g_dg_zeros + sl * 16), lds0 + OFF_X + piece * 1024);
            }
        }
    };
    if (blockIdx.x < ntile) issue(blockIdx.x, 0);
    for (int i0 = threadIdx.x; i0 < 128 * 36; i0 += 512 * 9) {     // 9 loads in flight per thread
        u32x4 v[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int i = min(i0 + u * 512, 128 * 36 - 1);
            v[u] = *(const u32x4*)(a.w + (size_t)i * 8);
        }
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int i = i0 + u * 512, r = i / 36, ch = i - r * 36;
            if (i < 128 * 36) *(u32x4*)(s_w + r * D3_WPITCH + ch * 8) = v[u];
        }
    }
    if constexpr (CORR) {
        if (threadIdx.x < 32) {
            const int c = threadIdx.x;
            double A, B;
            rep_sum2(a.ab, a.ab + a.ab_half, a.ab_reps, a.ab_rstride, c, A, B);
            const float Af = (float)(A / a.count), Bf = (float)(B / a.count);
            s_cc[c] = fmaf(Bf, a.xt[c], Af); s_cc[32 + c] = Bf * a.xs[c];
        }
    }
    // this lane's channel in each of the four 32-channel tiles
    float sc[4], sh[4], a1[4], a0[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = 32 * t + lr;
        sc[t] = a.scale[c]; sh[t] = a.shift[c]; a1[t] = a.invstd[c]; a0[t] = -a.mean[c] * a1[t];
    }
    __syncthreads();
    float red1[4] = {0.f, 0.f, 0.f, 0.f}, red2[4] = {0.f, 0.f, 0.f, 0.f};
    // pixel of row q (0..31: two tile rows of 16) of this wave in tile tp
    auto row_pixel = [&](unsigned tp, int q) -> size_t {
        const unsigned txi = tp % tilesX, r1 = tp / tilesX, tyi = r1 % tilesY, n = r1 / tilesY;
        return ((size_t)(n * a.H + tyi * 16 + 2 * wave + (q >> 4)) * a.W) + txi * 16 + (q & 15);
    };
    // z1 pieces (32 rows x 16 pieces of 16 bytes, eight per lane) are requested one PIXEL tile ahead, in front of the next halo requests
    // (vmcnt returns in order: behind them they would wait for the whole halo)
    u32x4 zq[8];
    auto request_z = [&](unsigned t) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int q = lane + 64 * (i & 1); zq[i] = *(const u32x4*)(a.z + row_pixel(t, q >> 2) * a.ldz + 32 * (i >> 1) + (q & 3) * 8); }
    };
    if (blockIdx.x < ntile) request_z(blockIdx.x);
    int kbuf = 0;
    for (unsigned tp = blockIdx.x; tp < ntile; tp += gridDim.x) {
        TSTAMP(58);
        mm_wait_vm<0>();
        TSTAMP(59);
        unsigned char* hb = d_smem + kbuf * D3_HALO_BYTES;
        if constexpr (CORR) {
            const unsigned txi = tp % tilesX, r1 = tp / tilesX, tyi = r1 % tilesY, n = r1 / tilesY;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int piece = wave + 8 * j;
                if (piece < D3_HALO_PIECES) {
                    const int hp = piece * 16 + (lane >> 2), sl = lane & 3;
                    const int hy = hp / 18, hx = hp - hy * 18;
                    const int iy = (int)tyi * 16 + hy - 1, ix = (int)txi * 16 + hx - 1;
                    const bool ok = hp < 324 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    if (ok) {
                        const int ch = sl ^ ((hp >> 2) & 3);
                        u32x4* slot = (u32x4*)(hb + piece * 1024 + lane * 16);
                        float gv[8], xv[8];
                        Vec16<u16>::unpack(*slot, gv); Vec16<u16>::unpack(*(const u32x4*)(d_smem + OFF_X + piece * 1024 + lane * 16), xv);
                        const f32x4 ca0 = *(const f32x4*)(s_cc + ch * 8), ca1 = *(const f32x4*)(s_cc + ch * 8 + 4);
                        const f32x4 cb0 = *(const f32x4*)(s_cc + 32 + ch * 8), cb1 = *(const f32x4*)(s_cc + 32 + ch * 8 + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { gv[e] -= fmaf(cb0[e], xv[e], ca0[e]); gv[4 + e] -= fmaf(cb1[e], xv[4 + e], ca1[e]); }
                        const u32x4 v = Vec16<u16>::pack(gv);
                        *slot = v;
                        if (hy >= 1 && hy <= 16 && hx >= 1 && hx <= 16)      // the tile's own pixels: what the conv2 weight gradient reads later
                            *(u32x4*)(a.gc + ((size_t)n * a.H * a.W + (size_t)iy * a.W + ix) * a.ldgc + ch * 8) = v;
                    }
                }
            }
        }
        TSTAMP(60);
        mm_barrier();
        TSTAMP(51);
        u32x4 gf[18];       // A fragments: [tap][k half]  (k = 16*h + 8*lh .. +8 of the 32 gradient channels), row = this lane's pixel
        {
            int hp0 = (2 * wave + (lr >> 4)) * 18 + (lr & 15);
            asm volatile("" : "+v"(hp0));                       // opaque: 36 loop-invariant LDS offsets would otherwise live in registers (spills)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int hp = hp0 + (tap / 3) * 18 + tap % 3, key = (hp >> 2) & 3;
                gf[2 * tap] = *(const u32x4*)(hb + hp * 64 + ((lh ^ key) << 4));
                gf[2 * tap + 1] = *(const u32x4*)(hb + hp * 64 + (((2 + lh) ^ key) << 4));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned tn = tp + gridDim.x;
        u32x4 zc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) zc[i] = zq[i];
        if (tn < ntile) { request_z(tn); issue(tn, kbuf ^ 1); }
        kbuf ^= 1;
        TSTAMP(52);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ct = 32 * t;
            // this channel tile's z1 pieces into the wave's LDS tile
#pragma unroll
            for (int i = 0; i < 2; ++i) { const int q = lane + 64 * i; *(u32x4*)(tl + (q >> 2) * D3H_TPITCH + (q & 3) * 16) = zc[2 * t + i]; }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // opaque per-tile base: the weight addresses are loop-invariant and the compiler would hoist all 72 fragments out of the tile loop (spills)
            int wlane = ((ct + lr) * D3_WPITCH + lh * 8) * 2;
            asm volatile("" : "+v"(wlane));
            const u16* wrow = (const u16*)((const unsigned char*)s_w + wlane);
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) {
                const u32x4 wf = *(const u32x4*)(wrow + ks * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, gf[ks]), __builtin_bit_cast(bf16x8_t, wf), acc, 0, 0, 0);
            }
            u16 zs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) zs[r] = *(const u16*)(tl + ((r & 3) + 8 * (r >> 2) + 4 * lh) * D3H_TPITCH + lr * 2);
            TSTAMP(54);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float zf = __uint_as_float((unsigned)zs[r] << 16);
                const bool keep = !a.relu | (fmaf(zf, sc[t], sh[t]) > 0.f);
                const float Gv = keep ? acc[r] : 0.f;
                s1 += Gv; s2 = fmaf(Gv, fmaf(zf, a1[t], a0[t]), s2);
                *(u16*)(tl + row * D3H_TPITCH + lr * 2) = __builtin_bit_cast(u16, (__bf16)Gv);
            }
            red1[t] += s1; red2[t] += s2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = lane + 64 * i;
                *(u32x4*)(a.y + row_pixel(tp, q >> 2) * a.ldy + ct + (q & 3) * 8) = *(const u32x4*)(tl + (q >> 2) * D3H_TPITCH + (q & 3) * 16);
            }
            TSTAMP(55);
        }
    }
    TSTAMP(57);
    // fold: the two lane halves, then the eight waves through the (now idle) weight area
    __syncthreads();
    float* s_red = (float*)(d_smem + OFF_W);                     // [wave][tile][kind][32]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float r1 = red1[t] + __shfl_xor(red1[t], 32, 64), r2 = red2[t] + __shfl_xor(red2[t], 32, 64);
        if (lh == 0) { s_red[((wave * 4 + t) * 2) * 32 + lr] = r1; s_red[((wave * 4 + t) * 2 + 1) * 32 + lr] = r2; }
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int kind = threadIdx.x >> 7, c = threadIdx.x & 127, t = c >> 5, l = c & 31;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += s_red[((w * 4 + t) * 2 + kind) * 32 + l];
        const size_t ro = (size_t)(blockIdx.x % a.reps) * a.rstride;
        atomicAdd(&a.sums[ro + kind * 128 + c], (double)v);
    }
}

