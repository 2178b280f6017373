// Cross-pixel outer-product sums on the matrix cores for kernels where one THREAD owns one PIXEL:
// every wave transposes its 64 pixels through LDS into [channel][pixel] bf16 tiles and multiplies tiles with
// mfma_f32_32x32x16_bf16 (K = pixels):  acc[i][j] += sum_p A[i][p] * B[j][p].
#pragma once
#include "common.h"

namespace saunet {

constexpr int GP = 72;            // LDS tile row pitch in bf16 elements: 64 pixels + 8 pad (144 B, 16-byte aligned rows)
constexpr int G_TILE = 32 * GP;   // a 32-row tile
constexpr int G_MISC = 8 * GP;    // an 8-row tile for leftover rows

__device__ __forceinline__ u16 to_bf16(float v) { return __builtin_bit_cast(u16, (__bf16)v); }

template <int C> __device__ __forceinline__ void load_row(const u16* __restrict__ p, float* f)
{
#pragma unroll
    for (int g = 0; g < C / 8; ++g) Vec16<u16>::unpack(*(const u32x4*)(p + 8 * g), f + 8 * g);
}
template <int C> __device__ __forceinline__ void store_row(u16* __restrict__ p, const float* f)
{
#pragma unroll
    for (int g = 0; g < C / 8; ++g) *(u32x4*)(p + 8 * g) = Vec16<u16>::pack(f + 8 * g);
}

// MFMA operand fragment: 8 consecutive pixels of row (lane & 31) of a [row][pixel] tile; rows >= nrows read as zero
__device__ __forceinline__ bf16x8_t tile_frag(const u16* tile, int lane, int ks, int nrows)
{
    const int r = lane & 31;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r < nrows) v = *(const u32x4*)(tile + r * GP + ks * 16 + (lane >> 5) * 8);
    return __builtin_bit_cast(bf16x8_t, v);
}
// acc[i][j] += sum over the wave's 64 pixels of A[i][p] * B[j][p]
__device__ __forceinline__ void tile_mma(const u16* ta, int ra, const u16* tb, int rb, int lane, f32x16& acc)
{
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tile_frag(ta, lane, ks, ra), tile_frag(tb, lane, ks, rb), acc, 0, 0, 0);
}
// add a wave's 32x32 accumulator into a float LDS tile
__device__ __forceinline__ void tile_flush(float* red, const f32x16& acc, int lane)
{
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(&red[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr], acc[r]);
}

}  // namespace saunet
