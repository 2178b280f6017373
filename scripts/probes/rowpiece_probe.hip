// How fast can a wave stream [pixel][channel] rows when each LANE owns one pixel row and touches it in 8-byte / 16-byte pieces
// (the MFMA accumulator layout of a transposed product) versus fully coalesced 16-byte-per-lane rows?
// build: hipcc --offload-arch=gfx950 -O3 rowpiece_probe.hip -o rowpiece_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// mode 0: coalesced: consecutive lanes read consecutive 16 B (row-major), RMW y += x
// mode 1: lane = pixel (32 per half-wave), 8-byte pieces: piece j at channel 4*lh + 8*j, 64-channel groups (8 pieces per lane)
// mode 2: lane = pixel, 16-byte pieces: piece j at channel 8*lh + 16*j (4 pieces per lane per 64 channels)
template <int MODE> __global__ __launch_bounds__(256) void k(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, unsigned P, int C, int ld)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
    if (MODE == 0) {
        const unsigned chunks = P * (C / 8);
        for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < chunks; i += gridDim.x * 256u) {
            const unsigned p = i / (C / 8), c = (i % (C / 8)) * 8;
            u32x4 a = *(const u32x4*)(x + (size_t)p * ld + c), b = *(const u32x4*)(y + (size_t)p * ld + c);
            b += a;
            *(u32x4*)(y + (size_t)p * ld + c) = b;
        }
    } else {
        const unsigned ntp = P / 32;
        for (unsigned tp = blockIdx.x * 4u + wave; tp < ntp; tp += gridDim.x * 4u) {
            const size_t p = tp * 32u + lr;
            for (int c0 = 0; c0 < C; c0 += 64) {
                if (MODE == 1) {
                    uint2 a[8], b[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { a[j] = *(const uint2*)(x + p * ld + c0 + 4 * lh + 8 * j); b[j] = *(const uint2*)(y + p * ld + c0 + 4 * lh + 8 * j); }
#pragma unroll
                    for (int j = 0; j < 8; ++j) { b[j].x += a[j].x; b[j].y += a[j].y; *(uint2*)(y + p * ld + c0 + 4 * lh + 8 * j) = b[j]; }
                } else {
                    u32x4 a[4], b[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[j] = *(const u32x4*)(x + p * ld + c0 + 8 * lh + 16 * j); b[j] = *(const u32x4*)(y + p * ld + c0 + 8 * lh + 16 * j); }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { b[j] += a[j]; *(u32x4*)(y + p * ld + c0 + 8 * lh + 16 * j) = b[j]; }
                }
            }
        }
    }
}

template <int MODE> float run(const uint16_t* x, uint16_t* y, unsigned P, int C, int ld, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, x, y, P, C, ld);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, x, y, P, C, ld);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main()
{
    const unsigned P = 524288; const int C = 192, ld = 256;
    uint16_t *x, *y;
    hipMalloc(&x, (size_t)P * ld * 2); hipMalloc(&y, (size_t)P * ld * 2);
    hipMemset(x, 0, (size_t)P * ld * 2); hipMemset(y, 0, (size_t)P * ld * 2);
    const double bytes = 3.0 * P * C * 2;
    for (int blocks : {512, 1024, 2048, 4096}) {
        float t0 = run<0>(x, y, P, C, ld, blocks), t1 = run<1>(x, y, P, C, ld, blocks), t2 = run<2>(x, y, P, C, ld, blocks);
        printf("blocks %4d: coalesced %.1f us (%.2f TB/s) | 8B pieces %.1f us (%.2f TB/s) | 16B pieces %.1f us (%.2f TB/s)\n", blocks,
               t0 * 1e3, bytes / t0 / 1e9, t1 * 1e3, bytes / t1 / 1e9, t2 * 1e3, bytes / t2 / 1e9);
    }
    return 0;
}
