// probe: the SUSTAINED dense bf16 matrix-core rate of this MI355X under its power limit (VERDICT r5 item 5 ii: "power-limited at 1.2-1.3 PF/s"
// needs a committed measurement).  Nothing but v_mfma_f32_32x32x16_bf16 from registers: 8 waves per CU (two per SIMD), four independent
// accumulators per wave, operands loaded once.  Three operand fills -- zeros, random normal bf16, random with the sign / exponent spread of
// real activations x weights -- because the power of a matrix pipe depends on the bits that toggle.  Each fill runs ~10 s in launches of
// ~40 ms; per one-second window: TFLOP/s and the effective shader clock (s_memtime ticks of one wave / wall time of its launch).
// scripts/mfma_ceiling.sh samples rocm-smi (sclk, socket power) next to it.
// build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/probes/mfma_ceiling.hip -o scripts/probes/mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int ITERS = 20000;          // x 8 MFMAs per iteration and wave

__global__ __launch_bounds__(512, 1) void mfma_burn(const u32x4* __restrict__ ops, float* __restrict__ sink, unsigned long long* __restrict__ ticks)
{
    const int tid = threadIdx.x, lane = tid & 63;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = ops[(i * 2) * 64 + lane]; b[i] = ops[(i * 2 + 1) * 64 + lane]; }
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[(i + u) & 3]), __builtin_bit_cast(bf16x8_t, b[i]), c[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += c[i][r];
    if (s == 1234.5678f) sink[0] = s;
    if (blockIdx.x == 0 && tid == 0) ticks[0] = t1 - t0;
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float gauss() { float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX; return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("# %s, %d CUs, max sclk %d MHz; one 512-thread workgroup per CU, %d x 8 mfma_f32_32x32x16_bf16 per wave and launch\n", p.gcnArchName, cus, p.clockRate / 1000, ITERS);
    u32x4* d_ops; float* d_sink; unsigned long long* d_ticks;
    hipMalloc(&d_ops, 8 * 64 * 16); hipMalloc(&d_sink, 4); hipMalloc(&d_ticks, 8);
    uint16_t h[8 * 64 * 8];
    const char* names[3] = {"zeros", "random normal bf16", "activations x weights (|N(0,1)| relu-like x N(0, 0.05))"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop_per_launch = (double)cus * 8 * ITERS * 8 * 2.0 * 32 * 32 * 16;
    for (int fill = 0; fill < 3; ++fill) {
        srand(7);
        for (int i = 0; i < 8 * 64 * 8; ++i) {
            const bool is_b = ((i / (64 * 8)) & 1) != 0;
            float v = fill == 0 ? 0.f : fill == 1 ? gauss() : (is_b ? 0.05f * gauss() : fmaxf(gauss(), 0.f));
            h[i] = f2bf(v);
        }
        hipMemcpy(d_ops, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_burn, dim3(cus), dim3(512), 0, 0, d_ops, d_sink, d_ticks);
        hipDeviceSynchronize();
        printf("## fill: %s\n#  t[s]   TFLOP/s   cycle counter [GHz] (= sclk / 2)   ms/launch\n", names[fill]);
        const auto start = std::chrono::steady_clock::now();
        double win_flop = 0, win_ms = 0, win_ticks = 0, tot_flop = 0, tot_ms = 0; int win = 1; double lo = 1e30, hi = 0;
        for (;;) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_burn, dim3(cus), dim3(512), 0, 0, d_ops, d_sink, d_ticks);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t; hipMemcpy(&t, d_ticks, 8, hipMemcpyDeviceToHost);
            win_flop += flop_per_launch; win_ms += ms; win_ticks += (double)t; tot_flop += flop_per_launch; tot_ms += ms;
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
            if (el >= win) {
                const double tf = win_flop / (win_ms * 1e-3) / 1e12;
                printf("%6.1f   %8.1f   %6.3f   %8.3f\n", el, tf, win_ticks / (win_ms * 1e-3) / 1e9, win_ms / (win_flop / flop_per_launch));
                lo = tf < lo ? tf : lo; hi = tf > hi ? tf : hi;
                win_flop = win_ms = win_ticks = 0; ++win;
            }
            if (el >= seconds) break;
        }
        printf("# %s: mean %.1f TFLOP/s over %.1f s of kernel time (windows %.1f ... %.1f) = %.3f of the 2500 TFLOP/s dense bf16 peak\n", names[fill],
               tot_flop / (tot_ms * 1e-3) / 1e12, tot_ms * 1e-3, lo, hi, tot_flop / (tot_ms * 1e-3) / 1e12 / 2500.0);
    }
    return 0;
}
