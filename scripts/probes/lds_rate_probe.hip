// probe: LDS fragment-read throughput per CU on gfx950 -- ds_read_b128 vs ds_read_b64 vs the transposing ds_read_b64_tr_b16, with the address
// pattern of the tiled weight-gradient kernel (conv_tile.hip tr_read2: 16-lane groups, 4 rows x 16 columns, row pitch 192 B).
// build: hipcc --offload-arch=gfx950 -O3 scripts/probes/lds_rate_probe.hip -o scripts/probes/lds_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
constexpr int ITERS = 4096, UNROLL = 8;

template <int MODE> __global__ __launch_bounds__(256) void k(unsigned* out, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 40 * 1024 / 4; i += 256) ((unsigned*)lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4, nhalf = lg & 1, khalf = lg >> 1;
    unsigned acc = 0;
    // per-wave base: rows 8*khalf + (li >> 2) (+4 for the second read), columns 16*nhalf + 4*(li & 3) elements
    const unsigned char* base = lds + wave * 16 * pitch + (8 * khalf + (li >> 2)) * pitch + (16 * nhalf + 4 * (li & 3)) * 2;
    const unsigned char* base128 = lds + wave * 16 * pitch + (lane & 31) * pitch + (lane >> 5) * 16;     // a plain 16-byte fragment read: row = lane % 32
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int off = ((it + u) & 7) * 64 * pitch / 8;       // walk a few tiles so that addresses change
            if constexpr (MODE == 0) {
                u32x4 v = *(const u32x4*)(base128 + off);
                acc += v[0] ^ v[3];
            } else if constexpr (MODE == 1) {
                u32x2 v = *(const u32x2*)(base + off);
                acc += v[0] ^ v[1];
            } else {
                s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off));
                acc += (unsigned)v[0] ^ (unsigned)v[3];
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE> static void run(const char* name, int bytes_per_lane, int blocks_per_cu, int pitch)
{
    unsigned* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 40 * 1024, 0, d, pitch);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 40 * 1024, 0, d, pitch);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)ITERS * UNROLL * 4 * blocks_per_cu;       // wave instructions per CU
    const double bytes_per_cu = instr_per_cu * 64 * bytes_per_lane;
    printf("%-22s pitch %3d  %d block(s)/CU: %7.3f ms  %6.2f ns per wave instruction per CU  %7.1f B/ns per CU (%.0f B/clk at 2.1 GHz)\n", name, pitch, blocks_per_cu, ms,
           ms * 1e6 / instr_per_cu, bytes_per_cu / (ms * 1e6), bytes_per_cu / (ms * 1e6) / 2.1);
    hipFree(d);
}

int main()
{
    for (int pitch : {192, 64}) {
        for (int b : {1, 2}) {
            run<0>("ds_read_b128", 16, b, pitch);
            run<1>("ds_read_b64", 8, b, pitch);
            run<2>("ds_read_b64_tr_b16", 8, b, pitch);
        }
    }
    return 0;
}
