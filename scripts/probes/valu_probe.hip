// cycles per instruction of the building blocks of the dense data-gradient epilogue on gfx950, measured with s_memtime:
// plain VALU, DPP adds (quad_perm / row_mirror / row_bcast), v_permlane32_swap, LDS float atomics with 2 active lanes, v_cvt_pk_bf16.
// Each test runs REP x UNROLL instructions in one wave (and, second column, with a second wave busy on the same SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o valu_probe valu_probe.hip && ./valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int CTRL> __device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(moved);
}

constexpr int REP = 256;

template <int TEST> __global__ void probe(float* out, unsigned long long* cyc, int waves_per_simd)
{
    __shared__ float lds[512];
    lds[threadIdx.x & 511] = 0.f;
    __syncthreads();
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (TEST == 0) v[i] = fmaf(v[i], 1.0001f, 0.5f);                       // plain VALU, 8 independent chains
            if (TEST == 1) v[i] = dpp_add<0xB1>(v[i]);                             // quad_perm
            if (TEST == 2) v[i] = dpp_add<0x140>(v[i]);                            // row_mirror
            if (TEST == 3) v[i] = dpp_add<0x142>(v[i]);                            // row_bcast:15
            if (TEST == 4) {                                                       // permlane32 swap of a pair
                auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[(i + 1) & 7]), false, false);
                v[i] = __uint_as_float(sw[0]); v[(i + 1) & 7] = __uint_as_float(sw[1]);
            }
            if (TEST == 5) { if ((lane & 31) == 31) atomicAdd(&lds[(lane >> 5) * 8 + i], v[i]); }      // LDS float atomic, 2 active lanes
            if (TEST == 6) { atomicAdd(&lds[lane * 8 + i], v[i]); }                                    // LDS float atomic, all lanes, distinct addresses
            if (TEST == 7) v[i] = __shfl_xor(v[i], 16, 64) + v[i];                                     // ds_bpermute / swizzle path
            if (TEST == 8) { v[i] = dpp_add<0xB1>(v[i]); v[i] = dpp_add<0x4E>(v[i]); v[i] = dpp_add<0x141>(v[i]); v[i] = dpp_add<0x140>(v[i]); v[i] = dpp_add<0x142>(v[i]); }  // the 5-step half-wave sum
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x & 511];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int TEST> void run(const char* name, int per_iter)
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    for (int w = 1; w <= 2; ++w) {
        // w waves per SIMD on ONE CU: 4*w waves in one block
        hipLaunchKernelGGL(probe<TEST>, dim3(1), dim3(256 * w), 0, 0, out, cyc, w);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(probe<TEST>, dim3(1), dim3(256 * w), 0, 0, out, cyc, w);
        hipDeviceSynchronize();
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-44s %d wave/SIMD: %7.2f cycles per instruction (%d per iteration)\n", name, w, (double)c / (REP * 8.0 * per_iter), per_iter);
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("v_fma_f32 (8 independent chains)", 1);
    run<1>("v_add_f32 dpp quad_perm", 1);
    run<2>("v_add_f32 dpp row_mirror", 1);
    run<3>("v_add_f32 dpp row_bcast:15", 1);
    run<4>("v_permlane32_swap (pair)", 1);
    run<5>("ds_add_f32, 2 active lanes", 1);
    run<6>("ds_add_f32, 64 lanes distinct", 1);
    run<7>("shfl_xor 16 + add", 1);
    run<8>("5-step DPP half-wave sum (per DPP add)", 5);
    return 0;
}
