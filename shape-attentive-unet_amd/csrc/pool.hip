// Selectable global pooling: avg / max / avgmax / avgmaxc over H x W of an NHWC activation in ONE pass.
// Replaces adaptive_avgmax_pool2d / AdaptiveAvgMaxPool2d (/root/reference/models/adaptive_avgmax_pool.py:19-40, 43-74;
// never imported by the reference's train.py -- SURVEY.md section 8 row A16) and nn.AdaptiveAvgPool2d(1) of SEModule
// (/root/reference/models/attention_blocks.py:32,50).
//
// Every block reduces a run of pixel rows of one image for all channels (16-byte channel chunks per lane, so a wave reads whole
// 128-byte row segments), keeping per-channel (sum, max, first argmax) in registers; the row lanes of the block are combined
// through LDS and one partial per (image, split) goes to the caller's workspace; a tiny second kernel folds the splits in
// ascending pixel order (max ties resolve to the FIRST pixel, like F.max_pool2d) and writes the pooled vector(s).
#include "common.h"

namespace saunet {

struct PoolPartial { float sum, max; int idx; };

template <typename T, int V>
__global__ __launch_bounds__(256) void global_pool_partial_kernel(const T* __restrict__ x, int HW, int C, int ld, int rows_per_block,
                                                                  float* __restrict__ ws_sum, float* __restrict__ ws_max, int* __restrict__ ws_idx)
{
    extern __shared__ unsigned char pool_smem[];
    const int n = blockIdx.x, split = blockIdx.y, splits = gridDim.y;
    const int r0 = split * rows_per_block, r1 = min(r0 + rows_per_block, HW);
    const int CH = C / V;
    float* s_sum = (float*)pool_smem;              // [rl][C]
    float* s_max = s_sum + 256 * V;
    int* s_idx = (int*)(s_max + 256 * V);
    for (int cb = 0; cb < CH; cb += 256) {
        const int cw = min(256, CH - cb), rl = 256 / cw;
        const int ch = cb + threadIdx.x % cw, rr = threadIdx.x / cw;
        float s[V], m[V]; int ix[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { s[j] = 0.f; m[j] = -__builtin_inff(); ix[j] = 0x7fffffff; }
        if (rr < rl) {
            for (int r = r0 + rr; r < r1; r += rl) {
                float f[V];
                if constexpr (V == 1) f[0] = Elem<T>::load(x + ((long)n * HW + r) * ld + ch);
                else Vec16<T>::unpack(*(const u32x4*)(x + ((long)n * HW + r) * ld + ch * V), f);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    s[j] += f[j];
                    if (f[j] > m[j] || (f[j] != f[j] && !(m[j] != m[j]))) { m[j] = f[j]; ix[j] = r; }   // first maximum; NaN propagates like torch
                }
            }
        }
        __syncthreads();
        if (rr < rl) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const int o = (rr * cw + (ch - cb)) * V + j;
                s_sum[o] = s[j]; s_max[o] = m[j]; s_idx[o] = ix[j];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < cw * V; i += 256) {
            float ss = 0.f, mm = -__builtin_inff(); int ii = 0x7fffffff;
            for (int q = 0; q < rl; ++q) {
                const float sv = s_sum[q * cw * V + i], mv = s_max[q * cw * V + i]; const int iv = s_idx[q * cw * V + i];
                ss += sv;
                const bool better = mv > mm || (mv == mm && iv < ii) || (mv != mv && !(mm != mm));
                if (better) { mm = mv; ii = iv; }
            }
            const long o = ((long)n * splits + split) * C + cb * V + i;
            ws_sum[o] = ss; ws_max[o] = mm; ws_idx[o] = ii;
        }
    }
}

// mode: 0 avg, 1 max, 2 avgmax (0.5*(avg+max)), 3 avgmaxc ([avg | max], 2C outputs per image)
__global__ __launch_bounds__(256) void global_pool_final_kernel(const float* __restrict__ ws_sum, const float* __restrict__ ws_max,
                                                                const int* __restrict__ ws_idx, int N, int splits, int C, int HW, int mode,
                                                                float* __restrict__ out, int* __restrict__ argmax)
{
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= (long)N * C) return;
    const int n = (int)(i / C), c = (int)(i - (long)n * C);
    float ss = 0.f, mm = -__builtin_inff(); int ii = 0x7fffffff;
    for (int s = 0; s < splits; ++s) {
        const long o = ((long)n * splits + s) * C + c;
        ss += ws_sum[o];
        const float mv = ws_max[o]; const int iv = ws_idx[o];
        if (mv > mm || (mv == mm && iv < ii) || (mv != mv && !(mm != mm))) { mm = mv; ii = iv; }
    }
    const float avg = ss / (float)HW;
    if (mode == 0) out[i] = avg;
    else if (mode == 1) out[i] = mm;
    else if (mode == 2) out[i] = 0.5f * (avg + mm);
    else { out[(long)n * 2 * C + c] = avg; out[(long)n * 2 * C + C + c] = mm; }
    if (argmax) argmax[i] = ii;
}

// dx[n,p,c] = davg/HW + (p == argmax[n,c]) * dmax  with (davg, dmax) taken from dy according to the mode
template <typename T>
__global__ __launch_bounds__(256) void global_pool_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ argmax, int HW, int C, int mode,
                                                              T* __restrict__ dx, int lddx, long total)
{
    const float inv = 1.f / (float)HW;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / C; const int c = (int)(i - p * C); const int n = (int)(p / HW), r = (int)(p - (long)n * HW);
        float da = 0.f, dm = 0.f;
        if (mode == 0) da = dy[(long)n * C + c];
        else if (mode == 1) dm = dy[(long)n * C + c];
        else if (mode == 2) da = dm = 0.5f * dy[(long)n * C + c];
        else { da = dy[(long)n * 2 * C + c]; dm = dy[(long)n * 2 * C + C + c]; }
        float g = da * inv;
        if (mode != 0 && argmax[(long)n * C + c] == r) g += dm;
        Elem<T>::store(dx + p * lddx + c, g);
    }
}

static int pool_splits(int HW)
{
    int splits = (HW + 255) / 256; if (splits > 64) splits = 64; if (splits < 1) splits = 1;
    return splits;
}

}  // namespace saunet

using namespace saunet;

extern "C" {

int64_t saunet_global_pool_workspace(int N, int HW, int C)
{
    if (N <= 0 || HW <= 0 || C <= 0) return -1;
    return (int64_t)N * pool_splits(HW) * C * 12;
}

int saunet_global_pool_forward(int dtype, int mode, const void* x, int N, int HW, int C, int ldx, float* out, int* argmax,
                               void* workspace, int64_t workspace_bytes, void* stream)
{
    if (mode < 0 || mode > 3) return set_error(SAUNET_UNSUPPORTED, "global_pool: mode %d (0 avg, 1 max, 2 avgmax, 3 avgmaxc)", mode);
    if (N <= 0 || HW <= 0 || C <= 0 || ldx < C) return set_error(SAUNET_BAD_SHAPE, "global_pool: N=%d HW=%d C=%d ld=%d", N, HW, C, ldx);
    const int splits = pool_splits(HW);
    const int64_t need = (int64_t)N * splits * C * 12;
    if (!workspace || workspace_bytes < need) return set_error(SAUNET_BAD_SHAPE, "global_pool: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    int rpb = (HW + splits - 1) / splits;
    const int gs = (HW + rpb - 1) / rpb;      // == splits unless HW is tiny; the final kernel folds `splits` slots, so keep them all written
    if (gs != splits) rpb = (HW + splits - 1) / splits;
    float* ws_sum = (float*)workspace; float* ws_max = ws_sum + (size_t)N * splits * C; int* ws_idx = (int*)(ws_max + (size_t)N * splits * C);
    hipStream_t st = (hipStream_t)stream;
    const int epc = dtype == SAUNET_BF16 ? 8 : 4;
    const bool vec = C % epc == 0 && ldx % epc == 0 && ((uintptr_t)x & 15) == 0;
    const int V = vec ? epc : 1;
    const size_t lds = (size_t)256 * V * 12;
    if (dtype == SAUNET_BF16) {
        if (vec) hipLaunchKernelGGL((global_pool_partial_kernel<u16, 8>), dim3(N, splits), dim3(256), lds, st, (const u16*)x, HW, C, ldx, rpb, ws_sum, ws_max, ws_idx);
        else hipLaunchKernelGGL((global_pool_partial_kernel<u16, 1>), dim3(N, splits), dim3(256), lds, st, (const u16*)x, HW, C, ldx, rpb, ws_sum, ws_max, ws_idx);
    } else if (dtype == SAUNET_F32) {
        if (vec) hipLaunchKernelGGL((global_pool_partial_kernel<float, 4>), dim3(N, splits), dim3(256), lds, st, (const float*)x, HW, C, ldx, rpb, ws_sum, ws_max, ws_idx);
        else hipLaunchKernelGGL((global_pool_partial_kernel<float, 1>), dim3(N, splits), dim3(256), lds, st, (const float*)x, HW, C, ldx, rpb, ws_sum, ws_max, ws_idx);
    } else return set_error(SAUNET_BAD_DTYPE, "global_pool: dtype %d", dtype);
    hipLaunchKernelGGL(global_pool_final_kernel, dim3((unsigned)(((long)N * C + 255) / 256)), dim3(256), 0, st, ws_sum, ws_max, ws_idx, N, splits, C, HW, mode, out, argmax);
    SAUNET_CHECK_LAUNCH("global_pool_forward");
    return SAUNET_OK;
}

int saunet_global_pool_backward(int dtype, int mode, const float* dy, const int* argmax, int N, int HW, int C, void* dx, int lddx, void* stream)
{
    if (mode < 0 || mode > 3) return set_error(SAUNET_UNSUPPORTED, "global_pool: mode %d", mode);
    if (mode != 0 && !argmax) return set_error(SAUNET_BAD_SHAPE, "global_pool_backward: max modes need the forward argmax");
    const long total = (long)N * HW * C;
    long b = (total + 255) / 256; if (b > 8192) b = 8192; if (b < 1) b = 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SAUNET_BF16) hipLaunchKernelGGL(global_pool_bwd_kernel<u16>, dim3((unsigned)b), dim3(256), 0, st, dy, argmax, HW, C, mode, (u16*)dx, lddx, total);
    else if (dtype == SAUNET_F32) hipLaunchKernelGGL(global_pool_bwd_kernel<float>, dim3((unsigned)b), dim3(256), 0, st, dy, argmax, HW, C, mode, (float*)dx, lddx, total);
    else return set_error(SAUNET_BAD_DTYPE, "global_pool: dtype %d", dtype);
    SAUNET_CHECK_LAUNCH("global_pool_backward");
    return SAUNET_OK;
}

}  // extern "C"
