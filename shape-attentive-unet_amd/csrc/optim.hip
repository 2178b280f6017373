// Multi-tensor optimiser / bucket kernels: one launch updates up to 96 parameter tensors.
// SGD with momentum + weight decay (torch.optim.SGD semantics, /root/reference/train.py:190-196),
// RAdam (/root/reference/radam.py:15-78) and gradient bucket pack/unpack for the RCCL all-reduce.
// Hyper-parameters come from a DEVICE array so a captured hipGraph can be replayed with a new
// learning rate (the host only rewrites that array).
#include "common.h"

namespace saunet {

// hyper (SGD):   [0] lr [1] momentum [2] weight_decay [3] first_step (1 -> buf = grad) [4] grad_scale
__global__ __launch_bounds__(256) void sgd_kernel(saunet_tensor_list tl, const float* __restrict__ hyper)
{
    const int t = blockIdx.y;
    float* p = (float*)tl.ptrs[0][t]; const float* g = (const float*)tl.ptrs[1][t]; float* m = (float*)tl.ptrs[2][t];
    const long n = tl.numel[t];
    const float lr = hyper[0], mom = hyper[1], wd = hyper[2], gs = hyper[4];
    const bool first = hyper[3] != 0.f;
    auto upd = [&](float pv, float gv, float mv, float& po, float& mo) {
        float d = gv * gs;
        if (wd != 0.f) d = fmaf(wd, pv, d);
        if (mom != 0.f) { const float b = first ? d : fmaf(mom, mv, d); mo = b; d = b; }
        po = fmaf(-lr, d, pv);
    };
    // 16-byte main loop when all three arrays allow it, scalar tail / fallback
    const bool vec = !(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m) & 15);
    const long n4 = vec ? n / 4 : 0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 pv = ((const f32x4*)p)[i], gv = ((const f32x4*)g)[i], mv = (mom != 0.f && !first) ? ((const f32x4*)m)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 po, mo = mv;
#pragma unroll
        for (int j = 0; j < 4; ++j) { float a, b = mv[j]; upd(pv[j], gv[j], mv[j], a, b); po[j] = a; mo[j] = b; }
        if (mom != 0.f) ((f32x4*)m)[i] = mo;
        ((f32x4*)p)[i] = po;
    }
    for (long i = n4 * 4 + blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float a, b = (mom != 0.f && !first) ? m[i] : 0.f;
        upd(p[i], g[i], b, a, b);
        if (mom != 0.f) m[i] = b;
        p[i] = a;
    }
}

// hyper (RAdam): [0] beta1 [1] beta2 [2] eps [3] wd*lr [4] step_size [5] rectified (N_sma >= 5) [6] grad_scale
__global__ __launch_bounds__(256) void radam_kernel(saunet_tensor_list tl, const float* __restrict__ hyper)
{
    const int t = blockIdx.y;
    float* p = (float*)tl.ptrs[0][t]; const float* g = (const float*)tl.ptrs[1][t];
    float* ea = (float*)tl.ptrs[2][t]; float* es = (float*)tl.ptrs[3][t];
    const long n = tl.numel[t];
    const float b1 = hyper[0], b2 = hyper[1], eps = hyper[2], wdlr = hyper[3], step = hyper[4], gs = hyper[6];
    const bool rect = hyper[5] != 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gr = g[i] * gs;
        const float v = b2 * es[i] + (1.f - b2) * gr * gr;
        const float m = b1 * ea[i] + (1.f - b1) * gr;
        es[i] = v; ea[i] = m;
        float w = p[i];
        if (wdlr != 0.f) w = fmaf(-wdlr, w, w);
        w = rect ? w - step * m / (sqrtf(v) + eps) : w - step * m;
        p[i] = w;
    }
}

// hyper (Adam):  [0] beta1 [1] beta2 [2] eps [3] weight_decay (L2, added to the gradient) [4] lr / (1 - beta1^t) [5] 1 / sqrt(1 - beta2^t)
//                [6] grad_scale     -- torch.optim.Adam (amsgrad=False) as /root/reference/train.py:197-201 builds it
__global__ __launch_bounds__(256) void adam_kernel(saunet_tensor_list tl, const float* __restrict__ hyper)
{
    const int t = blockIdx.y;
    float* p = (float*)tl.ptrs[0][t]; const float* g = (const float*)tl.ptrs[1][t];
    float* ea = (float*)tl.ptrs[2][t]; float* es = (float*)tl.ptrs[3][t];
    const long n = tl.numel[t];
    const float b1 = hyper[0], b2 = hyper[1], eps = hyper[2], wd = hyper[3], step = hyper[4], rbc2 = hyper[5], gs = hyper[6];
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float w = p[i];
        float gr = g[i] * gs;
        if (wd != 0.f) gr = fmaf(wd, w, gr);
        const float m = b1 * ea[i] + (1.f - b1) * gr;
        const float v = b2 * es[i] + (1.f - b2) * gr * gr;
        ea[i] = m; es[i] = v;
        p[i] = w - step * m / (sqrtf(v) * rbc2 + eps);
    }
}

__global__ __launch_bounds__(256) void bucket_copy_kernel(saunet_tensor_list tl, int pack, float scale)
{
    const int t = blockIdx.y;
    float* a = (float*)tl.ptrs[0][t]; float* b = (float*)tl.ptrs[1][t];
    const long n = tl.numel[t];
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (pack) b[i] = a[i]; else a[i] = b[i] * scale;
    }
}

}  // namespace saunet

using namespace saunet;

extern "C" {

int saunet_sgd_step(const saunet_tensor_list* tl, const float* hyper, void* stream)
{
    if (tl->count <= 0 || tl->count > 96) return set_error(SAUNET_BAD_SHAPE, "sgd: %d tensors", tl->count);
    long biggest = 1;
    for (int t = 0; t < tl->count; ++t) if (tl->numel[t] > biggest) biggest = tl->numel[t];
    long bx = (biggest / 4 + 511) / 512; if (bx > 2048) bx = 2048; if (bx < 1) bx = 1;      // ~2 vectors per thread for the largest tensor
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)bx, tl->count), dim3(256), 0, (hipStream_t)stream, *tl, hyper);
    SAUNET_CHECK_LAUNCH("sgd");
    return SAUNET_OK;
}

int saunet_radam_step(const saunet_tensor_list* tl, const float* hyper, void* stream)
{
    if (tl->count <= 0 || tl->count > 96) return set_error(SAUNET_BAD_SHAPE, "radam: %d tensors", tl->count);
    hipLaunchKernelGGL(radam_kernel, dim3(128, tl->count), dim3(256), 0, (hipStream_t)stream, *tl, hyper);
    SAUNET_CHECK_LAUNCH("radam");
    return SAUNET_OK;
}

int saunet_adam_step(const saunet_tensor_list* tl, const float* hyper, void* stream)
{
    if (tl->count <= 0 || tl->count > 96) return set_error(SAUNET_BAD_SHAPE, "adam: %d tensors", tl->count);
    hipLaunchKernelGGL(adam_kernel, dim3(128, tl->count), dim3(256), 0, (hipStream_t)stream, *tl, hyper);
    SAUNET_CHECK_LAUNCH("adam");
    return SAUNET_OK;
}

int saunet_bucket_copy(const saunet_tensor_list* tl, int pack, float scale, void* stream)
{
    if (tl->count <= 0 || tl->count > 96) return set_error(SAUNET_BAD_SHAPE, "bucket_copy: %d tensors", tl->count);
    hipLaunchKernelGGL(bucket_copy_kernel, dim3(128, tl->count), dim3(256), 0, (hipStream_t)stream, *tl, pack, scale);
    SAUNET_CHECK_LAUNCH("bucket_copy");
    return SAUNET_OK;
}

}  // extern "C"
