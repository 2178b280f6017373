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
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float d = g[i] * gs;
        if (wd != 0.f) d = fmaf(wd, p[i], d);
        if (mom != 0.f) {
            float b = first ? d : fmaf(mom, m[i], d);
            m[i] = b; d = b;
        }
        p[i] = fmaf(-lr, d, p[i]);
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
    hipLaunchKernelGGL(sgd_kernel, dim3(128, tl->count), dim3(256), 0, (hipStream_t)stream, *tl, hyper);
    SAUNET_CHECK_LAUNCH("sgd_step");
    return SAUNET_OK;
}

int saunet_radam_step(const saunet_tensor_list* tl, const float* hyper, void* stream)
{
    if (tl->count <= 0 || tl->count > 96) return set_error(SAUNET_BAD_SHAPE, "radam: %d tensors", tl->count);
    hipLaunchKernelGGL(radam_kernel, dim3(128, tl->count), dim3(256), 0, (hipStream_t)stream, *tl, hyper);
    SAUNET_CHECK_LAUNCH("radam_step");
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
