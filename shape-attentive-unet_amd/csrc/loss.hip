// Dual-task loss (weighted CE + soft Dice + BCE on the edge map) and the training metrics in ONE
// pass over logits [P,4] / edge [P,1]; backward in one more pass.
// Replaces DualLoss.forward (/root/reference/loss.py:149-159), dice_loss (:51-88) and
// SegmentationModuleBase.pixel_acc (/root/reference/models/models.py:51-74).
#include "common.h"

namespace saunet {

constexpr int NSUM = 23;
// sums: [0] sum w*nll [1] sum w [2..5] I_c [6..9] K_c [10] sum bce [11] acc_num [12] acc_den [13] unused
//       [14..16] |P_c & Y_c| c=1..3  [17..19] |Y_c|  [20..22] |P_c|
__constant__ float c_ce_w[4] = {1.f, 4.f, 5.f, 1.f};  // loss.py:130

__device__ __forceinline__ void softmax4(const float* z, float* p, float& m, float& logs)
{
    m = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3]));
    float e[4], s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { e[c] = expf(z[c] - m); s += e[c]; }
    logs = logf(s);
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < 4; ++c) p[c] = e[c] * inv;
}

template <typename T>
__global__ __launch_bounds__(256) void dual_loss_fwd_kernel(const T* __restrict__ logits, int ldl, const T* __restrict__ edge,
                                                            const int64_t* __restrict__ seg, const float* __restrict__ edge_t, long P,
                                                            double* __restrict__ sums)
{
    float acc[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) acc[k] = 0.f;
    for (long p = blockIdx.x * 256L + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
        float z[4], pr[4], m, logs;
#pragma unroll
        for (int c = 0; c < 4; ++c) z[c] = Elem<T>::load(logits + p * ldl + c);
        softmax4(z, pr, m, logs);
        const int y = (int)seg[p];
        float zy = 0.f, wy = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float t = (c == y) ? 1.f : 0.f;
            if (c == y) { zy = z[c]; wy = c_ce_w[c]; }
            acc[2 + c] += pr[c] * t;
            acc[6 + c] += pr[c] + t;
        }
        acc[0] += wy * (logs - (zy - m));
        acc[1] += wy;
        const float e = Elem<T>::load(edge + p), et = edge_t[p];
        acc[10] -= et * fmaxf(logf(e), -100.f) + (1.f - et) * fmaxf(logf(1.f - e), -100.f);
        // metrics: argmax(round(softmax)) -> the class with p > 0.5 (round-half-even: 0.5 -> 0), else 0
        int pred = 0;
#pragma unroll
        for (int c = 3; c >= 1; --c) if (pr[c] > 0.5f) pred = c;
        if (y >= 1) { acc[12] += 1.f; if (pred == y) acc[11] += 1.f; }
#pragma unroll
        for (int c = 1; c < 4; ++c) {
            const bool v = (y == c), q = (pred == c);
            if (v && q) acc[13 + c] += 1.f;
            if (v) acc[16 + c] += 1.f;
            if (q) acc[19 + c] += 1.f;
        }
    }
    __shared__ double red[4][NSUM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NSUM; ++k) {
        double v = wave_sum((double)acc[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NSUM) atomicAdd(&sums[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void dual_loss_finalize_kernel(const double* __restrict__ s, long P, float* __restrict__ loss, float* __restrict__ metrics)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float ce = (float)(s[0] / s[1]);
    float dsum = 0.f;
    for (int c = 0; c < 4; ++c) dsum += 2.f * (float)s[2 + c] / ((float)s[6 + c] + 1e-7f);
    const float dice = 1.f - dsum * 0.25f;
    const float bce = (float)(s[10] / (double)P);
    loss[0] = dice + ce + bce;
    if (metrics) {
        metrics[0] = (float)s[11] / ((float)s[12] + 1e-10f);
        for (int c = 1; c < 4; ++c) {
            float anb = (float)s[13 + c];
            metrics[c] = anb / ((float)s[16 + c] + (float)s[19 + c] - anb + 1e-10f);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dual_loss_bwd_kernel(const T* __restrict__ logits, int ldl, const T* __restrict__ edge,
                                                            const int64_t* __restrict__ seg, const float* __restrict__ edge_t, long P,
                                                            const double* __restrict__ sums, const float* __restrict__ dloss,
                                                            T* __restrict__ dlogits, int lddl, T* __restrict__ dedge)
{
    const float go = dloss ? dloss[0] : 1.f;
    const float inv_w = (float)(1.0 / sums[1]);
    float qa[4], qb[4];  // d dice / d p_c = -(1/4) * (2 t_c /(K_c+eps) - 2 I_c/(K_c+eps)^2) = qa*t_c + qb
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float K = (float)sums[6 + c] + 1e-7f, I = (float)sums[2 + c];
        qa[c] = -0.5f / K; qb[c] = 0.5f * I / (K * K);
    }
    const float inv_p = 1.f / (float)P;
    for (long p = blockIdx.x * 256L + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
        float z[4], pr[4], m, logs;
#pragma unroll
        for (int c = 0; c < 4; ++c) z[c] = Elem<T>::load(logits + p * ldl + c);
        softmax4(z, pr, m, logs);
        const int y = (int)seg[p];
        float wy = 0.f, q[4], dot = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c == y) wy = c_ce_w[c];
            q[c] = qb[c] + ((c == y) ? qa[c] : 0.f);
            dot = fmaf(pr[c], q[c], dot);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float g = inv_w * wy * (pr[c] - ((c == y) ? 1.f : 0.f)) + pr[c] * (q[c] - dot);
            Elem<T>::store(dlogits + p * lddl + c, g * go);
        }
        const float e = Elem<T>::load(edge + p), et = edge_t[p];
        Elem<T>::store(dedge + p, go * inv_p * (e - et) / fmaxf((1.f - e) * e, 1e-12f));
    }
}

// Inference head: probabilities = softmax over the C class logits of every pixel (float32, NHWC) and / or the predicted label
// = argmax (first maximum, like torch.argmax) -- SegmentationModule's test / inference branches (/root/reference/models/models.py:96-109:
// `torch.nn.functional.softmax(pred, dim=1)`) followed by train.py:47 / test_and_pack.py's argmax.
template <typename T, int C>
__global__ __launch_bounds__(256) void softmax_argmax_kernel(const T* __restrict__ logits, int ldl, long P, float* __restrict__ prob, int ldp,
                                                             int64_t* __restrict__ label)
{
    for (long p = blockIdx.x * 256L + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
        float z[C];
#pragma unroll
        for (int c = 0; c < C; ++c) z[c] = Elem<T>::load(logits + p * ldl + c);
        float m = z[0]; int am = 0;
#pragma unroll
        for (int c = 1; c < C; ++c) if (z[c] > m || (z[c] != z[c] && m == m)) { m = z[c]; am = c; }     // first maximum; a NaN wins (torch.argmax)
        if (prob) {
            float e[C], sum = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) { e[c] = expf(z[c] - m); sum += e[c]; }
            const float inv = 1.f / sum;
#pragma unroll
            for (int c = 0; c < C; ++c) prob[p * ldp + c] = e[c] * inv;
        }
        if (label) label[p] = am;
    }
}

// any class count (runtime loop, two passes over the pixel's logits): the num_class != 2 / 4 / 8 case of a drop-in user
template <typename T>
__global__ __launch_bounds__(256) void softmax_argmax_generic_kernel(const T* __restrict__ logits, int ldl, long P, int C, float* __restrict__ prob,
                                                                     int ldp, int64_t* __restrict__ label)
{
    for (long p = blockIdx.x * 256L + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
        const T* row = logits + p * ldl;
        float m = Elem<T>::load(row); int am = 0;
        for (int c = 1; c < C; ++c) { const float z = Elem<T>::load(row + c); if (z > m || (z != z && m == m)) { m = z; am = c; } }
        if (prob) {
            float sum = 0.f;
            for (int c = 0; c < C; ++c) sum += expf(Elem<T>::load(row + c) - m);
            const float inv = 1.f / sum;
            for (int c = 0; c < C; ++c) prob[p * ldp + c] = expf(Elem<T>::load(row + c) - m) * inv;
        }
        if (label) label[p] = am;
    }
}


// ---- SegmentationModuleBase.pixel_acc / .jaccard with the reference's own signatures (/root/reference/models/models.py:51-78): a prediction
// tensor and a label map in, ratios out.  Integer counts accumulate in 64-bit atomics (exact, hence order-independent and deterministic).
// kind: 0 float32, 1 bf16, 2 int64, 3 uint8 / bool
__device__ __forceinline__ float metric_load(const void* p, long i, int kind)
{
    if (kind == 0) return ((const float*)p)[i];
    if (kind == 1) return Elem<u16>::load((const u16*)p + i);
    if (kind == 2) return (float)((const int64_t*)p)[i];
    return (float)((const unsigned char*)p)[i];
}

constexpr int METRIC_MAXC = 16;

// counts: [0] sum valid * (argmax == label)  [1] sum valid  then per class c = 1 .. C-1: [2 + 3(c-1)] |P_c & Y_c|, [+1] |Y_c|, [+2] |P_c|
__global__ __launch_bounds__(256) void pixel_metrics_kernel(const void* __restrict__ pred, int kind, long sn, long sc, long sp,
                                                            const int64_t* __restrict__ label, long HW, long P, int C,
                                                            unsigned long long* __restrict__ counts)
{
    unsigned cnt[2 + 3 * (METRIC_MAXC - 1)];
    const int nc = 2 + 3 * (C - 1);
#pragma unroll
    for (int k = 0; k < 2 + 3 * (METRIC_MAXC - 1); ++k) cnt[k] = 0;
    for (long p = blockIdx.x * 256L + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
        const long n = p / HW, q = p - n * HW;
        const long base = n * sn + q * sp;
        float m = metric_load(pred, base, kind); int am = 0;
        for (int c = 1; c < C; ++c) { const float z = metric_load(pred, base + c * sc, kind); if (z > m) { m = z; am = c; } }   // torch.max: first maximum
        const long y = label[p];
        if (y >= 1) { cnt[1] += 1; if (am == y) cnt[0] += 1; }
#pragma unroll
        for (int c = 1; c < METRIC_MAXC; ++c) {
            if (c < C) {
                const bool v = (y == c), h = (am == c);
                cnt[2 + 3 * (c - 1)] += (v && h); cnt[3 + 3 * (c - 1)] += v; cnt[4 + 3 * (c - 1)] += h;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 2 + 3 * (METRIC_MAXC - 1); ++k) {
        if (k < nc) {
            unsigned v = cnt[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&counts[k], (unsigned long long)v);
        }
    }
}

__global__ void pixel_metrics_finalize_kernel(const unsigned long long* __restrict__ counts, int C, float* __restrict__ out)
{
    const int c = threadIdx.x;
    if (c == 0) out[0] = (float)counts[0] / ((float)counts[1] + 1e-10f);
    else if (c < C) {
        const float anb = (float)counts[2 + 3 * (c - 1)];
        const float j = anb / ((float)counts[3 + 3 * (c - 1)] + (float)counts[4 + 3 * (c - 1)] - anb + 1e-10f);
        out[c] = j <= 1.f ? j : 0.f;
    }
}

// sums: [0] sum (long(pred) & label)  [1] sum label  (int64);  psum: sum pred in float64 (the reference sums the ORIGINAL tensor)
__global__ __launch_bounds__(256) void binary_jaccard_kernel(const void* __restrict__ pred, int kind, const int64_t* __restrict__ label, long n,
                                                             long long* __restrict__ sums, double* __restrict__ psum)
{
    long long anb = 0, sl = 0; double sp = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float pv = metric_load(pred, i, kind);
        const long long pl = kind == 2 ? (long long)((const int64_t*)pred)[i] : (long long)pv;      // .long(): truncation toward zero
        const long long y = label[i];
        anb += pl & y; sl += y; sp += kind == 2 ? (double)pl : (double)pv;
    }
    for (int o = 32; o > 0; o >>= 1) { anb += __shfl_xor(anb, o); sl += __shfl_xor(sl, o); sp += __shfl_xor(sp, o); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long*)&sums[0], (unsigned long long)anb); atomicAdd((unsigned long long*)&sums[1], (unsigned long long)sl);
        atomicAdd(psum, sp);
    }
}

__global__ void binary_jaccard_finalize_kernel(const long long* __restrict__ sums, const double* __restrict__ psum, float* __restrict__ out)
{
    const float anb = (float)sums[0];
    out[0] = anb / ((float)psum[0] + (float)sums[1] - anb);
}

}  // namespace saunet

using namespace saunet;

extern "C" {

int saunet_dual_loss_forward(int dtype, const void* logits, int ldl, const void* edge, const int64_t* seg_t, const float* edge_t,
                             int64_t pixels, double* sums, void* stream)
{
    long blocks = (pixels + 255) / 256; if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SAUNET_F32) hipLaunchKernelGGL(dual_loss_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)logits, ldl, (const float*)edge, seg_t, edge_t, (long)pixels, sums);
    else if (dtype == SAUNET_BF16) hipLaunchKernelGGL(dual_loss_fwd_kernel<u16>, dim3((unsigned)blocks), dim3(256), 0, st, (const u16*)logits, ldl, (const u16*)edge, seg_t, edge_t, (long)pixels, sums);
    else return set_error(SAUNET_BAD_DTYPE, "dual_loss: dtype %d", dtype);
    SAUNET_CHECK_LAUNCH("dual_loss_forward");
    return SAUNET_OK;
}

int saunet_dual_loss_finalize(const double* sums, int64_t pixels, float* loss, float* metrics, void* stream)
{
    hipLaunchKernelGGL(dual_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, (long)pixels, loss, metrics);
    SAUNET_CHECK_LAUNCH("dual_loss_finalize");
    return SAUNET_OK;
}

int saunet_dual_loss_backward(int dtype, const void* logits, int ldl, const void* edge, const int64_t* seg_t, const float* edge_t,
                              int64_t pixels, const double* sums, const float* dloss, void* dlogits, int lddl, void* dedge, void* stream)
{
    long blocks = (pixels + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SAUNET_F32) hipLaunchKernelGGL(dual_loss_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)logits, ldl, (const float*)edge, seg_t, edge_t, (long)pixels, sums, dloss, (float*)dlogits, lddl, (float*)dedge);
    else if (dtype == SAUNET_BF16) hipLaunchKernelGGL(dual_loss_bwd_kernel<u16>, dim3((unsigned)blocks), dim3(256), 0, st, (const u16*)logits, ldl, (const u16*)edge, seg_t, edge_t, (long)pixels, sums, dloss, (u16*)dlogits, lddl, (u16*)dedge);
    else return set_error(SAUNET_BAD_DTYPE, "dual_loss: dtype %d", dtype);
    SAUNET_CHECK_LAUNCH("dual_loss_backward");
    return SAUNET_OK;
}

int saunet_softmax_argmax(int dtype, const void* logits, int ldl, int64_t pixels, int C, float* prob, int ldp, int64_t* label, void* stream)
{
    if (C < 1) return set_error(SAUNET_BAD_SHAPE, "softmax_argmax: %d classes", C);
    if (!prob && !label) return set_error(SAUNET_BAD_SHAPE, "softmax_argmax: no output requested");
    long b = (pixels + 255) / 256; if (b > 4096) b = 4096; if (b < 1) b = 1;
    hipStream_t st = (hipStream_t)stream;
    if (C != 2 && C != 4 && C != 8) {
        if (dtype == SAUNET_F32) hipLaunchKernelGGL(softmax_argmax_generic_kernel<float>, dim3((unsigned)b), dim3(256), 0, st, (const float*)logits, ldl, (long)pixels, C, prob, ldp, label);
        else if (dtype == SAUNET_BF16) hipLaunchKernelGGL(softmax_argmax_generic_kernel<u16>, dim3((unsigned)b), dim3(256), 0, st, (const u16*)logits, ldl, (long)pixels, C, prob, ldp, label);
        else return set_error(SAUNET_BAD_DTYPE, "softmax_argmax: dtype %d", dtype);
        SAUNET_CHECK_LAUNCH("softmax_argmax");
        return SAUNET_OK;
    }
#define SM(TT, CC) hipLaunchKernelGGL((softmax_argmax_kernel<TT, CC>), dim3((unsigned)b), dim3(256), 0, st, (const TT*)logits, ldl, (long)pixels, prob, ldp, label)
#define SMT(TT) do { if (C == 2) SM(TT, 2); else if (C == 4) SM(TT, 4); else SM(TT, 8); } while (0)
    if (dtype == SAUNET_F32) SMT(float);
    else if (dtype == SAUNET_BF16) SMT(u16);
    else return set_error(SAUNET_BAD_DTYPE, "softmax_argmax: dtype %d", dtype);
#undef SMT
#undef SM
    SAUNET_CHECK_LAUNCH("softmax_argmax");
    return SAUNET_OK;
}

int saunet_pixel_metrics(int kind, const void* pred, int64_t stride_n, int64_t stride_c, int64_t stride_p, const int64_t* label, int64_t N, int64_t HW,
                         int C, void* counts, float* out, void* stream)
{
    if (C < 1 || C > METRIC_MAXC) return set_error(SAUNET_BAD_SHAPE, "pixel_metrics: %d classes (1 .. %d)", C, METRIC_MAXC);
    if (kind < 0 || kind > 3) return set_error(SAUNET_BAD_DTYPE, "pixel_metrics: prediction kind %d", kind);
    if (!pred || !label || !counts || !out || N < 0 || HW < 1) return set_error(SAUNET_BAD_SHAPE, "pixel_metrics: incomplete arguments");
    const long P = (long)N * HW;
    long b = (P + 255) / 256; if (b > 1024) b = 1024; if (b < 1) b = 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pixel_metrics_kernel, dim3((unsigned)b), dim3(256), 0, st, pred, kind, (long)stride_n, (long)stride_c, (long)stride_p, label, (long)HW, P, C,
                       (unsigned long long*)counts);
    hipLaunchKernelGGL(pixel_metrics_finalize_kernel, dim3(1), dim3(64), 0, st, (const unsigned long long*)counts, C, out);
    SAUNET_CHECK_LAUNCH("pixel_metrics");
    return SAUNET_OK;
}

int saunet_binary_jaccard(int kind, const void* pred, const int64_t* label, int64_t n, void* sums, float* out, void* stream)
{
    if (kind < 0 || kind > 3) return set_error(SAUNET_BAD_DTYPE, "binary_jaccard: prediction kind %d", kind);
    if (!pred || !label || !sums || !out || n < 0) return set_error(SAUNET_BAD_SHAPE, "binary_jaccard: incomplete arguments");
    long b = (n + 255) / 256; if (b > 1024) b = 1024; if (b < 1) b = 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(binary_jaccard_kernel, dim3((unsigned)b), dim3(256), 0, st, pred, kind, label, (long)n, (long long*)sums, (double*)((long long*)sums + 2));
    hipLaunchKernelGGL(binary_jaccard_finalize_kernel, dim3(1), dim3(1), 0, st, (const long long*)sums, (const double*)((const long long*)sums + 2), out);
    SAUNET_CHECK_LAUNCH("binary_jaccard");
    return SAUNET_OK;
}

}  // extern "C"
