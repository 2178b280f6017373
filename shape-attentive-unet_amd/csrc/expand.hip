// Fused  Conv2d(1, C, 1x1) -> BatchNorm2d(C) -> ReLU  on a ONE-channel float32 map (SAUNet's `expand`, reference
// /root/reference/models/models.py:316,367: the sigmoid edge/canny fusion map broadcast to 32 decoder channels).
// With y_c = w_c*a + b_c every batch statistic of y is analytic in those of a:
//     mean_c = w_c*mu_a + b_c ,  var_c = w_c^2 * var_a ,  yhat_c = k_c*(a - mu_a) ,  k_c = w_c / sqrt(var_c + eps)
// so the layer is the per-pixel map  out_c = relu(A_c*a + B_c)  with  A_c = gamma_c*k_c ,  B_c = beta_c - A_c*mu_a,
// written once in the storage dtype (no float32 conv output, no separate BN / cast passes).  Backward is two passes
// over dout: per-channel  S1 = sum g ,  S2 = sum g*a  (g = dout*[out > 0]), then
//     da = sum_c A_c*g_c + D0 + D1*a   and the parameter gradients in closed form.
#include "common.h"

namespace saunet {

// coef layout [4][C]: A, B, k, scale(=gamma*invstd);  mv = {mu_a, var_a (biased)}
__global__ void expand_coeff_kernel(int C, const double* __restrict__ sum, const double* __restrict__ sumsq, int reps, int rstride, double count,
                                    const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float eps, float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
                                    float* __restrict__ coef, float* __restrict__ mv, int training)
{
    double mu = 0.0, var = 0.0;
    if (training) {
        double s = 0.0, q = 0.0;
        for (int r = 0; r < reps; ++r) { s += sum[(size_t)r * rstride]; q += sumsq[(size_t)r * rstride]; }
        mu = s / count; var = q / count - mu * mu; if (var < 0.0) var = 0.0;
    }
    if (threadIdx.x == 0) { mv[0] = (float)mu; mv[1] = (float)var; }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double wc = w[c], bc = b ? b[c] : 0.f;
        if (training) {
            const double mean_y = wc * mu + bc, var_y = wc * wc * var;
            const double invstd = 1.0 / sqrt(var_y + (double)eps);
            const double k = wc * invstd, sc = gamma[c] * invstd;
            coef[c] = (float)(sc * wc); coef[C + c] = (float)(beta[c] - sc * wc * mu); coef[2 * C + c] = (float)k; coef[3 * C + c] = (float)sc;
            if (rmean) rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean_y);
            if (rvar) rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * var_y * (count > 1.0 ? count / (count - 1.0) : 1.0));
        } else {
            const double sc = gamma[c] / sqrt((double)rvar[c] + (double)eps);
            coef[c] = (float)(sc * wc); coef[C + c] = (float)(beta[c] + sc * (bc - rmean[c])); coef[2 * C + c] = 0.f; coef[3 * C + c] = (float)sc;
        }
    }
}

// one thread per (pixel, 8-channel group)
template <typename T> __global__ __launch_bounds__(256)
void expand_fwd_kernel(const float* __restrict__ a, unsigned total, int CG, const float* __restrict__ coef, int C, T* __restrict__ y, int ldy, int relu)
{
    const float lo = relu ? 0.f : -__builtin_inff();
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned p = i / CG; const int c = (int)(i - p * CG) * 8;
        const float av = a[p];
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaxf(fmaf(coef[c + j], av, coef[C + c + j]), lo);
        T* q = y + (size_t)p * ldy + c;
        if constexpr (sizeof(T) == 2) *(u32x4*)q = Vec16<T>::pack(o);
        else { *(u32x4*)q = Vec16<T>::pack(o); *(u32x4*)(q + 4) = Vec16<T>::pack(o + 4); }
    }
}

template <typename T> __device__ __forceinline__ void load8v(const T* q, float* f)
{
    if constexpr (sizeof(T) == 2) Vec16<T>::unpack(*(const u32x4*)q, f);
    else { Vec16<T>::unpack(*(const u32x4*)q, f); Vec16<T>::unpack(*(const u32x4*)(q + 4), f + 4); }
}

// sums[rep][0][c] += sum_p g ,  sums[rep][1][c] += sum_p g*a ;  a thread keeps one 8-channel group and strides the pixels
template <typename T> __global__ __launch_bounds__(256)
void expand_bwd_reduce_kernel(const T* __restrict__ dy, int lddy, const float* __restrict__ a, unsigned P, int C, const float* __restrict__ coef, int relu,
                              double* __restrict__ sums, int reps, int rstride)
{
    extern __shared__ float e_red[];   // [2][C]
    for (int i = threadIdx.x; i < 2 * C; i += 256) e_red[i] = 0.f;
    __syncthreads();
    const int CG = C / 8, rl = 256 / CG;          // pixel lanes per block
    const int cg = threadIdx.x % CG, r0 = threadIdx.x / CG;
    if (r0 < rl) {
        const int c = cg * 8;
        float A[8], B[8], s1[8], s2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { A[j] = coef[c + j]; B[j] = coef[C + c + j]; s1[j] = 0.f; s2[j] = 0.f; }
        // four pixels per step: four independent 16-byte loads in flight per lane (one per step left the 16-step walk a chain of exposed latencies:
        // 1.4 TB/s); indices past the end re-read the last pixel and contribute zero
        const unsigned stride = gridDim.x * rl;
        for (unsigned p0 = blockIdx.x * rl + r0; p0 < P; p0 += 4 * stride) {
            float av[4], g[4][8]; bool live[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned p = p0 + u * stride; live[u] = p < P;
                const unsigned pp = live[u] ? p : P - 1;
                av[u] = a[pp];
                load8v(dy + (size_t)pp * lddy + c, g[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gj = (live[u] && (!relu || fmaf(A[j], av[u], B[j]) > 0.f)) ? g[u][j] : 0.f;
                    s1[j] += gj; s2[j] = fmaf(gj, av[u], s2[j]);
                }
        }
        // lanes of a wave that own the same channel group (lane % CG; CG = 4 or 8 divides 64): xor tree, then ONE LDS atomic per group and wave
        for (int off = CG; off < 64; off <<= 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[j] += __shfl_xor(s1[j], off, 64); s2[j] += __shfl_xor(s2[j], off, 64); }
        }
        if ((threadIdx.x & 63) < CG) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { atomicAdd(&e_red[c + j], s1[j]); atomicAdd(&e_red[C + c + j], s2[j]); }
        }
    }
    __syncthreads();
    const size_t ro = (size_t)(blockIdx.x % reps) * rstride;
    for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(&sums[ro + i], (double)e_red[i]);
}

// out: dw [C], db [C] (= 0), dgamma [C], dbeta [C], D = {D0, D1}
__global__ void expand_bwd_coeff_kernel(int C, const double* __restrict__ sums, int reps, int rstride, double count, const float* __restrict__ coef,
                                        const float* __restrict__ mv, float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dgamma,
                                        float* __restrict__ dbeta, float* __restrict__ D)
{
    __shared__ double s_d0, s_x;
    if (threadIdx.x == 0) { s_d0 = 0.0; s_x = 0.0; }
    __syncthreads();
    const double mu = mv[0], var = mv[1];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double S1 = 0.0, S2 = 0.0;
        for (int r = 0; r < reps; ++r) { S1 += sums[(size_t)r * rstride + c]; S2 += sums[(size_t)r * rstride + C + c]; }
        const double A = coef[c], k = coef[2 * C + c], sc = coef[3 * C + c];
        const double dg = k * (S2 - mu * S1);
        dgamma[c] = (float)dg; dbeta[c] = (float)S1;
        dw[c] = (float)(sc * (S2 - S1 * mu - dg * k * var));
        if (db) db[c] = 0.f;
        atomicAdd(&s_d0, A * S1 / count);
        atomicAdd(&s_x, A * k * dg / count);
    }
    __syncthreads();
    if (threadIdx.x == 0) { D[1] = (float)(-s_x); D[0] = (float)(-s_d0 + s_x * mu); }
}

// da[p] = sum_c A_c*g_c + D0 + D1*a ;  one thread per pixel
template <typename T, int C> __global__ __launch_bounds__(256)
void expand_bwd_apply_kernel(const T* __restrict__ dy, int lddy, const float* __restrict__ a, unsigned P, const float* __restrict__ coef, int relu,
                             const float* __restrict__ D, float* __restrict__ da)
{
    __shared__ float sA[C], sB[C];
    for (int i = threadIdx.x; i < C; i += 256) { sA[i] = coef[i]; sB[i] = coef[C + i]; }
    __syncthreads();
    const float D0 = D[0], D1 = D[1];
    for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < P; p += gridDim.x * 256u) {
        const float av = a[p];
        float s = fmaf(D1, av, D0);
#pragma unroll
        for (int cg = 0; cg < C / 8; ++cg) {
            float g[8];
            load8v(dy + (size_t)p * lddy + cg * 8, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float A = sA[cg * 8 + j];
                if (!relu || fmaf(A, av, sB[cg * 8 + j]) > 0.f) s = fmaf(A, g[j], s);
            }
        }
        da[p] = s;
    }
}

static int expand_check(const char* what, int dtype, int C, int64_t pixels, const void* t, int ld)
{
    if (dtype != SAUNET_F32 && dtype != SAUNET_BF16) return set_error(SAUNET_BAD_DTYPE, "%s: dtype %d", what, dtype);
    if (C <= 0 || C % 8 || C > 256) return set_error(SAUNET_UNSUPPORTED, "%s: C=%d (multiple of 8, <= 256)", what, C);
    const int epc = dtype == SAUNET_BF16 ? 8 : 4;
    if (ld % epc || ((uintptr_t)t & 15)) return set_error(SAUNET_BAD_ALIGN, "%s: rows must be 16-byte aligned", what);
    if (pixels <= 0 || pixels * (C / 8) >= (1LL << 32)) return set_error(SAUNET_BAD_SHAPE, "%s: %lld pixels", what, (long long)pixels);
    return SAUNET_OK;
}

}  // namespace saunet

using namespace saunet;

extern "C" {

int saunet_expand_coeff(int C, const double* sum, const double* sumsq, int replicas, int rstride, double count, const float* w, const float* b,
                        const float* gamma, const float* beta, float eps, float momentum, float* rmean, float* rvar, float* coef, float* mu_var,
                        int training, void* stream)
{
    if (!training && (rmean == nullptr || rvar == nullptr)) return set_error(SAUNET_BAD_SHAPE, "expand_coeff: eval mode needs running statistics");
    hipLaunchKernelGGL(expand_coeff_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, C, sum, sumsq, replicas > 0 ? replicas : 1, rstride, count, w, b, gamma, beta,
                       eps, momentum, rmean, rvar, coef, mu_var, training);
    SAUNET_CHECK_LAUNCH("expand_coeff");
    return SAUNET_OK;
}

int saunet_expand_forward(int dtype, const float* a, int64_t pixels, int C, const float* coef, void* y, int ldy, int relu, void* stream)
{
    if (int rc = expand_check("expand_forward", dtype, C, pixels, y, ldy)) return rc;
    const unsigned total = (unsigned)(pixels * (C / 8));
    long blocks = ((long)total + 255) / 256; if (blocks > 32768) blocks = 32768;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SAUNET_F32) hipLaunchKernelGGL(expand_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, a, total, C / 8, coef, C, (float*)y, ldy, relu);
    else hipLaunchKernelGGL(expand_fwd_kernel<u16>, dim3((unsigned)blocks), dim3(256), 0, st, a, total, C / 8, coef, C, (u16*)y, ldy, relu);
    SAUNET_CHECK_LAUNCH("expand_forward");
    return SAUNET_OK;
}

int saunet_expand_backward(int dtype, const void* dy, int lddy, const float* a, int64_t pixels, int C, const float* coef, const float* mu_var, int relu,
                           double* sums, int replicas, int rstride, float* dw, float* db, float* dgamma, float* dbeta, float* D, float* da, void* stream)
{
    if (int rc = expand_check("expand_backward", dtype, C, pixels, dy, lddy)) return rc;
    if (C != 32 && C != 64) return set_error(SAUNET_UNSUPPORTED, "expand_backward: C=%d (32 or 64)", C);
    hipStream_t st = (hipStream_t)stream;
    const int rl = 256 / (C / 8);
    long blocks = (pixels + rl - 1) / rl; if (blocks > 2048) blocks = 2048;
    const size_t lds = sizeof(float) * 2 * C;
    const int reps = replicas > 0 ? replicas : 1;
#define RED(TT) hipLaunchKernelGGL(expand_bwd_reduce_kernel<TT>, dim3((unsigned)blocks), dim3(256), lds, st, (const TT*)dy, lddy, a, (unsigned)pixels, C, coef, relu, sums, reps, rstride)
    if (dtype == SAUNET_F32) RED(float); else RED(u16);
#undef RED
    hipLaunchKernelGGL(expand_bwd_coeff_kernel, dim3(1), dim3(64), 0, st, C, sums, reps, rstride, (double)pixels, coef, mu_var, dw, db, dgamma, dbeta, D);
    long ab = (pixels + 255) / 256; if (ab > 8192) ab = 8192;
#define APP(TT, CC) hipLaunchKernelGGL((expand_bwd_apply_kernel<TT, CC>), dim3((unsigned)ab), dim3(256), 0, st, (const TT*)dy, lddy, a, (unsigned)pixels, coef, relu, D, da)
    if (dtype == SAUNET_F32) { if (C == 32) APP(float, 32); else APP(float, 64); }
    else { if (C == 32) APP(u16, 32); else APP(u16, 64); }
#undef APP
    SAUNET_CHECK_LAUNCH("expand_backward");
    return SAUNET_OK;
}

}  // extern "C"
