// BatchNorm (training + eval) as reduce / finalise / apply kernels for NHWC activations.
// Statistics are taken either in the producing convolution's epilogue (conv_igemm.hip) or by
// bn_stats_kernel below; the normalise(+ReLU) step is either folded into the consuming convolution's
// operand load (prologue) or materialised by affine_act_kernel.  Sums are float64 end to end.
// Replaces nn.BatchNorm2d / SynchronizedBatchNorm2d fwd+bwd (/root/reference/models/norm.py:16-22,
// lib/nn/modules/batchnorm.py:58-61).
#include "common.h"
#include <initializer_list>

namespace saunet {

// Work decomposition shared by the per-channel kernels: a block owns rows [p0,p1); channel chunks are
// processed in groups of <=256; inside a group thread -> (row lane, chunk) so a wave touches contiguous
// bytes.  V = elements per chunk (16-byte vectors when the view is aligned, else 1).
template <typename T, int V> struct ChunkIO {
    __device__ static __forceinline__ void load(const T* p, float* f)
    {
        if constexpr (V == 1) f[0] = Elem<T>::load(p);
        else Vec16<T>::unpack(*(const u32x4*)p, f);
    }
    __device__ static __forceinline__ void store(T* p, const float* f)
    {
        if constexpr (V == 1) Elem<T>::store(p, f[0]);
        else *(u32x4*)p = Vec16<T>::pack(f);
    }
};

template <typename T, int V>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, long P, int C, int ld, long rpb,
                                                       double* __restrict__ gsum, double* __restrict__ gsq, int reps, int rstride)
{
    extern __shared__ double s_red[];  // [2][C]
    for (int i = threadIdx.x; i < 2 * C; i += 256) s_red[i] = 0.0;
    __syncthreads();
    const long p0 = blockIdx.x * rpb, p1 = min(p0 + rpb, P);
    const int CH = C / V;
    for (int cb = 0; cb < CH; cb += 256) {
        const int cw = min(256, CH - cb), rl = 256 / cw;
        const int ch = cb + threadIdx.x % cw, r0 = threadIdx.x / cw;
        if (r0 >= rl) continue;
        double ds[V], dq[V];
        float fs[V], fq[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { ds[j] = dq[j] = 0.0; fs[j] = fq[j] = 0.f; }
        int cnt = 0;
        for (long p = p0 + r0; p < p1; p += rl) {
            float f[V];
            ChunkIO<T, V>::load(x + p * ld + ch * V, f);
#pragma unroll
            for (int j = 0; j < V; ++j) { fs[j] += f[j]; fq[j] = fmaf(f[j], f[j], fq[j]); }
            if (++cnt == 128) {
#pragma unroll
                for (int j = 0; j < V; ++j) { ds[j] += fs[j]; dq[j] += fq[j]; fs[j] = fq[j] = 0.f; }
                cnt = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            atomicAdd(&s_red[ch * V + j], ds[j] + (double)fs[j]);
            atomicAdd(&s_red[C + ch * V + j], dq[j] + (double)fq[j]);
        }
    }
    __syncthreads();
    const size_t ro = (size_t)(blockIdx.x % reps) * rstride;
    for (int c = threadIdx.x; c < C; c += 256) { atomicAdd(&gsum[ro + c], s_red[c]); atomicAdd(&gsq[ro + c], s_red[C + c]); }
}

__global__ void sum_replicas_kernel(double* __restrict__ base, int n, int reps, int rstride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = base[i];
    for (int r = 1; r < reps; ++r) s += base[(size_t)r * rstride + i];
    base[i] = s;
}

// xhat rows (saunet_bn_prologue.xhat) of C channels straight from their statistics
__global__ void bn_xhat_kernel(int C, const double* __restrict__ sum, const double* __restrict__ sq, int reps, int rstride, double count, float eps,
                               float* __restrict__ xhat, int ld)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double s1 = rep_sum(sum, reps, rstride, c), s2 = rep_sum(sq, reps, rstride, c);
    const double m = s1 / count;
    double v = s2 / count - m * m;
    if (v < 0.0) v = 0.0;
    const float mean = (float)m, is = (float)(1.0 / sqrt(v + (double)eps));
    xhat[c] = is; xhat[ld + c] = -mean * is; xhat[2 * ld + c] = mean; xhat[3 * ld + c] = is; xhat[4 * ld + c] = (float)v;
}

// the `writer` part of bn_prologue_fill on its own: used in front of convolution kernels that do not derive the coefficients themselves
__global__ __launch_bounds__(256) void bn_prologue_finalize_kernel(saunet_bn_prologue p, int Cin, float* __restrict__ scratch)
{
    // scratch: [2][cpad] never read (the fill wants a destination); chunked so any Cin works with one workgroup
    extern __shared__ float s_tmp[];
    bn_prologue_fill<256>(p, Cin, Cin, s_tmp, true);
}

__global__ void bn_finalize_kernel(int C, const double* __restrict__ sum, const double* __restrict__ sq, int reps, int rstride, double count,
                                   const float* __restrict__ cbias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_o,
                                   float* __restrict__ invstd_o, int training)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean, invstd;
    if (training) {
        const double s1 = rep_sum(sum, reps, rstride, c), s2 = rep_sum(sq, reps, rstride, c);
        double m = s1 / count;
        double var = s2 / count - m * m;
        if (var < 0.0) var = 0.0;
        if (cbias) m += (double)cbias[c];
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (rmean) {
            double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
        }
    } else {
        mean = rmean[c];
        invstd = 1.f / sqrtf(rvar[c] + eps);
    }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float s = g * invstd;
    scale[c] = s; shift[c] = b - mean * s;
    if (mean_o) mean_o[c] = mean;
    if (invstd_o) invstd_o[c] = invstd;
}

// SynchronizedBatchNorm2d in its multi-replica form (lib/nn/modules/batchnorm.py:118-139): statistics from the GLOBAL
// (all-reduced) sums, inv_std = clamp(biased var, eps)^-1/2, and the moving average kept as the pair
// (_tmp_running_* , _running_iter): tmp = tmp*(1-m) + stat; iter = iter*(1-m) + 1; running = tmp / iter.
__global__ void syncbn_finalize_kernel(int C, const double* __restrict__ sum, const double* __restrict__ sq, int reps, int rstride, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                       float* __restrict__ tmp_mean, float* __restrict__ tmp_var, float* __restrict__ iter,
                                       float* __restrict__ rmean, float* __restrict__ rvar,
                                       float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_o, float* __restrict__ invstd_o)
{
    const float keep = 1.f - momentum;
    const float it_new = iter ? iter[0] * keep + 1.f : 1.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double s1 = rep_sum(sum, reps, rstride, c), s2 = rep_sum(sq, reps, rstride, c);
        const double m = s1 / count;
        double sumvar = s2 - s1 * m;
        if (sumvar < 0.0) sumvar = 0.0;
        const double bias_var = sumvar / count, unb = sumvar / (count - 1.0);
        const float mean = (float)m;
        const float invstd = (float)(1.0 / sqrt(bias_var < (double)eps ? (double)eps : bias_var));
        if (tmp_mean) {
            const float tm = tmp_mean[c] * keep + mean, tv = tmp_var[c] * keep + (float)unb;
            tmp_mean[c] = tm; tmp_var[c] = tv;
            rmean[c] = tm / it_new; rvar[c] = tv / it_new;
        }
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        const float s = g * invstd;
        scale[c] = s; shift[c] = b - mean * s;
        if (mean_o) mean_o[c] = mean;
        if (invstd_o) invstd_o[c] = invstd;
    }
    __syncthreads();              // every thread has read iter[0]
    if (iter && threadIdx.x == 0) iter[0] = it_new;
}

// PRO: consumer-side BatchNorm finalize (round 6): scale / shift are derived in the kernel's prologue from the raw batch statistics the
// producing convolution accumulated (bn_prologue_fill: the arithmetic of bn_finalize_kernel, cooperatively into the LDS); workgroup 0 publishes the
// [4][C] parameter block the backward pass reads and updates the running statistics -- the saunet_bn_finalize launch in front of every
// conv -> BN -> act layer (33 single-workgroup launches per step) disappears.
template <typename T, int V, bool PRO = false>
__global__ __launch_bounds__(256) void affine_act_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const T* __restrict__ res, int ldr, int relu,
                                                         T* __restrict__ y, int ldy, long P, int C, long rpb, unsigned char* __restrict__ mask,
                                                         saunet_bn_prologue pro, const float* __restrict__ cbias)
{
    extern __shared__ float s_aff[];      // PRO: [2][C]
    if constexpr (PRO) {
        bn_prologue_fill<256>(pro, C, C, s_aff, blockIdx.x == 0, cbias);
        __syncthreads();
        scale = s_aff; shift = s_aff + C;
    }
    // mask (V == 8 only): one byte per 8-channel chunk, bit j = the ReLU let channel j through -- what the backward pass of a residual block
    // needs instead of re-reading the skip tensor (saunet_bn_backward_*_masked)
    const long p0 = blockIdx.x * rpb, p1 = min(p0 + rpb, P);
    const int CH = C / V;
    for (int cb = 0; cb < CH; cb += 256) {
        const int cw = min(256, CH - cb), rl = 256 / cw;
        const int ch = cb + threadIdx.x % cw, r0 = threadIdx.x / cw;
        if (r0 >= rl) continue;
        float s[V], t[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { s[j] = scale ? scale[ch * V + j] : 1.f; t[j] = shift ? shift[ch * V + j] : 0.f; }
        for (long p = p0 + r0; p < p1; p += rl) {
            float f[V], r[V];
            ChunkIO<T, V>::load(x + p * ldx + ch * V, f);
            if (res) ChunkIO<T, V>::load(res + p * ldr + ch * V, r);
            unsigned bits = 0;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float v = fmaf(f[j], s[j], t[j]);
                if (res) v += r[j];
                bits |= (v > 0.f ? 1u : 0u) << j;
                f[j] = relu ? fmaxf(v, 0.f) : v;
            }
            ChunkIO<T, V>::store(y + p * ldy + ch * V, f);
            if constexpr (V == 8) { if (mask) mask[p * CH + ch] = (unsigned char)bits; }
        }
    }
}

// affine_act with the SE squeeze fused in: while the normalised + activated tensor F = act(x*scale+shift) is written, its global
// average pool over H x W (nn.AdaptiveAvgPool2d(1) of SEModule, /root/reference/models/attention_blocks.py:32,50) is accumulated:
// per-thread sums -> LDS (two images at most per block: rpb <= HW) -> one float atomic per (image, channel) per block into the zeroed
// pooled[N][C], already scaled by 1/HW.  F is never re-read for the pool.
template <typename T, int V, bool PRO = false>
__global__ __launch_bounds__(256) void affine_act_pool_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int relu, T* __restrict__ y, int ldy, long P, int C,
                                                              long rpb, float* __restrict__ pooled, int HW, saunet_bn_prologue pro, const float* __restrict__ cbias)
{
    extern __shared__ float s_pool[];   // [2][C]  (PRO: + [2][C] coefficients)
    if constexpr (PRO) {
        bn_prologue_fill<256>(pro, C, C, s_pool + 2 * C, blockIdx.x == 0, cbias);
        scale = s_pool + 2 * C; shift = s_pool + 3 * C;
    }
    for (int i = threadIdx.x; i < 2 * C; i += 256) s_pool[i] = 0.f;
    __syncthreads();
    const long p0 = blockIdx.x * rpb, p1 = min(p0 + rpb, P);
    const long n0 = p0 / HW, boundary = (n0 + 1) * (long)HW;
    const int CH = C / V;
    for (int cb = 0; cb < CH; cb += 256) {
        const int cw = min(256, CH - cb), rl = 256 / cw;
        const int ch = cb + threadIdx.x % cw, r0 = threadIdx.x / cw;
        if (r0 >= rl) continue;
        float s[V], t[V], a0[V], a1[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { s[j] = scale[ch * V + j]; t[j] = shift[ch * V + j]; a0[j] = a1[j] = 0.f; }
        for (long p = p0 + r0; p < p1; p += rl) {
            float f[V];
            ChunkIO<T, V>::load(x + p * ldx + ch * V, f);
            const bool first = p < boundary;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float v = fmaf(f[j], s[j], t[j]);
                v = relu ? fmaxf(v, 0.f) : v;
                f[j] = v;
            }
            if constexpr (sizeof(T) == 2) {      // pool what the consumers will read: the bf16-rounded values
                const u32x4 q = Vec16<T>::pack(f);
                *(u32x4*)(y + p * ldy + ch * V) = q;
                Vec16<T>::unpack(q, f);
            } else ChunkIO<T, V>::store(y + p * ldy + ch * V, f);
#pragma unroll
            for (int j = 0; j < V; ++j) { if (first) a0[j] += f[j]; else a1[j] += f[j]; }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) { atomicAdd(&s_pool[ch * V + j], a0[j]); atomicAdd(&s_pool[C + ch * V + j], a1[j]); }
    }
    __syncthreads();
    const float inv = 1.f / (float)HW;
    const bool two = p1 > boundary;
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(&pooled[n0 * C + c], s_pool[c] * inv);
        if (two) atomicAdd(&pooled[(n0 + 1) * C + c], s_pool[C + c] * inv);
    }
}

struct BnBwdArgs {
    const void* dy; int lddy; const void* x; int ldx; const void* res; int ldr;
    const float* scale; const float* shift; const float* mean; const float* invstd; int relu;
    double* sums; int sreps, srstride; double count; int training, accumulate;
    void* dx; int lddx; void* dres; int lddres; float* dgamma; float* dbeta;
    long P; int C; long rpb;
    const unsigned char* mask;      // non-null (V == 8): the ReLU decisions as bits (affine_act's mask output); the pre-activation is not recomputed
};

template <typename T, int V> __global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdArgs a)
{
    extern __shared__ double s_red[];  // [2][C]
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) s_red[i] = 0.0;
    __syncthreads();
    const long p0 = blockIdx.x * a.rpb, p1 = min(p0 + a.rpb, a.P);
    const int CH = a.C / V;
    const T* dy = (const T*)a.dy; const T* x = (const T*)a.x; const T* res = (const T*)a.res;
    for (int cb = 0; cb < CH; cb += 256) {
        const int cw = min(256, CH - cb), rl = 256 / cw;
        const int ch = cb + threadIdx.x % cw, r0 = threadIdx.x / cw;
        if (r0 >= rl) continue;
        float s[V], t[V], mu[V], is[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            s[j] = a.scale[ch * V + j]; t[j] = a.shift[ch * V + j]; mu[j] = a.mean[ch * V + j]; is[j] = a.invstd[ch * V + j];
        }
        double d1[V], d2[V]; float f1[V], f2[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { d1[j] = d2[j] = 0.0; f1[j] = f2[j] = 0.f; }
        int cnt = 0;
        for (long p = p0 + r0; p < p1; p += rl) {
            float g[V], xv[V], r[V];
            ChunkIO<T, V>::load(dy + p * a.lddy + ch * V, g);
            ChunkIO<T, V>::load(x + p * a.ldx + ch * V, xv);
            if (res) ChunkIO<T, V>::load(res + p * a.ldr + ch * V, r);
            unsigned bits = 0xffu;
            if constexpr (V == 8) { if (a.mask) bits = a.mask[p * CH + ch]; }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float gv = g[j];
                if (V == 8 && a.mask) { if (!((bits >> j) & 1u)) gv = 0.f; }
                else if (a.relu) {
                    float o = fmaf(xv[j], s[j], t[j]);
                    if (res) o += r[j];
                    if (!(o > 0.f)) gv = 0.f;
                }
                f1[j] += gv;
                f2[j] = fmaf(gv, (xv[j] - mu[j]) * is[j], f2[j]);
            }
            if (++cnt == 128) {
#pragma unroll
                for (int j = 0; j < V; ++j) { d1[j] += f1[j]; d2[j] += f2[j]; f1[j] = f2[j] = 0.f; }
                cnt = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            atomicAdd(&s_red[ch * V + j], d1[j] + (double)f1[j]);
            atomicAdd(&s_red[a.C + ch * V + j], d2[j] + (double)f2[j]);
        }
    }
    __syncthreads();
    const size_t ro = (size_t)(blockIdx.x % a.sreps) * a.srstride;
    for (int c = threadIdx.x; c < 2 * a.C; c += 256) atomicAdd(&a.sums[ro + c], s_red[c]);
}

template <typename T, int V> __global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a)
{
    // the two per-channel sums, replicas added ONCE per block (cooperatively, into LDS) -- not once per thread and channel
    extern __shared__ float s_c12[];       // [2][C]: sum g / count , sum g*xhat / count
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) {
        const double v = rep_sum(a.sums, a.sreps, a.srstride, i);
        s_c12[i] = a.training ? (float)(v / a.count) : 0.f;
        if (blockIdx.x == 0 && a.dgamma) { if (i < a.C) a.dbeta[i] = (float)v; else a.dgamma[i - a.C] = (float)v; }
    }
    __syncthreads();
    const long p0 = blockIdx.x * a.rpb, p1 = min(p0 + a.rpb, a.P);
    const int CH = a.C / V;
    const T* dy = (const T*)a.dy; const T* x = (const T*)a.x; const T* res = (const T*)a.res;
    T* dx = (T*)a.dx; T* dres = (T*)a.dres;
    for (int cb = 0; cb < CH; cb += 256) {
        const int cw = min(256, CH - cb), rl = 256 / cw;
        const int ch = cb + threadIdx.x % cw, r0 = threadIdx.x / cw;
        if (r0 >= rl) continue;
        float s[V], t[V], mu[V], is[V], c1[V], c2[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int c = ch * V + j;
            s[j] = a.scale[c]; t[j] = a.shift[c]; mu[j] = a.mean[c]; is[j] = a.invstd[c];
            c1[j] = s_c12[c]; c2[j] = s_c12[a.C + c];
        }
        for (long p = p0 + r0; p < p1; p += rl) {
            float g[V], xv[V], r[V], o[V];
            ChunkIO<T, V>::load(dy + p * a.lddy + ch * V, g);
            ChunkIO<T, V>::load(x + p * a.ldx + ch * V, xv);
            if (res) ChunkIO<T, V>::load(res + p * a.ldr + ch * V, r);
            if (a.accumulate) ChunkIO<T, V>::load(dx + p * a.lddx + ch * V, o);
            unsigned bits = 0xffu;
            if constexpr (V == 8) { if (a.mask) bits = a.mask[p * CH + ch]; }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float gv = g[j];
                if (V == 8 && a.mask) { if (!((bits >> j) & 1u)) gv = 0.f; }
                else if (a.relu) {
                    float ov = fmaf(xv[j], s[j], t[j]);
                    if (res) ov += r[j];
                    if (!(ov > 0.f)) gv = 0.f;
                }
                g[j] = gv;
                float d = s[j] * (gv - c1[j] - (xv[j] - mu[j]) * is[j] * c2[j]);
                o[j] = a.accumulate ? o[j] + d : d;
            }
            ChunkIO<T, V>::store(dx + p * a.lddx + ch * V, o);
            if (dres) ChunkIO<T, V>::store(dres + p * a.lddres + ch * V, g);
        }
    }
}

__global__ void bn_bwd_coeff_kernel(int C, const double* __restrict__ sums, int reps, int rstride, double count, const float* __restrict__ scale,
                                    float* __restrict__ A, float* __restrict__ B, float* __restrict__ dgamma, float* __restrict__ dbeta, int training)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double s1 = rep_sum(sums, reps, rstride, c), s2 = rep_sum(sums, reps, rstride, C + c);
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    if (training) { A[c] += scale[c] * (float)(s1 / count); B[c] += scale[c] * (float)(s2 / count); }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_correct_kernel(T* __restrict__ dx, int lddx, const T* __restrict__ x, int ldx, const float* __restrict__ A,
                                                             const float* __restrict__ B, const float* __restrict__ xs, const float* __restrict__ xt,
                                                             long P, int C, long rpb)
{
    const long p0 = blockIdx.x * rpb, p1 = min(p0 + rpb, P);
    const int CH = C / V;
    for (int cb = 0; cb < CH; cb += 256) {
        const int cw = min(256, CH - cb), rl = 256 / cw;
        const int ch = cb + threadIdx.x % cw, r0 = threadIdx.x / cw;
        if (r0 >= rl) continue;
        float a[V], b[V], s[V], t[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { a[j] = A[ch * V + j]; b[j] = B[ch * V + j]; s[j] = xs[ch * V + j]; t[j] = xt[ch * V + j]; }
        for (long p = p0 + r0; p < p1; p += rl) {
            float d[V], xv[V];
            ChunkIO<T, V>::load(dx + p * lddx + ch * V, d);
            ChunkIO<T, V>::load(x + p * ldx + ch * V, xv);
#pragma unroll
            for (int j = 0; j < V; ++j) d[j] -= a[j] + b[j] * fmaf(xv[j], s[j], t[j]);
            ChunkIO<T, V>::store(dx + p * lddx + ch * V, d);
        }
    }
}

// bn_bwd_coeff + bn_bwd_correct of the NEXT chunk in one launch (training mode).  Inside a dense block's backward the coefficient kernel of
// layer l is always followed by the correction of the gradient chunk that layer l-1 consumes: the chunk's A / B only need this layer's two sums
// for those few channels, so every correcting block derives them itself (CW channels x replicas, one batched round trip) while the last
// blocks of the grid do the coefficient update for all C channels.  A / B are ping-pong buffers (a correcting block must not read a value
// the coefficient blocks have already updated).
struct CoeffCorrectArgs {
    int C; const double* sums; int reps, rstride; double count; const float* scale;
    const float* A_in; const float* B_in; float* A_out; float* B_out; float* dgamma; float* dbeta;
    void* dx; int lddx; const void* x; int ldx; int c_lo, CW; const float* xs; const float* xt; long P; long rpb; int nb_correct;
};

template <typename T, int V> __global__ __launch_bounds__(256) void bn_bwd_coeff_correct_kernel(CoeffCorrectArgs a)
{
    if ((int)blockIdx.x >= a.nb_correct) {
        const int c = ((int)blockIdx.x - a.nb_correct) * 256 + threadIdx.x;
        if (c >= a.C) return;
        const double s1 = rep_sum(a.sums, a.reps, a.rstride, c), s2 = rep_sum(a.sums, a.reps, a.rstride, a.C + c);
        a.dbeta[c] = (float)s1; a.dgamma[c] = (float)s2;
        const float sc = a.scale[c];
        a.A_out[c] = a.A_in[c] + sc * (float)(s1 / a.count); a.B_out[c] = a.B_in[c] + sc * (float)(s2 / a.count);
        return;
    }
    __shared__ float s_ab[4][256];
    for (int t = threadIdx.x; t < a.CW; t += 256) {
        const int c = a.c_lo + t;
        const double s1 = rep_sum(a.sums, a.reps, a.rstride, c), s2 = rep_sum(a.sums, a.reps, a.rstride, a.C + c);
        const float sc = a.scale[c];
        s_ab[0][t] = a.A_in[c] + sc * (float)(s1 / a.count); s_ab[1][t] = a.B_in[c] + sc * (float)(s2 / a.count);
        s_ab[2][t] = a.xs[c]; s_ab[3][t] = a.xt[c];
    }
    __syncthreads();
    const long p0 = blockIdx.x * a.rpb, p1 = min(p0 + a.rpb, a.P);
    const int CH = a.CW / V;
    T* dx = (T*)a.dx; const T* x = (const T*)a.x;
    for (int cb = 0; cb < CH; cb += 256) {
        const int cw = min(256, CH - cb), rl = 256 / cw;
        const int ch = cb + threadIdx.x % cw, r0 = threadIdx.x / cw;
        if (r0 >= rl) continue;
        float av[V], bv[V], sv[V], tv[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { av[j] = s_ab[0][ch * V + j]; bv[j] = s_ab[1][ch * V + j]; sv[j] = s_ab[2][ch * V + j]; tv[j] = s_ab[3][ch * V + j]; }
        for (long p = p0 + r0; p < p1; p += rl) {
            float d[V], xv[V];
            ChunkIO<T, V>::load(dx + p * a.lddx + ch * V, d);
            ChunkIO<T, V>::load(x + p * a.ldx + ch * V, xv);
#pragma unroll
            for (int j = 0; j < V; ++j) d[j] -= av[j] + bv[j] * fmaf(xv[j], sv[j], tv[j]);
            ChunkIO<T, V>::store(dx + p * a.lddx + ch * V, d);
        }
    }
}

// the chunk correction with the coefficients taken from the block's running float64 sums (dense_dgrad_kernel's epilogue accumulates them):
// y = d - (A + B * (x*xs + xt)),  A = sum_r ab[r][c] / count,  B = sum_r ab[r][half + c] / count.  y may alias d.
struct CorrectAbArgs {
    const void* d; int ldd; const void* x; int ldx; void* y; int ldy;
    const double* ab; int reps, rstride, half; double count; const float* xs; const float* xt; long P; int C; long rpb;
};
template <typename T, int V> __global__ __launch_bounds__(256) void bn_bwd_correct_ab_kernel(CorrectAbArgs a)
{
    __shared__ float s_ab[2][256];         // A + B*xt ,  B*xs
    for (int c = threadIdx.x; c < a.C; c += 256) {
        double A, B;
        rep_sum2(a.ab, a.ab + a.half, a.reps, a.rstride, c, A, B);
        const float Af = (float)(A / a.count), Bf = (float)(B / a.count);
        s_ab[0][c] = fmaf(Bf, a.xt[c], Af); s_ab[1][c] = Bf * a.xs[c];
    }
    __syncthreads();
    const long p0 = blockIdx.x * a.rpb, p1 = min(p0 + a.rpb, a.P);
    const int CH = a.C / V;
    const T* dd = (const T*)a.d; const T* x = (const T*)a.x; T* y = (T*)a.y;
    const int cw = min(256, CH), rl = 256 / cw;
    const int ch = threadIdx.x % cw, r0 = threadIdx.x / cw;
    if (r0 >= rl) return;
    float av[V], bv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { av[j] = s_ab[0][ch * V + j]; bv[j] = s_ab[1][ch * V + j]; }
    for (long p = p0 + r0; p < p1; p += rl) {
        float dv[V], xv[V];
        ChunkIO<T, V>::load(dd + p * a.ldd + ch * V, dv);
        ChunkIO<T, V>::load(x + p * a.ldx + ch * V, xv);
#pragma unroll
        for (int j = 0; j < V; ++j) dv[j] -= fmaf(bv[j], xv[j], av[j]);
        ChunkIO<T, V>::store(y + p * a.ldy + ch * V, dv);
    }
}

__global__ void bn_bwd_coeff_ab_kernel(int C, const double* __restrict__ sums, int reps, int rstride, const float* __restrict__ scale, double* __restrict__ ab,
                                       int half, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1, s2;
    rep_sum2(sums, sums + C, reps, rstride, c, s1, s2);
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    const double sc = (double)scale[c];
    ab[c] += sc * s1; ab[half + c] += sc * s2;
}

// ---- DenseNet transition with the pooling IN FRONT of the 1x1 convolution (round 6).  torchvision's _Transition is norm -> relu -> conv1x1 ->
// AvgPool2d(2, 2) (/root/reference/models/models.py:271 as sliced at :306-313); a pointwise convolution and an average pool are both linear and
// act on different axes, so they commute: pool(conv(a)) == conv(pool(a)).  Pooling the ACTIVATION first runs the convolution, its data gradient
// and its weight gradient on a quarter of the pixels and never materialises the full-resolution C/2-channel tensor in either direction.
//   forward :  a'[n, oy, ox, c] = 1/4 * sum_{2x2} relu(x * scale + shift)                       (one pass: read C full-res, write C quarter-res)
//   backward:  g[p, c] = [x * scale + shift > 0] * 1/4 * da'[p / 2, c];  sums[c] += g, sums[C + c] += g * xhat;  dx[p, c] = (scaled ? scale : 1) * g
//              (the reduce pass of the BatchNorm backward and the pool / ReLU backward in ONE pass; `scaled` = the dense block's linear form)
struct PoolBnArgs {
    const void* x; int ldx; const float* scale; const float* shift; const float* mean; const float* invstd;
    void* y; int ldy; const void* da; int ldda; void* dx; int lddx; int scaled;
    double* sums; int sreps, srstride;
    int Wo, W, C; long Po, rpb;
};

template <typename T, int V> __global__ __launch_bounds__(256) void bn_relu_avgpool2_fwd_kernel(PoolBnArgs a)
{
    const long p0 = blockIdx.x * a.rpb, p1 = min(p0 + a.rpb, a.Po);
    const int CH = a.C / V;
    const T* x = (const T*)a.x; T* y = (T*)a.y;
    for (int cb = 0; cb < CH; cb += 256) {
        const int cw = min(256, CH - cb), rl = 256 / cw;
        const int ch = cb + threadIdx.x % cw, r0 = threadIdx.x / cw;
        if (r0 >= rl) continue;
        float s[V], t[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { s[j] = a.scale[ch * V + j]; t[j] = a.shift[ch * V + j]; }
        for (long op = p0 + r0; op < p1; op += rl) {
            const long r = op / a.Wo; const int ox = (int)(op - r * a.Wo);
            const T* xb = x + ((2 * r) * a.W + 2 * ox) * (long)a.ldx + ch * V;
            float f[4][V], o[V];
            ChunkIO<T, V>::load(xb, f[0]); ChunkIO<T, V>::load(xb + a.ldx, f[1]);
            ChunkIO<T, V>::load(xb + (long)a.W * a.ldx, f[2]); ChunkIO<T, V>::load(xb + (long)(a.W + 1) * a.ldx, f[3]);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float v0 = fmaxf(fmaf(f[0][j], s[j], t[j]), 0.f), v1 = fmaxf(fmaf(f[1][j], s[j], t[j]), 0.f);
                const float v2 = fmaxf(fmaf(f[2][j], s[j], t[j]), 0.f), v3 = fmaxf(fmaf(f[3][j], s[j], t[j]), 0.f);
                o[j] = 0.25f * ((v0 + v1) + (v2 + v3));
            }
            ChunkIO<T, V>::store(y + op * a.ldy + ch * V, o);
        }
    }
}

template <typename T, int V> __global__ __launch_bounds__(256) void bn_relu_avgpool2_bwd_kernel(PoolBnArgs a)
{
    extern __shared__ double s_red[];  // [2][C]
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) s_red[i] = 0.0;
    __syncthreads();
    const long p0 = blockIdx.x * a.rpb, p1 = min(p0 + a.rpb, a.Po);
    const int CH = a.C / V;
    const T* x = (const T*)a.x; const T* da = (const T*)a.da; T* dx = (T*)a.dx;
    for (int cb = 0; cb < CH; cb += 256) {
        const int cw = min(256, CH - cb), rl = 256 / cw;
        const int ch = cb + threadIdx.x % cw, r0 = threadIdx.x / cw;
        if (r0 >= rl) continue;
        float s[V], t[V], mu[V], is[V], os[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            s[j] = a.scale[ch * V + j]; t[j] = a.shift[ch * V + j]; mu[j] = a.mean[ch * V + j]; is[j] = a.invstd[ch * V + j];
            os[j] = a.scaled ? s[j] : 1.f;
        }
        double d1[V], d2[V]; float f1[V], f2[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { d1[j] = d2[j] = 0.0; f1[j] = f2[j] = 0.f; }
        int cnt = 0;
        for (long op = p0 + r0; op < p1; op += rl) {
            const long r = op / a.Wo; const int ox = (int)(op - r * a.Wo);
            const long pb = (2 * r) * a.W + 2 * ox;
            float g[V], xv[4][V];
            ChunkIO<T, V>::load(da + op * a.ldda + ch * V, g);
#pragma unroll
            for (int q = 0; q < 4; ++q) ChunkIO<T, V>::load(x + (pb + (q >> 1) * (long)a.W + (q & 1)) * a.ldx + ch * V, xv[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float o[V];
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float gv = fmaf(xv[q][j], s[j], t[j]) > 0.f ? 0.25f * g[j] : 0.f;
                    f1[j] += gv;
                    f2[j] = fmaf(gv, (xv[q][j] - mu[j]) * is[j], f2[j]);
                    o[j] = os[j] * gv;
                }
                ChunkIO<T, V>::store(dx + (pb + (q >> 1) * (long)a.W + (q & 1)) * a.lddx + ch * V, o);
            }
            if ((cnt += 4) >= 128) {
#pragma unroll
                for (int j = 0; j < V; ++j) { d1[j] += f1[j]; d2[j] += f2[j]; f1[j] = f2[j] = 0.f; }
                cnt = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            atomicAdd(&s_red[ch * V + j], d1[j] + (double)f1[j]);
            atomicAdd(&s_red[a.C + ch * V + j], d2[j] + (double)f2[j]);
        }
    }
    __syncthreads();
    const size_t ro = (size_t)(blockIdx.x % a.sreps) * a.srstride;
    for (int c = threadIdx.x; c < 2 * a.C; c += 256) atomicAdd(&a.sums[ro + c], s_red[c]);
}

// dgamma / dbeta of all norm1 layers of a dense block: blockIdx.y = layer
__global__ __launch_bounds__(256) void dense_bn1_grads_kernel(saunet_dense_bn1_list l)
{
    const int layer = blockIdx.y, C = l.cin[layer];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s1, s2;
    rep_sum2(l.sums[layer], l.sums[layer] + C, l.replicas, l.rstride[layer], c, s1, s2);
    l.dbeta[layer][c] = (float)s1; l.dgamma[layer][c] = (float)s2;
}

static inline long rows_per_block(long P, int C, int V, int* blocks)
{
    // enough blocks to fill the chip, each with a few thousand elements per thread at most
    long nb = (P * (long)(C / V) + 256L * 8 - 1) / (256L * 8);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    long rpb = (P + nb - 1) / nb;
    *blocks = (int)((P + rpb - 1) / rpb);
    return rpb;
}

// vector path needs every view 16-byte aligned with channel counts/strides multiples of the chunk
static inline bool vec_ok(int dtype, int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs)
{
    const int epc = dtype == SAUNET_BF16 ? 8 : 4;
    if (C % epc) return false;
    for (int l : lds) if (l % epc) return false;
    for (const void* p : ptrs) if (((uintptr_t)p) & 15) return false;
    return true;
}

}  // namespace saunet

using namespace saunet;

#define DISPATCH_TV(dtype, vec, CALL)                                                     \
    do {                                                                                  \
        if ((dtype) == SAUNET_F32) { if (vec) { CALL(float, 4); } else { CALL(float, 1); } } \
        else if ((dtype) == SAUNET_BF16) { if (vec) { CALL(u16, 8); } else { CALL(u16, 1); } } \
        else return set_error(SAUNET_BAD_DTYPE, "dtype %d", (dtype));                     \
    } while (0)
// launch-log name of a <T, V> kernel: "base_kernel<unsigned short, 8>"
#define CHECK_LAUNCH_TV(base, dtype, vec)                                                 \
    do {                                                                                  \
        const KName kn__(base "_kernel", (dtype) == SAUNET_BF16 ? "unsigned short" : "float", (vec) ? ((dtype) == SAUNET_BF16 ? 8 : 4) : 1); \
        SAUNET_CHECK_LAUNCH(kn__.s);                                                      \
    } while (0)

namespace saunet {
int bn_backward_correct_ab(int dtype, const void* d, int ldd, const void* x, int ldx, void* y, int ldy, const double* ab, int ab_replicas,
                           int ab_rstride, int ab_half, double count, const float* xs, const float* xt, int64_t pixels, int C, hipStream_t st)
{
    if (C < 1 || C > 256 || pixels < 1 || !d || !x || !y || !ab || !xs || !xt || ab_replicas < 1 || count < 1.0)
        return set_error(SAUNET_BAD_SHAPE, "bn_backward_correct_ab: C=%d pixels=%ld", C, (long)pixels);
    const bool vec = vec_ok(dtype, C, {ldd, ldx, ldy}, {d, x, y});
    int blocks; const int V = vec ? (dtype == SAUNET_BF16 ? 8 : 4) : 1;
    CorrectAbArgs a{d, ldd, x, ldx, y, ldy, ab, ab_replicas, ab_rstride, ab_half, count, xs, xt, (long)pixels, C, 0};
    a.rpb = rows_per_block(pixels, C, V, &blocks);
#define CALL(TT, VV) hipLaunchKernelGGL((bn_bwd_correct_ab_kernel<TT, VV>), dim3(blocks), dim3(256), 0, st, a)
    DISPATCH_TV(dtype, vec, CALL);
#undef CALL
    CHECK_LAUNCH_TV("bn_bwd_correct_ab", dtype, vec);
    return SAUNET_OK;
}

int bn_prologue_finalize(const saunet_bn_prologue* p, int Cin, hipStream_t st)
{
    hipLaunchKernelGGL(bn_prologue_finalize_kernel, dim3(1), dim3(256), sizeof(float) * 2 * Cin, st, *p, Cin, nullptr);
    SAUNET_CHECK_LAUNCH("bn_prologue_finalize");
    return SAUNET_OK;
}
}  // namespace saunet

extern "C" {

int saunet_sum_replicas(double* base, int n, int replicas, int rstride, void* stream)
{
    if (replicas <= 1) return SAUNET_OK;
    hipLaunchKernelGGL(sum_replicas_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, base, n, replicas, rstride);
    SAUNET_CHECK_LAUNCH("sum_replicas");
    return SAUNET_OK;
}

int saunet_bn_stats(int dtype, const void* x, int64_t pixels, int C, int ld, double* sum, double* sumsq, int replicas, int rstride, void* stream)
{
    if (replicas < 1) replicas = 1;
    hipStream_t st = (hipStream_t)stream;
    if (C > 4096) return set_error(SAUNET_UNSUPPORTED, "bn_stats: C=%d > 4096", C);
    const bool vec = vec_ok(dtype, C, {ld}, {x});
    int blocks; const int V = vec ? (dtype == SAUNET_BF16 ? 8 : 4) : 1;
    long rpb = rows_per_block(pixels, C, V, &blocks);
#define CALL(TT, VV) hipLaunchKernelGGL((bn_stats_kernel<TT, VV>), dim3(blocks), dim3(256), 2 * C * sizeof(double), st, (const TT*)x, (long)pixels, C, ld, rpb, sum, sumsq, replicas, rstride)
    DISPATCH_TV(dtype, vec, CALL);
#undef CALL
    CHECK_LAUNCH_TV("bn_stats", dtype, vec);
    return SAUNET_OK;
}

int saunet_bn_xhat(int C, const double* sum, const double* sumsq, int replicas, int rstride, double count, float eps, float* xhat, int ld, void* stream)
{
    if (C < 1 || !sum || !sumsq || !xhat || ld < C) return set_error(SAUNET_BAD_SHAPE, "bn_xhat: C=%d ld=%d", C, ld);
    hipLaunchKernelGGL(bn_xhat_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, sum, sumsq, replicas < 1 ? 1 : replicas, rstride, count, eps, xhat, ld);
    SAUNET_CHECK_LAUNCH("bn_xhat");
    return SAUNET_OK;
}

int saunet_bn_finalize(int C, const double* sum, const double* sumsq, int replicas, int rstride, double count, const float* conv_bias,
                       const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* scale, float* shift,
                       float* mean, float* invstd, int training, void* stream)
{
    if (!training && (!running_mean || !running_var)) return set_error(SAUNET_BAD_SHAPE, "bn_finalize: eval mode needs running stats");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, sum, sumsq, replicas < 1 ? 1 : replicas, rstride, count, conv_bias,
                       gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd, training);
    SAUNET_CHECK_LAUNCH("bn_finalize");
    return SAUNET_OK;
}

int saunet_syncbn_finalize(int C, const double* sum, const double* sumsq, int replicas, int rstride, double count,
                           const float* gamma, const float* beta, float eps, float momentum,
                           float* tmp_running_mean, float* tmp_running_var, float* running_iter,
                           float* running_mean, float* running_var, float* scale, float* shift,
                           float* mean, float* invstd, void* stream)
{
    if (!sum || !sumsq || count <= 1.0) return set_error(SAUNET_BAD_SHAPE, "syncbn_finalize: needs sums and a global count > 1");
    if (tmp_running_mean && (!tmp_running_var || !running_iter || !running_mean || !running_var))
        return set_error(SAUNET_BAD_SHAPE, "syncbn_finalize: incomplete running-statistic buffers");
    hipLaunchKernelGGL(syncbn_finalize_kernel, dim3(1), dim3(C < 256 ? 64 * ((C + 63) / 64) : 256), 0, (hipStream_t)stream, C, sum, sumsq, replicas < 1 ? 1 : replicas,
                       rstride, count, gamma, beta, eps, momentum, tmp_running_mean, tmp_running_var, running_iter, running_mean, running_var,
                       scale, shift, mean, invstd);
    SAUNET_CHECK_LAUNCH("syncbn_finalize");
    return SAUNET_OK;
}

static int affine_act_impl(int dtype, const void* x, int ldx, const float* scale, const float* shift, const void* residual, int ldr, int relu,
                           void* y, int ldy, int64_t pixels, int C, unsigned char* mask, void* stream, const saunet_bn_prologue* pro = nullptr,
                           const float* cbias = nullptr)
{
    hipStream_t st = (hipStream_t)stream;
    const bool vec = residual ? vec_ok(dtype, C, {ldx, ldy, ldr}, {x, y, residual}) : vec_ok(dtype, C, {ldx, ldy}, {x, y});
    if (mask && !(vec && dtype == SAUNET_BF16))
        return set_error(SAUNET_UNSUPPORTED, "affine_act_mask: bf16, C and strides multiples of 8, 16-byte aligned views");
    int blocks; const int V = vec ? (dtype == SAUNET_BF16 ? 8 : 4) : 1;
    long rpb = rows_per_block(pixels, C, V, &blocks);
    if (pro) {
        if (!vec) return set_error(SAUNET_UNSUPPORTED, "affine_act_bn: vector path only (C, strides multiples of 8 bf16 / 4 f32 elements, 16-byte aligned views)");
        const size_t lds = sizeof(float) * 2 * C;
        if (dtype == SAUNET_BF16)
            hipLaunchKernelGGL((affine_act_kernel<u16, 8, true>), dim3(blocks), dim3(256), lds, st, (const u16*)x, ldx, nullptr, nullptr, (const u16*)residual, ldr, relu, (u16*)y, ldy, (long)pixels, C, rpb, mask, *pro, cbias);
        else if (dtype == SAUNET_F32)
            hipLaunchKernelGGL((affine_act_kernel<float, 4, true>), dim3(blocks), dim3(256), lds, st, (const float*)x, ldx, nullptr, nullptr, (const float*)residual, ldr, relu, (float*)y, ldy, (long)pixels, C, rpb, mask, *pro, cbias);
        else return set_error(SAUNET_BAD_DTYPE, "dtype %d", dtype);
        CHECK_LAUNCH_TV("affine_act", dtype, vec);
        return SAUNET_OK;
    }
    const saunet_bn_prologue none{};
#define CALL(TT, VV) hipLaunchKernelGGL((affine_act_kernel<TT, VV>), dim3(blocks), dim3(256), 0, st, (const TT*)x, ldx, scale, shift, (const TT*)residual, ldr, relu, (TT*)y, ldy, (long)pixels, C, rpb, mask, none, nullptr)
    DISPATCH_TV(dtype, vec, CALL);
#undef CALL
    CHECK_LAUNCH_TV("affine_act", dtype, vec);
    return SAUNET_OK;
}

static int check_bn_prologue(const saunet_bn_prologue* pro, int C, const char* who)
{
    if (!pro || !pro->gamma || !pro->beta || !pro->params || !pro->sum || !pro->sumsq || pro->count < 1.0 || pro->c_lo != 0 || C > 4096 ||
        (pro->running_mean == nullptr) != (pro->running_var == nullptr))
        return set_error(SAUNET_BAD_SHAPE, "%s: incomplete BatchNorm prologue descriptor (c_lo must be 0, C <= 4096)", who);
    return SAUNET_OK;
}

int saunet_affine_act_bn(int dtype, const void* x, int ldx, const saunet_bn_prologue* pro, const float* conv_bias, const void* residual, int ldr, int relu,
                         void* y, int ldy, int64_t pixels, int C, uint8_t* relu_mask, void* stream)
{
    if (int rc = check_bn_prologue(pro, C, "affine_act_bn")) return rc;
    if (relu_mask && !relu) return set_error(SAUNET_BAD_SHAPE, "affine_act_bn: a ReLU mask needs relu = 1");
    saunet_bn_prologue p = *pro;
    if (p.replicas < 1) p.replicas = 1;
    return affine_act_impl(dtype, x, ldx, nullptr, nullptr, residual, ldr, relu, y, ldy, pixels, C, relu_mask, stream, &p, conv_bias);
}

int saunet_affine_act(int dtype, const void* x, int ldx, const float* scale, const float* shift,
                      const void* residual, int ldr, int relu, void* y, int ldy, int64_t pixels, int C, void* stream)
{
    return affine_act_impl(dtype, x, ldx, scale, shift, residual, ldr, relu, y, ldy, pixels, C, nullptr, stream);
}

int saunet_affine_act_mask(int dtype, const void* x, int ldx, const float* scale, const float* shift,
                           const void* residual, int ldr, void* y, int ldy, int64_t pixels, int C, uint8_t* relu_mask, void* stream)
{
    if (!relu_mask) return set_error(SAUNET_BAD_SHAPE, "affine_act_mask: no mask buffer");
    return affine_act_impl(dtype, x, ldx, scale, shift, residual, ldr, 1, y, ldy, pixels, C, relu_mask, stream);
}

static int affine_act_pool_impl(int dtype, const void* x, int ldx, const float* scale, const float* shift, int relu, void* y, int ldy,
                                int64_t pixels, int C, float* pooled, int HW, void* stream, const saunet_bn_prologue* pro, const float* cbias = nullptr)
{
    hipStream_t st = (hipStream_t)stream;
    if ((!pro && (!scale || !shift)) || !pooled || HW <= 0 || pixels % HW) return set_error(SAUNET_BAD_SHAPE, "affine_act_pool: needs scale/shift/pooled and pixels %% HW == 0");
    const int epc = dtype == SAUNET_BF16 ? 8 : 4;
    if (!vec_ok(dtype, C, {ldx, ldy}, {x, y})) return set_error(SAUNET_BAD_ALIGN, "affine_act_pool: C, strides multiples of %d and 16-byte aligned views", epc);
    int blocks;
    long rpb = rows_per_block(pixels, C, epc, &blocks);
    if (rpb > HW) rpb = HW;                       // a block touches two images at most
    blocks = (int)((pixels + rpb - 1) / rpb);
    if (hipMemsetAsync(pooled, 0, sizeof(float) * (size_t)(pixels / HW) * C, st) != hipSuccess) return set_error(SAUNET_LAUNCH_FAILED, "affine_act_pool memset");
    const saunet_bn_prologue none{};
    if (pro) {
        const size_t lds = sizeof(float) * 4 * C;
        if (dtype == SAUNET_BF16) hipLaunchKernelGGL((affine_act_pool_kernel<u16, 8, true>), dim3(blocks), dim3(256), lds, st, (const u16*)x, ldx, nullptr, nullptr, relu, (u16*)y, ldy, (long)pixels, C, rpb, pooled, HW, *pro, cbias);
        else if (dtype == SAUNET_F32) hipLaunchKernelGGL((affine_act_pool_kernel<float, 4, true>), dim3(blocks), dim3(256), lds, st, (const float*)x, ldx, nullptr, nullptr, relu, (float*)y, ldy, (long)pixels, C, rpb, pooled, HW, *pro, cbias);
        else return set_error(SAUNET_BAD_DTYPE, "dtype %d", dtype);
    } else {
        const size_t lds = sizeof(float) * 2 * C;
        if (dtype == SAUNET_BF16) hipLaunchKernelGGL((affine_act_pool_kernel<u16, 8>), dim3(blocks), dim3(256), lds, st, (const u16*)x, ldx, scale, shift, relu, (u16*)y, ldy, (long)pixels, C, rpb, pooled, HW, none, nullptr);
        else if (dtype == SAUNET_F32) hipLaunchKernelGGL((affine_act_pool_kernel<float, 4>), dim3(blocks), dim3(256), lds, st, (const float*)x, ldx, scale, shift, relu, (float*)y, ldy, (long)pixels, C, rpb, pooled, HW, none, nullptr);
        else return set_error(SAUNET_BAD_DTYPE, "dtype %d", dtype);
    }
    SAUNET_CHECK_LAUNCH("affine_act_pool");
    return SAUNET_OK;
}

int saunet_affine_act_pool(int dtype, const void* x, int ldx, const float* scale, const float* shift, int relu, void* y, int ldy,
                           int64_t pixels, int C, float* pooled, int HW, void* stream)
{
    return affine_act_pool_impl(dtype, x, ldx, scale, shift, relu, y, ldy, pixels, C, pooled, HW, stream, nullptr);
}

int saunet_affine_act_pool_bn(int dtype, const void* x, int ldx, const saunet_bn_prologue* pro, const float* conv_bias, int relu, void* y, int ldy,
                              int64_t pixels, int C, float* pooled, int HW, void* stream)
{
    if (int rc = check_bn_prologue(pro, C, "affine_act_pool_bn")) return rc;
    saunet_bn_prologue p = *pro;
    if (p.replicas < 1) p.replicas = 1;
    return affine_act_pool_impl(dtype, x, ldx, nullptr, nullptr, relu, y, ldy, pixels, C, pooled, HW, stream, &p, conv_bias);
}

static int bn_backward_reduce_impl(int dtype, const void* dy, int lddy, const void* x, int ldx, const void* residual, int ldr,
                              const float* scale, const float* shift, const float* mean, const float* invstd,
                              int relu, double* sums, int replicas, int rstride, int64_t pixels, int C, const unsigned char* mask, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const bool vec = residual ? vec_ok(dtype, C, {lddy, ldx, ldr}, {dy, x, residual}) : vec_ok(dtype, C, {lddy, ldx}, {dy, x});
    if (mask && !(vec && dtype == SAUNET_BF16))
        return set_error(SAUNET_UNSUPPORTED, "bn_backward_reduce_masked: bf16, C and strides multiples of 8, 16-byte aligned views");
    BnBwdArgs a{}; a.mask = mask; a.dy = dy; a.lddy = lddy; a.x = x; a.ldx = ldx; a.res = residual; a.ldr = ldr; a.scale = scale; a.shift = shift;
    a.mean = mean; a.invstd = invstd; a.relu = relu; a.sums = sums; a.sreps = replicas < 1 ? 1 : replicas; a.srstride = rstride; a.P = pixels; a.C = C;
    int blocks; const int V = vec ? (dtype == SAUNET_BF16 ? 8 : 4) : 1;
    a.rpb = rows_per_block(pixels, C, V, &blocks);
#define CALL(TT, VV) hipLaunchKernelGGL((bn_bwd_reduce_kernel<TT, VV>), dim3(blocks), dim3(256), 2 * C * sizeof(double), st, a)
    DISPATCH_TV(dtype, vec, CALL);
#undef CALL
    CHECK_LAUNCH_TV("bn_bwd_reduce", dtype, vec);
    return SAUNET_OK;
}

int saunet_bn_backward_reduce(int dtype, const void* dy, int lddy, const void* x, int ldx, const void* residual, int ldr,
                              const float* scale, const float* shift, const float* mean, const float* invstd,
                              int relu, double* sums, int replicas, int rstride, int64_t pixels, int C, void* stream)
{
    return bn_backward_reduce_impl(dtype, dy, lddy, x, ldx, residual, ldr, scale, shift, mean, invstd, relu, sums, replicas, rstride, pixels, C, nullptr, stream);
}

int saunet_bn_backward_reduce_masked(int dtype, const void* dy, int lddy, const void* x, int ldx, const uint8_t* relu_mask,
                                     const float* scale, const float* shift, const float* mean, const float* invstd,
                                     double* sums, int replicas, int rstride, int64_t pixels, int C, void* stream)
{
    if (!relu_mask) return set_error(SAUNET_BAD_SHAPE, "bn_backward_reduce_masked: no mask");
    return bn_backward_reduce_impl(dtype, dy, lddy, x, ldx, nullptr, 0, scale, shift, mean, invstd, 1, sums, replicas, rstride, pixels, C, relu_mask, stream);
}

int saunet_bn_backward_coeff(int C, const double* sums, int sums_replicas, int sums_rstride, double count, const float* scale, float* A, float* B,
                             float* dgamma, float* dbeta, int training, void* stream)
{
    hipLaunchKernelGGL(bn_bwd_coeff_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, sums, sums_replicas < 1 ? 1 : sums_replicas, sums_rstride,
                       count, scale, A, B, dgamma, dbeta, training);
    SAUNET_CHECK_LAUNCH("bn_bwd_coeff");
    return SAUNET_OK;
}

int saunet_bn_backward_coeff_correct(int dtype, int C, const double* sums, int sums_replicas, int sums_rstride, double count, const float* scale,
                                     const float* A_in, const float* B_in, float* A_out, float* B_out, float* dgamma, float* dbeta,
                                     void* dx, int lddx, const void* x, int ldx, int c_lo, int c_hi,
                                     const float* xhat_scale, const float* xhat_shift, int64_t pixels, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int CW = c_hi - c_lo;
    if (C < 1 || c_lo < 0 || c_hi > C || CW < 1 || CW > 256) return set_error(SAUNET_BAD_SHAPE, "bn_backward_coeff_correct: chunk [%d, %d) of %d channels", c_lo, c_hi, C);
    if (A_in == A_out || B_in == B_out) return set_error(SAUNET_BAD_SHAPE, "bn_backward_coeff_correct: A / B must be ping-pong buffers");
    const bool vec = vec_ok(dtype, CW, {lddx, ldx}, {dx, x});
    int blocks; const int V = vec ? (dtype == SAUNET_BF16 ? 8 : 4) : 1;
    CoeffCorrectArgs a{C, sums, sums_replicas < 1 ? 1 : sums_replicas, sums_rstride, count, scale, A_in, B_in, A_out, B_out, dgamma, dbeta,
                       dx, lddx, x, ldx, c_lo, CW, xhat_scale, xhat_shift, (long)pixels, 0, 0};
    a.rpb = rows_per_block(pixels, CW, V, &blocks);
    a.nb_correct = blocks;
    const int total = blocks + (C + 255) / 256;
#define CALL(TT, VV) hipLaunchKernelGGL((bn_bwd_coeff_correct_kernel<TT, VV>), dim3(total), dim3(256), 0, st, a)
    DISPATCH_TV(dtype, vec, CALL);
#undef CALL
    CHECK_LAUNCH_TV("bn_bwd_coeff_correct", dtype, vec);
    return SAUNET_OK;
}

int saunet_bn_backward_correct_ab(int dtype, const void* d, int ldd, const void* x, int ldx, void* y, int ldy, const double* ab, int ab_replicas,
                                  int ab_rstride, int ab_half, double count, const float* xs, const float* xt, int64_t pixels, int C, void* stream)
{
    return bn_backward_correct_ab(dtype, d, ldd, x, ldx, y, ldy, ab, ab_replicas, ab_rstride, ab_half, count, xs, xt, pixels, C, (hipStream_t)stream);
}

int saunet_bn_backward_coeff_ab(int C, const double* sums, int sums_replicas, int sums_rstride, const float* scale, double* ab, int ab_half,
                                float* dgamma, float* dbeta, void* stream)
{
    if (C < 1 || !sums || !scale || !ab || sums_replicas < 1 || ab_half < C) return set_error(SAUNET_BAD_SHAPE, "bn_backward_coeff_ab: C=%d", C);
    hipLaunchKernelGGL(bn_bwd_coeff_ab_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, sums, sums_replicas, sums_rstride, scale, ab, ab_half,
                       dgamma, dbeta);
    SAUNET_CHECK_LAUNCH("bn_bwd_coeff_ab");
    return SAUNET_OK;
}

int saunet_dense_bn1_grads(const saunet_dense_bn1_list* l, void* stream)
{
    if (!l || l->count < 1 || l->count > SAUNET_DENSE_LAYERS_MAX || l->replicas < 1) return set_error(SAUNET_BAD_SHAPE, "dense_bn1_grads: %d layers", l ? l->count : 0);
    int cmax = 0;
    for (int i = 0; i < l->count; ++i) {
        if (!l->sums[i] || !l->dgamma[i] || !l->dbeta[i] || l->cin[i] < 1) return set_error(SAUNET_BAD_SHAPE, "dense_bn1_grads: layer %d is empty", i);
        if (l->cin[i] > cmax) cmax = l->cin[i];
    }
    hipLaunchKernelGGL(dense_bn1_grads_kernel, dim3((cmax + 255) / 256, l->count), dim3(256), 0, (hipStream_t)stream, *l);
    SAUNET_CHECK_LAUNCH("dense_bn1_grads");
    return SAUNET_OK;
}

int saunet_bn_backward_correct(int dtype, void* dx, int lddx, const void* x, int ldx, const float* A, const float* B,
                               const float* xhat_scale, const float* xhat_shift, int64_t pixels, int C, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const bool vec = vec_ok(dtype, C, {lddx, ldx}, {dx, x});
    int blocks; const int V = vec ? (dtype == SAUNET_BF16 ? 8 : 4) : 1;
    long rpb = rows_per_block(pixels, C, V, &blocks);
#define CALL(TT, VV) hipLaunchKernelGGL((bn_bwd_correct_kernel<TT, VV>), dim3(blocks), dim3(256), 0, st, (TT*)dx, lddx, (const TT*)x, ldx, A, B, xhat_scale, xhat_shift, (long)pixels, C, rpb)
    DISPATCH_TV(dtype, vec, CALL);
#undef CALL
    CHECK_LAUNCH_TV("bn_bwd_correct", dtype, vec);
    return SAUNET_OK;
}

static int bn_backward_apply_impl(int dtype, const void* dy, int lddy, const void* x, int ldx, const void* residual, int ldr,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             int relu, const double* sums, int sums_replicas, int sums_rstride, double count, int training, int accumulate,
                             void* dx, int lddx, void* dres, int lddres, float* dgamma, float* dbeta,
                             int64_t pixels, int C, const unsigned char* mask, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    bool vec = vec_ok(dtype, C, {lddy, ldx, lddx}, {dy, x, dx});
    if (residual) vec = vec && vec_ok(dtype, C, {ldr}, {residual});
    if (dres) vec = vec && vec_ok(dtype, C, {lddres}, {dres});
    if (mask && !(vec && dtype == SAUNET_BF16))
        return set_error(SAUNET_UNSUPPORTED, "bn_backward_apply_masked: bf16, C and strides multiples of 8, 16-byte aligned views");
    BnBwdArgs a{}; a.mask = mask; a.dy = dy; a.lddy = lddy; a.x = x; a.ldx = ldx; a.res = residual; a.ldr = ldr; a.scale = scale; a.shift = shift;
    a.mean = mean; a.invstd = invstd; a.relu = relu; a.sums = (double*)sums; a.sreps = sums_replicas < 1 ? 1 : sums_replicas; a.srstride = sums_rstride;
    a.count = count; a.training = training;
    a.accumulate = accumulate; a.dx = dx; a.lddx = lddx; a.dres = dres; a.lddres = lddres; a.dgamma = dgamma; a.dbeta = dbeta;
    a.P = pixels; a.C = C;
    int blocks; const int V = vec ? (dtype == SAUNET_BF16 ? 8 : 4) : 1;
    a.rpb = rows_per_block(pixels, C, V, &blocks);
#define CALL(TT, VV) hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, VV>), dim3(blocks), dim3(256), sizeof(float) * 2 * C, st, a)
    DISPATCH_TV(dtype, vec, CALL);
#undef CALL
    CHECK_LAUNCH_TV("bn_bwd_apply", dtype, vec);
    return SAUNET_OK;
}

int saunet_bn_backward_apply(int dtype, const void* dy, int lddy, const void* x, int ldx, const void* residual, int ldr,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             int relu, const double* sums, int sums_replicas, int sums_rstride, double count, int training, int accumulate,
                             void* dx, int lddx, void* dres, int lddres, float* dgamma, float* dbeta,
                             int64_t pixels, int C, void* stream)
{
    return bn_backward_apply_impl(dtype, dy, lddy, x, ldx, residual, ldr, scale, shift, mean, invstd, relu, sums, sums_replicas, sums_rstride, count, training,
                                  accumulate, dx, lddx, dres, lddres, dgamma, dbeta, pixels, C, nullptr, stream);
}

int saunet_bn_backward_apply_masked(int dtype, const void* dy, int lddy, const void* x, int ldx, const uint8_t* relu_mask,
                                    const float* scale, const float* shift, const float* mean, const float* invstd,
                                    const double* sums, int sums_replicas, int sums_rstride, double count, int training, int accumulate,
                                    void* dx, int lddx, void* dres, int lddres, float* dgamma, float* dbeta,
                                    int64_t pixels, int C, void* stream)
{
    if (!relu_mask) return set_error(SAUNET_BAD_SHAPE, "bn_backward_apply_masked: no mask");
    return bn_backward_apply_impl(dtype, dy, lddy, x, ldx, nullptr, 0, scale, shift, mean, invstd, 1, sums, sums_replicas, sums_rstride, count, training,
                                  accumulate, dx, lddx, dres, lddres, dgamma, dbeta, pixels, C, relu_mask, stream);
}

static int pool_bn_check(const char* who, int dtype, int N, int H, int W, int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs)
{
    if (N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 1 || C > 2048) return set_error(SAUNET_BAD_SHAPE, "%s: N=%d H=%d W=%d C=%d (even maps, C <= 2048)", who, N, H, W, C);
    if (dtype != SAUNET_BF16 && dtype != SAUNET_F32) return set_error(SAUNET_BAD_DTYPE, "%s: dtype %d", who, dtype);
    if (!vec_ok(dtype, C, lds, ptrs)) return set_error(SAUNET_UNSUPPORTED, "%s: vector path only (C, strides multiples of 8 bf16 / 4 f32 elements, 16-byte aligned views)", who);
    for (const void* p : ptrs) if (!p) return set_error(SAUNET_BAD_SHAPE, "%s: null operand", who);
    return SAUNET_OK;
}

int saunet_bn_relu_avgpool2(int dtype, const void* x, int ldx, const float* scale, const float* shift, void* y, int ldy,
                            int N, int H, int W, int C, void* stream)
{
    if (int rc = pool_bn_check("bn_relu_avgpool2", dtype, N, H, W, C, {ldx, ldy}, {x, y})) return rc;
    if (!scale || !shift) return set_error(SAUNET_BAD_SHAPE, "bn_relu_avgpool2: no coefficients");
    PoolBnArgs a{}; a.x = x; a.ldx = ldx; a.scale = scale; a.shift = shift; a.y = y; a.ldy = ldy; a.Wo = W / 2; a.W = W; a.C = C;
    a.Po = (long)N * (H / 2) * (W / 2);
    int blocks; const int V = dtype == SAUNET_BF16 ? 8 : 4;
    a.rpb = rows_per_block(a.Po * 4, C, V, &blocks); a.rpb = (a.rpb + 3) / 4; blocks = (int)((a.Po + a.rpb - 1) / a.rpb);
    if (dtype == SAUNET_BF16) hipLaunchKernelGGL((bn_relu_avgpool2_fwd_kernel<u16, 8>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((bn_relu_avgpool2_fwd_kernel<float, 4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    CHECK_LAUNCH_TV("bn_relu_avgpool2_fwd", dtype, true);
    return SAUNET_OK;
}

int saunet_bn_relu_avgpool2_backward(int dtype, const void* da, int ldda, const void* x, int ldx, const float* scale, const float* shift,
                                     const float* mean, const float* invstd, int scaled, void* dx, int lddx,
                                     double* sums, int replicas, int rstride, int N, int H, int W, int C, void* stream)
{
    if (int rc = pool_bn_check("bn_relu_avgpool2_backward", dtype, N, H, W, C, {ldda, ldx, lddx}, {da, x, dx})) return rc;
    if (!scale || !shift || !mean || !invstd || !sums) return set_error(SAUNET_BAD_SHAPE, "bn_relu_avgpool2_backward: incomplete arguments");
    PoolBnArgs a{}; a.x = x; a.ldx = ldx; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.da = da; a.ldda = ldda;
    a.dx = dx; a.lddx = lddx; a.scaled = scaled; a.sums = sums; a.sreps = replicas < 1 ? 1 : replicas; a.srstride = rstride;
    a.Wo = W / 2; a.W = W; a.C = C; a.Po = (long)N * (H / 2) * (W / 2);
    int blocks; const int V = dtype == SAUNET_BF16 ? 8 : 4;
    a.rpb = rows_per_block(a.Po * 4, C, V, &blocks); a.rpb = (a.rpb + 3) / 4; blocks = (int)((a.Po + a.rpb - 1) / a.rpb);
    const size_t lds = 2 * (size_t)C * sizeof(double);
    if (dtype == SAUNET_BF16) hipLaunchKernelGGL((bn_relu_avgpool2_bwd_kernel<u16, 8>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((bn_relu_avgpool2_bwd_kernel<float, 4>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, a);
    CHECK_LAUNCH_TV("bn_relu_avgpool2_bwd", dtype, true);
    return SAUNET_OK;
}

}  // extern "C"
