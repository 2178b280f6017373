// Bandwidth kernels of the SAUNet path: bilinear resampling (align_corners=True), 2x2 pooling,
// channel-slice copies (the only "cat"), sigmoid, the gate multiply, and the dual-attention tail
// (global average pool -> SE excitation -> (S+1)*F*se).  All NHWC, float32 or bf16 storage.
// Reference call sites: /root/reference/models/models.py:337-389, models/GSConv.py:53-57,
// models/attention_blocks.py:50-57,165-173,233-238.
#include "common.h"

namespace saunet {

// ------------------------------------------------------------------------------------------ bilinear
struct BilArgs {
    const void* src; void* dst; int N, H, W, C, lds, Ho, Wo, ldd; float sy, sx; int accumulate;
    FastDiv dcv, dwo, dho, dw, dh;     // channel groups, output width/height, input width/height
};

__device__ __forceinline__ void bil_coord(int o, float scale, int in, int& i0, int& i1, float& l1)
{
    float s = scale * (float)o;
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// V consecutive channels (one 16-byte vector, or a single element when V == 1) of one output pixel per thread:
// y = hy*(hx*v00 + lx*v01) + ly*(hx*v10 + lx*v11).  The kernel is write-bound (the source is 4..256x smaller).
template <typename T, int V> __global__ __launch_bounds__(256) void bilinear_fwd_kernel(BilArgs a)
{
    const unsigned CV = a.C / V;
    const unsigned total = (unsigned)a.N * a.Ho * a.Wo * CV;
    const T* x = (const T*)a.src; T* y = (T*)a.dst;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned t = a.dcv.div(i); const int c = (int)(i - t * CV) * V;
        const unsigned t2 = a.dwo.div(t); const int ow = (int)(t - t2 * a.Wo);
        const unsigned n = a.dho.div(t2); const int oh = (int)(t2 - n * a.Ho);
        int y0, y1, x0, x1; float ly, lx;
        bil_coord(oh, a.sy, a.H, y0, y1, ly); bil_coord(ow, a.sx, a.W, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const T* b = x + (size_t)n * a.H * a.W * a.lds + c;
        const T* p00 = b + ((size_t)y0 * a.W + x0) * a.lds; const T* p01 = b + ((size_t)y0 * a.W + x1) * a.lds;
        const T* p10 = b + ((size_t)y1 * a.W + x0) * a.lds; const T* p11 = b + ((size_t)y1 * a.W + x1) * a.lds;
        T* o = y + (size_t)t * a.ldd + c;
        if constexpr (V == 1) {
            float v00 = Elem<T>::load(p00), v01 = Elem<T>::load(p01), v10 = Elem<T>::load(p10), v11 = Elem<T>::load(p11);
            Elem<T>::store(o, hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11));
        } else {
            const u32x4 r00 = *(const u32x4*)p00, r01 = *(const u32x4*)p01, r10 = *(const u32x4*)p10, r11 = *(const u32x4*)p11;
            float f00[V], f01[V], f10[V], f11[V], r[V];
            Vec16<T>::unpack(r00, f00); Vec16<T>::unpack(r01, f01); Vec16<T>::unpack(r10, f10); Vec16<T>::unpack(r11, f11);
#pragma unroll
            for (int j = 0; j < V; ++j) r[j] = hy * (hx * f00[j] + lx * f01[j]) + ly * (hx * f10[j] + lx * f11[j]);
            *(u32x4*)o = Vec16<T>::pack(r);
        }
    }
}

// window of output rows (or columns) whose two sources can include input index i
__device__ __forceinline__ void bil_window(int i, float scale, int on, int& lo, int& hi)
{
    lo = 0; hi = on - 1;
    if (scale > 0.f) { lo = max(0, (int)floorf((i - 1) / scale)); hi = min(on - 1, (int)ceilf((i + 1) / scale)); }
}
__device__ __forceinline__ float bil_weight(int o, float scale, int in, int i)
{
    int i0, i1; float l; bil_coord(o, scale, in, i0, i1, l);
    return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
}

// gather form of the adjoint (deterministic, no atomics, dtype-agnostic): for every INPUT pixel visit the output
// pixels whose two source rows/cols can include it and re-derive their weights exactly.  V channels per thread.
template <typename T, int V> __global__ __launch_bounds__(256) void bilinear_bwd_kernel(BilArgs a)
{
    // here src = dy [N,Ho,Wo,C] (lds), dst = dx [N,H,W,C] (ldd)
    const unsigned CV = a.C / V;
    const unsigned total = (unsigned)a.N * a.H * a.W * CV;
    const T* dy = (const T*)a.src; T* dx = (T*)a.dst;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned t = a.dcv.div(i); const int c = (int)(i - t * CV) * V;
        const unsigned t2 = a.dw.div(t); const int ix = (int)(t - t2 * a.W);
        const unsigned n = a.dh.div(t2); const int iy = (int)(t2 - n * a.H);
        int oy_lo, oy_hi, ox_lo, ox_hi;
        bil_window(iy, a.sy, a.Ho, oy_lo, oy_hi); bil_window(ix, a.sx, a.Wo, ox_lo, ox_hi);
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        const T* b = dy + (size_t)n * a.Ho * a.Wo * a.lds + c;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const float wy = bil_weight(oy, a.sy, a.H, iy);
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const float wx = bil_weight(ox, a.sx, a.W, ix);
                if (wx == 0.f) continue;
                const float wgt = wy * wx;
                const T* q = b + ((size_t)oy * a.Wo + ox) * a.lds;
                if constexpr (V == 1) acc[0] = fmaf(wgt, Elem<T>::load(q), acc[0]);
                else {
                    float f[V];
                    Vec16<T>::unpack(*(const u32x4*)q, f);
#pragma unroll
                    for (int j = 0; j < V; ++j) acc[j] = fmaf(wgt, f[j], acc[j]);
                }
            }
        }
        T* o = dx + (size_t)t * a.ldd + c;
        if constexpr (V == 1) {
            if (a.accumulate) acc[0] += Elem<T>::load(o);
            Elem<T>::store(o, acc[0]);
        } else {
            if (a.accumulate) {
                float f[V];
                Vec16<T>::unpack(*(const u32x4*)o, f);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] += f[j];
            }
            *(u32x4*)o = Vec16<T>::pack(acc);
        }
    }
}

// Same adjoint for few-channel maps under a large zoom (the 1-channel c3/c4/c5 edge maps, 8..32x): one WAVE per
// (input pixel, channel); the lanes share the output window and a wave reduction finishes the sum.
template <typename T> __global__ __launch_bounds__(256) void bilinear_bwd_wave_kernel(BilArgs a)
{
    const unsigned total = (unsigned)a.N * a.H * a.W * a.C;
    const T* dy = (const T*)a.src; T* dx = (T*)a.dst;
    const int lane = threadIdx.x & 63;
    for (unsigned i = blockIdx.x * 4u + (threadIdx.x >> 6); i < total; i += gridDim.x * 4u) {
        const unsigned t = a.dcv.div(i); const int c = (int)(i - t * a.C);
        const unsigned t2 = a.dw.div(t); const int ix = (int)(t - t2 * a.W);
        const unsigned n = a.dh.div(t2); const int iy = (int)(t2 - n * a.H);
        int oy_lo, oy_hi, ox_lo, ox_hi;
        bil_window(iy, a.sy, a.Ho, oy_lo, oy_hi); bil_window(ix, a.sx, a.Wo, ox_lo, ox_hi);
        const int ww = ox_hi - ox_lo + 1, cnt = (oy_hi - oy_lo + 1) * ww;
        const T* b = dy + (size_t)n * a.Ho * a.Wo * a.lds + c;
        float acc = 0.f;
        for (int k = lane; k < cnt; k += 64) {
            const int r = k / ww, oy = oy_lo + r, ox = ox_lo + (k - r * ww);
            const float wgt = bil_weight(oy, a.sy, a.H, iy) * bil_weight(ox, a.sx, a.W, ix);
            acc = fmaf(wgt, Elem<T>::load(b + ((size_t)oy * a.Wo + ox) * a.lds), acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            T* o = dx + (size_t)t * a.ldd + c;
            if (a.accumulate) acc += Elem<T>::load(o);
            Elem<T>::store(o, acc);
        }
    }
}

// ------------------------------------------------------------------------------------------ 2x2 pooling
struct PoolArgs { const void* x; const void* dy; void* out; int N, H, W, C, ldx, lddy, ldo, is_max, accumulate; };

template <typename T> __global__ __launch_bounds__(256) void pool_fwd_kernel(PoolArgs a)
{
    const int Ho = a.H / 2, Wo = a.W / 2;
    const long total = (long)a.N * Ho * Wo * a.C;
    const T* x = (const T*)a.x; T* y = (T*)a.out;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c = (int)(i % a.C); long t = i / a.C;
        int ow = (int)(t % Wo); t /= Wo; int oh = (int)(t % Ho); int n = (int)(t / Ho);
        const T* b = x + (((long)n * a.H + 2 * oh) * a.W + 2 * ow) * a.ldx + c;
        float v0 = Elem<T>::load(b), v1 = Elem<T>::load(b + a.ldx), v2 = Elem<T>::load(b + (long)a.W * a.ldx), v3 = Elem<T>::load(b + (long)(a.W + 1) * a.ldx);
        float v = a.is_max ? fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)) : (v0 + v1 + v2 + v3) * 0.25f;
        Elem<T>::store(y + i / a.C * a.ldo + c, v);
    }
}

template <typename T> __global__ __launch_bounds__(256) void pool_bwd_kernel(PoolArgs a);
// 16-byte forms of the 2x2 pools (a thread owns one chunk of Vec16<T>::N channels of an output pixel, 32-bit index arithmetic): the scalar
// kernels above move 2-4 bytes per lane and instruction behind 64-bit div / mod and ran at half the HBM rate of their tensors
struct PoolVecArgs { PoolArgs a; unsigned chunks; FastDiv dch, dwo, dho; };
template <typename T> __global__ __launch_bounds__(256) void pool_fwd_vec_kernel(PoolVecArgs v)
{
    constexpr int E = Vec16<T>::N;
    const PoolArgs& a = v.a;
    const T* x = (const T*)a.x; T* y = (T*)a.out;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < v.chunks; i += gridDim.x * 256u) {
        const unsigned op = v.dch.div(i), ch = i - op * v.dch.d;
        unsigned t = v.dwo.div(op); const unsigned ow = op - t * v.dwo.d; const unsigned n = v.dho.div(t), oh = t - n * v.dho.d;
        const T* b = x + (((size_t)n * a.H + 2 * oh) * a.W + 2 * ow) * a.ldx + ch * E;
        float f0[E], f1[E], f2[E], f3[E], o[E];
        Vec16<T>::unpack(*(const u32x4*)b, f0); Vec16<T>::unpack(*(const u32x4*)(b + a.ldx), f1);
        Vec16<T>::unpack(*(const u32x4*)(b + (size_t)a.W * a.ldx), f2); Vec16<T>::unpack(*(const u32x4*)(b + (size_t)(a.W + 1) * a.ldx), f3);
#pragma unroll
        for (int j = 0; j < E; ++j) o[j] = a.is_max ? fmaxf(fmaxf(f0[j], f1[j]), fmaxf(f2[j], f3[j])) : (f0[j] + f1[j] + f2[j] + f3[j]) * 0.25f;
        *(u32x4*)(y + (size_t)op * a.ldo + ch * E) = Vec16<T>::pack(o);
    }
}
template <typename T> __global__ __launch_bounds__(256) void pool_bwd_vec_kernel(PoolVecArgs v)
{
    constexpr int E = Vec16<T>::N;
    const PoolArgs& a = v.a;
    const T* x = (const T*)a.x; const T* dy = (const T*)a.dy; T* dx = (T*)a.out;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < v.chunks; i += gridDim.x * 256u) {
        const unsigned op = v.dch.div(i), ch = i - op * v.dch.d;
        unsigned t = v.dwo.div(op); const unsigned ow = op - t * v.dwo.d; const unsigned n = v.dho.div(t), oh = t - n * v.dho.d;
        const size_t off = ((size_t)n * a.H + 2 * oh) * a.W + 2 * ow;
        float g[E], d[4][E];
        Vec16<T>::unpack(*(const u32x4*)(dy + (size_t)op * a.lddy + ch * E), g);
        if (a.is_max) {
            const T* b = x + off * a.ldx + ch * E;
            float f[4][E];
            Vec16<T>::unpack(*(const u32x4*)b, f[0]); Vec16<T>::unpack(*(const u32x4*)(b + a.ldx), f[1]);
            Vec16<T>::unpack(*(const u32x4*)(b + (size_t)a.W * a.ldx), f[2]); Vec16<T>::unpack(*(const u32x4*)(b + (size_t)(a.W + 1) * a.ldx), f[3]);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                int k = 0; float m = f[0][j];
#pragma unroll
                for (int q = 1; q < 4; ++q) if (f[q][j] > m) { m = f[q][j]; k = q; }   // first maximum wins (ATen order)
#pragma unroll
                for (int q = 0; q < 4; ++q) d[q][j] = (q == k) ? g[j] : 0.f;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < E; ++j) d[q][j] = 0.25f * g[j];
        }
        T* o = dx + off * a.ldo + ch * E;
        const size_t offs[4] = {0, (size_t)a.ldo, (size_t)a.W * a.ldo, (size_t)(a.W + 1) * a.ldo};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (a.accumulate) {
                float old[E];
                Vec16<T>::unpack(*(const u32x4*)(o + offs[q]), old);
#pragma unroll
                for (int j = 0; j < E; ++j) d[q][j] += old[j];
            }
            *(u32x4*)(o + offs[q]) = Vec16<T>::pack(d[q]);
        }
    }
}
static inline bool pool_vec_args(PoolVecArgs& v, const PoolArgs& a, int epc)
{
    const long chunks = (long)a.N * (a.H / 2) * (a.W / 2) * (a.C / epc);
    if (a.C % epc || a.ldx % epc || a.ldo % epc || (a.dy && a.lddy % epc) || chunks >= (1L << 32) || (((uintptr_t)a.x | (uintptr_t)a.out | (uintptr_t)a.dy) & 15)) return false;
    v.a = a; v.chunks = (unsigned)chunks;
    v.dch = FastDiv::make((unsigned)(a.C / epc)); v.dwo = FastDiv::make((unsigned)(a.W / 2)); v.dho = FastDiv::make((unsigned)(a.H / 2));
    return true;
}

template <typename T> __global__ __launch_bounds__(256) void pool_bwd_kernel(PoolArgs a)
{
    const int Ho = a.H / 2, Wo = a.W / 2;
    const long total = (long)a.N * Ho * Wo * a.C;
    const T* x = (const T*)a.x; const T* dy = (const T*)a.dy; T* dx = (T*)a.out;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c = (int)(i % a.C); long t = i / a.C;
        int ow = (int)(t % Wo); t /= Wo; int oh = (int)(t % Ho); int n = (int)(t / Ho);
        const long off = (((long)n * a.H + 2 * oh) * a.W + 2 * ow);
        const float g = Elem<T>::load(dy + i / a.C * a.lddy + c);
        float d[4];
        if (a.is_max) {
            const T* b = x + off * a.ldx + c;
            float v[4] = {Elem<T>::load(b), Elem<T>::load(b + a.ldx), Elem<T>::load(b + (long)a.W * a.ldx), Elem<T>::load(b + (long)(a.W + 1) * a.ldx)};
            int k = 0; float m = v[0];
#pragma unroll
            for (int j = 1; j < 4; ++j) if (v[j] > m) { m = v[j]; k = j; }   // first maximum wins (ATen order)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = (j == k) ? g : 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = 0.25f * g;
        }
        T* o = dx + off * a.ldo + c;
        const long offs[4] = {0, (long)a.ldo, (long)a.W * a.ldo, (long)(a.W + 1) * a.ldo};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = d[j];
            if (a.accumulate) v += Elem<T>::load(o + offs[j]);
            Elem<T>::store(o + offs[j], v);
        }
    }
}

// ------------------------------------------------------------------------------------------ im2col (stem conv)
// out[(n,oh,ow)][(kh,kw,c)] = x[n, oh*stride-pad+kh, ow*stride-pad+kw, c]  (0 outside): turns the 7x7 stride-2 stem conv
// on the 8-channel padded image into a K = 392 pointwise GEMM that runs on the tuned 1x1 forward / wgrad kernels
struct Im2colArgs { const void* x; void* out; int N, H, W, C, ldx, Ho, Wo, KH, KW, stride, pad, ldo; };
template <typename T> __global__ __launch_bounds__(256) void im2col_kernel(Im2colArgs a)
{
    constexpr int EPC = 16 / sizeof(T);
    const int cpp = a.C / EPC;                          // 16-byte chunks per pixel
    const long total = (long)a.N * a.Ho * a.Wo * a.KH * a.KW * cpp;
    const T* x = (const T*)a.x; T* o = (T*)a.out;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int ch = (int)(i % cpp); long t = i / cpp;
        int kw = (int)(t % a.KW); t /= a.KW; int kh = (int)(t % a.KH); t /= a.KH;
        int ow = (int)(t % a.Wo); long t2 = t / a.Wo; int oh = (int)(t2 % a.Ho); int n = (int)(t2 / a.Ho);
        int ih = oh * a.stride - a.pad + kh, iw = ow * a.stride - a.pad + kw;
        u32x4 v = {0u, 0u, 0u, 0u};
        if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
            v = *(const u32x4*)(x + (((long)n * a.H + ih) * a.W + iw) * a.ldx + ch * EPC);
        *(u32x4*)(o + t * a.ldo + ((kh * a.KW + kw) * cpp + ch) * EPC) = v;
    }
}

// ------------------------------------------------------------------------------------------ element-wise
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void copy_channels_kernel(const TS* __restrict__ s, int lds, TD* __restrict__ d, int ldd, long P, int C, int acc)
{
    const long total = P * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long p = i / C; int c = (int)(i - p * C);
        float v = Elem<TS>::load(s + p * lds + c);
        if (acc) v += Elem<TD>::load(d + p * ldd + c);
        Elem<TD>::store(d + p * ldd + c, v);
    }
}

// 8 consecutive elements as floats (two 16-byte vectors of f32, one of bf16)
__device__ __forceinline__ void load8(const float* p, float* f) { Vec16<float>::unpack(*(const u32x4*)p, f); Vec16<float>::unpack(*(const u32x4*)(p + 4), f + 4); }
__device__ __forceinline__ void load8(const u16* p, float* f) { Vec16<u16>::unpack(*(const u32x4*)p, f); }
__device__ __forceinline__ void store8(float* p, const float* f) { *(u32x4*)p = Vec16<float>::pack(f); *(u32x4*)(p + 4) = Vec16<float>::pack(f + 4); }
__device__ __forceinline__ void store8(u16* p, const float* f) { *(u32x4*)p = Vec16<u16>::pack(f); }

// vector form: rows are 16-byte aligned on both sides and C % 8 == 0; one thread moves 8 channels
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void copy_channels_vec_kernel(const TS* __restrict__ s, int lds, TD* __restrict__ d, int ldd, unsigned total, FastDiv dcv, int acc)
{
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned p = dcv.div(i); const int c = (int)(i - p * dcv.d) * 8;
        float f[8];
        load8(s + (size_t)p * lds + c, f);
        TD* o = d + (size_t)p * ldd + c;
        if (acc) {
            float g[8];
            load8(o, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += g[j];
        }
        store8(o, f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sigmoid_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, long P, int C)
{
    const long total = P * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long p = i / C; int c = (int)(i - p * C);
        float v = Elem<T>::load(x + p * ldx + c);
        Elem<T>::store(y + p * ldy + c, 1.f / (1.f + expf(-v)));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const T* __restrict__ y, int ldyy, const T* __restrict__ dy, int lddy,
                                                          T* __restrict__ dx, int lddx, long P, int C, int acc)
{
    const long total = P * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long p = i / C; int c = (int)(i - p * C);
        float s = Elem<T>::load(y + p * ldyy + c), g = Elem<T>::load(dy + p * lddy + c);
        float v = g * s * (1.f - s);
        if (acc) v += Elem<T>::load(dx + p * lddx + c);
        Elem<T>::store(dx + p * lddx + c, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gate_mul_fwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ al, T* __restrict__ y, int ldy, long P, int C)
{
    const long total = P * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long p = i / C; int c = (int)(i - p * C);
        Elem<T>::store(y + p * ldy + c, Elem<T>::load(x + p * ldx + c) * (Elem<T>::load(al + p) + 1.f));
    }
}
// one wave per pixel: dx = dy*(alpha+1);  dalpha = sum_c dy*x
template <typename T>
__global__ __launch_bounds__(256) void gate_mul_bwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ al, const T* __restrict__ dy, int lddy,
                                                           T* __restrict__ dx, int lddx, T* __restrict__ dal, long P, int C)
{
    const int lane = threadIdx.x & 63;
    for (long p = blockIdx.x * 4L + (threadIdx.x >> 6); p < P; p += (long)gridDim.x * 4) {
        const float a1 = Elem<T>::load(al + p) + 1.f;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) {
            float g = Elem<T>::load(dy + p * lddy + c);
            s = fmaf(g, Elem<T>::load(x + p * ldx + c), s);
            Elem<T>::store(dx + p * lddx + c, g * a1);
        }
        s = wave_sum(s);
        if (lane == 0) Elem<T>::store(dal + p, s);
    }
}

// ------------------------------------------------------------------------------------------ dual attention tail
// pooled[n][c] = mean over HW.  grid = (N, row splits); float atomics into zeroed pooled (scaled by 1/HW)
template <typename T>
__global__ __launch_bounds__(256) void global_avgpool_kernel(const T* __restrict__ x, int HW, int C, int ld, float* __restrict__ pooled, int rows_per_block)
{
    const int n = blockIdx.x;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, HW);
    const float inv = 1.f / (float)HW;
    for (int cb = 0; cb < C; cb += 256) {
        const int cw = min(256, C - cb), rl = 256 / cw;
        const int c = cb + threadIdx.x % cw, rr = threadIdx.x / cw;
        if (rr >= rl) continue;
        float s = 0.f;
        for (int r = r0 + rr; r < r1; r += rl) s += Elem<T>::load(x + ((long)n * HW + r) * ld + c);
        atomicAdd(&pooled[n * C + c], s * inv);
    }
}

// one block per sample: hidden = relu(W1 pooled + b1); se = sigmoid(W2 hidden + b2)
__global__ __launch_bounds__(256) void se_excite_kernel(const float* __restrict__ pooled, int C, int Cr, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ hidden, float* __restrict__ se)
{
    __shared__ float sh[64];
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* pv = pooled + (long)n * C;
    for (int j = wave; j < Cr; j += 4) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = fmaf(w1[(long)j * C + c], pv[c], s);
        s = wave_sum(s);
        if (lane == 0) { float h = fmaxf(s + b1[j], 0.f); sh[j] = h; hidden[(long)n * Cr + j] = h; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = b2[c];
        for (int j = 0; j < Cr; ++j) s = fmaf(w2[(long)c * Cr + j], sh[j], s);
        se[(long)n * C + c] = 1.f / (1.f + expf(-s));
    }
}

__global__ __launch_bounds__(256) void se_excite_bwd_kernel(const float* __restrict__ pooled, const float* __restrict__ hidden, const float* __restrict__ se,
                                                            const float* __restrict__ dse, int C, int Cr, const float* __restrict__ w1,
                                                            const float* __restrict__ w2, float* __restrict__ dpooled, float* __restrict__ dw1,
                                                            float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2)
{
    extern __shared__ float sm[];   // dz2[C], dz1[Cr]
    float* dz2 = sm; float* dz1 = sm + C;
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* pv = pooled + (long)n * C; const float* hv = hidden + (long)n * Cr;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = se[(long)n * C + c];
        float d = dse[(long)n * C + c] * s * (1.f - s);
        dz2[c] = d;
        atomicAdd(&db2[c], d);
        for (int j = 0; j < Cr; ++j) atomicAdd(&dw2[(long)c * Cr + j], d * hv[j]);
    }
    __syncthreads();
    for (int j = wave; j < Cr; j += 4) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = fmaf(w2[(long)c * Cr + j], dz2[c], s);
        s = wave_sum(s);
        if (lane == 0) { float d = hv[j] > 0.f ? s : 0.f; dz1[j] = d; atomicAdd(&db1[j], d); }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int j = 0; j < Cr; ++j) { s = fmaf(w1[(long)j * C + c], dz1[j], s); atomicAdd(&dw1[(long)j * C + c], dz1[j] * pv[c]); }
        dpooled[(long)n * C + c] = s;
    }
}

// out = (S+1) * F * se[n][c]
template <typename T>
__global__ __launch_bounds__(256) void att_combine_fwd_kernel(const T* __restrict__ F, int ldf, const T* __restrict__ S, const float* __restrict__ se,
                                                              T* __restrict__ out, int ldo, int HW, int C, long total)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long p = i / C; int c = (int)(i - p * C); int n = (int)(p / HW);
        float v = (Elem<T>::load(S + p) + 1.f) * Elem<T>::load(F + p * ldf + c) * se[(long)n * C + c];
        Elem<T>::store(out + p * ldo + c, v);
    }
}
// 16-byte vectorised forms (bf16, C % 8 == 0, C <= 512, aligned rows): a thread owns one 8-channel chunk of a pixel.  The scalar kernels move
// 2 bytes per lane and instruction and ran at 35 % (forward) / 40 % (backward) of the HBM rate of their tensors.
__global__ __launch_bounds__(256) void att_combine_fwd_vec_kernel(const u16* __restrict__ F, int ldf, const u16* __restrict__ S, const float* __restrict__ se,
                                                                  u16* __restrict__ out, int ldo, int HW, int CH, long chunks)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < chunks; i += (long)gridDim.x * 256) {
        const long p = i / CH; const int ch = (int)(i - p * CH); const int n = (int)(p / HW);
        const float s1 = Elem<u16>::load(S + p) + 1.f;
        float f[8];
        Vec16<u16>::unpack(*(const u32x4*)(F + p * ldf + ch * 8), f);
        const float* e = se + (long)n * CH * 8 + ch * 8;
        const f32x4 e0 = *(const f32x4*)e, e1 = *(const f32x4*)(e + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] *= s1 * e0[j]; f[4 + j] *= s1 * e1[j]; }
        *(u32x4*)(out + p * ldo + ch * 8) = Vec16<u16>::pack(f);
    }
}
// block = (sample n, pixel range).  CH = C / 8 chunks per pixel (a power of two <= 64): the 256 threads cover 256 / CH pixels per step, a thread
// keeps the dse partial of ITS chunk in 8 registers; dS = sum over the CH lanes of a pixel (xor tree); the per-chunk partials of the lanes of a
// wave are folded by an xor tree over the pixel lanes, then one slot per wave in LDS, summed in wave order: no float atomics on LDS.
__global__ __launch_bounds__(256) void att_combine_bwd_vec_kernel(const u16* __restrict__ F, int ldf, const u16* __restrict__ S, const float* __restrict__ se,
                                                                  const u16* __restrict__ dout, int lddo, u16* __restrict__ dF, int lddf, u16* __restrict__ dS,
                                                                  float* __restrict__ dse, int HW, int CH, int rows_per_block)
{
    extern __shared__ float s_slot[];  // [4][C]
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = CH * 8, ch = threadIdx.x % CH, sub = threadIdx.x / CH, ppi = 256 / CH;      // pixels per iteration
    const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, HW);
    float e[8], part[8];
    {
        const float* ev = se + (long)n * C + ch * 8;
        const f32x4 e0 = *(const f32x4*)ev, e1 = *(const f32x4*)(ev + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { e[j] = e0[j]; e[4 + j] = e1[j]; part[j] = 0.f; part[4 + j] = 0.f; }
    }
    for (int rb = r0; rb < r1; rb += ppi) {
        const int r = rb + sub; const bool live = r < r1;
        const long p = (long)n * HW + (live ? r : r0);
        const float s1 = Elem<u16>::load(S + p) + 1.f;
        float g[8], f[8], o[8];
        Vec16<u16>::unpack(*(const u32x4*)(dout + p * lddo + ch * 8), g);
        Vec16<u16>::unpack(*(const u32x4*)(F + p * ldf + ch * 8), f);
        float ds = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (!live) g[j] = 0.f;
            o[j] = g[j] * s1 * e[j];
            ds = fmaf(g[j] * f[j], e[j], ds);
            part[j] = fmaf(g[j] * s1, f[j], part[j]);
        }
        if (live) *(u32x4*)(dF + p * lddf + ch * 8) = Vec16<u16>::pack(o);
        for (int off = 1; off < CH; off <<= 1) ds += __shfl_xor(ds, off, 64);     // the CH lanes of a pixel are contiguous and CH-aligned
        if (live && ch == 0) Elem<u16>::store(dS + p, ds);
    }
    // lanes of a wave with the same chunk: lane = ch + CH * k
    for (int off = CH; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[j] += __shfl_xor(part[j], off, 64);
    }
    if (lane < CH) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s_slot[wave * C + ch * 8 + j] = part[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(&dse[(long)n * C + c], (s_slot[c] + s_slot[C + c]) + (s_slot[2 * C + c] + s_slot[3 * C + c]));
}

// block = (sample n, row range): dF = dout*(S+1)*se ; dS[p] = sum_c dout*F*se ; dse[n][c] += sum_p dout*(S+1)*F
template <typename T>
__global__ __launch_bounds__(256) void att_combine_bwd_kernel(const T* __restrict__ F, int ldf, const T* __restrict__ S, const float* __restrict__ se,
                                                              const T* __restrict__ dout, int lddo, T* __restrict__ dF, int lddf, T* __restrict__ dS,
                                                              float* __restrict__ dse, int HW, int C, int rows_per_block)
{
    extern __shared__ float s_dse[];  // [C]
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < C; c += 256) s_dse[c] = 0.f;
    __syncthreads();
    const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, HW);
    const float* sev = se + (long)n * C;
    // a wave walks rows; lanes walk channels; per-lane partial dse kept in registers for up to 16 channel strips
    float part[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) part[k] = 0.f;
    for (int r = r0 + wave; r < r1; r += 4) {
        const long p = (long)n * HW + r;
        const float s1 = Elem<T>::load(S + p) + 1.f;
        float ds = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                float g = Elem<T>::load(dout + p * lddo + c), f = Elem<T>::load(F + p * ldf + c), e = sev[c];
                Elem<T>::store(dF + p * lddf + c, g * s1 * e);
                ds = fmaf(g * f, e, ds);
                part[k] = fmaf(g * s1, f, part[k]);
            }
        }
        ds = wave_sum(ds);
        if (lane == 0) Elem<T>::store(dS + p, ds);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int c = lane + 64 * k; if (c < C) atomicAdd(&s_dse[c], part[k]); }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(&dse[(long)n * C + c], s_dse[c]);
}

template <typename T>
__global__ __launch_bounds__(256) void add_pooled_grad_kernel(T* __restrict__ dF, int lddf, const float* __restrict__ dpooled, int HW, int C, long total)
{
    const float inv = 1.f / (float)HW;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long p = i / C; int c = (int)(i - p * C); int n = (int)(p / HW);
        T* o = dF + p * lddf + c;
        Elem<T>::store(o, Elem<T>::load(o) + dpooled[(long)n * C + c] * inv);
    }
}

struct PooledVecArgs { void* dF; const float* dpooled; unsigned chunks; int lddf, C; float inv; FastDiv dch, dhw; };
template <typename T> __global__ __launch_bounds__(256) void add_pooled_grad_vec_kernel(PooledVecArgs v)
{
    constexpr int E = Vec16<T>::N;
    T* dF = (T*)v.dF;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < v.chunks; i += gridDim.x * 256u) {
        const unsigned p = v.dch.div(i), ch = i - p * v.dch.d, n = v.dhw.div(p);
        T* o = dF + (size_t)p * v.lddf + ch * E;
        float f[E];
        Vec16<T>::unpack(*(const u32x4*)o, f);
        const float* d = v.dpooled + (size_t)n * v.C + ch * E;
#pragma unroll
        for (int j = 0; j < E; ++j) f[j] = fmaf(d[j], v.inv, f[j]);
        *(u32x4*)o = Vec16<T>::pack(f);
    }
}

static inline int grid_for(long total) { long b = (total + 255) / 256; if (b > 8192) b = 8192; if (b < 1) b = 1; return (int)b; }
// the 16-byte kernels: bf16, 8 .. 64 chunks per pixel (a power of two), aligned rows
static inline bool att_vec_ok(int dtype, int C, int ld_a, int ld_b, const void* a, const void* b)
{
    const int ch = C / 8;
    return dtype == SAUNET_BF16 && C % 8 == 0 && ch >= 1 && ch <= 64 && (ch & (ch - 1)) == 0 && ld_a % 8 == 0 && ld_b % 8 == 0 && !(((uintptr_t)a | (uintptr_t)b) & 15);
}

}  // namespace saunet

using namespace saunet;

#define DISPATCH_T(dtype, CALL)                                            \
    do {                                                                   \
        if ((dtype) == SAUNET_F32) { CALL(float); }                        \
        else if ((dtype) == SAUNET_BF16) { CALL(u16); }                    \
        else return set_error(SAUNET_BAD_DTYPE, "dtype %d", (dtype));      \
    } while (0)

extern "C" {

static int bil_args(BilArgs& a, const void* src, void* dst, int N, int H, int W, int C, int lds, int Ho, int Wo, int ldd, int accumulate, int V, bool over_input)
{
    const long total = (long)N * (over_input ? (long)H * W : (long)Ho * Wo) * (C / V);
    if (total >= (1L << 32)) return set_error(SAUNET_UNSUPPORTED, "bilinear: %ld work items exceed 32-bit indexing", total);
    a = BilArgs{src, dst, N, H, W, C, lds, Ho, Wo, ldd, Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f, accumulate,
                FastDiv::make((unsigned)(C / V)), FastDiv::make((unsigned)Wo), FastDiv::make((unsigned)Ho), FastDiv::make((unsigned)W), FastDiv::make((unsigned)H)};
    return SAUNET_OK;
}

int saunet_bilinear_forward(int dtype, const void* x, int N, int H, int W, int C, int ldx, void* y, int Ho, int Wo, int ldy, void* stream)
{
    const int epc = dtype == SAUNET_BF16 ? 8 : 4;
    const bool vec = C % epc == 0 && ldx % epc == 0 && ldy % epc == 0 && !(((uintptr_t)x | (uintptr_t)y) & 15);
    BilArgs a;
    if (int rc = bil_args(a, x, y, N, H, W, C, ldx, Ho, Wo, ldy, 0, vec ? epc : 1, false)) return rc;
    const long total = (long)N * Ho * Wo * (vec ? C / epc : C);
    long blocks = (total + 255) / 256; if (blocks > 32768) blocks = 32768; if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream; dim3 g((unsigned)blocks);
    if (dtype == SAUNET_F32) { if (vec) hipLaunchKernelGGL((bilinear_fwd_kernel<float, 4>), g, dim3(256), 0, st, a); else hipLaunchKernelGGL((bilinear_fwd_kernel<float, 1>), g, dim3(256), 0, st, a); }
    else if (dtype == SAUNET_BF16) { if (vec) hipLaunchKernelGGL((bilinear_fwd_kernel<u16, 8>), g, dim3(256), 0, st, a); else hipLaunchKernelGGL((bilinear_fwd_kernel<u16, 1>), g, dim3(256), 0, st, a); }
    else return set_error(SAUNET_BAD_DTYPE, "dtype %d", dtype);
    SAUNET_CHECK_LAUNCH("bilinear_fwd");
    return SAUNET_OK;
}

int saunet_bilinear_backward(int dtype, const void* dy, int N, int Ho, int Wo, int C, int lddy, void* dx, int H, int W, int lddx, int accumulate, void* stream)
{
    const int epc = dtype == SAUNET_BF16 ? 8 : 4;
    const bool vec = C % epc == 0 && lddy % epc == 0 && lddx % epc == 0 && !(((uintptr_t)dy | (uintptr_t)dx) & 15);
    // outputs gathered per input element; few channels under a large zoom leave too few threads -> one wave per element
    const long window = ((long)(2 * ((Ho + H - 1) / H) + 1)) * (2 * ((Wo + W - 1) / W) + 1);
    const bool wave = !vec && window >= 64;
    BilArgs a;
    if (int rc = bil_args(a, dy, dx, N, H, W, C, lddy, Ho, Wo, lddx, accumulate, vec ? epc : 1, true)) return rc;
    const long total = (long)N * H * W * (vec ? C / epc : C);
    hipStream_t st = (hipStream_t)stream;
    if (wave) {
        long blocks = (total + 3) / 4; if (blocks > 32768) blocks = 32768;
#define CALL(TT) hipLaunchKernelGGL(bilinear_bwd_wave_kernel<TT>, dim3((unsigned)blocks), dim3(256), 0, st, a)
        DISPATCH_T(dtype, CALL);
#undef CALL
    }
    else if (dtype == SAUNET_F32) { if (vec) hipLaunchKernelGGL((bilinear_bwd_kernel<float, 4>), dim3(grid_for(total)), dim3(256), 0, st, a); else hipLaunchKernelGGL((bilinear_bwd_kernel<float, 1>), dim3(grid_for(total)), dim3(256), 0, st, a); }
    else if (dtype == SAUNET_BF16) { if (vec) hipLaunchKernelGGL((bilinear_bwd_kernel<u16, 8>), dim3(grid_for(total)), dim3(256), 0, st, a); else hipLaunchKernelGGL((bilinear_bwd_kernel<u16, 1>), dim3(grid_for(total)), dim3(256), 0, st, a); }
    else return set_error(SAUNET_BAD_DTYPE, "dtype %d", dtype);
    SAUNET_CHECK_LAUNCH("bilinear_bwd");
    return SAUNET_OK;
}

int saunet_im2col(int dtype, const void* x, int N, int H, int W, int C, int ldx, int KH, int KW, int stride, int pad, void* out, int ldo, void* stream)
{
    const int epc = dtype == SAUNET_BF16 ? 8 : 4;
    if (C % epc || ldx % epc || ldo % epc || (((uintptr_t)x | (uintptr_t)out) & 15)) return set_error(SAUNET_BAD_ALIGN, "im2col: 16-byte channel chunks required");
    Im2colArgs a{x, out, N, H, W, C, ldx, (H + 2 * pad - KH) / stride + 1, (W + 2 * pad - KW) / stride + 1, KH, KW, stride, pad, ldo};
    const long total = (long)N * a.Ho * a.Wo * KH * KW * (C / epc);
#define CALL(TT) hipLaunchKernelGGL(im2col_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a)
    DISPATCH_T(dtype, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("im2col");
    return SAUNET_OK;
}

int saunet_pool2x2_forward(int dtype, int is_max, const void* x, int N, int H, int W, int C, int ldx, void* y, int ldy, void* stream)
{
    if ((H | W) & 1) return set_error(SAUNET_BAD_SHAPE, "pool2x2: odd size %dx%d", H, W);
    PoolArgs a{x, nullptr, y, N, H, W, C, ldx, 0, ldy, is_max, 0};
    const long total = (long)N * (H / 2) * (W / 2) * C;
    PoolVecArgs pv;
    if ((dtype == SAUNET_BF16 || dtype == SAUNET_F32) && pool_vec_args(pv, a, dtype == SAUNET_BF16 ? 8 : 4)) {
        const dim3 g(grid_for(pv.chunks));
        if (dtype == SAUNET_BF16) hipLaunchKernelGGL(pool_fwd_vec_kernel<u16>, g, dim3(256), 0, (hipStream_t)stream, pv);
        else hipLaunchKernelGGL(pool_fwd_vec_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, pv);
        SAUNET_CHECK_LAUNCH("pool2x2_forward");
        return SAUNET_OK;
    }
#define CALL(TT) hipLaunchKernelGGL(pool_fwd_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a)
    DISPATCH_T(dtype, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("pool2x2_forward");
    return SAUNET_OK;
}

int saunet_pool2x2_backward(int dtype, int is_max, const void* x, const void* dy, int N, int H, int W, int C, int ldx, int lddy, void* dx, int lddx, int accumulate, void* stream)
{
    if ((H | W) & 1) return set_error(SAUNET_BAD_SHAPE, "pool2x2: odd size %dx%d", H, W);
    PoolArgs a{x, dy, dx, N, H, W, C, ldx, lddy, lddx, is_max, accumulate};
    const long total = (long)N * (H / 2) * (W / 2) * C;
    PoolVecArgs pv;
    if ((dtype == SAUNET_BF16 || dtype == SAUNET_F32) && pool_vec_args(pv, a, dtype == SAUNET_BF16 ? 8 : 4)) {
        const dim3 g(grid_for(pv.chunks));
        if (dtype == SAUNET_BF16) hipLaunchKernelGGL(pool_bwd_vec_kernel<u16>, g, dim3(256), 0, (hipStream_t)stream, pv);
        else hipLaunchKernelGGL(pool_bwd_vec_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, pv);
        SAUNET_CHECK_LAUNCH("pool2x2_backward");
        return SAUNET_OK;
    }
#define CALL(TT) hipLaunchKernelGGL(pool_bwd_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a)
    DISPATCH_T(dtype, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("pool2x2_backward");
    return SAUNET_OK;
}

int saunet_copy_channels(int dtype_src, int dtype_dst, const void* src, int lds, void* dst, int ldd, int64_t pixels, int C, int accumulate, void* stream)
{
    const long total = pixels * C;
    hipStream_t st = (hipStream_t)stream; dim3 g(grid_for(total)), b(256);
    if ((dtype_src != SAUNET_F32 && dtype_src != SAUNET_BF16) || (dtype_dst != SAUNET_F32 && dtype_dst != SAUNET_BF16))
        return set_error(SAUNET_BAD_DTYPE, "copy_channels: dtypes %d %d", dtype_src, dtype_dst);
    const int as = dtype_src == SAUNET_BF16 ? 8 : 4, ad = dtype_dst == SAUNET_BF16 ? 8 : 4;
    const bool vec = C % 8 == 0 && lds % as == 0 && ldd % ad == 0 && !(((uintptr_t)src | (uintptr_t)dst) & 15) && total / 8 < (1L << 32);
    if (vec) {
        const unsigned tv = (unsigned)(total / 8);
        long blocks = ((long)tv + 255) / 256; if (blocks > 32768) blocks = 32768; if (blocks < 1) blocks = 1;
        dim3 gv((unsigned)blocks);
        const FastDiv dcv = FastDiv::make((unsigned)(C / 8));
        if (dtype_src == SAUNET_F32 && dtype_dst == SAUNET_F32) hipLaunchKernelGGL((copy_channels_vec_kernel<float, float>), gv, b, 0, st, (const float*)src, lds, (float*)dst, ldd, tv, dcv, accumulate);
        else if (dtype_src == SAUNET_F32) hipLaunchKernelGGL((copy_channels_vec_kernel<float, u16>), gv, b, 0, st, (const float*)src, lds, (u16*)dst, ldd, tv, dcv, accumulate);
        else if (dtype_dst == SAUNET_F32) hipLaunchKernelGGL((copy_channels_vec_kernel<u16, float>), gv, b, 0, st, (const u16*)src, lds, (float*)dst, ldd, tv, dcv, accumulate);
        else hipLaunchKernelGGL((copy_channels_vec_kernel<u16, u16>), gv, b, 0, st, (const u16*)src, lds, (u16*)dst, ldd, tv, dcv, accumulate);
    }
    else if (dtype_src == SAUNET_F32 && dtype_dst == SAUNET_F32) hipLaunchKernelGGL((copy_channels_kernel<float, float>), g, b, 0, st, (const float*)src, lds, (float*)dst, ldd, (long)pixels, C, accumulate);
    else if (dtype_src == SAUNET_F32) hipLaunchKernelGGL((copy_channels_kernel<float, u16>), g, b, 0, st, (const float*)src, lds, (u16*)dst, ldd, (long)pixels, C, accumulate);
    else if (dtype_dst == SAUNET_F32) hipLaunchKernelGGL((copy_channels_kernel<u16, float>), g, b, 0, st, (const u16*)src, lds, (float*)dst, ldd, (long)pixels, C, accumulate);
    else hipLaunchKernelGGL((copy_channels_kernel<u16, u16>), g, b, 0, st, (const u16*)src, lds, (u16*)dst, ldd, (long)pixels, C, accumulate);
    SAUNET_CHECK_LAUNCH("copy_channels");
    return SAUNET_OK;
}

int saunet_sigmoid_forward(int dtype, const void* x, int ldx, void* y, int ldy, int64_t pixels, int C, void* stream)
{
    const long total = pixels * C;
#define CALL(TT) hipLaunchKernelGGL(sigmoid_fwd_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const TT*)x, ldx, (TT*)y, ldy, (long)pixels, C)
    DISPATCH_T(dtype, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("sigmoid_forward");
    return SAUNET_OK;
}

int saunet_sigmoid_backward(int dtype, const void* y, int ldyy, const void* dy, int lddy, void* dx, int lddx, int64_t pixels, int C, int accumulate, void* stream)
{
    const long total = pixels * C;
#define CALL(TT) hipLaunchKernelGGL(sigmoid_bwd_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const TT*)y, ldyy, (const TT*)dy, lddy, (TT*)dx, lddx, (long)pixels, C, accumulate)
    DISPATCH_T(dtype, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("sigmoid_backward");
    return SAUNET_OK;
}

int saunet_gate_mul_forward(int dtype, const void* x, int ldx, const void* alpha, void* y, int ldy, int64_t pixels, int C, void* stream)
{
    const long total = pixels * C;
#define CALL(TT) hipLaunchKernelGGL(gate_mul_fwd_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const TT*)x, ldx, (const TT*)alpha, (TT*)y, ldy, (long)pixels, C)
    DISPATCH_T(dtype, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("gate_mul_forward");
    return SAUNET_OK;
}

int saunet_gate_mul_backward(int dtype, const void* x, int ldx, const void* alpha, const void* dy, int lddy,
                             void* dx, int lddx, void* dalpha, int64_t pixels, int C, void* stream)
{
    long blocks = (pixels + 3) / 4; if (blocks > 16384) blocks = 16384;
#define CALL(TT) hipLaunchKernelGGL(gate_mul_bwd_kernel<TT>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const TT*)x, ldx, (const TT*)alpha, (const TT*)dy, lddy, (TT*)dx, lddx, (TT*)dalpha, (long)pixels, C)
    DISPATCH_T(dtype, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("gate_mul_backward");
    return SAUNET_OK;
}

int saunet_global_avgpool(int dtype, const void* x, int N, int HW, int C, int ldx, float* pooled, void* stream)
{
    int splits = (HW + 255) / 256; if (splits > 64) splits = 64;
    int rpb = (HW + splits - 1) / splits; splits = (HW + rpb - 1) / rpb;
    if (hipMemsetAsync(pooled, 0, sizeof(float) * (size_t)N * C, (hipStream_t)stream) != hipSuccess) return set_error(SAUNET_LAUNCH_FAILED, "avgpool memset");
#define CALL(TT) hipLaunchKernelGGL(global_avgpool_kernel<TT>, dim3(N, splits), dim3(256), 0, (hipStream_t)stream, (const TT*)x, HW, C, ldx, pooled, rpb)
    DISPATCH_T(dtype, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("global_avgpool");
    return SAUNET_OK;
}

int saunet_se_excite(const float* pooled, int N, int C, int Cr, const float* w1, const float* b1,
                     const float* w2, const float* b2, float* hidden, float* se, void* stream)
{
    if (Cr > 64) return set_error(SAUNET_UNSUPPORTED, "se_excite: reduced channels %d > 64", Cr);
    hipLaunchKernelGGL(se_excite_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, pooled, C, Cr, w1, b1, w2, b2, hidden, se);
    SAUNET_CHECK_LAUNCH("se_excite");
    return SAUNET_OK;
}

int saunet_se_excite_backward(const float* pooled, const float* hidden, const float* se, const float* dse, int N, int C, int Cr,
                              const float* w1, const float* w2, float* dpooled, float* dw1, float* db1, float* dw2, float* db2, void* stream)
{
    hipLaunchKernelGGL(se_excite_bwd_kernel, dim3(N), dim3(256), sizeof(float) * (C + Cr), (hipStream_t)stream, pooled, hidden, se, dse, C, Cr, w1, w2,
                       dpooled, dw1, db1, dw2, db2);
    SAUNET_CHECK_LAUNCH("se_excite_backward");
    return SAUNET_OK;
}

int saunet_att_combine_forward(int dtype, const void* F, int ldf, const void* S, const float* se, void* out, int ldo, int N, int HW, int C, void* stream)
{
    const long total = (long)N * HW * C;
    if (att_vec_ok(dtype, C, ldf, ldo, F, out)) {
        hipLaunchKernelGGL(att_combine_fwd_vec_kernel, dim3(grid_for(total / 8)), dim3(256), 0, (hipStream_t)stream, (const u16*)F, ldf, (const u16*)S, se, (u16*)out, ldo,
                           HW, C / 8, total / 8);
        SAUNET_CHECK_LAUNCH("att_combine_forward");
        return SAUNET_OK;
    }
#define CALL(TT) hipLaunchKernelGGL(att_combine_fwd_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const TT*)F, ldf, (const TT*)S, se, (TT*)out, ldo, HW, C, total)
    DISPATCH_T(dtype, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("att_combine_forward");
    return SAUNET_OK;
}

int saunet_att_combine_backward(int dtype, const void* F, int ldf, const void* S, const float* se, const void* dout, int lddo,
                                void* dF, int lddf, void* dS, float* dse, int N, int HW, int C, void* stream)
{
    if (C > 1024) return set_error(SAUNET_UNSUPPORTED, "att_combine_backward: C=%d > 1024", C);
    int splits = (HW + 63) / 64; if (splits > 128) splits = 128;
    int rpb = (HW + splits - 1) / splits; splits = (HW + rpb - 1) / rpb;
    if (hipMemsetAsync(dse, 0, sizeof(float) * (size_t)N * C, (hipStream_t)stream) != hipSuccess) return set_error(SAUNET_LAUNCH_FAILED, "att memset");
    if (att_vec_ok(dtype, C, ldf, lddo, F, dout) && lddf % 8 == 0 && !((uintptr_t)dF & 15)) {
        const int CH = C / 8, ppi = 256 / CH;
        int vsplits = (HW + 4 * ppi - 1) / (4 * ppi); if (vsplits > 64) vsplits = 64; if (vsplits < 1) vsplits = 1;      // >= 4 iterations per block
        int vrpb = (HW + vsplits - 1) / vsplits; vrpb = (vrpb + ppi - 1) / ppi * ppi; vsplits = (HW + vrpb - 1) / vrpb;
        hipLaunchKernelGGL(att_combine_bwd_vec_kernel, dim3(N, vsplits), dim3(256), sizeof(float) * 4 * C, (hipStream_t)stream, (const u16*)F, ldf, (const u16*)S, se,
                           (const u16*)dout, lddo, (u16*)dF, lddf, (u16*)dS, dse, HW, CH, vrpb);
        SAUNET_CHECK_LAUNCH("att_combine_backward");
        return SAUNET_OK;
    }
#define CALL(TT) hipLaunchKernelGGL(att_combine_bwd_kernel<TT>, dim3(N, splits), dim3(256), sizeof(float) * C, (hipStream_t)stream, (const TT*)F, ldf, (const TT*)S, se, (const TT*)dout, lddo, (TT*)dF, lddf, (TT*)dS, dse, HW, C, rpb)
    DISPATCH_T(dtype, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("att_combine_backward");
    return SAUNET_OK;
}

int saunet_add_pooled_grad(int dtype, void* dF, int lddf, const float* dpooled, int N, int HW, int C, void* stream)
{
    const long total = (long)N * HW * C;
    const int epc = dtype == SAUNET_BF16 ? 8 : 4;
    if ((dtype == SAUNET_BF16 || dtype == SAUNET_F32) && C % epc == 0 && lddf % epc == 0 && !((uintptr_t)dF & 15) && total / epc < (1L << 32)) {
        PooledVecArgs v{dF, dpooled, (unsigned)(total / epc), lddf, C, 1.f / (float)HW, FastDiv::make((unsigned)(C / epc)), FastDiv::make((unsigned)HW)};
        const dim3 g(grid_for(v.chunks));
        if (dtype == SAUNET_BF16) hipLaunchKernelGGL(add_pooled_grad_vec_kernel<u16>, g, dim3(256), 0, (hipStream_t)stream, v);
        else hipLaunchKernelGGL(add_pooled_grad_vec_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, v);
        SAUNET_CHECK_LAUNCH("add_pooled_grad");
        return SAUNET_OK;
    }
#define CALL(TT) hipLaunchKernelGGL(add_pooled_grad_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (TT*)dF, lddf, dpooled, HW, C, total)
    DISPATCH_T(dtype, CALL);
#undef CALL
    SAUNET_CHECK_LAUNCH("add_pooled_grad");
    return SAUNET_OK;
}

}  // extern "C"
