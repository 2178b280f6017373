// C-ABI front for the convolution family: weight packing, dispatch between the MFMA implicit-GEMM
// kernels (conv_igemm.hip) and the small-channel pointwise kernels below, bias gradient.
#include "common.h"
#include <string.h>
#include "mma_tiles.h"

namespace saunet {

static thread_local char g_err[512] = "";
static thread_local char g_launches[2][256] = {"", ""};
static thread_local int g_launch_cur = 0;
void note_launch(const char* name)
{
    char* b = g_launches[g_launch_cur];
    const size_t used = strlen(b), n = strlen(name);
    if (used + n + 2 < sizeof(g_launches[0])) { if (used) b[used] = '+'; memcpy(b + used + (used ? 1 : 0), name, n + 1); }
}
int set_error(int code, const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}

bool dense_dgrad_supported(const saunet_conv_desc* d, const float* bias, const float* ps, const saunet_bn_epilogue* epi);
int dense_dgrad_forward(const saunet_conv_desc* d, const void* x, const void* w, void* y, const saunet_bn_epilogue* epi, hipStream_t st);
bool dense_dgrad3_supported(const saunet_conv_desc* d, const float* bias, const float* ps, const saunet_bn_epilogue* epi);
int dense_dgrad3_forward(const saunet_conv_desc* d, const void* x, const void* w, void* y, const saunet_bn_epilogue* epi, hipStream_t st);
bool igemm_supported(const saunet_conv_desc* d);
int igemm_forward(const saunet_conv_desc* d, const void* x, const void* w, const float* bias, const float* ps, const float* psh,
                  void* y, double* ssum, double* ssq, const saunet_bn_epilogue* epi, hipStream_t st, const saunet_bn_prologue* bnp = nullptr);
int igemm_wgrad(const saunet_conv_desc* d, const void* x, const void* dy, const float* ps, const float* psh, float* dw, hipStream_t st);
bool tile_fwd_supported(const saunet_conv_desc* d);
bool tile_fwd_accumulate_supported(const saunet_conv_desc* d);
bool mm_fwd_supported(const saunet_conv_desc* d, const void* x, const void* w, const void* y, const float* ps, const saunet_bn_epilogue* epi);
bool mm_convt_supported(const saunet_conv_desc* d, const void* x, const void* w, const void* y, const float* ps, const saunet_bn_epilogue* epi);
int mm_convt_forward(const saunet_conv_desc* d, const void* x, const void* w, const float* bias, void* y, double* ssum, double* ssq, hipStream_t st);
int mm_forward(const saunet_conv_desc* d, const void* x, const void* w, const float* bias, void* y, double* ssum, double* ssq, hipStream_t st);
int64_t mm_forward_workspace(const saunet_conv_desc* d);
bool dense_conv2_small_supported(const saunet_conv_desc* d, const void* x, const void* w, const void* y, const float* bias);
int dense_conv2_small_forward(const saunet_conv_desc* d, const void* x, const void* w, void* y, double* ssum, double* ssq, const saunet_bn_prologue* bnp,
                              hipStream_t st);
bool dense_conv1_small_supported(const saunet_conv_desc* d, const void* x, const void* w, const void* y, const float* bias);
int dense_conv1_small_forward(const saunet_conv_desc* d, const void* x, const void* w, void* y, double* ssum, double* ssq, const saunet_bn_prologue* bnp,
                              hipStream_t st);
bool tile_wgrad_supported(const saunet_conv_desc* d);
bool tile_wgrad_unaligned_supported(const saunet_conv_desc* d);
// MFMA tile wgrad with scalar staging for odd channel counts: correct, but measured SLOWER than pointwise_wgrad_kernel
// on the shape-stream layers (scalar 2-byte global loads dominate) -> kept off until the staging is made cooperative
constexpr bool kUnalignedTileWgrad = false;
int tile_wgrad(const saunet_conv_desc* d, const void* x, const void* dy, const float* ps, const float* psh, float* dw,
               void* ws, size_t ws_bytes, size_t* need, bool aligned, hipStream_t st, saunet_wgrad_pending* pend = nullptr);
int wgrad_reduce_multi(const saunet_wgrad_reduce_list* l, hipStream_t st);
bool tile_wgrad_convt_supported(const saunet_conv_desc* d);
int tile_wgrad_convt(const saunet_conv_desc* d, const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, size_t* need, hipStream_t st,
                     saunet_wgrad_pending* pend);
bool tile_wgrad_grouped_supported(const saunet_wgrad_group* s);
int tile_wgrad_grouped(const saunet_wgrad_group* s, void* ws, size_t ws_bytes, size_t* need, hipStream_t st);
int tile_forward(const saunet_conv_desc* d, const void* x, const void* w, const float* bias, const float* ps, const float* psh,
                 void* y, double* ssum, double* ssq, const saunet_bn_epilogue* epi, hipStream_t st, const saunet_bn_prologue* bnp = nullptr);
int bn_prologue_finalize(const saunet_bn_prologue* p, int Cin, hipStream_t st);

// ---------------------------------------------------------------------------------------- packing
template <typename T>
__global__ void pack_weight_kernel(int mode, const float* __restrict__ w, int Co, int Ci, int KH, int KW, T* __restrict__ out)
{
    const long total = (long)Co * Ci * KH * KW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float v;
        if (mode == SAUNET_PACK_FWD) {            // out[co][kh][kw][ci] <- w[co][ci][kh][kw]
            int ci = i % Ci; long t = i / Ci; int kw = t % KW; t /= KW; int kh = t % KH; int co = t / KH;
            v = w[(((long)co * Ci + ci) * KH + kh) * KW + kw];
        } else if (mode == SAUNET_PACK_DGRAD) {   // out[ci][kh'][kw'][co] <- w[co][ci][KH-1-kh'][KW-1-kw']
            int co = i % Co; long t = i / Co; int kw = t % KW; t /= KW; int kh = t % KH; int ci = t / KH;
            v = w[(((long)co * Ci + ci) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
        } else if (mode == SAUNET_PACK_CONVT_FWD) {  // w[ci][co][4][4] (Co = out ch, Ci = in ch) -> [ph][pw][co][th][tw][ci]
            int ci = i % Ci; long t = i / Ci; int tw = t % 2; t /= 2; int th = t % 2; t /= 2; int co = t % Co; t /= Co;
            int pw = t % 2, ph = t / 2;
            int kh = (1 - ph) + 2 * th, kw = (1 - pw) + 2 * tw;
            v = w[(((long)ci * Co + co) * 4 + kh) * 4 + kw];
        } else {                                  // CONVT_DGRAD: w[ci][co][4][4] -> out[ci][kh][kw][co]
            int co = i % Co; long t = i / Co; int kw = t % 4; t /= 4; int kh = t % 4; int ci = t / 4;
            v = w[(((long)ci * Co + co) * 4 + kh) * 4 + kw];
        }
        Elem<T>::store(out + i, v);
    }
}

// all packings of one training step in a few launches: blockIdx.y = entry.
// Every packing is a permutation of the source viewed as S[A][B][T] (T = KH * KW taps, contiguous; A, B = the parameter's first two dimensions):
//   FWD          [Co][Ci][T] -> [Co][T][Ci]                    fastest output index = B          (pattern 1)
//   CONVT_DGRAD  [Ci][Co][16] -> [Ci][kh][kw][Co]              fastest output index = B          (pattern 1)
//   DGRAD        [Co][Ci][T] -> [Ci][T flipped][Co]            fastest output index = A          (pattern 2)
//   CONVT_FWD    [Ci][Co][16] -> [ph][pw][Co][th][tw][Ci]      fastest output index = A          (pattern 2)
// Round 1-3: one element per thread, gathered straight from the source -- for pattern 2 every lane of a wave touched its own cache line
// (stride Ci * T * 4 bytes), 57 us per launch and 0.27 ms per step for 190 MB of traffic.  Now a block moves a tile [AA][BB][T] through LDS:
// rows of BB * T contiguous floats in (>= 256 bytes per row), runs of >= 64 consecutive output elements out.
template <typename T, int TAPS, int AA, int BB, bool OUT_A>
__device__ __forceinline__ void pack_tile(const float* __restrict__ w, T* __restrict__ out, unsigned mode, unsigned A, unsigned B, unsigned a0, unsigned b0, float* s)
{
    constexpr int ROW = BB * TAPS, PITCH = ROW + 1, E = AA * ROW;
    for (int e = threadIdx.x; e < E; e += 256) {
        const unsigned ai = e / ROW, r = e - ai * ROW;
        const unsigned a = a0 + ai, bt = b0 * TAPS + r;
        s[ai * PITCH + r] = (a < A && bt < B * TAPS) ? w[(size_t)a * B * TAPS + bt] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += 256) {
        unsigned ai, bi, t;
        if (OUT_A) { ai = e % AA; const unsigned q = e / AA; t = q % TAPS; bi = q / TAPS; }
        else { bi = e % BB; const unsigned q = e / BB; t = q % TAPS; ai = q / TAPS; }
        const unsigned a = a0 + ai, b = b0 + bi;
        if (a >= A || b >= B) continue;
        const float v = s[ai * PITCH + bi * TAPS + t];
        size_t o;
        if (mode == SAUNET_PACK_FWD || mode == SAUNET_PACK_CONVT_DGRAD) o = ((size_t)a * TAPS + t) * B + b;
        else if (mode == SAUNET_PACK_DGRAD) o = ((size_t)b * TAPS + (TAPS - 1 - t)) * A + a;
        else {   // CONVT_FWD: a = ci, b = co, t = kh * 4 + kw;  kh = (1 - ph) + 2 th, kw = (1 - pw) + 2 tw
            const unsigned kh = t >> 2, kw = t & 3, ph = 1 - (kh & 1), th = kh >> 1, pw = 1 - (kw & 1), tw = kw >> 1;
            o = ((((size_t)(ph * 2 + pw) * B + b) * 2 + th) * 2 + tw) * A + a;
        }
        Elem<T>::store(out + o, v);
    }
    __syncthreads();
}

template <typename T, int TAPS> __device__ __forceinline__ void pack_entry(const float* w, T* out, unsigned mode, unsigned A, unsigned B, float* s)
{
    const bool out_a = mode == SAUNET_PACK_DGRAD || mode == SAUNET_PACK_CONVT_FWD;
    // tile shapes: ~2000 elements; the contiguous source run (BB * TAPS floats) is at least 256 bytes, the output run (BB or AA elements) 64
    constexpr int BB1 = TAPS >= 32 ? 8 : 64, AA1 = (2304 / (BB1 * TAPS)) > 0 ? 2304 / (BB1 * TAPS) : 1;
    constexpr int AA2 = 64, BB2 = TAPS == 1 ? 64 : (TAPS <= 9 ? 8 : (TAPS <= 16 ? 4 : 1));
    if (out_a) {
        const unsigned ta = (A + AA2 - 1) / AA2, tb = (B + BB2 - 1) / BB2;
        for (unsigned t = blockIdx.x; t < ta * tb; t += gridDim.x) pack_tile<T, TAPS, AA2, BB2, true>(w, out, mode, A, B, (t % ta) * AA2, (t / ta) * BB2, s);
    } else {
        const unsigned ta = (A + AA1 - 1) / AA1, tb = (B + BB1 - 1) / BB1;
        for (unsigned t = blockIdx.x; t < ta * tb; t += gridDim.x) pack_tile<T, TAPS, AA1, BB1, false>(w, out, mode, A, B, (t / tb) * AA1, (t % tb) * BB1, s);
    }
}

template <typename T> __global__ __launch_bounds__(256) void pack_weight_multi_kernel(saunet_pack_list pl)
{
    __shared__ float s[4800];      // the largest tile: 64 rows x (8 x 9 + 1) floats
    const int e = blockIdx.y;
    const unsigned mode = pl.mode[e], Co = pl.dims[e][0], Ci = pl.dims[e][1], KH = pl.dims[e][2], KW = pl.dims[e][3];
    const float* __restrict__ w = (const float*)pl.src[e];
    T* __restrict__ out = (T*)pl.dst[e];
    const bool convt = mode == SAUNET_PACK_CONVT_FWD || mode == SAUNET_PACK_CONVT_DGRAD;
    const unsigned A = convt ? Ci : Co, B = convt ? Co : Ci, taps = KH * KW;      // ConvTranspose2d parameters are [Ci][Co][4][4]
    if (taps == 1) pack_entry<T, 1>(w, out, mode, A, B, s);
    else if (taps == 9) pack_entry<T, 9>(w, out, mode, A, B, s);
    else if (taps == 16) pack_entry<T, 16>(w, out, mode, A, B, s);
    else if (taps == 49) pack_entry<T, 49>(w, out, mode, A, B, s);
    else {      // any other kernel size: one element per thread
        const unsigned total = Co * Ci * taps;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            float v;
            if (mode == SAUNET_PACK_FWD) {
                unsigned ci = i % Ci, t = i / Ci; unsigned kw = t % KW; t /= KW; unsigned kh = t % KH, co = t / KH;
                v = w[((co * Ci + ci) * KH + kh) * KW + kw];
            } else {
                unsigned co = i % Co, t = i / Co; unsigned kw = t % KW; t /= KW; unsigned kh = t % KH, ci = t / KH;
                v = w[((co * Ci + ci) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
            }
            Elem<T>::store(out + i, v);
        }
    }
}

// ---------------------------------------------------------------------------------------- pointwise (1x1) small-channel path
struct PwArgs {
    const void* x; const void* w; void* y; const float* bias; const float* ps; const float* psh;
    double* ssum; double* ssq;
    long P; int Cin, Cout, ldx, ldy, pro_relu; int srep, srstride; int vec_out, vec_in, act_relu;
};

// Each block owns ITEMS = 256*PW_IT consecutive (pixel, cout) items; thread t handles items t, t+256, ...
// When 256 % Cout == 0 a thread always sees the same cout, so BN statistics are accumulated in registers and
// leave the block as ONE LDS atomic per thread.
constexpr int PW_IT = 8;
template <typename T> __global__ __launch_bounds__(256) void pointwise_fwd_kernel(PwArgs a)
{
    __shared__ float s_sum[64], s_sq[64];
    const bool stats = a.ssum != nullptr;
    if (stats && threadIdx.x < 64) { s_sum[threadIdx.x] = 0.f; s_sq[threadIdx.x] = 0.f; }
    if (stats) __syncthreads();
    const long total = a.P * a.Cout;
    const long base = (long)blockIdx.x * (256 * PW_IT);
    const bool fixed_co = (256 % a.Cout) == 0;
    float rs = 0.f, rq = 0.f;
    int co_last = 0;
    const float relu_lo = a.pro_relu ? 0.f : -__builtin_inff();
    for (int it = 0; it < PW_IT; ++it) {
        const long idx = base + it * 256 + threadIdx.x;
        if (idx >= total) break;
        const long p = idx / a.Cout; const int co = (int)(idx - p * a.Cout);
        const T* xr = (const T*)a.x + p * a.ldx;
        const T* wr = (const T*)a.w + (long)co * a.Cin;
        float acc = 0.f;
        if (a.ps != nullptr) {
            for (int c = 0; c < a.Cin; ++c) {
                float v = fmaxf(fmaf(Elem<T>::load(xr + c), a.ps[c], a.psh[c]), relu_lo);
                acc = fmaf(v, Elem<T>::load(wr + c), acc);
            }
        } else {
            for (int c = 0; c < a.Cin; ++c) acc = fmaf(Elem<T>::load(xr + c), Elem<T>::load(wr + c), acc);
        }
        if (stats) {
            if (fixed_co) { rs += acc; rq = fmaf(acc, acc, rq); co_last = co; }
            else { atomicAdd(&s_sum[co], acc); atomicAdd(&s_sq[co], acc * acc); }
        }
        { const float o = acc + (a.bias ? a.bias[co] : 0.f); Elem<T>::store((T*)a.y + p * a.ldy + co, a.act_relu ? fmaxf(o, 0.f) : o); }
    }
    if (stats) {
        if (fixed_co) { atomicAdd(&s_sum[co_last], rs); atomicAdd(&s_sq[co_last], rq); }
        __syncthreads();
        if (threadIdx.x < a.Cout) {
            const size_t ro = (size_t)(blockIdx.x % a.srep) * a.srstride;
            atomicAdd(&a.ssum[ro + threadIdx.x], (double)s_sum[threadIdx.x]);
            atomicAdd(&a.ssq[ro + threadIdx.x], (double)s_sq[threadIdx.x]);
        }
    }
}

// Small-channel pointwise conv, one PIXEL per thread: the pixel's Cin values are read once into registers,
// weights / bias / prologue live in LDS (broadcast reads).  Used when Cin <= CIN_PAD <= 64 and Cout <= 64.
// BN statistics: per-output-channel wave reduction -> LDS -> one float64 atomic per channel per block.
template <typename T, int CIN_PAD> __global__ __launch_bounds__(256) void pointwise_small_fwd_kernel(PwArgs a)
{
    extern __shared__ float sm[];
    float* sw = sm;                          // [Cout][CIN_PAD]
    float* sb = sw + a.Cout * CIN_PAD;       // [Cout]
    float* sps = sb + a.Cout;                // [CIN_PAD] x2
    float* ssum = sps + 2 * CIN_PAD;         // [Cout] x2
    const bool stats = a.ssum != nullptr, has_pro = a.ps != nullptr;
    for (int i = threadIdx.x; i < a.Cout * CIN_PAD; i += 256) {
        int co = i / CIN_PAD, ci = i - co * CIN_PAD;
        sw[i] = ci < a.Cin ? Elem<T>::load((const T*)a.w + (long)co * a.Cin + ci) : 0.f;
    }
    for (int i = threadIdx.x; i < a.Cout; i += 256) { sb[i] = a.bias ? a.bias[i] : 0.f; ssum[i] = 0.f; ssum[a.Cout + i] = 0.f; }
    for (int i = threadIdx.x; i < CIN_PAD; i += 256) {
        sps[i] = (has_pro && i < a.Cin) ? a.ps[i] : 1.f;
        sps[CIN_PAD + i] = (has_pro && i < a.Cin) ? a.psh[i] : 0.f;
    }
    __syncthreads();
    const float relu_lo = a.pro_relu ? 0.f : -__builtin_inff();
    const int lane = threadIdx.x & 63;
    const long pstride = (long)gridDim.x * 256;
    const long pend = ((a.P + 255) / 256) * 256;   // whole waves iterate together (wave reductions below)
    for (long p = blockIdx.x * 256L + threadIdx.x; p < pend; p += pstride) {
        const bool live = p < a.P;
        float xv[CIN_PAD];
        const T* xr = (const T*)a.x + (live ? p : 0) * a.ldx;
        constexpr int EPI = 16 / sizeof(T);
        if (a.vec_in) {             // Cin % EPI == 0 and 16-byte aligned rows: whole vectors instead of element loads
#pragma unroll
            for (int c0 = 0; c0 < CIN_PAD; c0 += EPI) {
                float f[EPI];
                if (c0 < a.Cin) Vec16<T>::unpack(*(const u32x4*)(xr + c0), f);
#pragma unroll
                for (int j = 0; j < EPI; ++j) if (c0 + j < CIN_PAD) xv[c0 + j] = (c0 < a.Cin) ? f[j] : 0.f;
            }
        } else {
#pragma unroll
            for (int c = 0; c < CIN_PAD; ++c) xv[c] = (c < a.Cin) ? Elem<T>::load(xr + c) : 0.f;
        }
#pragma unroll
        for (int c = 0; c < CIN_PAD; ++c) {
            float v = xv[c];
            if (has_pro) v = fmaxf(fmaf(v, sps[c], sps[CIN_PAD + c]), relu_lo);
            xv[c] = (c < a.Cin) ? v : 0.f;
        }
        T* yr = (T*)a.y + (live ? p : 0) * a.ldy;
        auto dot = [&](int co) {
            const float* wr = sw + co * CIN_PAD;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CIN_PAD; c += 4) {
                f32x4 w4 = *(const f32x4*)(wr + c);
                acc = fmaf(w4[0], xv[c], acc); acc = fmaf(w4[1], xv[c + 1], acc);
                acc = fmaf(w4[2], xv[c + 2], acc); acc = fmaf(w4[3], xv[c + 3], acc);
            }
            if (stats) {
                float v = live ? acc : 0.f;
                float s = wave_sum(v), q = wave_sum(v * v);
                if (lane == 0) { atomicAdd(&ssum[co], s); atomicAdd(&ssum[a.Cout + co], q); }
            }
            return a.act_relu ? fmaxf(acc + sb[co], 0.f) : acc + sb[co];
        };
        constexpr int EPC = 16 / sizeof(T);
        if (a.vec_out) {            // Cout % EPC == 0, 16-byte aligned output rows: one vector store per EPC channels
            for (int co = 0; co < a.Cout; co += EPC) {
                float o[EPC];
#pragma unroll
                for (int j = 0; j < EPC; ++j) o[j] = dot(co + j);
                if (live) *(u32x4*)(yr + co) = Vec16<T>::pack(o);
            }
        } else {
            for (int co = 0; co < a.Cout; ++co) {
                const float v = dot(co);
                if (live) Elem<T>::store(yr + co, v);
            }
        }
    }
    if (stats) {
        __syncthreads();
        const size_t ro = (size_t)(blockIdx.x % a.srep) * a.srstride;
        for (int i = threadIdx.x; i < a.Cout; i += 256) {
            atomicAdd(&a.ssum[ro + i], (double)ssum[i]);
            atomicAdd(&a.ssq[ro + i], (double)ssum[a.Cout + i]);
        }
    }
}

// Cout small, Cin large (the C -> 1 projections c3/c4/c5): one WAVE per (pixel, cout), lanes stride the channels
template <typename T> __global__ __launch_bounds__(256) void pointwise_dot_kernel(PwArgs a)
{
    const int lane = threadIdx.x & 63;
    const long total = a.P * a.Cout;
    const float relu_lo = a.pro_relu ? 0.f : -__builtin_inff();
    for (long i = blockIdx.x * 4L + (threadIdx.x >> 6); i < total; i += (long)gridDim.x * 4) {
        const long p = i / a.Cout; const int co = (int)(i - p * a.Cout);
        const T* xr = (const T*)a.x + p * a.ldx;
        const T* wr = (const T*)a.w + (long)co * a.Cin;
        float acc = 0.f;
        for (int c = lane; c < a.Cin; c += 64) {
            float v = Elem<T>::load(xr + c);
            if (a.ps) v = fmaxf(fmaf(v, a.ps[c], a.psh[c]), relu_lo);
            acc = fmaf(v, Elem<T>::load(wr + c), acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) { const float o = acc + (a.bias ? a.bias[co] : 0.f); Elem<T>::store((T*)a.y + p * a.ldy + co, a.act_relu ? fmaxf(o, 0.f) : o); }
    }
}

// Cin == 1: y[p][co] = x[p]*w[co] + b[co]  (the dgrad of the C -> 1 projections c3/c4/c5/phi/fuse): pure streaming
template <typename T> __global__ __launch_bounds__(256) void pointwise_cin1_kernel(PwArgs a)
{
    const unsigned total = (unsigned)(a.P * a.Cout);
    const float relu_lo = a.pro_relu ? 0.f : -__builtin_inff();
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned p = i / (unsigned)a.Cout, co = i - p * (unsigned)a.Cout;
        float v = Elem<T>::load((const T*)a.x + (size_t)p * a.ldx);
        if (a.ps) v = fmaxf(fmaf(v, a.ps[0], a.psh[0]), relu_lo);
        { const float o = fmaf(v, Elem<T>::load((const T*)a.w + co), a.bias ? a.bias[co] : 0.f); Elem<T>::store((T*)a.y + (size_t)p * a.ldy + co, a.act_relu ? fmaxf(o, 0.f) : o); }
    }
}

// dw[co][ci] += sum_p dy[p][co] * a[p][ci]: rows staged in LDS, each thread owns <= WPT weights
struct PwWgradArgs {
    const void* x; const void* dy; float* dw; const float* ps; const float* psh;
    long P; int Cin, Cout, ldx, lddy, pro_relu; long pix_per_block; long sM, sN; int rows;
    float* part;      // narrow kernel: per-split partial gradients [splits][Cout][Cin + 1] (nullptr: float atomics into dw / dbias)
    float* dbias;     // narrow kernel: optional bias gradient sum_p dy[p][co] -- the "ones" input channel, index Cin of a partial row
};
template <typename T, int WPT> __global__ __launch_bounds__(256) void pointwise_wgrad_kernel(PwWgradArgs a)
{
    const int ROWS = a.rows;
    extern __shared__ float sbuf[];  // [ROWS][Cin] then [ROWS][Cout]
    float* sx = sbuf; float* sd = sbuf + ROWS * a.Cin;
    const int nW = a.Cin * a.Cout;
    float acc[WPT];
    int wco[WPT], wci[WPT];
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
        int w = threadIdx.x + k * 256;
        acc[k] = 0.f; wco[k] = (w < nW) ? w / a.Cin : -1; wci[k] = (w < nW) ? w % a.Cin : 0;
    }
    const long p0 = blockIdx.x * a.pix_per_block;
    const long p1 = min(p0 + a.pix_per_block, a.P);
    for (long pb = p0; pb < p1; pb += ROWS) {
        const int rows = (int)min((long)ROWS, p1 - pb);
        for (int i = threadIdx.x; i < rows * a.Cin; i += 256) {
            int r = i / a.Cin, c = i - r * a.Cin;
            float v = Elem<T>::load((const T*)a.x + (pb + r) * a.ldx + c);
            if (a.ps != nullptr) { v = fmaf(v, a.ps[c], a.psh[c]); if (a.pro_relu) v = fmaxf(v, 0.f); }
            sx[i] = v;
        }
        for (int i = threadIdx.x; i < rows * a.Cout; i += 256) {
            int r = i / a.Cout, c = i - r * a.Cout;
            sd[i] = Elem<T>::load((const T*)a.dy + (pb + r) * a.lddy + c);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            if (wco[k] >= 0) {
                float s = acc[k];
                for (int r = 0; r < rows; ++r) s = fmaf(sd[r * a.Cout + wco[k]], sx[r * a.Cin + wci[k]], s);
                acc[k] = s;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k)
        if (wco[k] >= 0) atomicAdd(a.dw + wco[k] * a.sM + wci[k] * a.sN, acc[k]);
}

// Few outputs, many inputs, few pixels (the C -> 1 projections c3/c4/c5/phi at 1/8..1/32 resolution): a weighted channel
// sum.  thread = input channel (coalesced rows), blockIdx.y = pixel split, float atomics into the zeroed dw.
template <typename T> __global__ __launch_bounds__(256) void pointwise_wgrad_fewout_kernel(PwWgradArgs a)
{
    const int ci = blockIdx.x * 256 + threadIdx.x;
    if (ci >= a.Cin) return;
    const long p0 = blockIdx.y * a.pix_per_block, p1 = min(p0 + a.pix_per_block, a.P);
    const float sc = a.ps ? a.ps[ci] : 1.f, sh = a.ps ? a.psh[ci] : 0.f;
    const float relu_lo = (a.ps && a.pro_relu) ? 0.f : -__builtin_inff();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (long p = p0; p < p1; ++p) {
        const float v = fmaxf(fmaf(Elem<T>::load((const T*)a.x + p * a.ldx + ci), sc, sh), relu_lo);
#pragma unroll
        for (int co = 0; co < 4; ++co)
            if (co < a.Cout) acc[co] = fmaf(Elem<T>::load((const T*)a.dy + p * a.lddy + co), v, acc[co]);
    }
#pragma unroll
    for (int co = 0; co < 4; ++co)
        if (co < a.Cout) atomicAdd(a.dw + co * a.sM + ci * a.sN, acc[co]);
}

// Few outputs (Cout <= 4) at ANY resolution and channel count (round 6): the C -> 1 side outputs c3 / c4 / c5 / phi / cw, `fuse` (2 -> 1, float32)
// and `final` (32 -> 4).  dw[co][ci] = sum_p dy[p][co] * a[p][ci] is a weighted row sum: a thread owns one V-element chunk of the row (16 bytes
// where the layout allows) and walks the pixels with a stride of RL rows, four rows in flight; the block folds its RL partial rows through LDS
// in row order (no LDS atomics) and leaves one float atomic per weight.  The kernels it replaces ran one THREAD per weight over LDS-staged rows
// (2 -> 1 at 256 x 256: two busy threads per block, 111 us for 25 MB) or one thread per channel over 32 serial pixels (23 - 45 us for 1 - 4 MB).
template <typename T, int V> struct NarrowVec;
template <> struct NarrowVec<float, 4> { typedef u32x4 type; };
template <> struct NarrowVec<float, 2> { typedef __attribute__((ext_vector_type(2))) unsigned int type; };
template <> struct NarrowVec<float, 1> { typedef unsigned int type; };
template <> struct NarrowVec<u16, 8> { typedef u32x4 type; };
template <> struct NarrowVec<u16, 4> { typedef __attribute__((ext_vector_type(2))) unsigned int type; };
template <> struct NarrowVec<u16, 2> { typedef unsigned int type; };
template <> struct NarrowVec<u16, 1> { typedef unsigned short type; };
template <typename T, int V> __device__ __forceinline__ void narrow_load(const T* p, float* f)
{
    typedef typename NarrowVec<T, V>::type VT;
    const VT v = *(const VT*)p;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < V; ++j) f[j] = __uint_as_float(((const unsigned*)&v)[j]);
    } else if constexpr (V == 1) f[0] = __uint_as_float((unsigned)v << 16);
    else {
#pragma unroll
        for (int j = 0; j < V / 2; ++j) { const unsigned w = ((const unsigned*)&v)[j]; f[2 * j] = bf16_lo(w); f[2 * j + 1] = bf16_hi(w); }
    }
}

template <typename T, int V> __global__ __launch_bounds__(256) void pointwise_wgrad_narrow_kernel(PwWgradArgs a)
{
    extern __shared__ float s_part[];                 // [RL][CHB][4][V]
    const int CH = a.Cin / V;                         // chunks per row
    const int c0 = blockIdx.x * 256;                  // first chunk of this block (Cin > 256 V: several chunk tiles)
    const int CHB = min(256, CH - c0), RL = 256 / CHB;
    const int c = threadIdx.x % CHB, r = threadIdx.x / CHB;
    const bool live = r < RL;
    const int ci = (c0 + c) * V;
    float sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { sc[j] = a.ps ? a.ps[ci + j] : 1.f; sh[j] = a.ps ? a.psh[ci + j] : 0.f; }
    const float relu_lo = (a.ps && a.pro_relu) ? 0.f : -__builtin_inff();
    float acc[4][V];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};              // bias gradient: the threads of chunk 0 own it
#pragma unroll
    for (int co = 0; co < 4; ++co)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[co][j] = 0.f;
    const long p0 = blockIdx.y * a.pix_per_block, p1 = min(p0 + a.pix_per_block, a.P);
    const T* __restrict__ x = (const T*)a.x; const T* __restrict__ dy = (const T*)a.dy;
    int cco[4];                                      // outputs beyond Cout re-read the last one; their sums are never stored
#pragma unroll
    for (int co = 0; co < 4; ++co) cco[co] = min(co, a.Cout - 1);
    if (live) {
        long p = p0 + r;
        for (; p + 3L * RL < p1; p += 4L * RL) {       // four rows in flight
            float xv[4][V], dv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                narrow_load<T, V>(x + (p + (long)u * RL) * a.ldx + ci, xv[u]);
#pragma unroll
                for (int co = 0; co < 4; ++co) dv[u][co] = Elem<T>::load(dy + (p + (long)u * RL) * a.lddy + cco[co]);      // unconditional (clamped) loads: a load under a branch waits vmcnt(0)
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float v = fmaxf(fmaf(xv[u][j], sc[j], sh[j]), relu_lo);
#pragma unroll
                    for (int co = 0; co < 4; ++co) acc[co][j] = fmaf(dv[u][co], v, acc[co][j]);
                }
#pragma unroll
                for (int co = 0; co < 4; ++co) bsum[co] += dv[u][co];
            }
        }
        for (; p < p1; p += RL) {
            float xv[V];
            narrow_load<T, V>(x + p * a.ldx + ci, xv);
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                const float d = Elem<T>::load(dy + p * a.lddy + cco[co]);
                bsum[co] += d;
#pragma unroll
                for (int j = 0; j < V; ++j) acc[co][j] = fmaf(d, fmaxf(fmaf(xv[j], sc[j], sh[j]), relu_lo), acc[co][j]);
            }
        }
#pragma unroll
        for (int co = 0; co < 4; ++co)
#pragma unroll
            for (int j = 0; j < V; ++j) s_part[((r * CHB + c) * 4 + co) * V + j] = acc[co][j];
    }
    __syncthreads();
    // fold the RL partial rows in row order; one atomic per weight and block
    for (int i = threadIdx.x; i < CHB * 4 * V; i += 256) {
        const int cc = i / (4 * V), co = (i / V) & 3, j = i % V;
        if (co >= a.Cout) continue;
        float s = 0.f;
        for (int rr = 0; rr < RL; ++rr) s += s_part[((rr * CHB + cc) * 4 + co) * V + j];
        const int ci_o = (c0 + cc) * V + j;
        if (a.part) a.part[((size_t)blockIdx.y * a.Cout + co) * (a.Cin + 1) + ci_o] = s;      // ordered second stage: pointwise_wgrad_narrow_reduce_kernel
        else atomicAdd(a.dw + co * a.sM + (long)ci_o * a.sN, s);
    }
    if (a.dbias != nullptr && blockIdx.x == 0) {        // bias gradient: the RL row-threads of chunk 0, folded in row order through the (now free) LDS
        __syncthreads();
        if (live && c == 0) {
#pragma unroll
            for (int co = 0; co < 4; ++co) s_part[r * 4 + co] = bsum[co];
        }
        __syncthreads();
        if ((int)threadIdx.x < a.Cout) {
            float s = 0.f;
            for (int rr = 0; rr < RL; ++rr) s += s_part[rr * 4 + threadIdx.x];
            if (a.part) a.part[((size_t)blockIdx.y * a.Cout + threadIdx.x) * (a.Cin + 1) + a.Cin] = s;
            else atomicAdd(a.dbias + threadIdx.x, s);
        }
    }
}

// dw[co][ci] += sum over the pixel splits of part[split][co][ci]: one wave per weight, lanes stride the splits, fixed-shape fold (deterministic)
// (a partial row has Cin + 1 entries: the last one is the bias gradient, written to dbias when the caller asked for it)
__global__ __launch_bounds__(256) void pointwise_wgrad_narrow_reduce_kernel(const float* __restrict__ part, int splits, int nW, int Cin, float* __restrict__ dw, long sM, long sN,
                                                                            float* __restrict__ dbias)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= nW) return;
    const int co = w / (Cin + 1), ci = w - co * (Cin + 1);
    if (ci == Cin && dbias == nullptr) return;
    float s = 0.f;
    for (int k = lane; k < splits; k += 64) s += part[(size_t)k * nW + w];
    s = wave_sum(s);
    if (lane == 0) { if (ci < Cin) dw[co * sM + ci * sN] += s; else dbias[co] += s; }
}

// Small channel counts at full resolution (d2/d3/fuse/final of the shape stream and head: Cin <= 32*CIT, Cout <= 32, bf16):
// one thread per pixel loads its rows with 16-byte vectors, a wave transposes its 64 pixels through LDS and the outer
// products run on the matrix cores (mma_tiles.h); one float atomic per weight per block.
template <int CIT, bool VD> __global__ __launch_bounds__(256) void pointwise_wgrad_mma_kernel(PwWgradArgs a)
{
    extern __shared__ u16 w_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u16* tX = w_lds + wave * (CIT + 1) * G_TILE; u16* tD = tX + CIT * G_TILE;
    const u16* __restrict__ x = (const u16*)a.x; const u16* __restrict__ dy = (const u16*)a.dy;
    f32x16 acc[CIT];
#pragma unroll
    for (int t = 0; t < CIT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const unsigned P = (unsigned)a.P;
    for (unsigned base = blockIdx.x * 256u; base < P; base += gridDim.x * 256u) {
        const unsigned p = base + threadIdx.x; const bool live = p < P; const size_t pp = live ? p : 0;
#pragma unroll
        for (int g = 0; g < CIT * 4; ++g) {
            if (g * 8 < a.Cin) {
                const u32x4 v = *(const u32x4*)(x + pp * a.ldx + g * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) { tX[(g * 8 + 2 * j) * GP + lane] = (u16)(v[j] & 0xffffu); tX[(g * 8 + 2 * j + 1) * GP + lane] = (u16)(v[j] >> 16); }
            }
        }
        if constexpr (VD) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g * 8 < a.Cout) {
                    u32x4 v = *(const u32x4*)(dy + pp * a.lddy + g * 8);
                    if (!live) v = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int j = 0; j < 4; ++j) { tD[(g * 8 + 2 * j) * GP + lane] = (u16)(v[j] & 0xffffu); tD[(g * 8 + 2 * j + 1) * GP + lane] = (u16)(v[j] >> 16); }
                }
            }
        } else {
            for (int c = 0; c < a.Cout; ++c) tD[c * GP + lane] = live ? dy[pp * a.lddy + c] : (u16)0;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < CIT; ++t) tile_mma(tD, a.Cout, tX + t * G_TILE, a.Cin - 32 * t, lane, acc[t]);
        __syncthreads();
    }
    float* red = (float*)w_lds;
    for (int i = threadIdx.x; i < CIT * 1024; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < CIT; ++t) tile_flush(red + t * 1024, acc[t], lane);
    __syncthreads();
    for (int i = threadIdx.x; i < CIT * 1024; i += 256) {
        const int t = i >> 10, co = (i >> 5) & 31, ci = t * 32 + (i & 31);
        if (co < a.Cout && ci < a.Cin) atomicAdd(a.dw + co * a.sM + ci * a.sN, red[i]);
    }
}

// out[c] += sum_p x[p][c]: per-thread float partials (flushed to double every 128 rows), one LDS double
// atomic per thread, one global double atomic per channel per block
template <typename T> __global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ x, long P, int C, int ld,
                                                                                long rows_per_block, double* __restrict__ out)
{
    extern __shared__ double s_red[];
    for (int i = threadIdx.x; i < C; i += 256) s_red[i] = 0.0;
    __syncthreads();
    const long p0 = blockIdx.x * rows_per_block, p1 = min(p0 + rows_per_block, P);
    for (int cb = 0; cb < C; cb += 256) {
        const int cw = min(256, C - cb);
        const int rl = 256 / cw;
        const int c = threadIdx.x % cw, r0 = threadIdx.x / cw;
        if (r0 >= rl) continue;
        double s = 0.0; float fs = 0.f; int cnt = 0;
        for (long p = p0 + r0; p < p1; p += rl) {
            fs += Elem<T>::load(x + p * ld + cb + c);
            if (++cnt == 128) { s += fs; fs = 0.f; cnt = 0; }
        }
        atomicAdd(&s_red[cb + c], s + (double)fs);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(&out[c], s_red[c]);
}

// Vector form: 16-byte chunks per lane (whole 128-byte row segments per wave), float partials flushed to double every 64 rows, LDS reduce,
// one double atomic per channel per block.  CV = channels of the reduction as the kernel sees them: C itself, or -- for dense tensors with
// fewer channels than a chunk holds (the one-channel side outputs c3/c4/c5, phi) -- the V interleaved "virtual channels" of a flat view,
// which the last step folds back modulo C.
template <typename T, int V> __global__ __launch_bounds__(256) void channel_sum_vec_kernel(const T* __restrict__ x, long rows, int CV, int ld, int C,
                                                                                          long rows_per_block, double* __restrict__ out)
{
    extern __shared__ double s_red[];
    for (int i = threadIdx.x; i < CV; i += 256) s_red[i] = 0.0;
    __syncthreads();
    const long p0 = blockIdx.x * rows_per_block, p1 = min(p0 + rows_per_block, rows);
    const int CH = CV / V;
    for (int cb = 0; cb < CH; cb += 256) {
        const int cw = min(256, CH - cb), rl = 256 / cw;
        const int ch = cb + threadIdx.x % cw, r0 = threadIdx.x / cw;
        if (r0 >= rl) continue;
        float fs[V]; double ds[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { fs[j] = 0.f; ds[j] = 0.0; }
        int cnt = 0;
        for (long p = p0 + r0; p < p1; p += rl) {
            float f[V];
            Vec16<T>::unpack(*(const u32x4*)(x + p * ld + ch * V), f);
#pragma unroll
            for (int j = 0; j < V; ++j) fs[j] += f[j];
            if (++cnt == 64) {
#pragma unroll
                for (int j = 0; j < V; ++j) { ds[j] += fs[j]; fs[j] = 0.f; }
                cnt = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) atomicAdd(&s_red[ch * V + j], ds[j] + (double)fs[j]);
    }
    __syncthreads();
    if (CV == C) { for (int c = threadIdx.x; c < C; c += 256) atomicAdd(&out[c], s_red[c]); }
    else if (threadIdx.x < C) {
        double s = 0.0;
        for (int j = threadIdx.x; j < CV; j += C) s += s_red[j];
        atomicAdd(&out[threadIdx.x], s);
    }
}

static bool is_pointwise(const saunet_conv_desc* d)
{
    return d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && !d->transposed && d->Ho == d->H && d->Wo == d->W;
}

// A pointwise conv over maps that are not multiples of the 16x16 pixel tile (8x8, 24x40, ...) is re-described as N' images of
// 16x16 pixels when the pixel count allows: the tiled weight-gradient kernel only needs 256-pixel groups of contiguous rows.
static bool dense_pointwise_rows(const saunet_conv_desc* d, saunet_conv_desc* out)
{
    const long P = (long)d->N * d->H * d->W;
    if ((d->H % 16 == 0 && d->W % 16 == 0) || P % 256 != 0) return false;
    *out = *d;
    out->N = (int)(P / 256); out->H = out->W = out->Ho = out->Wo = 16;
    return true;
}

}  // namespace saunet

using namespace saunet;

extern "C" {

const char* saunet_last_error(void) { return g_err; }
const char* saunet_launch_log(void)
{
    const char* r = g_launches[g_launch_cur];
    g_launch_cur ^= 1;
    g_launches[g_launch_cur][0] = 0;
    return r;
}
int saunet_version(void) { return SAUNET_ABI_VERSION; }

int saunet_init(int device)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return set_error(SAUNET_LAUNCH_FAILED, "no device %d", device);
    return prop.multiProcessorCount;
}

int saunet_pack_weight(int mode, int dtype, const float* w, int Co, int Ci, int KH, int KW, void* out, void* stream)
{
    if ((mode == SAUNET_PACK_CONVT_FWD || mode == SAUNET_PACK_CONVT_DGRAD) && (KH != 4 || KW != 4))
        return set_error(SAUNET_UNSUPPORTED, "pack: conv-transpose packing needs 4x4 kernels");
    const long total = (long)Co * Ci * KH * KW;
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SAUNET_F32) hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, st, mode, w, Co, Ci, KH, KW, (float*)out);
    else if (dtype == SAUNET_BF16) hipLaunchKernelGGL(pack_weight_kernel<u16>, dim3(blocks), dim3(256), 0, st, mode, w, Co, Ci, KH, KW, (u16*)out);
    else return set_error(SAUNET_BAD_DTYPE, "pack: dtype %d", dtype);
    SAUNET_CHECK_LAUNCH("pack_weight");
    return SAUNET_OK;
}

int saunet_pack_weight_multi(const saunet_pack_list* pl, int dtype, void* stream)
{
    if (pl->count <= 0 || pl->count > 64) return set_error(SAUNET_BAD_SHAPE, "pack_multi: %d entries", pl->count);
    hipStream_t st = (hipStream_t)stream;
    long biggest = 1;
    for (int e = 0; e < pl->count; ++e) {
        const long t = (long)pl->dims[e][0] * pl->dims[e][1] * pl->dims[e][2] * pl->dims[e][3];
        if (t >= (1L << 31)) return set_error(SAUNET_BAD_SHAPE, "pack_multi: entry %d has %ld elements", e, t);
        if (t > biggest) biggest = t;
        const bool convt = pl->mode[e] == SAUNET_PACK_CONVT_FWD || pl->mode[e] == SAUNET_PACK_CONVT_DGRAD;
        if (convt && (pl->dims[e][2] != 4 || pl->dims[e][3] != 4)) return set_error(SAUNET_UNSUPPORTED, "pack_multi: conv-transpose packing needs 4x4 kernels");
    }
    long bx = (biggest + 2047) / 2048; if (bx > 1024) bx = 1024;     // one ~2000-element tile per block for the largest entry; small entries' extra blocks exit at once
    if (dtype == SAUNET_F32) hipLaunchKernelGGL(pack_weight_multi_kernel<float>, dim3((unsigned)bx, pl->count), dim3(256), 0, st, *pl);
    else if (dtype == SAUNET_BF16) hipLaunchKernelGGL(pack_weight_multi_kernel<u16>, dim3((unsigned)bx, pl->count), dim3(256), 0, st, *pl);
    else return set_error(SAUNET_BAD_DTYPE, "pack_multi: dtype %d", dtype);
    SAUNET_CHECK_LAUNCH("pack_weight_multi");
    return SAUNET_OK;
}

int saunet_conv2d_forward(const saunet_conv_desc* d, const void* x, const void* w, const float* bias,
                          const float* ps, const float* psh, void* y, double* ssum, double* ssq, void* stream)
{
    return saunet_conv2d_forward_ex(d, x, w, bias, ps, psh, y, ssum, ssq, nullptr, stream);
}

int64_t saunet_conv2d_forward_workspace(const saunet_conv_desc* d)
{
    return mm_forward_workspace(d);
}

int saunet_conv2d_accumulate_supported(const saunet_conv_desc* d)
{
    return igemm_supported(d) && tile_fwd_accumulate_supported(d) ? 1 : 0;
}

int saunet_conv2d_forward_ex(const saunet_conv_desc* d, const void* x, const void* w, const float* bias,
                             const float* ps, const float* psh, void* y, double* ssum, double* ssq,
                             const saunet_bn_epilogue* epi, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->ldx < d->Cin || d->ldy < d->Cout)
        return set_error(SAUNET_BAD_SHAPE, "conv: bad shape N=%d Cin=%d ldx=%d Cout=%d ldy=%d", d->N, d->Cin, d->ldx, d->Cout, d->ldy);
    if (!d->transposed) {
        int ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
        if (ho != d->Ho || wo != d->Wo) return set_error(SAUNET_BAD_SHAPE, "conv: Ho/Wo %dx%d != %dx%d", d->Ho, d->Wo, ho, wo);
    }
    if ((ps == nullptr) != (psh == nullptr)) return set_error(SAUNET_BAD_SHAPE, "conv: prologue needs scale and shift");
    if (ssum == nullptr && dense_dgrad_supported(d, bias, ps, epi)) return dense_dgrad_forward(d, x, w, y, epi, st);
    if (ssum == nullptr && dense_dgrad3_supported(d, bias, ps, epi)) return dense_dgrad3_forward(d, x, w, y, epi, st);
    if (epi && epi->bn_x == nullptr) {
        // no BatchNorm epilogue: the structure only carries `accumulate` (y += conv), which the resident 3x3 kernel implements
        if (!epi->accumulate) epi = nullptr;
        else {
            if (!saunet_conv2d_accumulate_supported(d)) return set_error(SAUNET_UNSUPPORTED, "conv: y += conv has no kernel for this geometry (saunet_conv2d_accumulate_supported)");
            return tile_forward(d, x, w, bias, ps, psh, y, ssum, ssq, epi, st);
        }
    }
    if (mm_fwd_supported(d, x, w, y, ps, epi)) return mm_forward(d, x, w, bias, y, ssum, ssq, st);
    if (mm_convt_supported(d, x, w, y, ps, epi)) return mm_convt_forward(d, x, w, bias, y, ssum, ssq, st);
    if (epi && epi->relu_mask && !(igemm_supported(d) && !tile_fwd_supported(d) && !d->transposed && d->dtype == SAUNET_BF16 && d->Cout % 8 == 0))
        return set_error(SAUNET_UNSUPPORTED, "conv: the bit-mask BN epilogue is implemented for bf16 1x1 data gradients on the implicit-GEMM path");
    if (igemm_supported(d)) {
        if (epi && (((uintptr_t)epi->bn_x & 15) || epi->ld_bn_x % (d->dtype == SAUNET_BF16 ? 8 : 4)))
            return set_error(SAUNET_BAD_ALIGN, "conv: bn epilogue tensor must be 16-byte aligned");
        if (tile_fwd_supported(d)) {
            if (epi && epi->accumulate) return set_error(SAUNET_UNSUPPORTED, "conv: accumulating BN epilogue is implemented for 1x1 dgrads");
            return tile_forward(d, x, w, bias, ps, psh, y, ssum, ssq, epi, st);
        }
        return igemm_forward(d, x, w, bias, ps, psh, y, ssum, ssq, epi, st);
    }
    if (epi) return set_error(SAUNET_UNSUPPORTED, "conv: the BN-backward epilogue needs the MFMA path");
    if (!is_pointwise(d)) return set_error(SAUNET_UNSUPPORTED, "conv: %dx%d s%d Cin=%d Cout=%d has no kernel", d->KH, d->KW, d->stride, d->Cin, d->Cout);
    if (ssum != nullptr && d->Cout > 64) return set_error(SAUNET_UNSUPPORTED, "pointwise stats need Cout <= 64");
    PwArgs a{x, w, y, bias, ps, psh, ssum, ssq, (long)d->N * d->H * d->W, d->Cin, d->Cout, d->ldx, d->ldy, d->pro_relu,
             d->stat_replicas > 1 ? d->stat_replicas : 1, d->stat_rstride, 0, 0, d->epi_relu};
    {
        const int epc = d->dtype == SAUNET_BF16 ? 8 : 4;
        a.vec_out = d->Cout % epc == 0 && d->ldy % epc == 0 && !((uintptr_t)y & 15);
        a.vec_in = d->Cin % epc == 0 && d->ldx % epc == 0 && !((uintptr_t)x & 15);
    }
    if (d->Cin <= 64 && d->Cout <= 64) {
        const int cpad = d->Cin <= 4 ? 4 : d->Cin <= 8 ? 8 : d->Cin <= 16 ? 16 : d->Cin <= 36 ? 36 : 64;
        long blocks = (a.P + 255) / 256; if (blocks > 2048) blocks = 2048;
        const size_t lds = sizeof(float) * ((size_t)d->Cout * cpad + 3 * d->Cout + 2 * cpad);
#define PWS(TT, CP) hipLaunchKernelGGL((pointwise_small_fwd_kernel<TT, CP>), dim3((unsigned)blocks), dim3(256), lds, st, a)
#define PWS_T(TT) do { if (cpad == 4) PWS(TT, 4); else if (cpad == 8) PWS(TT, 8); else if (cpad == 16) PWS(TT, 16); else if (cpad == 36) PWS(TT, 36); else PWS(TT, 64); } while (0)
        if (d->dtype == SAUNET_F32) PWS_T(float);
        else if (d->dtype == SAUNET_BF16) PWS_T(u16);
        else return set_error(SAUNET_BAD_DTYPE, "conv: dtype %d", d->dtype);
#undef PWS_T
#undef PWS
        SAUNET_CHECK_LAUNCH("pointwise_small_fwd");
        return SAUNET_OK;
    }
    long total = a.P * a.Cout;
    if (d->Cin == 1 && ssum == nullptr && total < (1L << 31)) {
        long blocks = (total + 1023) / 1024; if (blocks > 8192) blocks = 8192;
        if (d->dtype == SAUNET_F32) hipLaunchKernelGGL(pointwise_cin1_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, a);
        else if (d->dtype == SAUNET_BF16) hipLaunchKernelGGL(pointwise_cin1_kernel<u16>, dim3((unsigned)blocks), dim3(256), 0, st, a);
        else return set_error(SAUNET_BAD_DTYPE, "conv: dtype %d", d->dtype);
        SAUNET_CHECK_LAUNCH("pointwise_cin1");
        return SAUNET_OK;
    }
    if (d->Cin >= 128 && d->Cout <= 4 && ssum == nullptr) {
        long blocks = (total + 3) / 4; if (blocks > 16384) blocks = 16384;
        if (d->dtype == SAUNET_F32) hipLaunchKernelGGL(pointwise_dot_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, a);
        else if (d->dtype == SAUNET_BF16) hipLaunchKernelGGL(pointwise_dot_kernel<u16>, dim3((unsigned)blocks), dim3(256), 0, st, a);
        else return set_error(SAUNET_BAD_DTYPE, "conv: dtype %d", d->dtype);
        SAUNET_CHECK_LAUNCH("pointwise_dot");
        return SAUNET_OK;
    }
    dim3 grid((unsigned)((total + 256 * PW_IT - 1) / (256 * PW_IT)));
    if (d->dtype == SAUNET_F32) hipLaunchKernelGGL(pointwise_fwd_kernel<float>, grid, dim3(256), 0, st, a);
    else if (d->dtype == SAUNET_BF16) hipLaunchKernelGGL(pointwise_fwd_kernel<u16>, grid, dim3(256), 0, st, a);
    else return set_error(SAUNET_BAD_DTYPE, "conv: dtype %d", d->dtype);
    SAUNET_CHECK_LAUNCH("pointwise_fwd");
    return SAUNET_OK;
}

int saunet_conv2d_forward_bnpro(const saunet_conv_desc* d, const void* x, const void* w, const float* bias, const saunet_bn_prologue* pro,
                                void* y, double* ssum, double* ssq, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!pro || !pro->gamma || !pro->beta || !pro->params || !pro->sum || !pro->sumsq || pro->count < 1.0 || pro->c_lo < 0 || pro->c_lo > d->Cin ||
        (pro->c_lo > 0 && (!pro->xhat || pro->ld_xhat < d->Cin)) || (pro->running_mean == nullptr) != (pro->running_var == nullptr))
        return set_error(SAUNET_BAD_SHAPE, "conv_bnpro: incomplete BatchNorm prologue descriptor");
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->ldx < d->Cin || d->ldy < d->Cout)
        return set_error(SAUNET_BAD_SHAPE, "conv: bad shape N=%d Cin=%d ldx=%d Cout=%d ldy=%d", d->N, d->Cin, d->ldx, d->Cout, d->ldy);
    saunet_bn_prologue p = *pro;
    if (p.replicas < 1) p.replicas = 1;
    if (d->pro_relu && d->Ho == d->H && d->Wo == d->W && dense_conv1_small_supported(d, x, w, y, bias))
        return dense_conv1_small_forward(d, x, w, y, ssum, ssq, &p, st);      // DenseNet conv1 on the low-resolution blocks (csrc/dense_fwd.hip)
    if (p.c_lo == 0 && dense_conv2_small_supported(d, x, w, y, bias))
        return dense_conv2_small_forward(d, x, w, y, ssum, ssq, &p, st);      // DenseNet conv2 on the low-resolution blocks
    if (!d->transposed && igemm_supported(d)) {
        int ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
        if (ho != d->Ho || wo != d->Wo) return set_error(SAUNET_BAD_SHAPE, "conv: Ho/Wo %dx%d != %dx%d", d->Ho, d->Wo, ho, wo);
        if (tile_fwd_supported(d)) return tile_forward(d, x, w, bias, nullptr, nullptr, y, ssum, ssq, nullptr, st, &p);
        return igemm_forward(d, x, w, bias, nullptr, nullptr, y, ssum, ssq, nullptr, st, &p);
    }
    // kernels that take ready-made coefficients: finalise with one small launch, then the ordinary forward
    if (int rc = bn_prologue_finalize(&p, d->Cin, st)) return rc;
    return saunet_conv2d_forward_ex(d, x, w, bias, p.params, p.params + d->Cin, y, ssum, ssq, nullptr, stream);
}

// A/B switch: SAUNET_CONVT_WGRAD_DIRECT=0 sends ConvTranspose2d weight gradients back to the generic path
static bool convt_direct(const saunet_conv_desc* d)
{
    static const bool on = ab_env_on("SAUNET_CONVT_WGRAD_DIRECT");
    return on && tile_wgrad_convt_supported(d);
}

// few-output pointwise layers (Cout <= 4): pointwise_wgrad_narrow_kernel.  Plan: chunk width V (elements, <= 16 bytes: the widest power of two
// dividing Cin and the row stride, with x aligned to it), chunk tiles of 256, pixel splits of about three 4-row batches per thread.
static bool narrow_wgrad_applies(const saunet_conv_desc* d)
{
    return is_pointwise(d) && !igemm_supported(d) && d->Cout >= 1 && d->Cout <= 4 && (d->dtype == SAUNET_F32 || d->dtype == SAUNET_BF16);
}
static void narrow_wgrad_plan(const saunet_conv_desc* d, const void* x, int* V_out, int* ctiles_out, long* splits_out, long* ppb_out)
{
    const int esz = d->dtype == SAUNET_BF16 ? 2 : 4;
    int V = 16 / esz;
    while (V > 1 && (d->Cin % V || d->ldx % V || ((uintptr_t)x % (size_t)(V * esz)))) V >>= 1;
    const int CH = d->Cin / V, ctiles = (CH + 255) / 256, rl = 256 / (CH < 256 ? CH : 256);
    const long P = (long)d->N * d->H * d->W;
    long splits = (P + 12L * rl - 1) / (12L * rl);
    if (splits > 2048 / ctiles) splits = 2048 / ctiles;
    if (splits < 1) splits = 1;
    const long ppb = (P + splits - 1) / splits;
    *V_out = V; *ctiles_out = ctiles; *splits_out = (P + ppb - 1) / ppb; *ppb_out = ppb;
}

int64_t saunet_conv2d_wgrad_workspace(const saunet_conv_desc* d)
{
    if (narrow_wgrad_applies(d)) {           // (the query sees no pointer: it assumes 16-byte aligned operands, the widest chunk = the most splits)
        int V, ctiles; long splits, ppb;
        narrow_wgrad_plan(d, nullptr, &V, &ctiles, &splits, &ppb);
        return (int64_t)splits * (d->Cin + 1) * d->Cout * (int64_t)sizeof(float);
    }
    if (convt_direct(d)) {
        size_t need = 0;
        int rc = tile_wgrad_convt(d, nullptr, nullptr, nullptr, nullptr, 0, &need, nullptr, nullptr);
        return rc == SAUNET_OK ? (int64_t)need : (int64_t)rc;
    }
    saunet_conv_desc flat;
    if (is_pointwise(d) && dense_pointwise_rows(d, &flat)) d = &flat;
    if (igemm_supported(d) && tile_wgrad_supported(d)) {
        size_t need = 0;
        int rc = tile_wgrad(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &need, true, nullptr);
        return rc == SAUNET_OK ? (int64_t)need : (int64_t)rc;
    }
    if (kUnalignedTileWgrad && !igemm_supported(d) && tile_wgrad_unaligned_supported(d)) {
        size_t need = 0;
        int rc = tile_wgrad(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &need, false, nullptr);
        return rc == SAUNET_OK ? (int64_t)need : (int64_t)rc;
    }
    return 0;
}

int saunet_conv2d_wgrad(const saunet_conv_desc* d, const void* x, const void* dy, const float* ps, const float* psh, float* dw,
                        void* workspace, int64_t workspace_bytes, void* stream)
{
    return saunet_conv2d_wgrad_deferred(d, x, dy, ps, psh, dw, workspace, workspace_bytes, nullptr, stream);
}

int64_t saunet_conv2d_wgrad_grouped_workspace(const saunet_wgrad_group* g)
{
    size_t need = 0;
    int rc = tile_wgrad_grouped(g, nullptr, 0, &need, nullptr);
    return rc == SAUNET_OK ? (int64_t)need : (int64_t)rc;
}

int saunet_conv2d_wgrad_grouped(const saunet_wgrad_group* g, void* workspace, int64_t workspace_bytes, void* stream)
{
    return tile_wgrad_grouped(g, workspace, (size_t)workspace_bytes, nullptr, (hipStream_t)stream);
}

int saunet_wgrad_reduce_multi(const saunet_wgrad_reduce_list* l, void* stream)
{
    if (l->count < 1 || l->count > SAUNET_WGRAD_REDUCE_MAX) return set_error(SAUNET_BAD_SHAPE, "wgrad_reduce_multi: %d entries", l->count);
    for (int e = 0; e < l->count; ++e)
        if (!l->item[e].ws || !l->item[e].dw || l->item[e].wsize < 1 || l->item[e].groups < 1 || (l->item[e].taps != 0 && l->item[e].taps != 9 && l->item[e].taps != 16) ||
            (l->item[e].taps > 0 && l->item[e].wsize % l->item[e].taps))
            return set_error(SAUNET_BAD_SHAPE, "wgrad_reduce_multi: entry %d is empty or malformed", e);
    return wgrad_reduce_multi(l, (hipStream_t)stream);
}

static int conv2d_wgrad_impl(const saunet_conv_desc* d, const void* x, const void* dy, const float* ps, const float* psh, float* dw, float* dbias,
                             void* workspace, int64_t workspace_bytes, saunet_wgrad_pending* pending, void* stream);

int saunet_conv2d_wgrad_deferred(const saunet_conv_desc* d, const void* x, const void* dy, const float* ps, const float* psh, float* dw,
                                 void* workspace, int64_t workspace_bytes, saunet_wgrad_pending* pending, void* stream)
{
    return conv2d_wgrad_impl(d, x, dy, ps, psh, dw, nullptr, workspace, workspace_bytes, pending, stream);
}

int saunet_conv2d_wgrad_bias_supported(const saunet_conv_desc* d)
{
    saunet_conv_desc flat;
    if (is_pointwise(d) && dense_pointwise_rows(d, &flat)) d = &flat;
    return narrow_wgrad_applies(d) ? 1 : 0;
}

int saunet_conv2d_wgrad_bias(const saunet_conv_desc* d, const void* x, const void* dy, const float* ps, const float* psh, float* dw, float* dbias,
                             void* workspace, int64_t workspace_bytes, void* stream)
{
    if (!dbias) return set_error(SAUNET_BAD_SHAPE, "wgrad_bias: no bias gradient buffer");
    if (!saunet_conv2d_wgrad_bias_supported(d)) return set_error(SAUNET_UNSUPPORTED, "wgrad_bias: geometry not served (saunet_conv2d_wgrad_bias_supported)");
    return conv2d_wgrad_impl(d, x, dy, ps, psh, dw, dbias, workspace, workspace_bytes, nullptr, stream);
}

static int conv2d_wgrad_impl(const saunet_conv_desc* d, const void* x, const void* dy, const float* ps, const float* psh, float* dw, float* dbias,
                             void* workspace, int64_t workspace_bytes, saunet_wgrad_pending* pending, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (pending) { pending->ws = nullptr; pending->dw = nullptr; pending->wsize = 0; pending->groups = 0; pending->taps = 0; }
    // (operands that are not 16-byte aligned -- an odd channel slice handed in through the C API -- take the generic path below; the workspace
    // query cannot see pointers and sizes for the direct kernel, which is the larger of the two)
    if (ps == nullptr && convt_direct(d) && (((uintptr_t)x | (uintptr_t)dy) & 15) == 0)
        return tile_wgrad_convt(d, x, dy, dw, workspace, (size_t)workspace_bytes, nullptr, st, pending);
    saunet_conv_desc flat;
    if (is_pointwise(d) && dense_pointwise_rows(d, &flat)) d = &flat;     // pixels are just rows for a 1x1 conv: any map shape tiles
    if (igemm_supported(d)) {
        if (tile_wgrad_supported(d)) return tile_wgrad(d, x, dy, ps, psh, dw, workspace, (size_t)workspace_bytes, nullptr, true, st, pending);
        return igemm_wgrad(d, x, dy, ps, psh, dw, st);
    }
    if (kUnalignedTileWgrad && tile_wgrad_unaligned_supported(d))
        return tile_wgrad(d, x, dy, ps, psh, dw, workspace, (size_t)workspace_bytes, nullptr, false, st, pending);
    if (!is_pointwise(d)) return set_error(SAUNET_UNSUPPORTED, "wgrad: %dx%d Cin=%d Cout=%d has no kernel", d->KH, d->KW, d->Cin, d->Cout);
    const int nW = d->Cin * d->Cout;
    PwWgradArgs a{x, dy, dw, ps, psh, (long)d->N * d->H * d->W, d->Cin, d->Cout, d->ldx, d->ldy, d->pro_relu, 0, (long)d->Cin, 1, 32, nullptr, nullptr};
    if (narrow_wgrad_applies(d)) {
        int V, ctiles; long splits;
        narrow_wgrad_plan(d, x, &V, &ctiles, &splits, &a.pix_per_block);
        const int nWp = d->Cout * (d->Cin + 1);                // a partial row carries the bias gradient as input channel Cin
        const size_t need = (size_t)splits * nWp * sizeof(float);
        a.dbias = dbias;
        // with caller scratch: per-split partials + an ordered reduce (deterministic, no contended atomics); without: one float atomic per weight and block
        a.part = (workspace != nullptr && (size_t)workspace_bytes >= need && (splits > 1 || dbias != nullptr)) ? (float*)workspace : nullptr;
        const size_t lds = sizeof(float) * 256 * 4 * V;
#define PW_NARROW(TT, VV) hipLaunchKernelGGL((pointwise_wgrad_narrow_kernel<TT, VV>), dim3(ctiles, (unsigned)splits), dim3(256), lds, st, a)
        if (d->dtype == SAUNET_F32) { if (V == 4) PW_NARROW(float, 4); else if (V == 2) PW_NARROW(float, 2); else PW_NARROW(float, 1); }
        else { if (V == 8) PW_NARROW(u16, 8); else if (V == 4) PW_NARROW(u16, 4); else if (V == 2) PW_NARROW(u16, 2); else PW_NARROW(u16, 1); }
#undef PW_NARROW
        SAUNET_CHECK_LAUNCH("pointwise_wgrad_narrow");
        if (a.part) {
            hipLaunchKernelGGL(pointwise_wgrad_narrow_reduce_kernel, dim3((nWp + 3) / 4), dim3(256), 0, st, a.part, (int)splits, nWp, d->Cin, dw, a.sM, a.sN, dbias);
            SAUNET_CHECK_LAUNCH("pointwise_wgrad_narrow_reduce");
        }
        return SAUNET_OK;
    }
    if (d->Cout <= 4 && d->Cin >= 64) {
        const int ctiles = (d->Cin + 255) / 256;
        long splits = 1024 / ctiles; if (splits > (a.P + 15) / 16) splits = (a.P + 15) / 16; if (splits < 1) splits = 1;
        a.pix_per_block = (a.P + splits - 1) / splits;
        splits = (a.P + a.pix_per_block - 1) / a.pix_per_block;
        if (d->dtype == SAUNET_F32) hipLaunchKernelGGL(pointwise_wgrad_fewout_kernel<float>, dim3(ctiles, (unsigned)splits), dim3(256), 0, st, a);
        else if (d->dtype == SAUNET_BF16) hipLaunchKernelGGL(pointwise_wgrad_fewout_kernel<u16>, dim3(ctiles, (unsigned)splits), dim3(256), 0, st, a);
        else return set_error(SAUNET_BAD_DTYPE, "wgrad: dtype %d", d->dtype);
        SAUNET_CHECK_LAUNCH("pointwise_wgrad_fewout");
        return SAUNET_OK;
    }
    if (d->dtype == SAUNET_BF16 && ps == nullptr && d->Cin % 8 == 0 && d->ldx % 8 == 0 && !((uintptr_t)x & 15) && d->Cin <= 64 && d->Cout <= 32 &&
        a.P >= 4096 && a.P < (1L << 32) - (1 << 20)) {
        const bool vd = d->Cout % 8 == 0 && d->ldy % 8 == 0 && !((uintptr_t)dy & 15);
        const int cit = d->Cin <= 32 ? 1 : 2;
        long blocks = (a.P + 255) / 256; if (blocks > 512) blocks = 512;
        const size_t lds = sizeof(u16) * 4 * (cit + 1) * G_TILE;
#define PW_MMA(CIT, VD) hipLaunchKernelGGL((pointwise_wgrad_mma_kernel<CIT, VD>), dim3((unsigned)blocks), dim3(256), lds, st, a)
        if (cit == 1) { if (vd) PW_MMA(1, true); else PW_MMA(1, false); }
        else { if (vd) PW_MMA(2, true); else PW_MMA(2, false); }
#undef PW_MMA
        SAUNET_CHECK_LAUNCH("pointwise_wgrad_mma");
        return SAUNET_OK;
    }
    long blocks = (a.P + 255) / 256; if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
    a.pix_per_block = ((a.P + blocks - 1) / blocks + 31) / 32 * 32;
    blocks = (a.P + a.pix_per_block - 1) / a.pix_per_block;
    a.rows = 32;
    while (a.rows > 1 && (size_t)a.rows * (d->Cin + d->Cout) * sizeof(float) > 60 * 1024) a.rows /= 2;
    size_t lds = (size_t)a.rows * (d->Cin + d->Cout) * sizeof(float);
    if (lds > 64 * 1024) return set_error(SAUNET_UNSUPPORTED, "pointwise wgrad: Cin+Cout too large (%d)", d->Cin + d->Cout);
#define PW_WGRAD(TT, WPT) hipLaunchKernelGGL((pointwise_wgrad_kernel<TT, WPT>), dim3((unsigned)blocks), dim3(256), lds, st, a)
    if (d->dtype == SAUNET_F32) {
        if (nW <= 256) PW_WGRAD(float, 1); else if (nW <= 1280) PW_WGRAD(float, 5); else if (nW <= 4096) PW_WGRAD(float, 16);
        else return set_error(SAUNET_UNSUPPORTED, "pointwise wgrad: %d weights", nW);
    } else if (d->dtype == SAUNET_BF16) {
        if (nW <= 256) PW_WGRAD(u16, 1); else if (nW <= 1280) PW_WGRAD(u16, 5); else if (nW <= 4096) PW_WGRAD(u16, 16);
        else return set_error(SAUNET_UNSUPPORTED, "pointwise wgrad: %d weights", nW);
    } else return set_error(SAUNET_BAD_DTYPE, "wgrad: dtype %d", d->dtype);
#undef PW_WGRAD
    SAUNET_CHECK_LAUNCH("pointwise_wgrad");
    return SAUNET_OK;
}

int saunet_channel_sum(int dtype, const void* x, int64_t pixels, int C, int ld, double* out, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    long blocks = (pixels * C + 8191) / 8192; if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
    long rpb = (pixels + blocks - 1) / blocks;
    blocks = (pixels + rpb - 1) / rpb;
    const size_t lds = sizeof(double) * C;
    if (C > 4096) return set_error(SAUNET_UNSUPPORTED, "channel_sum: C=%d", C);
    {
        const int V = dtype == SAUNET_BF16 ? 8 : 4;
        long rows = pixels; int cv = C, ldv = ld;
        bool vec = ((uintptr_t)x & 15) == 0 && (dtype == SAUNET_BF16 || dtype == SAUNET_F32);
        if (C % V == 0 && ld % V == 0) { /* whole chunks per pixel */ }
        else if (C < V && V % C == 0 && ld == C && (pixels * C) % V == 0) { rows = pixels * C / V; cv = V; ldv = V; }      // dense few-channel tensor: flat view
        else vec = false;
        if (vec) {
            long nb = (rows * (cv / V) + 2047) / 2048; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
            long rp = (rows + nb - 1) / nb; nb = (rows + rp - 1) / rp;
            const size_t l2 = sizeof(double) * cv;
            if (dtype == SAUNET_F32) hipLaunchKernelGGL((channel_sum_vec_kernel<float, 4>), dim3((unsigned)nb), dim3(256), l2, st, (const float*)x, rows, cv, ldv, C, rp, out);
            else hipLaunchKernelGGL((channel_sum_vec_kernel<u16, 8>), dim3((unsigned)nb), dim3(256), l2, st, (const u16*)x, rows, cv, ldv, C, rp, out);
            SAUNET_CHECK_LAUNCH("channel_sum");
            return SAUNET_OK;
        }
    }
    if (dtype == SAUNET_F32) hipLaunchKernelGGL(channel_sum_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, st, (const float*)x, (long)pixels, C, ld, rpb, out);
    else if (dtype == SAUNET_BF16) hipLaunchKernelGGL(channel_sum_kernel<u16>, dim3((unsigned)blocks), dim3(256), lds, st, (const u16*)x, (long)pixels, C, ld, rpb, out);
    else return set_error(SAUNET_BAD_DTYPE, "channel_sum: dtype %d", dtype);
    SAUNET_CHECK_LAUNCH("channel_sum");
    return SAUNET_OK;
}

}  // extern "C"
