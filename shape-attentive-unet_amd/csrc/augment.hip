// Training-time augmentation of ACDC slices on the device.  Replaces the loader's per-slice CPU chain
//   PaddingCenterCrop(256) -> RandomHorizontallyFlip -> RandomVerticallyFlip -> RandomRotate(180)   /root/reference/data/augmentations.py:223-264, 308-331, 392-412
//   augment_gamma -> per-slice z-score                                                              /root/reference/data/ac17_dataloader.py:22-57, 139-150
//   random_elastic_deformation(alpha=500, sigma=20) on [image, mask] (order-1, mode='nearest')        /root/reference/data/ac17_dataloader.py:196-216, 260-287
// with four kernels over a zero-padded batch of raw slices:
//   augment_geometric   crop/pad, flips and the rotation are ONE gather per output pixel (bilinear for the image, nearest for the mask)
//   augment_gamma_zs    one workgroup per slice: min/max -> gamma curve -> mean/std -> z-score (three sweeps over an L2-resident slice)
//   gauss_blur_rows/cols + elastic_warp   displacement fields = separable Gaussian of uniform noise (zero boundary) * alpha; bilinear warp
//   uniform_noise       counter-based hash generator (the reference draws from an unseeded numpy RandomState: only the distribution matters)
// The edge ground truth of the warped mask is saunet_mask_to_edges (canny.hip).
#include "common.h"

namespace saunet {

struct GeoParams { int h, w, oy, ox, hflip, vflip, rotate; float cosa, sina; };

// source lookup of crop-space integer pixel (yc, xc): un-flip, shift by the crop/pad offset, zero outside the slice
__device__ __forceinline__ float geo_fetch(const float* __restrict__ src, int ld, const GeoParams& g, int S, int yc, int xc)
{
    if (g.hflip) xc = S - 1 - xc;
    if (g.vflip) yc = S - 1 - yc;
    const int ys = yc + g.oy, xs = xc + g.ox;
    return ((unsigned)ys < (unsigned)g.h && (unsigned)xs < (unsigned)g.w) ? src[(long)ys * ld + xs] : 0.f;
}

__global__ __launch_bounds__(256) void augment_geometric_kernel(const float* __restrict__ img, const float* __restrict__ seg, int Hm, int Wm,
                                                                const GeoParams* __restrict__ params, int B, int S,
                                                                float* __restrict__ out_img, float* __restrict__ out_seg)
{
    const long total = (long)B * S * S;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % S); long t = i / S; const int y = (int)(t % S); const int b = (int)(t / S);
        const GeoParams g = params[b];
        const float* si = img + (long)b * Hm * Wm;
        const float* ss = seg + (long)b * Hm * Wm;
        float vi, vs;
        if (!g.rotate) { vi = geo_fetch(si, Wm, g, S, y, x); vs = geo_fetch(ss, Wm, g, S, y, x); }
        else {
            // torchvision.transforms.functional.affine (inverse matrix about center = S/2 + 0.5) evaluated by PIL at pixel centres
            const float c = 0.5f * S + 0.5f;
            const float X = x + 0.5f - c, Y = y + 0.5f - c;
            const float xin = g.cosa * X + g.sina * Y + c, yin = -g.sina * X + g.cosa * Y + c;
            vi = 0.f; vs = 0.f;
            if (xin >= 0.f && xin < (float)S && yin >= 0.f && yin < (float)S) {
                vs = geo_fetch(ss, Wm, g, S, (int)floorf(yin), (int)floorf(xin));           // NEAREST
                const float xf = xin - 0.5f, yf = yin - 0.5f;                              // BILINEAR with edge-clipped neighbours
                const int x0 = (int)floorf(xf), y0 = (int)floorf(yf);
                const float dx = xf - x0, dy = yf - y0;
                const int xa = min(max(x0, 0), S - 1), xb = min(max(x0 + 1, 0), S - 1), ya = min(max(y0, 0), S - 1), yb = min(max(y0 + 1, 0), S - 1);
                const float v00 = geo_fetch(si, Wm, g, S, ya, xa), v01 = geo_fetch(si, Wm, g, S, ya, xb);
                const float v10 = geo_fetch(si, Wm, g, S, yb, xa), v11 = geo_fetch(si, Wm, g, S, yb, xb);
                const float r0 = v00 + (v01 - v00) * dx, r1 = v10 + (v11 - v10) * dx;
                vi = r0 + (r1 - r0) * dy;
            }
        }
        out_img[i] = vi; out_seg[i] = vs;
    }
}

__device__ __forceinline__ double block_sum(double v, double* sh)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
    return s;
}
__device__ __forceinline__ float block_minmax(float v, bool is_max, float* sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float u = __shfl_xor(v, o, 64); v = is_max ? fmaxf(v, u) : fminf(v, u); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) s = is_max ? fmaxf(s, sh[i]) : fminf(s, sh[i]);
    return s;
}

// one workgroup per slice: x <- zscore(gamma_curve(x)).  gamma[b] <= 0 skips the gamma curve (z-score only).
__global__ __launch_bounds__(1024) void augment_gamma_zs_kernel(float* __restrict__ x, int npix, const float* __restrict__ gamma)
{
    __shared__ double shd[16];
    __shared__ float shf[16];
    float* p = x + (long)blockIdx.x * npix;
    const float gm = gamma[blockIdx.x];
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (int i = threadIdx.x; i < npix; i += 1024) { const float v = p[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    mn = block_minmax(mn, false, shf); mx = block_minmax(mx, true, shf);
    const float rng = mx - mn;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < npix; i += 1024) {
        float v = p[i];
        if (gm > 0.f) { v = powf((v - mn) / (rng + 1e-7f), gm) * rng + mn; p[i] = v; }
        s1 += v; s2 += (double)v * v;
    }
    s1 = block_sum(s1, shd); s2 = block_sum(s2, shd);
    const double mu = s1 / npix;
    double var = s2 / npix - mu * mu; if (var < 0.0) var = 0.0;
    const float mf = (float)mu, inv = (float)(1.0 / (sqrt(var) + 1e-10));
    for (int i = threadIdx.x; i < npix; i += 1024) p[i] = (p[i] - mf) * inv;
}

// uniform [0, 1) noise from a counter hash (two rounds of a 32-bit finaliser over (seed, index))
__device__ __forceinline__ unsigned int hash32(unsigned int v)
{
    v ^= v >> 16; v *= 0x7feb352dU; v ^= v >> 15; v *= 0x846ca68bU; v ^= v >> 16;
    return v;
}
__global__ __launch_bounds__(256) void uniform_noise_kernel(unsigned long long seed, float* __restrict__ out, long n)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned int h = hash32((unsigned int)i ^ hash32((unsigned int)(i >> 32) + (unsigned int)seed) ^ (unsigned int)(seed >> 32) * 0x9e3779b9U);
        out[i] = (float)(h >> 8) * (1.0f / 16777216.0f);
    }
}

// separable Gaussian, zero boundary (scipy.ndimage.gaussian_filter(..., mode='constant', cval=0)); wts[0..R] = normalised half kernel;
// `affine`: in = 2*u - 1 on the fly (the noise field), out scaled by `scale` (alpha) in the second pass
__global__ __launch_bounds__(256) void gauss_blur_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, const float* __restrict__ wts,
                                                         int R, int along_x, int affine, float scale)
{
    const long total = (long)B * H * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W); long t = i / W; const int y = (int)(t % H); const int b = (int)(t / H);
        const float* p = in + (long)b * H * W;
        float s = 0.f;
        for (int k = -R; k <= R; ++k) {
            const int yy = along_x ? y : y + k, xx = along_x ? x + k : x;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                float v = p[(long)yy * W + xx];
                if (affine) v = 2.f * v - 1.f;
                s = fmaf(wts[k < 0 ? -k : k], v, s);
            }
        }
        out[i] = s * scale;
    }
}

__device__ __forceinline__ float bilinear_clamped(const float* __restrict__ p, int H, int W, float r, float c)
{
    r = fminf(fmaxf(r, 0.f), (float)(H - 1)); c = fminf(fmaxf(c, 0.f), (float)(W - 1));      // mode='nearest'
    const int r0 = (int)floorf(r), c0 = (int)floorf(c);
    const int r1 = min(r0 + 1, H - 1), c1 = min(c0 + 1, W - 1);
    const float fr = r - r0, fc = c - c0;
    const float v00 = p[(long)r0 * W + c0], v01 = p[(long)r0 * W + c1], v10 = p[(long)r1 * W + c0], v11 = p[(long)r1 * W + c1];
    return (v00 * (1.f - fc) + v01 * fc) * (1.f - fr) + (v10 * (1.f - fc) + v11 * fc) * fr;
}

// out(r, c) = in(r + dr(r, c), c + dc(r, c)) for image and mask (both order 1, like the reference); apply[b] == 0 copies the slice through.
// Mask outputs: seg_f (the interpolated float mask), seg_l = trunc (what `.long()` makes of it in the loss), seg_e = the value if it is an exact
// class id 1..3 else 0 (what `mask == c` sees in mask_to_edges).
__global__ __launch_bounds__(256) void elastic_warp_kernel(const float* __restrict__ img, const float* __restrict__ seg, const float* __restrict__ dr,
                                                           const float* __restrict__ dc, const int* __restrict__ apply, int B, int H, int W,
                                                           float* __restrict__ out_img, float* __restrict__ seg_f, int64_t* __restrict__ seg_l,
                                                           int64_t* __restrict__ seg_e)
{
    const long total = (long)B * H * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % W); long t = i / W; const int r = (int)(t % H); const int b = (int)(t / H);
        float vi, vs;
        if (apply[b]) {
            const float rr = r + dr[i], cc = c + dc[i];
            vi = bilinear_clamped(img + (long)b * H * W, H, W, rr, cc);
            vs = bilinear_clamped(seg + (long)b * H * W, H, W, rr, cc);
        } else { vi = img[i]; vs = seg[i]; }
        out_img[i] = vi;
        if (seg_f) seg_f[i] = vs;
        const float tr = truncf(vs);
        if (seg_l) seg_l[i] = (int64_t)tr;
        if (seg_e) seg_e[i] = (vs == tr && tr >= 1.f && tr <= 3.f) ? (int64_t)tr : 0;
    }
}

static inline unsigned blocks_for(long total) { long b = (total + 255) / 256; if (b > 8192) b = 8192; if (b < 1) b = 1; return (unsigned)b; }

}  // namespace saunet

using namespace saunet;

extern "C" {

int saunet_augment_geometric(const float* img, const float* seg, int B, int Hm, int Wm, const void* params, int S, float* out_img, float* out_seg, void* stream)
{
    if (B <= 0 || Hm <= 0 || Wm <= 0 || S <= 0 || !params) return set_error(SAUNET_BAD_SHAPE, "augment_geometric: B=%d Hm=%d Wm=%d S=%d", B, Hm, Wm, S);
    hipLaunchKernelGGL(augment_geometric_kernel, dim3(blocks_for((long)B * S * S)), dim3(256), 0, (hipStream_t)stream, img, seg, Hm, Wm, (const GeoParams*)params, B, S, out_img, out_seg);
    SAUNET_CHECK_LAUNCH("augment_geometric");
    return SAUNET_OK;
}

int saunet_augment_gamma_zscore(float* x, int B, int npix, const float* gamma, void* stream)
{
    if (B <= 0 || npix <= 0) return set_error(SAUNET_BAD_SHAPE, "augment_gamma_zscore: B=%d npix=%d", B, npix);
    hipLaunchKernelGGL(augment_gamma_zs_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, npix, gamma);
    SAUNET_CHECK_LAUNCH("augment_gamma_zscore");
    return SAUNET_OK;
}

int saunet_uniform_noise(uint64_t seed, float* out, int64_t n, void* stream)
{
    hipLaunchKernelGGL(uniform_noise_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, (unsigned long long)seed, out, (long)n);
    SAUNET_CHECK_LAUNCH("uniform_noise");
    return SAUNET_OK;
}

int saunet_gauss_blur(const float* in, float* tmp, float* out, int B, int H, int W, const float* weights, int radius, int affine_2u_minus_1, float scale, void* stream)
{
    if (B <= 0 || H <= 0 || W <= 0 || radius < 0 || !weights) return set_error(SAUNET_BAD_SHAPE, "gauss_blur: B=%d H=%d W=%d R=%d", B, H, W, radius);
    const unsigned nb = blocks_for((long)B * H * W);
    hipStream_t st = (hipStream_t)stream;
    // scipy filters axis 0 first, then axis 1 (rows direction = along y first); the result is the same either way up to rounding
    hipLaunchKernelGGL(gauss_blur_kernel, dim3(nb), dim3(256), 0, st, in, tmp, B, H, W, weights, radius, 0, affine_2u_minus_1, 1.f);
    hipLaunchKernelGGL(gauss_blur_kernel, dim3(nb), dim3(256), 0, st, tmp, out, B, H, W, weights, radius, 1, 0, scale);
    SAUNET_CHECK_LAUNCH("gauss_blur");
    return SAUNET_OK;
}

int saunet_elastic_warp(const float* img, const float* seg, const float* drow, const float* dcol, const int* apply, int B, int H, int W,
                        float* out_img, float* seg_f, int64_t* seg_long, int64_t* seg_edge, void* stream)
{
    if (B <= 0 || H <= 0 || W <= 0 || !apply) return set_error(SAUNET_BAD_SHAPE, "elastic_warp: B=%d H=%d W=%d", B, H, W);
    hipLaunchKernelGGL(elastic_warp_kernel, dim3(blocks_for((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, img, seg, drow, dcol, apply, B, H, W, out_img, seg_f, seg_long, seg_edge);
    SAUNET_CHECK_LAUNCH("elastic_warp");
    return SAUNET_OK;
}

}  // extern "C"
