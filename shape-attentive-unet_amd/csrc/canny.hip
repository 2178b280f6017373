// Canny branch on the device: removes the D2H -> cv2.Canny -> H2D round trip the reference performs
// inside SAUNet.forward (/root/reference/models/models.py:359-363).  Integer arithmetic, bit-exact with
// oracle/canny.c:  gray = uint8(trunc(mean_c x)) (mod 256) ; Sobel 3x3 (replicate border) ; L1 magnitude ;
// non-maximum suppression with the fixed-point tan(22.5deg) sector test ; hysteresis (low, high) ;
// output {0,255}.
#include "common.h"

namespace saunet {

#define CANNY_SHIFT 15
#define CANNY_TG22 13573

__device__ __forceinline__ int gray_u8(const float* __restrict__ img, int n, int H, int W, int y, int x)
{
    y = min(max(y, 0), H - 1); x = min(max(x, 0), W - 1);
    const long plane = (long)H * W;
    const float* b = img + (long)n * 3 * plane + (long)y * W + x;
    float s = b[0] + b[plane];
    s = s + b[2 * plane];
    float m = __fdiv_rn(s, 3.0f);
    return ((int)m) & 0xFF;   // trunc toward zero, low 8 bits (x86 numpy float32 -> uint8)
}

// pass 1: Sobel + magnitude.  work[0] = mag, work[1] = (dx & 0xffff) | (dy << 16)
__global__ __launch_bounds__(256) void canny_sobel_kernel(const float* __restrict__ img, int N, int H, int W, int32_t* __restrict__ work)
{
    const long total = (long)N * H * W;
    const long plane = (long)H * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int x = (int)(i % W); long t = i / W; int y = (int)(t % H); int n = (int)(t / H);
        int p[3][3];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) p[dy + 1][dx + 1] = gray_u8(img, n, H, W, y + dy, x + dx);
        int gx = (p[0][2] + 2 * p[1][2] + p[2][2]) - (p[0][0] + 2 * p[1][0] + p[2][0]);
        int gy = (p[2][0] + 2 * p[2][1] + p[2][2]) - (p[0][0] + 2 * p[0][1] + p[0][2]);
        int32_t* w = work + (long)n * 3 * plane;
        w[(long)y * W + x] = abs(gx) + abs(gy);
        w[plane + (long)y * W + x] = (gx & 0xffff) | (gy << 16);
    }
}

__device__ __forceinline__ int mag_at(const int32_t* mag, int H, int W, int y, int x)
{
    return ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? mag[(long)y * W + x] : 0;
}

// pass 2: NMS -> map (work[2]): 1 = not an edge, 0 = weak candidate, 2 = strong edge
__global__ __launch_bounds__(256) void canny_nms_kernel(int N, int H, int W, int low, int high, int32_t* __restrict__ work)
{
    const long total = (long)N * H * W;
    const long plane = (long)H * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int x = (int)(i % W); long t = i / W; int y = (int)(t % H); int n = (int)(t / H);
        const int32_t* mag = work + (long)n * 3 * plane;
        const int m = mag[(long)y * W + x];
        int out = 1;
        if (m > low) {
            const int32_t d = mag[plane + (long)y * W + x];
            const int xs = (int)(short)(d & 0xffff), ys = d >> 16;
            const int ax = abs(xs), ay = abs(ys) << CANNY_SHIFT;
            const int tg22x = ax * CANNY_TG22;
            bool keep;
            if (ay < tg22x) keep = (m > mag_at(mag, H, W, y, x - 1)) && (m >= mag_at(mag, H, W, y, x + 1));
            else {
                const int tg67x = tg22x + (ax << (CANNY_SHIFT + 1));
                if (ay > tg67x) keep = (m > mag_at(mag, H, W, y - 1, x)) && (m >= mag_at(mag, H, W, y + 1, x));
                else {
                    const int s = ((xs ^ ys) < 0) ? -1 : 1;
                    keep = (m > mag_at(mag, H, W, y - 1, x - s)) && (m > mag_at(mag, H, W, y + 1, x + s));
                }
            }
            if (keep) out = (m > high) ? 2 : 0;
        }
        work[(long)n * 3 * plane + 2 * plane + (long)y * W + x] = out;
    }
}

// pass 3: hysteresis by in-place relaxation (monotone: 0 -> 2 only), one workgroup per image, until a
// sweep changes nothing; then emit {0,255}.  Result = pixels 8-connected to a strong pixel through
// candidates, identical to the stack-based flood fill of the CPU algorithm.
// LDS = true: the whole map of the image lives in LDS as bytes (H*W <= 152 KiB, e.g. 256x256 = 64 KiB) and the sweeps
// never touch memory; otherwise the sweeps relax the int32 map in global memory.
template <typename T, bool LDS>
__global__ __launch_bounds__(1024) void canny_hyst_kernel(int H, int W, int32_t* __restrict__ work, T* __restrict__ out, int list_cap)
{
    extern __shared__ unsigned char h_map[];
    const long plane = (long)H * W;
    volatile int32_t* gmap = work + (long)blockIdx.x * 3 * plane + 2 * plane;
    volatile unsigned char* lmap = h_map;
    __shared__ int changed, ncand;
    const int npix = H * W;
    // weak candidates (map value 0) are a few percent of the pixels: the sweeps walk a compact list of them (built once, in LDS behind
    // the byte map) instead of the whole image; if the list overflows its LDS budget the sweeps fall back to scanning every pixel
    unsigned int* list = (unsigned int*)(h_map + ((npix + 15) & ~15));
    if (threadIdx.x == 0) ncand = 0;
    if constexpr (LDS) {
        for (int i = threadIdx.x; i < npix; i += 1024) lmap[i] = (unsigned char)gmap[i];
    }
    __syncthreads();
    auto at = [&](int i) -> int { if constexpr (LDS) return lmap[i]; else return gmap[i]; };
    bool use_list = false;
    if constexpr (LDS) {
        if (list_cap > 0) {
            for (int i = threadIdx.x; i < npix; i += 1024)
                if (lmap[i] == 0) { const int pos = atomicAdd(&ncand, 1); if (pos < list_cap) list[pos] = (unsigned)i; }
            __syncthreads();
            use_list = ncand <= list_cap;
        }
    }
    const int nwork = use_list ? ncand : npix;
    for (int it = 0; it < npix; ++it) {
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        int any = 0;
        for (int k = threadIdx.x; k < nwork; k += 1024) {
            const int i = use_list ? (int)list[k] : k;
            if (at(i) != 0) continue;
            const int y = i / W, x = i - y * W;
            bool hit = false;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if ((dy | dx) != 0 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && at(yy * W + xx) == 2) hit = true;
                }
            if (hit) { if constexpr (LDS) lmap[i] = 2; else gmap[i] = 2; any = 1; }
        }
        if (any) changed = 1;
        __threadfence_block();
        __syncthreads();
        const int c = changed;
        __syncthreads();
        if (!c) break;
    }
    T* o = out + (long)blockIdx.x * plane;
    for (int i = threadIdx.x; i < npix; i += 1024) Elem<T>::store(o + i, at(i) == 2 ? 255.f : 0.f);
}

// Edge ground truth on the device (replaces the loader's three Euclidean distance transforms per slice,
// /root/reference/data/ac17_dataloader.py:236-258): a pixel is an edge iff, for some class c in 1..3, a pixel of the
// opposite membership lies within Euclidean distance 2 (EDT(m) + EDT(1-m) <= 2); outside the image counts as
// "not class c" within the loader's one-pixel zero pad.
__global__ __launch_bounds__(256) void mask_to_edges_kernel(const int64_t* __restrict__ seg, int N, int H, int W, int num_classes, float* __restrict__ edge)
{
    const long total = (long)N * H * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W); long t = i / W; const int y = (int)(t % H); const int n = (int)(t / H);
        const int64_t* s = seg + (long)n * H * W;
        const int lab = (int)s[(long)y * W + x];
        bool e = false;
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                if (dy * dy + dx * dx > 4 || (dy == 0 && dx == 0)) continue;
                const int yy = y + dy, xx = x + dx;
                if (yy < -1 || yy > H || xx < -1 || xx > W) continue;             // beyond the one-pixel pad: no partner
                const bool inside = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                const int other = inside ? (int)s[(long)yy * W + xx] : 0;
                // membership differs for some class c in 1..num_classes  <=>  labels differ and one of them is a class
                if (other != lab && ((lab >= 1 && lab <= num_classes) || (other >= 1 && other <= num_classes))) e = true;
            }
        edge[i] = e ? 1.f : 0.f;
    }
}

// Test-set post-processing (/root/reference/test_and_pack.py:31-76): undo_crop followed by the order-0 resize back to the original
// grid, as ONE gather over the predicted labels:  out[z][Y][X] = p[z][floor((Y+.5)*h/H)][floor((X+.5)*w/W)]  with the un-cropped map
// p[y][x] = pred[by0 + y - top][bx0 + x - left] inside the pasted window (cw x ch at (left, top)) and 0 outside.
__global__ __launch_bounds__(256) void labels_uncrop_resize_kernel(const int64_t* __restrict__ pred, int Z, int th, int tw, int bx0, int by0, int cw, int ch,
                                                                   int left, int top, int w, int h, int W, int H, unsigned char* __restrict__ out)
{
    const long total = (long)Z * H * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int X = (int)(i % W); long t = i / W; const int Y = (int)(t % H); const int z = (int)(t / H);
        int y = (int)(((2L * Y + 1) * h) / (2L * H)), x = (int)(((2L * X + 1) * w) / (2L * W));     // floor((Y + 0.5) * h / H), exact in integers
        y = min(y, h - 1); x = min(x, w - 1);
        const int sy = y - top, sx = x - left;
        unsigned char v = 0;
        if ((unsigned)sy < (unsigned)ch && (unsigned)sx < (unsigned)cw) v = (unsigned char)pred[((long)z * th + by0 + sy) * tw + bx0 + sx];
        out[i] = v;
    }
}

}  // namespace saunet

using namespace saunet;

extern "C" int saunet_labels_uncrop_resize(const int64_t* pred, int Z, int th, int tw, int bx0, int by0, int cw, int ch, int left, int top,
                                           int w, int h, int W, int H, unsigned char* out, void* stream)
{
    if (Z <= 0 || th <= 0 || tw <= 0 || w <= 0 || h <= 0 || W <= 0 || H <= 0 || bx0 < 0 || by0 < 0 || bx0 + cw > tw || by0 + ch > th)
        return set_error(SAUNET_BAD_SHAPE, "labels_uncrop_resize: window %dx%d at (%d,%d) outside the %dx%d prediction", cw, ch, bx0, by0, tw, th);
    const long total = (long)Z * H * W;
    long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(labels_uncrop_resize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, Z, th, tw, bx0, by0, cw, ch, left, top, w, h, W, H, out);
    SAUNET_CHECK_LAUNCH("labels_uncrop_resize");
    return SAUNET_OK;
}

extern "C" int saunet_mask_to_edges(const int64_t* seg, int N, int H, int W, int num_classes, float* edge, void* stream)
{
    const long total = (long)N * H * W;
    long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(mask_to_edges_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, seg, N, H, W, num_classes, edge);
    SAUNET_CHECK_LAUNCH("mask_to_edges");
    return SAUNET_OK;
}


extern "C" int saunet_canny(int dtype, const float* image, int N, int H, int W, int low, int high, void* out, int32_t* work, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (low > high) { int t = low; low = high; high = t; }
    const long total = (long)N * H * W;
    long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(canny_sobel_kernel, dim3((unsigned)blocks), dim3(256), 0, st, image, N, H, W, work);
    hipLaunchKernelGGL(canny_nms_kernel, dim3((unsigned)blocks), dim3(256), 0, st, N, H, W, low, high, work);
    if (dtype != SAUNET_F32 && dtype != SAUNET_BF16) return set_error(SAUNET_BAD_DTYPE, "canny: dtype %d", dtype);
    const size_t map_bytes = (size_t)H * W;
    if (map_bytes <= 152 * 1024) {
        static DeviceOnce attr;
        if (attr.first()) {
            (void)hipFuncSetAttribute((const void*)canny_hyst_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
            (void)hipFuncSetAttribute((const void*)canny_hyst_kernel<u16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        }
        // candidate list behind the byte map: up to a quarter of the pixels (weak candidates are typically < 5 %), within the LDS budget
        const size_t map_al = (map_bytes + 15) & ~(size_t)15;
        size_t cap = map_bytes / 4;
        if (map_al + cap * 4 > 156 * 1024) cap = (156 * 1024 - map_al) / 4;
        const size_t lds = map_al + cap * 4;
        if (dtype == SAUNET_F32) hipLaunchKernelGGL((canny_hyst_kernel<float, true>), dim3(N), dim3(1024), lds, st, H, W, work, (float*)out, (int)cap);
        else hipLaunchKernelGGL((canny_hyst_kernel<u16, true>), dim3(N), dim3(1024), lds, st, H, W, work, (u16*)out, (int)cap);
    } else {
        if (dtype == SAUNET_F32) hipLaunchKernelGGL((canny_hyst_kernel<float, false>), dim3(N), dim3(1024), 0, st, H, W, work, (float*)out, 0);
        else hipLaunchKernelGGL((canny_hyst_kernel<u16, false>), dim3(N), dim3(1024), 0, st, H, W, work, (u16*)out, 0);
    }
    SAUNET_CHECK_LAUNCH("canny");
    return SAUNET_OK;
}
