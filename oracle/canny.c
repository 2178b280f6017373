/* Canny edge detector, CPU restatement -- TEST INFRASTRUCTURE (see oracle/__init__.py).
 *
 * The reference calls the third-party  cv2.Canny(im_arr[i], 10, 100)
 * (/root/reference/models/models.py:362; requirements.txt:6 `opencv-python`, no
 * version pin).  OpenCV is neither vendored nor installed, so this file restates
 * OpenCV's published algorithm for 8-bit single-channel input, aperture 3,
 * L2gradient=false:
 *   1. Sobel 3x3 dx, dy in int16, BORDER_REPLICATE
 *   2. magnitude = |dx| + |dy|   (magnitude outside the image = 0)
 *   3. non-maximum suppression with the fixed-point tan(22.5 deg) sector test
 *      (TG22 = round(0.41421356 * 2^15)), keeping pixels with mag > low
 *   4. hysteresis: candidates 8-connected to a pixel with mag > high survive
 *   5. output 255 on edges, 0 elsewhere
 * PARITY UNPINNED against real OpenCV; pinned by the known-answer tests in
 * tests/test_oracle_canny.py only.
 *
 * Also here: the float -> uint8 conversion that precedes it in the reference,
 *   im = np.mean(x.cpu().numpy(), axis=1).astype(np.uint8)   (models.py:359)
 * which on x86 numpy is  trunc-to-int32 then low 8 bits  (z-scored inputs are
 * negative about half the time: -1.0 -> 255, -2.3 -> 254, 0.6 -> 0).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CANNY_SHIFT 15
#define TG22 13573 /* (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5) */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* mean over 3 channels then the reference's uint8 cast.  x: [3][H][W] float32 */
void saunet_oracle_gray_u8(const float* x, int H, int W, uint8_t* out)
{
    const int n = H * W;
    for (int i = 0; i < n; ++i) {
        /* numpy reduces the length-3 axis sequentially in float32, then divides by 3 */
        float s = x[i] + x[n + i];
        s = s + x[2 * n + i];
        float m = s / 3.0f;
        int32_t t = (int32_t)m; /* trunc toward zero (|m| << 2^31 for z-scored data) */
        out[i] = (uint8_t)(t & 0xFF);
    }
}

/* img: [H][W] uint8 -> edges: [H][W] uint8 in {0,255}.  Returns 0, or -1 on OOM. */
int saunet_oracle_canny(const uint8_t* img, int H, int W, int low, int high, uint8_t* edges)
{
    if (low > high) { int t = low; low = high; high = t; }
    const int n = H * W;
    int* mag = (int*)calloc((size_t)(H + 2) * (W + 2), sizeof(int));
    short* dxs = (short*)malloc(sizeof(short) * n);
    short* dys = (short*)malloc(sizeof(short) * n);
    uint8_t* map = (uint8_t*)malloc((size_t)(H + 2) * (W + 2));
    int* stack = (int*)malloc(sizeof(int) * (size_t)n);
    if (!mag || !dxs || !dys || !map || !stack) { free(mag); free(dxs); free(dys); free(map); free(stack); return -1; }
    const int ms = W + 2;
#define P(y, x) ((int)img[clampi((y), 0, H - 1) * W + clampi((x), 0, W - 1)])
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int gx = (P(y - 1, x + 1) + 2 * P(y, x + 1) + P(y + 1, x + 1)) -
                     (P(y - 1, x - 1) + 2 * P(y, x - 1) + P(y + 1, x - 1));
            int gy = (P(y + 1, x - 1) + 2 * P(y + 1, x) + P(y + 1, x + 1)) -
                     (P(y - 1, x - 1) + 2 * P(y - 1, x) + P(y - 1, x + 1));
            dxs[y * W + x] = (short)gx;
            dys[y * W + x] = (short)gy;
            mag[(y + 1) * ms + (x + 1)] = abs(gx) + abs(gy);
        }
#undef P
    /* map: 1 = not an edge, 0 = candidate, 2 = edge */
    memset(map, 1, (size_t)(H + 2) * (W + 2));
    int sp = 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int* m0 = mag + (y + 1) * ms + (x + 1);
            int m = m0[0];
            if (m <= low) continue;
            int xs = dxs[y * W + x], ys = dys[y * W + x];
            int ax = abs(xs);
            int ay = abs(ys) << CANNY_SHIFT;
            int tg22x = ax * TG22;
            int keep = 0;
            if (ay < tg22x) {
                keep = (m > m0[-1] && m >= m0[1]);
            } else {
                int tg67x = tg22x + (ax << (CANNY_SHIFT + 1));
                if (ay > tg67x) {
                    keep = (m > m0[-ms] && m >= m0[ms]);
                } else {
                    int s = ((xs ^ ys) < 0) ? -1 : 1;
                    keep = (m > m0[-ms - s] && m > m0[ms + s]);
                }
            }
            if (!keep) continue;
            int mi = (y + 1) * ms + (x + 1);
            if (m > high) { map[mi] = 2; stack[sp++] = mi; }
            else map[mi] = 0;
        }
    static const int dyo[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
    static const int dxo[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
    while (sp > 0) {
        int mi = stack[--sp];
        for (int k = 0; k < 8; ++k) {
            int ni = mi + dyo[k] * ms + dxo[k];
            if (map[ni] == 0) { map[ni] = 2; stack[sp++] = ni; }
        }
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            edges[y * W + x] = (map[(y + 1) * ms + (x + 1)] == 2) ? 255 : 0;
    free(mag); free(dxs); free(dys); free(map); free(stack);
    return 0;
}
