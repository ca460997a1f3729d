// The strong view of an unlabeled batch, generated ON THE DEVICE from the weak view.
//
// Replaces, per image, the cv2 chain of the reference's data-loader workers (utils/datasets_ssod.py:520-570):
//   random_perspective_with_M  -> cv2.warpAffine(img, M[:2], borderValue 114)      utils/datasets_ssod.py:902-945
//   augment_hsv                -> BGR2HSV, three 256-entry LUTs, HSV2BGR            utils/augmentations.py:48-61
//   cutout                     -> rectangles filled with one colour each             utils/augmentations.py:382-398
//   np.flipud / np.fliplr                                                            utils/datasets_ssod.py:552-563
// The random draws (matrix, gains, rectangles, flips) stay on the host (efficientteacher_amd/utils/augment.py, the same
// recipe as the reference); this kernel is the per-pixel work, one thread per output pixel, all stages fused: the output
// pixel is traced back through the flips, takes a cutout colour if a rectangle covers it (the LAST covering rectangle, as
// successive assignments do), else is sampled from the weak view through the inverse affine map and colour-jittered.
//
// PARITY UNPINNED: cv2 is not installed in the build image and the reference holds no image goldens, so the arithmetic
// below RESTATES OpenCV's published 8-bit algorithms -- warpAffine's fixed-point bilinear sampling (10-bit coordinates,
// 5-bit fractions, 15-bit weights, constant border per tap), the integer RGB->HSV of cvtColor (hsv_shift 12 division
// tables, H in [0,180)) and the float HSV->RGB with cvRound -- and is tested for its invariants only.
#include "et_device.h"
#include "../../include/et_hip.h"
#include <math.h>

#define AUG_MAX_CUT 32

struct AugArgs {
    const unsigned char* src;   // (B,3,H,W) weak view, RGB planes
    unsigned char* dst;         // (B,3,H,W) strong view
    const double* minv;         // [B][6] dst -> src affine map (OpenCV's inverted matrix)
    const unsigned char* lut;   // [B][3][256] hue / sat / val look-up tables, or NULL (no colour jitter)
    const int* cut;             // [B][AUG_MAX_CUT][7] x0,y0,x1,y1 (half open), r,g,b ; unused entries x1 <= x0
    const int* flags;           // [B][3] n_cut, flipud, fliplr
    int B, H, W;
    int border;                 // 114
};

__device__ __forceinline__ int aug_round(double v) { return (int)rint(v); }       // cvRound: half to even (default rounding mode)

__global__ __launch_bounds__(256) void strong_view_kernel(AugArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long plane = (long long)a.H * a.W;
    if (i >= (long long)a.B * plane) return;
    const int b = (int)(i / plane);
    const int rem = (int)(i - (long long)b * plane);
    const int oy = rem / a.W, ox = rem - oy * a.W;
    const int* fl = a.flags + b * 3;
    const int y = fl[1] ? a.H - 1 - oy : oy, x = fl[2] ? a.W - 1 - ox : ox;      // position before the flips
    int r, g, bl;
    int hit = -1;
    const int* cuts = a.cut + (size_t)b * AUG_MAX_CUT * 7;
    for (int k = 0; k < fl[0] && k < AUG_MAX_CUT; ++k) {
        const int* c = cuts + k * 7;
        if (x >= c[0] && x < c[2] && y >= c[1] && y < c[3]) hit = k;
    }
    if (hit >= 0) {
        r = cuts[hit * 7 + 4]; g = cuts[hit * 7 + 5]; bl = cuts[hit * 7 + 6];
    } else {
        // ---- cv2.warpAffine, INTER_LINEAR, BORDER_CONSTANT: AB_BITS 10, INTER_BITS 5, weights * 2^15
        const double* M = a.minv + b * 6;
        const int round_delta = 16;                                           // AB_SCALE / INTER_TAB_SIZE / 2
        const int X0 = aug_round((M[1] * y + M[2]) * 1024.0) + round_delta, Y0 = aug_round((M[4] * y + M[5]) * 1024.0) + round_delta;
        const int X = (X0 + aug_round(M[0] * x * 1024.0)) >> 5, Y = (Y0 + aug_round(M[3] * x * 1024.0)) >> 5;
        const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        const unsigned char* base = a.src + (size_t)b * 3 * plane;
        int px[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const unsigned char* p = base + ch * plane;
            auto at = [&](int yy, int xx) -> int {
                return ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) ? (int)p[(size_t)yy * a.W + xx] : a.border;
            };
            const int v = w00 * at(sy, sx) + w01 * at(sy, sx + 1) + w10 * at(sy + 1, sx) + w11 * at(sy + 1, sx + 1);
            px[ch] = (v + (1 << 14)) >> 15;
        }
        r = px[0]; g = px[1]; bl = px[2];
        if (a.lut) {
            // ---- cvtColor RGB -> HSV (8 bit, H in [0,180)): integer division tables with hsv_shift = 12
            const int v = max(max(r, g), bl), vmin = min(min(r, g), bl);
            const int diff = v - vmin;
            const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
            const int sdiv = v ? aug_round((255 << 12) / (double)v) : 0;
            const int hdiv = diff ? aug_round((180 << 12) / (6.0 * diff)) : 0;
            const int s = (diff * sdiv + (1 << 11)) >> 12;
            int h = (vr & (g - bl)) + (~vr & ((vg & (bl - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
            h = (h * hdiv + (1 << 11)) >> 12;
            h += h < 0 ? 180 : 0;
            const unsigned char* L = a.lut + (size_t)b * 768;
            const int h2 = L[h & 255], s2 = L[256 + s], v2 = L[512 + v];
            // ---- cvtColor HSV -> RGB (8 bit): float sector arithmetic, cvRound to uchar
            const float fs = s2 * (1.f / 255.f), fv = v2 * (1.f / 255.f);
            float fr, fg, fb;
            if (s2 == 0) {
                fr = fg = fb = fv;
            } else {
                float hh = h2 * (6.f / 180.f);
                int sector = (int)floorf(hh);
                hh -= sector;
                if ((unsigned)sector >= 6u) { sector = 0; hh = 0.f; }
                const float tab[4] = {fv, fv * (1.f - fs), fv * (1.f - fs * hh), fv * (1.f - fs * (1.f - hh))};
                const int sd[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};      // b, g, r
                fb = tab[sd[sector][0]]; fg = tab[sd[sector][1]]; fr = tab[sd[sector][2]];
            }
            r = min(max(aug_round(fr * 255.f), 0), 255);
            g = min(max(aug_round(fg * 255.f), 0), 255);
            bl = min(max(aug_round(fb * 255.f), 0), 255);
        }
    }
    unsigned char* o = a.dst + (size_t)b * 3 * plane + (size_t)oy * a.W + ox;
    o[0] = (unsigned char)r; o[plane] = (unsigned char)g; o[2 * plane] = (unsigned char)bl;
}

extern "C" int et_strong_view_u8(const uint8_t* weak, uint8_t* strong, int B, int H, int W, const double* minv, const uint8_t* lut,
                                 const int* cutouts, const int* flags, int border_value, et_stream_t stream) {
    if (!weak || !strong || !minv || !cutouts || !flags) return -1;
    if (B <= 0 || H <= 0 || W <= 0) return -2;
    AugArgs a;
    a.src = weak; a.dst = strong; a.minv = minv; a.lut = lut; a.cut = cutouts; a.flags = flags;
    a.B = B; a.H = H; a.W = W; a.border = border_value;
    hipLaunchKernelGGL(strong_view_kernel, dim3(et_cdiv((long long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, a);
    ET_CHECK_LAUNCH();
    return 0;
}

// ---- 4-image mosaic (load_mosaic_with_M, utils/datasets_ssod.py:732-782) -----------------------------------------------------
// The reference pastes four images around a random centre into a 2s x 2s canvas of border value 114 and halves it with
// cv2.resize(img4, (s, s)).  At this exact 2:1 ratio OpenCV's INTER_LINEAR resize takes its fast INTER_AREA path: an output
// pixel is (a + b + c + d + 2) >> 2 of its 2 x 2 source pixels.  Here one thread produces one output pixel (three planes)
// straight from the four source images -- the canvas is never materialised: each of the 2 x 2 canvas positions belongs to the
// quadrant given by its side of the centre and reads tile i at (cy - y1a + y1b, cx - x1a + x1b) when it lies inside the
// tile's pasted rectangle, else the border value.  tiles: [B][4][8] int64 = {device pointer of the (3, h, w) uint8 planes, h, w,
// x1a, y1a, x2a, y2a, x1b << 32 | y1b} (efficientteacher_amd/utils/augment.py mosaic_layout).  PARITY: the placement is pinned on
// the reference (tests/golden/mosaic.npz); the 2:1 resampling restates OpenCV's published algorithm and is unpinned (no cv2).
__global__ __launch_bounds__(256) void mosaic4_kernel(const long long* __restrict__ tiles, unsigned char* __restrict__ out, int B, int S,
                                                      int border) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long plane = (long long)S * S;
    if (i >= (long long)B * plane) return;
    const int b = (int)(i / plane);
    const int rem = (int)(i - (long long)b * plane);
    const int oy = rem / S, ox = rem - oy * S;
    const long long* T = tiles + (size_t)b * 32;
    // the centre: x2a of the top-left tile is xc, its y2a is yc (mosaic_layout)
    const int xc = (int)T[5], yc = (int)T[6];
    int acc[3] = {2, 2, 2};                                  // the rounding term of (a + b + c + d + 2) >> 2
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int cy = 2 * oy + dy, cx = 2 * ox + dx;
            const int q = (cy >= yc ? 2 : 0) + (cx >= xc ? 1 : 0);
            const long long* t = T + q * 8;
            const int x1a = (int)t[3], y1a = (int)t[4], x2a = (int)t[5], y2a = (int)t[6];
            if (cx >= x1a && cx < x2a && cy >= y1a && cy < y2a) {
                const unsigned char* src = (const unsigned char*)(uintptr_t)t[0];
                const int h = (int)t[1], w = (int)t[2];
                const int x1b = (int)(t[7] >> 32), y1b = (int)(t[7] & 0xffffffffll);
                const size_t o = (size_t)(cy - y1a + y1b) * w + (cx - x1a + x1b), pl = (size_t)h * w;
                acc[0] += src[o]; acc[1] += src[pl + o]; acc[2] += src[2 * pl + o];
            } else {
                acc[0] += border; acc[1] += border; acc[2] += border;
            }
        }
    unsigned char* o = out + (size_t)b * 3 * plane + (size_t)oy * S + ox;
    o[0] = (unsigned char)(acc[0] >> 2); o[plane] = (unsigned char)(acc[1] >> 2); o[2 * plane] = (unsigned char)(acc[2] >> 2);
}

extern "C" int et_mosaic4_u8(const int64_t* tiles, uint8_t* out, int B, int S, int border_value, et_stream_t stream) {
    if (!tiles || !out) return -1;
    if (B <= 0 || S <= 0 || (long long)B * S * S >= (1ll << 40)) return -2;
    hipLaunchKernelGGL(mosaic4_kernel, dim3(et_cdiv((long long)B * S * S, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)tiles, out, B, S, border_value);
    ET_CHECK_LAUNCH();
    return 0;
}
