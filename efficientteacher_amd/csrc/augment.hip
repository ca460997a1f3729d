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
