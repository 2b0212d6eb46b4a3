/* CPU ORACLE (test infrastructure, not product code): OpenCV's 8-bit INTER_LINEAR `cv2.resize`, restated in plain scalar C
 * from the published algorithm of opencv 4.8 modules/imgproc/src/resize.cpp (the reference calls it at
 * easy_ViTPose/inference.py:316 through opencv-python==4.8.0.76, requirements.txt:25 -- a third-party dependency that is
 * absent from /root/reference and from this image, hence PARITY UNPINNED against the OpenCV binary itself):
 *
 *   - destination pixel centre (d + 0.5) * scale - 0.5 with scale = 1 / (dsize / ssize), evaluated in double, stored as float;
 *     s = floor, f = fraction; s < 0 -> (s, f) = (0, 0); s >= ssize - 1 -> (s, f) = (ssize - 1, 0)
 *   - 11-bit fixed-point coefficients (INTER_RESIZE_COEF_BITS = 11): a1 = cvRound(f * 2048), a0 = cvRound((1 - f) * 2048),
 *     cvRound = round half to even; both as int16
 *   - horizontal pass into int32: row[d] = src[s] * a0 + src[s + 1] * a1 (src[s] * 2048 where the second tap is outside)
 *   - vertical pass: dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
 *   - exactly 2x down-scaling in both directions is routed to the 2x2 box average (INTER_AREA fast path): (a + b + c + d + 2) >> 2
 *
 * Independent of easy_vitpose_amd/cropprep.py (numpy, vectorised) and of the HIP kernel: those two are the things checked. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static int round_half_even(float v) { return (int)lrintf(v); /* default FE_TONEAREST = ties to even, what cvRound does */ }

static void axis(int dsize, int ssize, int* s0, int* s1, int* a0, int* a1, double* scale_out) {
    const double inv = (double)dsize / (double)ssize;
    const double scale = 1.0 / inv;
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= ssize - 1) { s = ssize - 1; f = 0.f; }
        s0[d] = s;
        s1[d] = s + 1 < ssize ? s + 1 : ssize - 1;
        a1[d] = round_half_even(f * 2048.f);
        a0[d] = round_half_even((1.f - f) * 2048.f);
    }
    *scale_out = scale;
}

/* src: uint8 [sh, sw, ch] -> dst: uint8 [dh, dw, ch]; returns 0 on success */
int resize_linear_u8_ref(const uint8_t* src, int sh, int sw, int ch, uint8_t* dst, int dh, int dw) {
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || ch <= 0) return 1;
    if (sh == dh && sw == dw) {
        for (long i = 0; i < (long)sh * sw * ch; ++i) dst[i] = src[i];
        return 0;
    }
    int* sx0 = malloc(sizeof(int) * 4 * dw);
    int* sy0 = malloc(sizeof(int) * 4 * dh);
    if (!sx0 || !sy0) { free(sx0); free(sy0); return 2; }
    int *sx1 = sx0 + dw, *ax0 = sx0 + 2 * dw, *ax1 = sx0 + 3 * dw;
    int *sy1 = sy0 + dh, *ay0 = sy0 + 2 * dh, *ay1 = sy0 + 3 * dh;
    double scx, scy;
    axis(dw, sw, sx0, sx1, ax0, ax1, &scx);
    axis(dh, sh, sy0, sy1, ay0, ay1, &scy);
    if (scx == 2.0 && scy == 2.0) {
        for (int y = 0; y < dh; ++y)
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < ch; ++c) {
                    const uint8_t* p = src + ((long)(2 * y) * sw + 2 * x) * ch + c;
                    dst[((long)y * dw + x) * ch + c] = (uint8_t)((p[0] + p[ch] + p[(long)sw * ch] + p[(long)sw * ch + ch] + 2) >> 2);
                }
    } else {
        for (int y = 0; y < dh; ++y)
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < ch; ++c) {
                    const uint8_t* r0 = src + (long)sy0[y] * sw * ch + c;
                    const uint8_t* r1 = src + (long)sy1[y] * sw * ch + c;
                    const int32_t h0 = r0[(long)sx0[x] * ch] * ax0[x] + r0[(long)sx1[x] * ch] * ax1[x];
                    const int32_t h1 = r1[(long)sx0[x] * ch] * ax0[x] + r1[(long)sx1[x] * ch] * ax1[x];
                    const int32_t v = (((ay0[y] * (h0 >> 4)) >> 16) + ((ay1[y] * (h1 >> 4)) >> 16) + 2) >> 2;
                    dst[((long)y * dw + x) * ch + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                }
    }
    free(sx0); free(sy0);
    return 0;
}
