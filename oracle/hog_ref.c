/* Plain-C restatement of the MaskFeat HOG target -- TEST INFRASTRUCTURE ONLY.
 *
 * Restates reference dataset.py:39-45 (extract_hog_features) on top of the
 * skimage-0.18.3 arithmetic described in oracle/hog_oracle.py.  Differences
 * from the NumPy restatement are deliberate: the orientation bin is decided
 * with exact sign tests on the integer gradients instead of atan2 (integer
 * gradients can only touch the 0-degree boundary), which is the
 * form the HIP kernel uses; tests/test_hog_oracle.py proves both forms agree
 * for every possible gradient pair in [-255,255]^2.
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libhogref.so oracle/hog_ref.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORIENT 9
#define CELL 8

/* cos/sin of 20k degrees, k = 1..8, rounded-to-nearest doubles */
static const double BC[8] = {
    0.93969262078590838, 0.76604444311897804, 0.5, 0.17364817766693035,
    -0.17364817766693035, -0.5, -0.76604444311897804, -0.93969262078590838};
static const double BS[8] = {
    0.34202014332566873, 0.64278760968653933, 0.8660254037844386, 0.98480775301220806,
    0.98480775301220806, 0.8660254037844386, 0.64278760968653933, 0.34202014332566873};

int vtx_ref_hog_bin(int g_row, int g_col) {
    if (g_row == 0 && g_col == 0) return 0;
    if (g_row < 0 || (g_row == 0 && g_col < 0)) { g_row = -g_row; g_col = -g_col; }
    int b = 0;
    for (int k = 0; k < 8; ++k)
        b += (BC[k] * (double)g_row - BS[k] * (double)g_col >= 0.0) ? 1 : 0;
    return b;
}

/* img: [H,W,3] uint8 interleaved (one frame).  out: [H/16, W/16, 108] float64.
 * bins (optional): [3,H,W] int32. */
void vtx_ref_hog_frame(const uint8_t* img, int H, int W, double* out, int32_t* bins) {
    int nr = H / CELL, nc = W / CELL;
    for (int ch = 0; ch < 3; ++ch)
        for (int cr = 0; cr < nr; ++cr)
            for (int cc = 0; cc < nc; ++cc) {
                float tot[ORIENT];
                for (int i = 0; i < ORIENT; ++i) tot[i] = 0.0f;
                for (int py = 0; py < CELL; ++py)
                    for (int px = 0; px < CELL; ++px) {
                        int y = cr * CELL + py, x = cc * CELL + px;
                        int gr = 0, gc = 0;
                        if (y > 0 && y < H - 1)
                            gr = (int)img[((y + 1) * W + x) * 3 + ch] - (int)img[((y - 1) * W + x) * 3 + ch];
                        if (x > 0 && x < W - 1)
                            gc = (int)img[(y * W + x + 1) * 3 + ch] - (int)img[(y * W + x - 1) * 3 + ch];
                        int b = vtx_ref_hog_bin(gr, gc);
                        if (bins) bins[(ch * H + y) * W + x] = b;
                        double mag = hypot((double)gc, (double)gr);   /* numpy.hypot == libm hypot (not always == sqrt) */
                        tot[b] = (float)((double)tot[b] + mag);
                    }
                double h[ORIENT];
                for (int i = 0; i < ORIENT; ++i) h[i] = (double)(tot[i] / 64.0f);
                /* numpy pairwise sum of 9 squares */
                double s = ((h[0]*h[0] + h[1]*h[1]) + (h[2]*h[2] + h[3]*h[3])) +
                           ((h[4]*h[4] + h[5]*h[5]) + (h[6]*h[6] + h[7]*h[7]));
                s += h[8]*h[8];
                double nrm = sqrt(s + 1e-5 * 1e-5);
                int ph = cr >> 1, dh = cr & 1, pw = cc >> 1, dw = cc & 1;
                double* o = out + ((size_t)(ph * (nc / 2) + pw)) * 108 + dh * 54 + dw * 27 + ch * 9;
                for (int i = 0; i < ORIENT; ++i) o[i] = h[i] / nrm;
            }
}
