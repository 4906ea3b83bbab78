/* ORACLE (test infrastructure, not product code): plain-C closed form of the reference's
 * image-space PSF application, float64 accumulation.  An independent second restatement
 * next to oracle/conv.py; tests compare both with the golden vectors.
 *
 *   oracle_render_psf_map  <- deeplens/render_psf.py:31-73   (reflect pad, flipped PSF per patch)
 *   oracle_local_psf_render <- deeplens/render_psf.py:76-107 (replicate pad, no flip, per-pixel PSF)
 */
#include <stddef.h>

static int reflect(int i, int n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; return i; }
static int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

void oracle_render_psf_map(const float* img, const float* psf_map, float* out,
                           int B, int C, int H, int W, int grid, int ks) {
    const int G = grid * ks, pad = ks / 2;
    for (int bc = 0; bc < B * C; ++bc) {
        const int c = bc % C;
        const float* plane = img + (size_t)bc * H * W;
        for (int i = 0; i < grid; ++i) {
            /* Python: int(i / grid * H), float64 (render_psf.py:65-66) */
            const int y0 = (int)((double)i / grid * H), y1 = (int)((double)(i + 1) / grid * H);
            for (int j = 0; j < grid; ++j) {
                const int x0 = (int)((double)j / grid * W), x1 = (int)((double)(j + 1) / grid * W);
                const float* k = psf_map + ((size_t)c * G + (size_t)i * ks) * G + (size_t)j * ks;
                for (int y = y0; y < y1; ++y)
                    for (int x = x0; x < x1; ++x) {
                        double acc = 0.0;
                        for (int u = 0; u < ks; ++u) {
                            const float* row = plane + (size_t)reflect(y - pad + u, H) * W;
                            for (int v = 0; v < ks; ++v)
                                acc += (double)row[reflect(x - pad + v, W)] * (double)k[(size_t)(ks - 1 - u) * G + (ks - 1 - v)];
                        }
                        out[(size_t)bc * H * W + (size_t)y * W + x] = (float)acc;
                    }
            }
        }
    }
}

void oracle_local_psf_render(const float* img, const float* psf, float* out,
                             int B, int C, int H, int W, int ks) {
    const int pad = ks / 2;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const float* plane = img + ((size_t)b * C + c) * H * W;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const float* k = psf + (((size_t)b * H + y) * W + x) * ks * ks;
                    double acc = 0.0;
                    for (int u = 0; u < ks; ++u)
                        for (int v = 0; v < ks; ++v)
                            acc += (double)plane[(size_t)clampi(y - pad + u, H) * W + clampi(x - pad + v, W)] * (double)k[u * ks + v];
                    out[((size_t)b * C + c) * H * W + (size_t)y * W + x] = (float)acc;
                }
        }
}
