// GPU-side input pipeline: the transform chain of aihc_utils/image_transform.py:50-84 -
//   Resize((S, S)) -> RandomHorizontalFlip -> RandomRotation -> RandomCrop / CenterCrop -> ToTensor -> Normalize
// fused into ONE gather kernel over decoded uint8 HWC images of ragged sizes, bit-exact against Pillow's integer arithmetic
// (what torchvision's PIL backend calls): the separable antialiased BILINEAR resample in 22-bit fixed point with a uint8
// intermediate after the horizontal pass, NEAREST rotation through libImaging's 16.16 fixed-point affine map, fill 0.
// One thread per output pixel (3 channels); the host supplies, per sample, the coefficient tables of its two axes and the affine
// increments (computed once per distinct size in double, as Pillow does), so the device code is integer-only up to the final
// (v / 255 - mean) / std.  HBM-bound by the f32 CHW output (602 KB per 224^2 image) plus the touched part of the source.
#include "common.cuh"
#include "kernels.h"

namespace mfvit {

namespace {

constexpr int PREC = 32 - 8 - 2;   // Pillow PRECISION_BITS

__device__ __forceinline__ int clip8(int v) {
    v >>= PREC;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// desc[s][20]: 0 src byte offset of the (cropped) source's top-left pixel, 1 in_h, 2 in_w, 3 x-table offset (int32 units), 4 y-table offset, 5 ksize_x, 6 ksize_y,
//              7 flip, 8 rot mode (0 none, 1 affine, 2 / 3 / 4 = transpose 90 / 180 / 270), 9..14 a0 a1 a2 a3 a4 a5 (16.16), 15 crop_i << 32 | crop_j,
//              16 source row pitch in bytes (a RandomResizedCrop box is a window of a wider image), 17..19 unused
// table row xx of an axis: [xmin, count, k[0..ksize)]
__global__ __launch_bounds__(256) void input_transform_kernel(const unsigned char* __restrict__ src, const long long* __restrict__ desc,
                                                              const int* __restrict__ tables, int S, int crop, float m0, float m1, float m2,
                                                              float s0, float s1, float s2, float* __restrict__ out) {
    const int smp = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= crop * crop) return;
    const long long* d = desc + (long)smp * 20;
    const long pitch = d[16];
    const int in_h = (int)d[1], in_w = (int)d[2];
    const unsigned char* img = src + d[0];
    const int x = pix % crop, y = pix / crop;
    const int xc = x + (int)(d[15] & 0xffffffffll), yc = y + (int)(d[15] >> 32);
    // inverse rotation: position in the flipped, resized S x S image
    int xr = xc, yr = yc;
    bool inside = true;
    const int mode = (int)d[8];
    if (mode == 1) {
        const long long xx = d[11] + d[10] * yc + d[9] * xc, yy = d[14] + d[13] * yc + d[12] * xc;
        xr = (int)(xx >> 16);
        yr = (int)(yy >> 16);
        inside = xr >= 0 && xr < S && yr >= 0 && yr < S;
    } else if (mode == 2) {        // ROTATE_90 (counter-clockwise): out[i][j] = in[j][S-1-i]
        xr = S - 1 - yc; yr = xc;
    } else if (mode == 3) {
        xr = S - 1 - xc; yr = S - 1 - yc;
    } else if (mode == 4) {        // ROTATE_270: out[i][j] = in[S-1-j][i]
        xr = yc; yr = S - 1 - xc;
    }
    int v0 = 0, v1 = 0, v2 = 0;    // fill colour 0
    if (inside) {
        const int xf = d[7] ? S - 1 - xr : xr;
        const int ksx = (int)d[5], ksy = (int)d[6];
        const int* tx = tables + d[3] + (long)xf * (2 + ksx);
        const int* ty = tables + d[4] + (long)yr * (2 + ksy);
        const bool hres = in_w != S, vres = in_h != S;
        const int x0 = hres ? tx[0] : xf, nx = hres ? tx[1] : 1;
        const int y0 = vres ? ty[0] : yr, ny = vres ? ty[1] : 1;
        int a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
        for (int r = 0; r < ny; ++r) {
            const unsigned char* row = img + (long)(y0 + r) * pitch + (long)x0 * 3;
            int h0, h1, h2;
            if (hres) {
                int b0 = 1 << (PREC - 1), b1 = b0, b2 = b0;
                for (int k = 0; k < nx; ++k) {
                    const int c = tx[2 + k];
                    b0 += row[3 * k] * c;
                    b1 += row[3 * k + 1] * c;
                    b2 += row[3 * k + 2] * c;
                }
                h0 = clip8(b0), h1 = clip8(b1), h2 = clip8(b2);       // the horizontal pass lands in a uint8 image
            } else {
                h0 = row[0], h1 = row[1], h2 = row[2];
            }
            if (vres) {
                const int c = ty[2 + r];
                a0 += h0 * c, a1 += h1 * c, a2 += h2 * c;
            } else {
                v0 = h0, v1 = h1, v2 = h2;
            }
        }
        if (vres) v0 = clip8(a0), v1 = clip8(a1), v2 = clip8(a2);
    }
    // ToTensor (uint8 -> float / 255) and Normalize ((x - mean) / std), IEEE f32 like the torch ops they replace
    float* o = out + ((long)smp * 3) * crop * crop + pix;
    o[0] = __fdiv_rn(__fdiv_rn((float)v0, 255.0f) - m0, s0);
    o[(long)crop * crop] = __fdiv_rn(__fdiv_rn((float)v1, 255.0f) - m1, s1);
    o[2L * crop * crop] = __fdiv_rn(__fdiv_rn((float)v2, 255.0f) - m2, s2);
}

}  // namespace

int input_transform(const unsigned char* src, const long long* desc, const int* tables, int n, int S, int crop, const float* mean,
                    const float* stdv, float* out, hipStream_t st) {
    if (n <= 0 || S <= 0 || crop <= 0 || crop > S) return MFVIT_EINVAL;
    MFVIT_LAUNCH(input_transform_kernel, dim3((crop * crop + 255) / 256, n), dim3(256), 0, st, src, desc, tables, S, crop, mean[0], mean[1],
                 mean[2], stdv[0], stdv[1], stdv[2], out);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

}  // namespace mfvit
