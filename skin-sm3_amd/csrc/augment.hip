// SimCLR augmentation of the SM3 pre-training input on the GPU (gfx950): the per-sample transform chain the reference
// builds from torchvision / PIL on DataLoader workers (tools/backbone_train.py:448-466)
//
//   RandomResizedCrop(size, scale=(0.5, 1)) -> RandomApply(ColorJitter(0.8, 0.8, 0.8, 0.2), p=0.8)
//   -> RandomGrayscale(0.2) -> RandomHorizontalFlip -> RandomApply(GaussianBlur(3x3, sigma in [0.1, 2]), p=0.5)
//   -> ToTensor -> Normalize(mean, std)
//
// applied to a batch of decoded RGB images that is already resident in HBM ([B, Hs, Ws, 3] uint8, as cv2 / PIL
// deliver them: src/utils/data/datasets.py:508-533), producing the NCHW fp32 batch the encoder's stem reads.  The
// random parameters (crop box, flip, jitter order and factors, grayscale, blur sigma) are drawn by the host mirror
// (sm3hip/augment.py) with the sampling rules of torchvision 0.13 (requirements.txt:3) and passed in per sample; the
// arithmetic here is that of torchvision's float-tensor functional ops (functional_tensor.py: _blend, rgb_to_grayscale,
// _rgb2hsv/_hsv2rgb, gaussian_blur with reflect padding) and PIL's antialiased bilinear resample (ImagingResample's
// triangle filter with support max(scale, 1)).  HBM-bound element-wise passes; the 4-view batch of a step costs ~1 ms.
#include "common.h"

namespace {

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float gray3(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }

// crop box (top i, left j, height h, width w) of the source -> [H, W] by antialiased bilinear, optional horizontal flip;
// out [B, 3, H, W] in [0, 1]
__global__ __launch_bounds__(256) void aug_resized_crop_kernel(const uint8_t* __restrict__ src, int Hs, int Ws,
                                                               const int* __restrict__ box, const uint8_t* __restrict__ flip,
                                                               float* __restrict__ out, int H, int W) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int oy = idx / W, ox0 = idx - oy * W;
    const int ox = flip[b] ? W - 1 - ox0 : ox0;  // flip(resize(crop)) : output column ox0 shows resized column W-1-ox0
    const int ci = box[4 * b + 0], cj = box[4 * b + 1], ch = box[4 * b + 2], cw = box[4 * b + 3];
    const float sy = (float)ch / (float)H, sx = (float)cw / (float)W;
    const float fsy = fmaxf(sy, 1.f), fsx = fmaxf(sx, 1.f);  // filter scale; bilinear support = 1 * filterscale
    const float cy = (oy + 0.5f) * sy, cx = (ox + 0.5f) * sx;
    const int ymin = max(0, (int)(cy - fsy + 0.5f)), ymax = min(ch, (int)(cy + fsy + 0.5f));
    const int xmin = max(0, (int)(cx - fsx + 0.5f)), xmax = min(cw, (int)(cx + fsx + 0.5f));
    float wysum = 0.f, wxsum = 0.f;
    for (int y = ymin; y < ymax; ++y) wysum += fmaxf(0.f, 1.f - fabsf((y - cy + 0.5f) / fsy));
    for (int x = xmin; x < xmax; ++x) wxsum += fmaxf(0.f, 1.f - fabsf((x - cx + 0.5f) / fsx));
    float acc[3] = {0.f, 0.f, 0.f};
    const uint8_t* sb = src + (size_t)b * Hs * Ws * 3;
    for (int y = ymin; y < ymax; ++y) {
        const float wy = fmaxf(0.f, 1.f - fabsf((y - cy + 0.5f) / fsy)) / wysum;
        const uint8_t* row = sb + ((size_t)(ci + y) * Ws + cj) * 3;
        float r[3] = {0.f, 0.f, 0.f};
        for (int x = xmin; x < xmax; ++x) {
            const float wx = fmaxf(0.f, 1.f - fabsf((x - cx + 0.5f) / fsx)) / wxsum;
            r[0] += wx * row[3 * x + 0];
            r[1] += wx * row[3 * x + 1];
            r[2] += wx * row[3 * x + 2];
        }
        acc[0] += wy * r[0];
        acc[1] += wy * r[1];
        acc[2] += wy * r[2];
    }
    float* ob = out + (size_t)b * 3 * H * W + idx;
#pragma unroll
    for (int c = 0; c < 3; ++c) ob[(size_t)c * H * W] = acc[c] * (1.f / 255.f);  // ToTensor's scaling
}

// mean of the grayscale image (ColorJitter's contrast blends with it): one block per image
__global__ __launch_bounds__(1024) void aug_gray_mean_kernel(const float* __restrict__ img, int HW, float* __restrict__ mean) {
    __shared__ float red[16];
    const float* p = img + (size_t)blockIdx.x * 3 * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += 1024) s += gray3(p[i], p[HW + i], p[2 * HW + i]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i];
        mean[blockIdx.x] = t / (float)HW;
    }
}

// one position of ColorJitter's randomly ordered chain: op[b] in {0 none, 1 brightness, 2 contrast, 3 saturation, 4 hue}
__global__ __launch_bounds__(256) void aug_color_op_kernel(float* __restrict__ img, int HW, const int* __restrict__ op,
                                                           const float* __restrict__ factor, const float* __restrict__ gmean) {
    const int b = blockIdx.y;
    const int o = op[b];
    if (o == 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    float* p = img + (size_t)b * 3 * HW + i;
    float r = p[0], g = p[HW], bl = p[2 * HW];
    const float f = factor[b];
    if (o == 1) {  // _blend(img, 0, f)
        r = clamp01(r * f); g = clamp01(g * f); bl = clamp01(bl * f);
    } else if (o == 2) {  // _blend(img, mean(gray(img)), f)
        const float m = (1.f - f) * gmean[b];
        r = clamp01(f * r + m); g = clamp01(f * g + m); bl = clamp01(f * bl + m);
    } else if (o == 3) {  // _blend(img, gray(img), f)
        const float m = (1.f - f) * gray3(r, g, bl);
        r = clamp01(f * r + m); g = clamp01(f * g + m); bl = clamp01(f * bl + m);
    } else {  // hue: rgb -> hsv, h = (h + f) mod 1, hsv -> rgb   (functional_tensor._rgb2hsv / _hsv2rgb)
        const float maxc = fmaxf(r, fmaxf(g, bl)), minc = fminf(r, fminf(g, bl));
        const bool eqc = maxc == minc;
        const float cr = maxc - minc;
        const float s = cr / (eqc ? 1.f : maxc);
        const float crd = eqc ? 1.f : cr;
        const float rc = (maxc - r) / crd, gc = (maxc - g) / crd, bc = (maxc - bl) / crd;
        const float hr = (maxc == r) ? (bc - gc) : 0.f;
        const float hg = ((maxc == g) && (maxc != r)) ? (2.f + rc - bc) : 0.f;
        const float hb = ((maxc != g) && (maxc != r)) ? (4.f + gc - rc) : 0.f;
        float h = (hr + hg + hb) / 6.f + 1.f;
        h = h - floorf(h);           // fmod(h, 1)
        h = h + f;
        h = h - floorf(h);           // (h + f) % 1
        const float v = maxc;
        const float h6 = h * 6.f;
        const float fi = floorf(h6);
        const float fr = h6 - fi;
        const int ii = ((int)fi) % 6;
        const float pp = clamp01(v * (1.f - s)), qq = clamp01(v * (1.f - s * fr)), tt = clamp01(v * (1.f - s * (1.f - fr)));
        switch (ii) {
            case 0: r = v; g = tt; bl = pp; break;
            case 1: r = qq; g = v; bl = pp; break;
            case 2: r = pp; g = v; bl = tt; break;
            case 3: r = pp; g = qq; bl = v; break;
            case 4: r = tt; g = pp; bl = v; break;
            default: r = v; g = pp; bl = qq; break;
        }
    }
    p[0] = r; p[HW] = g; p[2 * HW] = bl;
}

// RandomGrayscale -> GaussianBlur(3x3, reflect) -> Normalize, one pass: out = (blur(gray?(img)) - mean) / std
__global__ __launch_bounds__(256) void aug_finish_kernel(const float* __restrict__ img, int H, int W,
                                                         const uint8_t* __restrict__ gray, const float* __restrict__ sigma,
                                                         float m0, float m1, float m2, float s0, float s1, float s2,
                                                         float* __restrict__ out) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    if (idx >= HW) return;
    const int y = idx / W, x = idx - y * W;
    const float* p = img + (size_t)b * 3 * HW;
    const bool gr = gray[b] != 0;
    const float sg = sigma[b];  // <= 0: no blur
    float v[3];
    if (sg > 0.f) {
        const float e = __expf(-0.5f / (sg * sg));  // kernel1d = [e, 1, e] / (1 + 2e)
        const float wc = 1.f / (1.f + 2.f * e), ws = e * wc;
        const int ym = (y == 0) ? (H > 1 ? 1 : 0) : y - 1, yp = (y == H - 1) ? (H > 1 ? H - 2 : 0) : y + 1;  // reflect
        const int xm = (x == 0) ? (W > 1 ? 1 : 0) : x - 1, xp = (x == W - 1) ? (W > 1 ? W - 2 : 0) : x + 1;
        const int ys[3] = {ym, y, yp}, xs[3] = {xm, x, xp};
        const float wt[3] = {ws, wc, ws};
        v[0] = v[1] = v[2] = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c2 = 0; c2 < 3; ++c2) {
                const int o = ys[a] * W + xs[c2];
                float r = p[o], g = p[HW + o], bl = p[2 * HW + o];
                if (gr) r = g = bl = gray3(r, g, bl);
                const float w2 = wt[a] * wt[c2];
                v[0] += w2 * r; v[1] += w2 * g; v[2] += w2 * bl;
            }
    } else {
        float r = p[idx], g = p[HW + idx], bl = p[2 * HW + idx];
        if (gr) r = g = bl = gray3(r, g, bl);
        v[0] = r; v[1] = g; v[2] = bl;
    }
    float* ob = out + (size_t)b * 3 * HW + idx;
    ob[0] = (v[0] - m0) / s0;
    ob[HW] = (v[1] - m1) / s1;
    ob[2 * HW] = (v[2] - m2) / s2;
}

}  // namespace

extern "C" int sm3_aug_resized_crop(const uint8_t* src, int B, int Hs, int Ws, const int32_t* box, const uint8_t* flip,
                                    float* out, int H, int W, void* stream) {
    if (!src || !box || !flip || !out || B <= 0 || Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0 || B > 65535) return SM3_EINVAL;
    hipLaunchKernelGGL(aug_resized_crop_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, src, Hs, Ws,
                       box, flip, out, H, W);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_aug_color_op(float* img, int B, int H, int W, const int32_t* op, const float* factor, float* gray_mean,
                                void* stream) {
    if (!img || !op || !factor || !gray_mean || B <= 0 || H <= 0 || W <= 0 || B > 65535) return SM3_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(aug_gray_mean_kernel, dim3(B), dim3(1024), 0, st, img, H * W, gray_mean);
    SM3_CHECK_LAUNCH();
    hipLaunchKernelGGL(aug_color_op_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, st, img, H * W, op, factor, gray_mean);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_aug_finish(const float* img, int B, int H, int W, const uint8_t* gray, const float* sigma,
                              const float* mean3, const float* std3, float* out, void* stream) {
    if (!img || !gray || !sigma || !mean3 || !std3 || !out || B <= 0 || H <= 0 || W <= 0 || B > 65535) return SM3_EINVAL;
    if (std3[0] == 0.f || std3[1] == 0.f || std3[2] == 0.f) return SM3_EINVAL;
    hipLaunchKernelGGL(aug_finish_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, img, H, W, gray,
                       sigma, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out);
    SM3_CHECK_LAUNCH();
    return 0;
}
