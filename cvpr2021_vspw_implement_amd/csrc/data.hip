// Device side of the VSPW input pipeline: everything the reference's datasets do to a decoded
// frame between PIL.Image.open and the tensor the model sees (dataset2.py:852-1048 BaseDataset_longclip, :657-850
// BaseDataset_clip, :154-342 / :344-490 the test datasets):
//   * multi-scale augmentation: PIL Image.resize(BILINEAR) of the RGB frame and Image.resize(NEAREST) of the mask
//     (dataset2.py:1019-1026).  Pillow's 8-bit resampler is a separable fixed-point filter (22 fractional bits,
//     horizontal pass rounded to uint8, then vertical); the host computes Pillow's bounds / integer coefficient tables
//     and nearest-neighbour index tables (they are O(size)), the kernels apply them - bit-exact with Pillow;
//   * horizontal flip, zero / 255 padding to the crop size, the random crop shared by the T frames of a clip,
//     /255, ImageNet mean/std normalisation, NCHW->NHWC, and the label remap 0->255, v->v-1 (dataset2.py:921-977),
//     in ONE gather pass that writes the frame's slot of the [B][crop][crop][3] batch tensor.
// All HBM-bound byte gathers: 3 B read + 12 B written per output pixel (+1 B / 4 B for the label).
#include "common.h"

// One separable pass of Pillow's ImagingResample{Horizontal,Vertical}_8bpc on interleaved u8 pixels.
// axis 0: out[y][x] = sum_k in[y][xmin(x)+k] * kk[x][k]   (out width = n_out, height = h)
// axis 1: out[y][x] = sum_k in[ymin(y)+k][x] * kk[y][k]   (out height = n_out, width = w)
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                          const int* __restrict__ bounds, const int* __restrict__ kk,
                                                          int ksize, int in_h, int in_w, int out_h, int out_w, int ch,
                                                          int axis, int flip) {
    const long long total = (long long)out_h * out_w * ch;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c = (int)(i % ch);
        const long long p = i / ch;
        const int x = (int)(p % out_w), y = (int)(p / out_w);
        const int o = axis == 0 ? x : y;
        const int lo = bounds[2 * o], n = bounds[2 * o + 1];
        const int* k = kk + (size_t)o * ksize;
        int ss = 1 << (22 - 1);
        // flip: the source is read mirrored along x (the reference flips the PIL image before resizing it)
        if (axis == 0) {
            const uint8_t* s = in + (size_t)y * in_w * ch + c;
            for (int t = 0; t < n; ++t) {
                const int sx = flip ? in_w - 1 - (lo + t) : lo + t;
                ss += (int)s[(size_t)sx * ch] * k[t];
            }
        } else {
            const int sx = flip ? in_w - 1 - x : x;
            const uint8_t* s = in + ((size_t)lo * in_w + sx) * ch + c;
            for (int t = 0; t < n; ++t) ss += (int)s[(size_t)t * in_w * ch] * k[t];
        }
        ss >>= 22;  // Pillow clip8(): arithmetic shift, clamp to [0, 255]
        out[i] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
    }
}

// Pillow's nearest-neighbour resize (ImagingScaleAffine): out[y][x] = in[ytab[y]][xtab[x]] (host-computed tables).
__global__ __launch_bounds__(256) void gather_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                        const int* __restrict__ xtab, const int* __restrict__ ytab,
                                                        int in_w, int out_h, int out_w, int flip) {
    const long long total = (long long)out_h * out_w;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int x = (int)(i % out_w), y = (int)(i / out_w);
        const int sx = flip ? in_w - 1 - xtab[x] : xtab[x];
        out[i] = in[(size_t)ytab[y] * in_w + sx];
    }
}

// Flip + pad + crop + normalise one frame into its batch slot.  Output pixel (oy, ox) reads source pixel
// (oy + crop_y - pad_h, ox' + crop_x - pad_w) with ox' mirrored when flip; outside the frame the image is 0 (before
// normalisation, dataset2.py:928-936) and the label 255.  img_out: NHWC fp32 [out_h][out_w][3] slot;
// lab_out: fp32 [out_h][out_w] slot (segm_transform: 0 -> 255, v -> v-1, then float; dataset2.py:970-977).
__global__ __launch_bounds__(256) void frame_transform_kernel(const uint8_t* __restrict__ img,
                                                              const uint8_t* __restrict__ lab, int h, int w, int flip,
                                                              int pad_h, int pad_w, int crop_y, int crop_x, int out_h,
                                                              int out_w, float m0, float m1, float m2, float s0,
                                                              float s1, float s2, float* __restrict__ img_out,
                                                              float* __restrict__ lab_out) {
    const long long total = (long long)out_h * out_w;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int ox = (int)(i % out_w), oy = (int)(i / out_w);
        const int sy = oy + crop_y - pad_h;
        int sx = ox + crop_x - pad_w;
        const bool in = (sy >= 0) & (sy < h) & (sx >= 0) & (sx < w);
        if (flip) sx = w - 1 - sx;
        float r = 0.f, g = 0.f, b = 0.f;
        unsigned l = 255u;
        if (in) {
            const uint8_t* p = img + ((size_t)sy * w + sx) * 3;
            r = (float)p[0] / 255.f;  // np.float32(np.array(img)) / 255.
            g = (float)p[1] / 255.f;
            b = (float)p[2] / 255.f;
            if (lab) l = lab[(size_t)sy * w + sx];
        }
        if (img_out) {
            float* o = img_out + i * 3;
            o[0] = (r - m0) / s0;  // transforms.Normalize: sub_(mean).div_(std)
            o[1] = (g - m1) / s1;
            o[2] = (b - m2) / s2;
        }
        if (lab_out) {
            if (l == 0u) l = 255u;
            l = (l - 1u) & 255u;  // uint8 arithmetic of the reference
            if (l == 254u) l = 255u;
            lab_out[i] = (float)l;
        }
    }
}

extern "C" int vspw_resample_u8(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* kk, int ksize,
                                int in_h, int in_w, int out_h, int out_w, int channels, int axis, int flip,
                                void* stream) {
    if (!in || !out || !bounds || !kk || ksize <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 ||
        channels <= 0 || (axis != 0 && axis != 1))
        return VSPW_EINVAL;
    if ((axis == 0 && out_h != in_h) || (axis == 1 && out_w != in_w)) return VSPW_EINVAL;
    const long long total = (long long)out_h * out_w * channels;
    hipLaunchKernelGGL(resample_u8_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, vspw_stream(stream), in, out,
                       bounds, kk, ksize, in_h, in_w, out_h, out_w, channels, axis, flip);
    return vspw_launch_status();
}

extern "C" int vspw_gather_u8(const uint8_t* in, uint8_t* out, const int32_t* xtab, const int32_t* ytab, int in_w,
                              int out_h, int out_w, int flip, void* stream) {
    if (!in || !out || !xtab || !ytab || in_w <= 0 || out_h <= 0 || out_w <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(gather_u8_kernel, dim3(vspw_stream_grid((long long)out_h * out_w, 256)), dim3(256), 0,
                       vspw_stream(stream), in, out, xtab, ytab, in_w, out_h, out_w, flip);
    return vspw_launch_status();
}

extern "C" int vspw_frame_transform(const uint8_t* img, const uint8_t* lab, int h, int w, int flip,
                                    int pad_h, int pad_w, int crop_y, int crop_x, int out_h, int out_w,
                                    const float* mean3, const float* std3, float* img_out, float* lab_out,
                                    void* stream) {
    if (!img || h <= 0 || w <= 0 || out_h <= 0 || out_w <= 0 || pad_h < 0 || pad_w < 0 || !mean3 || !std3 ||
        (!img_out && !lab_out))
        return VSPW_EINVAL;
    hipLaunchKernelGGL(frame_transform_kernel, dim3(vspw_stream_grid((long long)out_h * out_w, 256)), dim3(256), 0,
                       vspw_stream(stream), img, lab, h, w, flip, pad_h, pad_w, crop_y, crop_x, out_h, out_w, mean3[0],
                       mean3[1], mean3[2], std3[0], std3[1], std3[2], img_out, lab_out);
    return vspw_launch_status();
}
