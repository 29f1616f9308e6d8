// Small bandwidth-bound helpers of the Matching stage.
#include "common.hpp"

namespace pds {

// cat([left, S_d(right)], dim=1) for a range of disparities (reference matching.py:50-61).
// out [d_count, batch, 2C, h, w]
__global__ __launch_bounds__(256) void shift_concat_kernel(const float* __restrict__ left,
                                                           const float* __restrict__ right,
                                                           float* __restrict__ out, int batch, int C, int h,
                                                           int w, int d_begin, int d_count) {
    const size_t row_count = (size_t)d_count * batch * 2 * C * h;
    for (size_t row = blockIdx.x; row < row_count; row += gridDim.x) {
        size_t r = row;
        const int y = (int)(r % h);
        r /= h;
        const int c2 = (int)(r % (2 * C));
        r /= (2 * C);
        const int b = (int)(r % batch);
        const int d = d_begin + (int)(r / batch);
        float* dst = out + row * w;
        if (c2 < C) {
            const float* src = left + (((size_t)b * C + c2) * h + y) * w;
            for (int x = threadIdx.x; x < w; x += 256) dst[x] = src[x];
        } else {
            const float* src = right + (((size_t)b * C + (c2 - C)) * h + y) * w;
            for (int x = threadIdx.x; x < w; x += 256) dst[x] = (x >= d) ? src[x - d] : 0.f;
        }
    }
}

int launch_shift_concat(const float* left, const float* right, float* out, int batch, int channels, int h,
                        int w, int d_begin, int d_count, hipStream_t s) {
    const size_t rows = (size_t)d_count * batch * 2 * channels * h;
    const unsigned grid = (unsigned)(rows < 65536 ? rows : 65536);
    hipLaunchKernelGGL(shift_concat_kernel, dim3(grid), dim3(256), 0, s, left, right, out, batch, channels, h, w,
                       d_begin, d_count);
    return check_launch("shift_concat");
}

// Layer 0 of MatchingOperation is linear and un-normalised (reference matching.py:80-83), so
//   conv0(cat[L, S_d R])[x] = A[x] + G[x-d]              A = conv_L(L) + bias, G = conv_R(R~)
// with G defined on u = x-d in [-1, w-1] (stored at column u+1) and 0 for u < -1; at x = w-1, d >= 1
// the tap that would read R[w-d] sees the crop of the shifted image, so G2 (conv_R without its
// dx = +1 taps) replaces G (verified against the reference to 4.8e-7).
// x0 layout [batch, C, d_count, h, w].
// A, G, G2 all have row stride w + 1 and channel stride `cstride`; A points at column 1 of its rows
// (it is the convolution of the left descriptor padded by one zero column, like G).
__global__ __launch_bounds__(256) void l0_combine_kernel(const float* __restrict__ A,
                                                         const float* __restrict__ G,
                                                         const float* __restrict__ G2, size_t cstride,
                                                         float* __restrict__ x0, int C, int h, int w,
                                                         int d_begin, int d_count) {
    // grid: x = row tile, y = local disparity, z = b*C + c
    const int bc = blockIdx.z, dl = blockIdx.y;
    const int d = d_begin + dl;
    const size_t px = (size_t)h * w;
    const float* a = A + (size_t)bc * cstride;
    const float* g = G + (size_t)bc * cstride;
    const float* g2 = G2 + (size_t)bc * cstride;
    float* dst = x0 + ((size_t)bc * d_count + dl) * px;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < px; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i % w);
        const int u = x - d;
        float v = a[(size_t)y * (w + 1) + x];
        if (u >= -1) {
            const size_t off = (size_t)y * (w + 1) + (u + 1);
            v += (x == w - 1 && d >= 1) ? g2[off] : g[off];
        }
        dst[i] = v;
    }
}

int launch_l0_combine(const float* A, const float* G, const float* G2, size_t cstride, float* x0, int batch,
                      int channels, int h, int w, int d_begin, int d_count, hipStream_t s) {
    const size_t px = (size_t)h * w;
    unsigned bx = (unsigned)((px + 255) / 256);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(l0_combine_kernel, dim3(bx, d_count, batch * channels), dim3(256), 0, s, A, G, G2, cstride, x0,
                       channels, h, w, d_begin, d_count);
    return check_launch("l0_combine");
}

// Inputs of the three layer-0 convolutions as ONE volume [B*C][3 planes][h][w+1]: plane 0 = left descriptor,
// planes 1 and 2 = right descriptor, each padded with one zero column on the left.
__global__ __launch_bounds__(256) void l0_stack_inputs_kernel(const float* __restrict__ left,
                                                              const float* __restrict__ right,
                                                              float* __restrict__ out, size_t bc_count, int h, int w) {
    const size_t total = bc_count * 3 * h * (size_t)(w + 1);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % (w + 1));
        size_t r = i / (w + 1);
        const int y = (int)(r % h);
        r /= h;
        const int p = (int)(r % 3);
        const size_t bc = r / 3;
        const float* src = p == 0 ? left : right;
        out[i] = x == 0 ? 0.f : src[(bc * h + y) * w + x - 1];
    }
}

int launch_l0_stack_inputs(const float* left, const float* right, float* out, size_t bc_count, int h, int w,
                           hipStream_t s) {
    const size_t total = bc_count * 3 * h * (size_t)(w + 1);
    unsigned bx = (unsigned)((total + 255) / 256);
    if (bx > 8192) bx = 8192;
    hipLaunchKernelGGL(l0_stack_inputs_kernel, dim3(bx), dim3(256), 0, s, left, right, out, bc_count, h, w);
    return check_launch("l0_stack_inputs");
}

__global__ __launch_bounds__(256) void pad_left1_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        size_t rows, int w) {
    const size_t total = rows * (size_t)(w + 1);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / (w + 1);
        const int x = (int)(i % (w + 1));
        out[i] = x == 0 ? 0.f : in[r * w + x - 1];
    }
}

int launch_pad_left1(const float* in, float* out, size_t rows, int w, hipStream_t s) {
    const size_t total = rows * (size_t)(w + 1);
    unsigned bx = (unsigned)((total + 255) / 256);
    if (bx > 8192) bx = 8192;
    hipLaunchKernelGGL(pad_left1_kernel, dim3(bx), dim3(256), 0, s, in, out, rows, w);
    return check_launch("pad_left1");
}

// w0 [Cout, 2C, 3, 3] -> wl = w0[:, :C], wr = w0[:, C:], wr2 = wr with kw == 2 zeroed
__global__ __launch_bounds__(256) void split_first_weights_kernel(const float* __restrict__ w0,
                                                                  const float* __restrict__ b0,
                                                                  float* __restrict__ wl,
                                                                  float* __restrict__ wr,
                                                                  float* __restrict__ wr2,
                                                                  float* __restrict__ bias3, int cout, int C) {
    const int total = cout * C * 9;
    // bias3 [3][cout]: the layer bias belongs to the left term only
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * cout; i += gridDim.x * 256)
        bias3[i] = i < cout ? b0[i] : 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int t = i % 9;
        const int c = (i / 9) % C;
        const int o = i / (9 * C);
        const float l = w0[((size_t)o * 2 * C + c) * 9 + t];
        const float r = w0[((size_t)o * 2 * C + C + c) * 9 + t];
        wl[i] = l;
        wr[i] = r;
        wr2[i] = (t % 3 == 2) ? 0.f : r;
    }
}

int launch_split_first_weights(const float* w0, const float* b0, float* wl, float* wr, float* wr2, float* bias3,
                               int cout, int cin_half, hipStream_t s) {
    const int total = cout * cin_half * 9;
    hipLaunchKernelGGL(split_first_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w0, b0, wl, wr, wr2,
                       bias3, cout, cin_half);
    return check_launch("split_first_weights");
}

}  // namespace pds
