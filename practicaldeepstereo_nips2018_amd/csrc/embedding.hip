// Kernels of the descriptor network (reference practical_deep_stereo/embedding.py:11-65) that are not
// shared with MatchingOperation:
//
//   * statistics of the raw image for the parameter-free InstanceNorm2d at embedding.py:32, taken over
//     the zero-padded image SizeAdapter.pad would have produced (size_adapter.py:29-43) without
//     materialising that image;
//   * "space to depth": a k5 s2 p2 convolution over [C, H, W] is exactly a k3 s1 p1 convolution over the
//     four pixel-parity sub-images stacked on the channel axis, [4C, ceil(H/2), ceil(W/2)], with the kernel
//     re-indexed as  ky = 2*ty + a,  kx = 2*tx + b  (taps with ky == 5 or kx == 5 are zero).  That lets the
//     two convolutional_block_5x5_stride_2 layers (network_blocks.py:86-92) run on the stride-1 MFMA
//     kernel of conv2d_mfma.hip.  The re-layout applies the producer's deferred InstanceNorm, the
//     SizeAdapter offset and the literal zero padding in the same pass.
#include "common.hpp"

namespace pds {

namespace {

constexpr int kStatChunk = 8192;  // pixels per statistics record

__global__ __launch_bounds__(256) void image_stats_kernel(const float* __restrict__ img, size_t plane, int chunks,
                                                          double* __restrict__ partials) {
    const int nc = blockIdx.y, chunk = blockIdx.x;
    const float* p = img + (size_t)nc * plane;
    const size_t begin = (size_t)chunk * kStatChunk;
    const size_t end = begin + kStatChunk < plane ? begin + kStatChunk : plane;
    double s = 0.0, q = 0.0;
    for (size_t i = begin + threadIdx.x; i < end; i += 256) {
        const double v = p[i];
        s += v;
        q += v * v;
    }
    __shared__ double red[4][2];
    s = wave_sum(s);
    q = wave_sum(q);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave][0] = s;
        red[wave][1] = q;
    }
    __syncthreads();
    if (threadIdx.x < 2)
        partials[((size_t)nc * chunks + chunk) * 2 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// in: [N, C, H, W] (deferred-normalised source), virtually padded by `top` rows / `left` columns of zeros
// (the zeros are part of the image: they are normalised like any other pixel).
// out: [N, 4C, h2, w2], channel (a*2 + b)*C + c holds pixel (2i + a, 2j + b) of the padded image; positions past
// the padded image are literal zeros (the convolution's own padding).
__global__ __launch_bounds__(256) void space_to_depth_kernel(const Src a, int C, int H, int W, int top, int left,
                                                             int h2, int w2, float* __restrict__ out,
                                                             float* __restrict__ bound_out) {
    const int nc = blockIdx.y;
    const int n = nc / C, c = nc % C;
    // the range certificate travels with the values (Src::bound): a re-layout changes none of them
    if (bound_out && blockIdx.x == 0 && nc == 0 && threadIdx.x == 0) {
        float m = 0.f;
        for (int i = 0; i < a.bound_n; ++i) m = fmaxf(m, a.bound[i] == a.bound[i] ? fabsf(a.bound[i]) : __builtin_inff());
        *bound_out = m;
    }
    float sc = 1.f, sh = 0.f;
    if (a.scale) {
        sc = a.scale[nc];
        sh = a.shift[nc];
    }
    const float* p = a.p + (size_t)nc * H * W;
    const int Hp = H + top, Wp = W + left;
    const size_t plane2 = (size_t)h2 * w2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane2; i += (size_t)gridDim.x * 256) {
        const int oy = (int)(i / w2), ox = (int)(i % w2);
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int y = 2 * oy + pa, x = 2 * ox + pb;
                float v = 0.f;
                if (y < Hp && x < Wp) {
                    const int yy = y - top, xx = x - left;
                    const float raw = (yy >= 0 && xx >= 0) ? p[(size_t)yy * W + xx] : 0.f;
                    v = fmaf(sc, raw, sh);
                }
                out[((size_t)n * 4 * C + (pa * 2 + pb) * C + c) * plane2 + i] = v;
            }
        }
    }
}

// adjoint of space_to_depth without padding offsets: g [N, 4C, h2, w2] -> out [N, C, H, W]
__global__ __launch_bounds__(256) void depth_to_space_kernel(const float* __restrict__ g, int C, int H, int W, int h2,
                                                             int w2, float* __restrict__ out) {
    const int nc = blockIdx.y;
    const int n = nc / C, c = nc % C;
    const size_t plane = (size_t)H * W, plane2 = (size_t)h2 * w2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i % W);
        const int ch = ((y & 1) * 2 + (x & 1)) * C + c;
        out[(size_t)nc * plane + i] = g[((size_t)n * 4 * C + ch) * plane2 + (size_t)(y >> 1) * w2 + (x >> 1)];
    }
}

// Adjoint of the image head (embedding.py:32 under autograd): g [N, 4C, h2, w2], the gradient of the space-to-depth
// tensor, goes back through depth-to-space and the parameter-free InstanceNorm2d of the PADDED image to
// d image [N, C, H, W].  With xh = (x - mean) * rstd over the Hp x Wp padded plane (pad pixels are ordinary members
// of the statistics, x = 0) :  dx_i = rstd * (g_i - mean_j(g_j) - xh_i * mean_j(g_j xh_j)), j over the padded plane.
// One workgroup per (n, c): both plane sums in fp64, then the apply pass over the real pixels.
__global__ __launch_bounds__(1024) void image_grad_kernel(const float* __restrict__ g, const float* __restrict__ img,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int C, int H, int W, int top,
                                                          int left, int h2, int w2, float* __restrict__ out) {
    const int nc = blockIdx.x;
    const int n = nc / C, c = nc % C;
    const float sc = scale[nc], sh = shift[nc];   // no affine: scale = rstd, shift = -mean * rstd
    const int Hp = H + top, Wp = W + left;
    const size_t plane2 = (size_t)h2 * w2, padded = (size_t)Hp * Wp;
    const float* gp = g + (size_t)n * 4 * C * plane2;
    const float* p = img + (size_t)nc * H * W;
    auto grad_at = [&](int y, int x) {
        return gp[(size_t)(((y & 1) * 2 + (x & 1)) * C + c) * plane2 + (size_t)(y >> 1) * w2 + (x >> 1)];
    };
    double s1 = 0.0, s2 = 0.0;
    for (size_t i = threadIdx.x; i < padded; i += 1024) {
        const int y = (int)(i / Wp), x = (int)(i % Wp);
        const int yy = y - top, xx = x - left;
        const float raw = (yy >= 0 && xx >= 0) ? p[(size_t)yy * W + xx] : 0.f;
        const float gv = grad_at(y, x);
        s1 += (double)gv;
        s2 += (double)gv * (double)fmaf(sc, raw, sh);
    }
    __shared__ double red[16][2];
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave][0] = s1;
        red[wave][1] = s2;
    }
    __syncthreads();
    double t1 = 0.0, t2 = 0.0;
    for (int i = 0; i < 16; ++i) {
        t1 += red[i][0];
        t2 += red[i][1];
    }
    const float m1 = (float)(t1 / (double)padded), m2 = (float)(t2 / (double)padded);
    for (size_t i = threadIdx.x; i < (size_t)H * W; i += 1024) {
        const int yy = (int)(i / W), xx = (int)(i % W);
        const float xh = fmaf(sc, p[i], sh);
        out[(size_t)nc * H * W + i] = sc * (grad_at(yy + top, xx + left) - m1 - xh * m2);
    }
}

// w5 [K, C, 5, 5] -> w3 [K, 4C, 3, 3]
__global__ void s2d_weights_kernel(const float* __restrict__ w5, float* __restrict__ w3, int K, int C) {
    const int total = K * 4 * C * 9;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int tx = i % 3, ty = (i / 3) % 3, ch = (i / 9) % (4 * C), k = i / (36 * C);
        const int par = ch / C, c = ch % C;
        const int ky = 2 * ty + (par >> 1), kx = 2 * tx + (par & 1);
        w3[i] = (ky < 5 && kx < 5) ? w5[((size_t)(k * C + c) * 5 + ky) * 5 + kx] : 0.f;
    }
}

// gradient of the above: g3 [K, 4C, 3, 3] -> g5 [K, C, 5, 5] (a gather: every 5x5 tap has exactly one image)
__global__ void s2d_weights_bwd_kernel(const float* __restrict__ g3, float* __restrict__ g5, int K, int C,
                                       int accumulate) {
    const int total = K * C * 25;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int kx = i % 5, ky = (i / 5) % 5, c = (i / 25) % C, k = i / (25 * C);
        const int ch = ((ky & 1) * 2 + (kx & 1)) * C + c;
        const float v = g3[((size_t)(k * 4 * C + ch) * 3 + (ky >> 1)) * 3 + (kx >> 1)];
        g5[i] = accumulate ? g5[i] + v : v;
    }
}

}  // namespace

int image_stats_chunks(int h, int w) { return (int)(((size_t)h * w + kStatChunk - 1) / kStatChunk); }

int launch_image_stats(const float* img, int nc, int h, int w, double* partials, hipStream_t s) {
    const int chunks = image_stats_chunks(h, w);
    hipLaunchKernelGGL(image_stats_kernel, dim3(chunks, nc), dim3(256), 0, s, img, (size_t)h * w, chunks, partials);
    return check_launch("image_stats");
}

int launch_space_to_depth(const Src& a, int n, int c, int h, int w, int top, int left, float* out, hipStream_t s,
                          float* bound_out) {
    if (bound_out && !a.bound) return set_error(-1, "space_to_depth: the source carries no range bound to pass on");
    const int h2 = (h + top + 1) / 2, w2 = (w + left + 1) / 2;
    size_t bx = ((size_t)h2 * w2 + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(space_to_depth_kernel, dim3((unsigned)bx, n * c), dim3(256), 0, s, a, c, h, w, top, left, h2, w2,
                       out, bound_out);
    return check_launch("space_to_depth");
}

int launch_depth_to_space(const float* g, int n, int c, int h, int w, float* out, hipStream_t s) {
    const int h2 = (h + 1) / 2, w2 = (w + 1) / 2;
    size_t bx = ((size_t)h * w + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(depth_to_space_kernel, dim3((unsigned)bx, n * c), dim3(256), 0, s, g, c, h, w, h2, w2, out);
    return check_launch("depth_to_space");
}

int launch_image_grad(const float* g, const float* img, const float* scale, const float* shift, int n, int c, int h,
                      int w, int top, int left, float* out, hipStream_t s) {
    const int h2 = (h + top + 1) / 2, w2 = (w + left + 1) / 2;
    hipLaunchKernelGGL(image_grad_kernel, dim3(n * c), dim3(1024), 0, s, g, img, scale, shift, c, h, w, top, left, h2, w2,
                       out);
    return check_launch("image_grad");
}

int launch_s2d_weights(const float* w5, float* w3, int cout, int cin, hipStream_t s) {
    const int total = cout * 4 * cin * 9;
    hipLaunchKernelGGL(s2d_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w5, w3, cout, cin);
    return check_launch("s2d_weights");
}

int launch_s2d_weights_bwd(const float* g3, float* g5, int cout, int cin, int accumulate, hipStream_t s) {
    const int total = cout * cin * 25;
    hipLaunchKernelGGL(s2d_weights_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, g3, g5, cout, cin,
                       accumulate);
    return check_launch("s2d_weights_bwd");
}

}  // namespace pds
