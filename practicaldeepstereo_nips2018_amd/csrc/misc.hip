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

// ---------------------------------------------------------------------------------------------------
// Layer 1 factorisation.  The first residual block starts with a convolution and layer 0 has no activation
// (reference matching.py:80-88, network_blocks.py:139-141), so with x0[d] = A + Gs_d (A = conv_L(left) + b0,
// Gs_d[x] = G[x-d] for x-d >= -1, G2[x-d] at x = w-1 when d >= 1, else 0):
//   conv1(x0[d])[x] = B[x] + T_d[x],   B = conv1(A) + b1 (zero padded),
//   T_d[x] = sum_dx [x+dx <= w-1] W1[dx] * Gs_d[x+dx]
//          = H [u]   (u = x-d)  in general: plain conv of the G row on u in [-2, w-1] (zero beyond both ends)
//          = Ha[u]   at x = w-2, d >= 1:  taps dx=-1,0 on G, tap dx=+1 on G2   (x+1 = w-1 is the fixed-up column)
//          = Hb[u]   at x = w-1, d >= 1:  tap dx=-1 on G, tap dx=0 on G2, tap dx=+1 is padding
//          = 0       for u < -2,
//   and for d = 0 (no shift: x+dx = -1 is image padding although G[-1] exists):  T_0[x] = H0[x], the conv of G with
//   its u = -1 entry zeroed.
// Five single-plane 128 -> 64 convolutions (inputs [A;0], [G;0], [G;G2], [G;G2], [G0;0] with tap-masked weight
// sets) run as ONE 5-plane launch of conv2d_mfma; l1_combine then forms LeakyReLU(B + T_d) for every disparity plane and the
// InstanceNorm partial sums -- a 122-GFLOP convolution becomes a 425 MB streaming write.
// Column layout of the 4-plane tensors: width w + 2, column = x + 2 for A / B and u + 2 for G / H rows.
// ---------------------------------------------------------------------------------------------------
// y3 [bc][3][h][w+1] (A at column x+1, G/G2 at column u+1) -> x4 [b][2C][5][h][w+2]
__global__ __launch_bounds__(256) void l1_stack_inputs_kernel(const float* __restrict__ y3, float* __restrict__ x4,
                                                              int batch, int C, int h, int w) {
    const int W2 = w + 2;
    const size_t total = (size_t)batch * 2 * C * kL1Planes * h * W2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int col = (int)(i % W2);
        size_t r = i / W2;
        const int y = (int)(r % h);
        r /= h;
        const int p = (int)(r % kL1Planes);
        r /= kL1Planes;
        const int c2 = (int)(r % (2 * C));
        const int b = (int)(r / (2 * C));
        float v = 0.f;
        const int c = c2 < C ? c2 : c2 - C;
        const float* src = y3 + (((size_t)b * C + c) * 3) * h * (w + 1) + (size_t)y * (w + 1);
        if (c2 < C) {
            if (p == 0) {
                if (col >= 2) v = src[col - 1];                                   // A[x], x = col - 2, stored at x + 1
            } else if (col >= (p == 4 ? 2 : 1)) {                                 // plane 4: G0 = G without u = -1
                v = src[(size_t)h * (w + 1) + col - 1];                           // G[u], u = col - 2, stored at u + 1
            }
        } else if ((p == 2 || p == 3) && col >= 1) {
            v = src[(size_t)2 * h * (w + 1) + col - 1];                           // G2[u]
        }
        x4[i] = v;
    }
}

// weight sets [5][Cout][2C][3][3] and bias sets [5][Cout] from W1 [Cout][C][3][3], b1
__global__ __launch_bounds__(256) void l1_weights_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                                         float* __restrict__ w4, float* __restrict__ bias4, int cout,
                                                         int C) {
    const int total = kL1Planes * cout * 2 * C * 9;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < kL1Planes * cout; i += gridDim.x * 256)
        bias4[i] = i < cout ? b1[i] : 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int t = i % 9;
        const int c2 = (i / 9) % (2 * C);
        const int o = (i / (9 * 2 * C)) % cout;
        const int set = i / (9 * 2 * C * cout);
        const int dx = t % 3;  // 0,1,2 <-> -1,0,+1
        const bool second = c2 >= C;
        const float wv = w1[((size_t)o * C + (second ? c2 - C : c2)) * 9 + t];
        bool keep;
        if (set <= 1 || set == 4) keep = !second;                   // [A;0], [G;0] and [G0;0]
        else if (set == 2) keep = second ? dx == 2 : dx != 2;       // Ha: -1,0 on G ; +1 on G2
        else keep = second ? dx == 1 : dx == 0;                     // Hb: -1 on G ; 0 on G2 ; +1 dropped
        w4[i] = keep ? wv : 0.f;
    }
}

// t1[n,o,d,y,x] = LeakyReLU(B + T_d) and partial sums (records [(n*C+o)*D + d][tile] x {sum, sumsq}).
// One thread owns four consecutive x of one row for ALL disparity planes: B is loaded once; the four H values of a
// thread, H[x - d] for its x, form a window that slides by one column per plane, the entering value being the
// left neighbour lane's right-most one (whole-wave DPP shift; only lane 0 of a wave and the first quad of a row load
// it, and the last quad loads its two special columns from Ha / Hb).  The only streaming traffic is the 16-byte
// stores of t1; per-plane statistics are reduced with DPP row sums (no LDS permutes).
constexpr int kL1MaxPlanes = 64;  // disparity planes handled per launch (statistics scratch in LDS)
__device__ __forceinline__ float l1_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;  // lane 15 of every 16-lane row holds the row's total
}

__global__ __launch_bounds__(256) void l1_combine_kernel(const float* __restrict__ y4, float* __restrict__ t1,
                                                         double* __restrict__ partials, int C, int h, int w,
                                                         int d_begin, int d_first, int d_launch, int d_count) {
    // grid: x = tile over (y, x/4), y = n*C + o ; y4 [n][C][5][h][w+2]
    const int nc = blockIdx.y, tile = blockIdx.x, tiles = gridDim.x;
    const int W2 = w + 2;
    const size_t px = (size_t)h * w;
    const float* Bp = y4 + (size_t)nc * kL1Planes * h * W2;
    const float* Hp = Bp + (size_t)h * W2;
    const float* Ha = Hp + (size_t)h * W2;
    const float* Hb = Ha + (size_t)h * W2;
    const float* H0 = Hb + (size_t)h * W2;
    __shared__ float red[kL1MaxPlanes][16][2];   // [plane][wave * 4 + DPP row]
    const int xq = (w + 3) / 4;
    const bool vec = (w & 3) == 0;
    const int qi = tile * 256 + threadIdx.x;
    const bool active = qi < h * xq;
    const int y = active ? qi / xq : 0, xi = active ? qi - y * xq : 0, xb = xi * 4;
    const size_t row = (size_t)y * W2;
    float bq[4], hw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bq[k] = (active && xb + k < w) ? Bp[row + xb + k + 2] : 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool loads_left = xi == 0 || lane == 0;
    const int d0 = d_begin + d_first;  // disparity of the first plane of this launch
    // window for the first plane: hw[k] = Hp[x_k - d0] (column u + 2 holds H[u], u >= -2)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int u = xb + k - d0;
        hw[k] = (active && xb + k < w && u >= -2) ? Hp[row + u + 2] : 0.f;
    }
    for (int di = 0; di < d_launch; ++di) {
        const int dl = d_first + di;
        const int d = d_begin + dl;
        float r[4], s = 0.f, q = 0.f;
        float tk[4] = {hw[0], hw[1], hw[2], hw[3]};
        if (d == 0) {  // the image border is padding at zero disparity: its own H plane
#pragma unroll
            for (int k = 0; k < 4; ++k) tk[k] = (active && xb + k < w) ? H0[row + xb + k + 2] : 0.f;
        } else {
            // the two right-most columns of the image take their terms from Ha / Hb
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = xb + k, u = x - d;
                if (active && x >= w - 2 && x < w) tk[k] = u >= -2 ? (x == w - 2 ? Ha : Hb)[row + u + 2] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = 0.f;
            if (active && xb + k < w) {
                v = bq[k] + tk[k];
                v = v > 0.f ? v : v * kLeakySlope;
                s += v;
                q = fmaf(v, v, q);
            }
            r[k] = v;
        }
        if (active) {
            float* o = t1 + ((size_t)nc * d_count + dl) * px + (size_t)y * w + xb;
            if (vec) {
                *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (xb + k < w) o[k] = r[k];
            }
        }
        s = l1_row16_sum(s);
        q = l1_row16_sum(q);
        if ((lane & 15) == 15) {
            red[di][wave * 4 + (lane >> 4)][0] = s;
            red[di][wave * 4 + (lane >> 4)][1] = q;
        }
        // slide the window to disparity d + 1
        const float from_left = __builtin_bit_cast(
            float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, hw[3]), 0x138, 0xf, 0xf, true));
        hw[3] = hw[2];
        hw[2] = hw[1];
        hw[1] = hw[0];
        hw[0] = from_left;
        if (loads_left) {
            const int u = xb - (d + 1);
            hw[0] = (active && u >= -2) ? Hp[row + u + 2] : 0.f;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < d_launch * 2; i += 256) {
        const int di = i >> 1, k = i & 1;
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) v += (double)red[di][j][k];
        partials[((((size_t)nc * d_count + d_first + di) * tiles) + tile) * 2 + k] = v;
    }
}

int launch_l1_stack_inputs(const float* y3, float* x4, int batch, int channels, int h, int w, hipStream_t s) {
    const size_t total = (size_t)batch * 2 * channels * kL1Planes * h * (w + 2);
    unsigned bx = (unsigned)((total + 255) / 256);
    if (bx > 8192) bx = 8192;
    hipLaunchKernelGGL(l1_stack_inputs_kernel, dim3(bx), dim3(256), 0, s, y3, x4, batch, channels, h, w);
    return check_launch("l1_stack_inputs");
}

int launch_l1_weights(const float* w1, const float* b1, float* w4, float* bias4, int cout, int channels,
                      hipStream_t s) {
    const int total = kL1Planes * cout * 2 * channels * 9;
    hipLaunchKernelGGL(l1_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w1, b1, w4, bias4, cout,
                       channels);
    return check_launch("l1_weights");
}

int l1_combine_tiles(int h, int w) {
    const size_t quads = (size_t)h * ((w + 3) / 4);
    return (int)((quads + 255) / 256);  // one quad per thread
}

int launch_l1_combine(const float* y4, float* t1, double* partials, int batch, int channels, int h, int w,
                      int d_begin, int d_count, hipStream_t s) {
    for (int first = 0; first < d_count; first += kL1MaxPlanes) {
        const int n = d_count - first < kL1MaxPlanes ? d_count - first : kL1MaxPlanes;
        hipLaunchKernelGGL(l1_combine_kernel, dim3(l1_combine_tiles(h, w), batch * channels), dim3(256), 0, s, y4, t1,
                           partials, channels, h, w, d_begin, first, n, d_count);
    }
    return check_launch("l1_combine");
}

}  // namespace pds
