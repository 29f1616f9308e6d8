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
                                                         int d_begin, int d_count, float* __restrict__ amax) {
    __shared__ float red[16];
    float seen = 0.f, poison = 0.f;   // largest |x0| of this thread: the range certificate of the plain result (Src::bound)
    // grid: x = row tile, y = local disparity, z = b*C + c
    const int bc = blockIdx.z, dl = blockIdx.y;
    const int d = d_begin + dl;
    const size_t px = (size_t)h * w;
    const float* a = A + (size_t)bc * cstride;
    const float* g = G + (size_t)bc * cstride;
    const float* g2 = G2 + (size_t)bc * cstride;
    float* dst = x0 + ((size_t)bc * d_count + dl) * px;
    if ((w & 3) == 0 && (reinterpret_cast<uintptr_t>(x0) & 15) == 0) {
        // four consecutive columns per thread, one 16-byte store, four quads in flight (the element loop below with its
        // division per element wrote 1.9 TB/s)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int wq = w >> 2, quads = h * wq;
#pragma unroll 4
        for (int q = threadIdx.x; q < quads; q += 256) {
            const int y = q / wq, xb = 4 * (q - y * wq);
            const float* ar = a + (size_t)y * (w + 1) + xb;
            const float* gr = g + (size_t)y * (w + 1) + (xb - d + 1);   // column u + 1 of u = x - d
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int x = xb + e, u = x - d;
                float t = ar[e];
                const float gv = u >= -1 ? gr[e] : 0.f;
                t += (x == w - 1 && d >= 1 && u >= -1) ? g2[(size_t)y * (w + 1) + (u + 1)] : gv;
                v[e] = t;
                seen = fmaxf(seen, fabsf(t));
                poison = fmaf(t, 0.f, poison);   // NaN / inf stick (fmaxf alone drops a NaN)
            }
            *reinterpret_cast<f32x4*>(dst + (size_t)y * w + xb) = v;
        }
        if (amax) block_amax_record(poison == poison ? seen : __builtin_inff(), amax + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, red);
        return;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < px; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i % w);
        const int u = x - d;
        float v = a[(size_t)y * (w + 1) + x];
        if (u >= -1) {
            const size_t off = (size_t)y * (w + 1) + (u + 1);
            v += (x == w - 1 && d >= 1) ? g2[off] : g[off];
        }
        dst[i] = v;
        seen = fmaxf(seen, fabsf(v));
        poison = fmaf(v, 0.f, poison);
    }
    if (amax) block_amax_record(poison == poison ? seen : __builtin_inff(), amax + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, red);
}

// one workgroup per (batch entry, channel, plane): the amax records (one per workgroup) stay a few thousand
int l0_combine_records(int batch, int channels, int d_count) { return batch * channels * d_count; }

int launch_l0_combine(const float* A, const float* G, const float* G2, size_t cstride, float* x0, int batch,
                      int channels, int h, int w, int d_begin, int d_count, hipStream_t s, float* amax) {
    hipLaunchKernelGGL(l0_combine_kernel, dim3(1, d_count, batch * channels), dim3(256), 0, s, A, G, G2, cstride, x0,
                       channels, h, w, d_begin, d_count, amax);
    return check_launch("l0_combine");
}

// ---- adjoint of l0_combine (training: pds_matching_bwd) -----------------------------------------------------------
// g [B, C, d_count, h, w] = d loss / d x0  ->  the gradients of the three planes x0 was formed from, each as a
// contiguous single-plane tensor [B*C][h][w + 1] in the column convention of the forward (A at column x + 1, G / G2
// at column u + 1, u = x - d >= -1):
//   gA[x]  = sum_d g[d][x]
//   gG[u]  = sum over the planes d with 0 <= u + d <= w - 1 that read G there (not the x = w - 1, d >= 1 case)
//   gG2[u] = g[d*][w - 1] for the one plane d* = w - 1 - u >= 1 of this call that read G2[u], else 0
// Written as  gy_a = gA,  gy_gs = gG + gG2,  gy_g = gG: the weight gradient of the right half is
// wgrad(R~, gG + gG2) for the taps dx <= 0 and wgrad(R~, gG) for dx = +1 (G2 = conv_R without its dx = +1 taps).
// One thread per column j = u + 1 = x + 1 of one (b, c, y) row; consecutive threads read consecutive addresses of
// every plane, so the 425 MB gradient streams through once for gA and once more (mostly from L2) for gG.
__global__ __launch_bounds__(256) void l0_combine_bwd_kernel(const float* __restrict__ g, float* __restrict__ gy_a,
                                                             float* __restrict__ gy_gs, float* __restrict__ gy_g,
                                                             float* __restrict__ gy_g2, int h, int w, int d_begin,
                                                             int d_count) {
    const int bc = blockIdx.z, y = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j > w) return;
    const size_t px = (size_t)h * w;
    const float* row = g + (size_t)bc * d_count * px + (size_t)y * w;
    const int x = j - 1, u = j - 1;
    float sa = 0.f, sg = 0.f, sg2 = 0.f;
    // branch-free body, eight planes (sixteen loads) in flight: the loop of conditional loads ran at 1.6 TB/s
    const int xa = max(x, 0);
#pragma unroll 8
    for (int dl = 0; dl < d_count; ++dl) {
        const int d = d_begin + dl;
        const float* p = row + (size_t)dl * px;
        const int xs = u + d;   // the column of plane d that read G[u] (or G2[u])
        const bool in = xs >= 0 && xs <= w - 1;
        const float va = p[xa];
        const float v = p[min(max(xs, 0), w - 1)];
        sa += x >= 0 ? va : 0.f;
        const bool edge = xs == w - 1 && d >= 1;
        sg2 += (in && edge) ? v : 0.f;
        sg += (in && !edge) ? v : 0.f;
    }
    const size_t o = ((size_t)bc * h + y) * (w + 1) + j;
    gy_a[o] = sa;
    gy_gs[o] = sg + sg2;
    gy_g[o] = sg;
    gy_g2[o] = sg2;
}

int launch_l0_combine_bwd(const float* g, float* gy_a, float* gy_gs, float* gy_g, float* gy_g2, int batch,
                          int channels, int h, int w, int d_begin, int d_count, hipStream_t s) {
    hipLaunchKernelGGL(l0_combine_bwd_kernel, dim3((w + 1 + 255) / 256, h, batch * channels), dim3(256), 0, s, g, gy_a,
                       gy_gs, gy_g, gy_g2, h, w, d_begin, d_count);
    return check_launch("l0_combine_bwd");
}

// d conv0.weight [cout][2C][9] from the three single-plane weight gradients (see above): left half = dwl, right half =
// dws for the taps dx <= 0 and dwg for dx = +1
__global__ __launch_bounds__(256) void first_weight_grads_kernel(const float* __restrict__ dwl,
                                                                 const float* __restrict__ dws,
                                                                 const float* __restrict__ dwg, float* __restrict__ dw0,
                                                                 int cout, int C) {
    const int total = cout * C * 9;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int t = i % 9;
        const int c = (i / 9) % C;
        const int o = i / (9 * C);
        dw0[((size_t)o * 2 * C + c) * 9 + t] = dwl[i];
        dw0[((size_t)o * 2 * C + C + c) * 9 + t] = (t % 3 == 2) ? dwg[i] : dws[i];
    }
}

int launch_first_weight_grads(const float* dwl, const float* dws, const float* dwg, float* dw0, int cout, int cin_half,
                              hipStream_t s) {
    const int total = cout * cin_half * 9;
    hipLaunchKernelGGL(first_weight_grads_kernel, dim3((total + 255) / 256), dim3(256), 0, s, dwl, dws, dwg, dw0, cout,
                       cin_half);
    return check_launch("first_weight_grads");
}

// out[r][x] = a[r][x + 1] (+ b[r][x + 1]): drops the zero column the layer-0 planes carry on their left
__global__ __launch_bounds__(256) void crop_left1_add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             float* __restrict__ out, size_t rows, int w) {
    const size_t total = rows * (size_t)w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / w;
        const int x = (int)(i % w);
        const size_t o = r * (w + 1) + x + 1;
        out[i] = a[o] + (b ? b[o] : 0.f);
    }
}

int launch_crop_left1_add(const float* a, const float* b, float* out, size_t rows, int w, hipStream_t s) {
    const size_t total = rows * (size_t)w;
    unsigned bx = (unsigned)((total + 255) / 256);
    if (bx > 8192) bx = 8192;
    hipLaunchKernelGGL(crop_left1_add_kernel, dim3(bx), dim3(256), 0, s, a, b, out, rows, w);
    return check_launch("crop_left1_add");
}

// Inputs of the layer-0 convolutions as ONE volume [B*C][planes][h][w+pad]: plane 0 = left descriptor,
// planes 1 (and 2) = right descriptor, each behind `pad` zero columns on the left.
__global__ __launch_bounds__(256) void l0_stack_inputs_kernel(const float* __restrict__ left,
                                                              const float* __restrict__ right,
                                                              float* __restrict__ out, size_t bc_count, int h, int w,
                                                              int planes, int pad) {
    const int wp = w + pad;
    const size_t total = bc_count * planes * h * (size_t)wp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % wp);
        size_t r = i / wp;
        const int y = (int)(r % h);
        r /= h;
        const int p = (int)(r % planes);
        const size_t bc = r / planes;
        const float* src = p == 0 ? left : right;
        out[i] = x < pad ? 0.f : src[(bc * h + y) * w + x - pad];
    }
}

// the same for even w and pad (the column form: pad = 2): one workgroup per output row, float2 per thread, no index
// arithmetic beyond the block coordinates (round 6: the generic kernel above spends 17 us on 18 MB in divisions)
__global__ __launch_bounds__(128) void l0_stack_rows_kernel(const float* __restrict__ left, const float* __restrict__ right,
                                                            float* __restrict__ out, int h, int w, int planes, int pad) {
    const int y = blockIdx.x, p = blockIdx.y % planes;
    const size_t bc = blockIdx.y / planes;
    const int wp = w + pad;
    const float* src = (p == 0 ? left : right) + (bc * h + y) * (size_t)w;
    float* dst = out + ((bc * planes + p) * h + y) * (size_t)wp;
    for (int x = 2 * threadIdx.x; x < wp; x += 256) {
        float2 v = make_float2(0.f, 0.f);
        if (x >= pad) v = *reinterpret_cast<const float2*>(src + x - pad);
        *reinterpret_cast<float2*>(dst + x) = v;
    }
}

int launch_l0_stack_inputs(const float* left, const float* right, float* out, size_t bc_count, int h, int w,
                           int planes, int pad, hipStream_t s) {
    if ((w & 1) == 0 && (pad & 1) == 0 && h <= 65535 && bc_count * planes <= 65535) {
        hipLaunchKernelGGL(l0_stack_rows_kernel, dim3(h, (unsigned)(bc_count * planes)), dim3(128), 0, s, left, right, out,
                           h, w, planes, pad);
        return check_launch("l0_stack_inputs");
    }
    const size_t total = bc_count * planes * h * (size_t)(w + pad);
    unsigned bx = (unsigned)((total + 255) / 256);
    if (bx > 8192) bx = 8192;
    hipLaunchKernelGGL(l0_stack_inputs_kernel, dim3(bx), dim3(256), 0, s, left, right, out, bc_count, h, w, planes, pad);
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

// ---------------------------------------------------------------------------------------------------
// Column form of the two factorisations (round 2).  G2, Ha, Hb and H0 differ from G / H only through the taps of ONE
// kernel column, and each is read at ONE column per disparity plane (u = w-1-d, w-2-d) or at x = 0, so instead of
// whole extra planes (three of the five 128-channel planes of the launch above, one of the three of layer 0):
//   G2[u]          = G[u]  - sum_{ic,dy} Wr[dy][+1] R[y+dy][u+1]                                   (u = w-1-d, d >= 1)
//   Ha[u] - H[u]   =  sum W1[dy][+1] D[y+dy][u+1]                       D = G2 - G                 (u = w-2-d)
//   Hb[u] - H[u]   =  sum W1[dy][0]  D[y+dy][u] - sum W1[dy][+1] G[y+dy][u+1]                      (u = w-1-d)
//   H0[0] - H[0]   = -sum W1[dy][-1] G[y+dy][-1]
// (sums over the 64 input channels and the three kernel rows; rows and columns outside the images are zero).  Only
// A, G (layer 0) and B, H (layer 1) remain full 64-channel planes; the corrections are a few thousand short dot
// products.  Weight columns are re-laid out as wcol[dx][ic][dy][oc] so that a wave reads them through scalar loads.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void column_weights_kernel(const float* __restrict__ wt, int cin_total, int cin_off,
                                                             int C, int cout, float* __restrict__ out) {
    const int total = 3 * C * 3 * cout;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int oc = i % cout;
        const int dy = (i / cout) % 3;
        const int ic = (i / (3 * cout)) % C;
        const int dx = i / (3 * cout * C);
        out[i] = wt[(((size_t)oc * cin_total + cin_off + ic) * 3 + dy) * 3 + dx];
    }
}

int launch_column_weights(const float* wt, int cin_total, int cin_off, int channels, int cout, float* out,
                          hipStream_t s) {
    const int total = 9 * channels * cout;
    hipLaunchKernelGGL(column_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, wt, cin_total, cin_off,
                       channels, cout, out);
    return check_launch("column_weights");
}

constexpr int kColOcg = 8;      // output channels per thread
constexpr int kColSlices = 8;   // the input channels are split over the 8 waves of a workgroup (these kernels are a few
                                // thousand short dot products: bound by load latency, so spread them as wide as possible)

// G2 at the columns the disparity planes [d_lo, d_lo + count) (all >= 1, <= w) read: u = w - 1 - d.  G / G2: row
// stride rs, column u + co.  Also zeroes the first `zero_cols` columns of A (same strides): with two padding columns
// the layer-0 output doubles as the input of the layer-1 launch, which needs literal zeros left of the image.
// grid: x = items (y, j) / 64, y = batch * cout / kColOcg; 512 threads: wave = slice of input channels, lane = item
__global__ __launch_bounds__(512) void l0_column_fix_kernel(const float* __restrict__ G, const float* __restrict__ R,
                                                            const float* __restrict__ wcol, float* __restrict__ G2,
                                                            float* __restrict__ A, int zero_cols, size_t cstride,
                                                            int rs, int co, int C, int cout, int h, int w, int d_lo,
                                                            int count) {
    __shared__ float red[kColSlices][kColOcg][64];
    const int lane = threadIdx.x & 63;
    const int slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int groups = cout / kColOcg;
    const int b = blockIdx.y / groups, ocg = blockIdx.y % groups;
    if (blockIdx.x == 0 && zero_cols > 0) {
        for (int i = threadIdx.x; i < kColOcg * h * zero_cols; i += 512) {
            const int k = i / (h * zero_cols), r = i % (h * zero_cols);
            A[(size_t)(b * cout + ocg * kColOcg + k) * cstride + (size_t)(r / zero_cols) * rs + r % zero_cols] = 0.f;
        }
    }
    const int item = blockIdx.x * 64 + lane;
    const bool live = item < h * count;
    const int y = live ? item / count : 0, d = d_lo + (live ? item % count : 0);
    const int xr = w - d;  // column of R under the dropped tap (u + 1), in [0, w-1]
    float acc[kColOcg];
#pragma unroll
    for (int k = 0; k < kColOcg; ++k) acc[k] = 0.f;
    const float* wc = wcol + (size_t)2 * C * 3 * cout + ocg * kColOcg;  // dx = +1
    const int per = (C + kColSlices - 1) / kColSlices;
    const int ic_end = (slice + 1) * per < C ? (slice + 1) * per : C;
    for (int ic = slice * per; ic < ic_end; ++ic) {
        const float* r = R + ((size_t)(b * C + ic) * h) * w + xr;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            const float v = (live && yy >= 0 && yy < h) ? r[(size_t)yy * w] : 0.f;
            const float* wk = wc + ((size_t)ic * 3 + dy) * cout;
#pragma unroll
            for (int k = 0; k < kColOcg; ++k) acc[k] = fmaf(wk[k], v, acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < kColOcg; ++k) red[slice][k][lane] = acc[k];
    __syncthreads();
    {
        const int k = threadIdx.x >> 6;  // 8 waves <-> 8 output channels
        float v = 0.f;
#pragma unroll
        for (int sl = 0; sl < kColSlices; ++sl) v += red[sl][k][lane];
        if (live) {
            const size_t off = (size_t)(b * cout + ocg * kColOcg + k) * cstride + (size_t)y * rs + (xr - 1 + co);
            G2[off] = G[off] - v;
        }
    }
}

int launch_l0_column_fix(const float* G, const float* R, const float* wcol, float* G2, float* A, int zero_cols,
                         size_t cstride, int rs, int co, int batch, int channels, int cout, int h, int w, int d_begin,
                         int d_count, hipStream_t s) {
    static_assert(kColOcg == kColSlices, "the final reduction maps one wave to one output channel");
    const int d_lo = d_begin > 1 ? d_begin : 1;
    int d_hi = d_begin + d_count - 1;
    if (d_hi > w) d_hi = w;
    int count = d_hi - d_lo + 1;
    if (count < 0) count = 0;
    if (count == 0 && zero_cols == 0) return 0;
    const int bx = count > 0 ? (h * count + 63) / 64 : 1;
    hipLaunchKernelGGL(l0_column_fix_kernel, dim3(bx, batch * (cout / kColOcg)), dim3(512), 0, s, G, R, wcol, G2, A,
                       zero_cols, cstride, rs, co, channels, cout, h, w, d_lo, count);
    return check_launch("l0_column_fix");
}

// corr [nc][h][d_count][2] = {Ha - H at x = w-2, Hb - H at x = w-1} of disparity plane d (0 where the plane does not
// read them), corr0 [nc][h] = H0 - H at x = 0 (only when d_begin == 0).  G / G2: row stride rs, column u + co.
__global__ __launch_bounds__(512) void l1_column_terms_kernel(const float* __restrict__ G,
                                                              const float* __restrict__ G2,
                                                              const float* __restrict__ wcol,
                                                              float* __restrict__ corr, float* __restrict__ corr0,
                                                              size_t cstride, int rs, int co, int C, int cout, int h,
                                                              int w, int d_begin, int d_count) {
    __shared__ float red[kColSlices][2 * kColOcg][64];
    const int lane = threadIdx.x & 63;
    const int slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int groups = cout / kColOcg;
    const int b = blockIdx.y / groups, ocg = blockIdx.y % groups;
    // items: the planes d >= 1 of every row first (blocks [0, main_blocks)), then -- when the call includes d = 0 -- one
    // item per row for corr0 (the remaining blocks): the two kinds never share a wave, so no wave runs both loops
    const int dl0 = d_begin == 0 ? 1 : 0;                 // first local plane with d >= 1
    const int per_row = d_count - dl0;
    const int main_blocks = (h * per_row + 63) / 64;
    const bool zero_block = (int)blockIdx.x >= main_blocks;
    const int item = (zero_block ? (int)blockIdx.x - main_blocks : (int)blockIdx.x) * 64 + lane;
    const bool live = item < (zero_block ? h : h * per_row);
    const int y = live ? (zero_block ? item : item / per_row) : 0;
    const int dl = (live && !zero_block) ? dl0 + item % per_row : 0, d = d_begin + dl;
    const size_t wstep = (size_t)C * 3 * cout;
    const float* w_m = wcol + ocg * kColOcg;   // dx = -1
    const float* w_0 = w_m + wstep;            // dx = 0
    const float* w_p = w_0 + wstep;            // dx = +1
    const int per = (C + kColSlices - 1) / kColSlices;
    const int ic_begin = slice * per, ic_end = (slice + 1) * per < C ? (slice + 1) * per : C;
    float accA[kColOcg], accB[kColOcg];
#pragma unroll
    for (int k = 0; k < kColOcg; ++k) accA[k] = accB[k] = 0.f;
    if (zero_block) {
        if (live)
        for (int ic = ic_begin; ic < ic_end; ++ic) {
            const float* g = G + (size_t)(b * C + ic) * cstride + (co - 1);   // u = -1
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = y + dy - 1;
                const float v = (yy >= 0 && yy < h) ? g[(size_t)yy * rs] : 0.f;
                const float* wk = w_m + ((size_t)ic * 3 + dy) * cout;
#pragma unroll
                for (int k = 0; k < kColOcg; ++k) accA[k] = fmaf(-wk[k], v, accA[k]);
            }
        }
    } else if (live) {
        // u = w - 1 - d: D[u] lives at storage column u + co (valid for u >= -1, i.e. d <= w), G[u + 1] next to it
        // (valid for d <= w + 1; d >= 1 keeps it inside the row)
        const int u = w - 1 - d;
        const bool has_d = u >= -1, has_g = u + 1 >= -1;
        // four input channels at a time: all 36 loads are issued before the first FMA needs one (the plain loop waited
        // for a round trip to L2 per channel row: 35 us for a few thousand dot products)
        for (int ic0 = ic_begin; ic0 < ic_end; ic0 += 4) {
            float dvv[4][3], gnn[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ch_ok = ic0 + j < ic_end;
                const size_t base = (size_t)(b * C + (ch_ok ? ic0 + j : ic_begin)) * cstride;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = y + dy - 1;
                    const bool row_ok = ch_ok && yy >= 0 && yy < h;
                    const size_t off = base + (size_t)(row_ok ? yy : 0) * rs + co;
                    const float g2v = G2[off + (has_d ? u : 0)], gv = G[off + (has_d ? u : 0)];
                    const float gnv = G[off + (has_g ? u + 1 : 0)];
                    dvv[j][dy] = (row_ok && has_d) ? g2v - gv : 0.f;
                    gnn[j][dy] = (row_ok && has_g) ? gnv : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ic = ic0 + j < ic_end ? ic0 + j : ic_begin;   // (values are zero for the surplus)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float* wk0 = w_0 + ((size_t)ic * 3 + dy) * cout;
                    const float* wkp = w_p + ((size_t)ic * 3 + dy) * cout;
#pragma unroll
                    for (int k = 0; k < kColOcg; ++k) {
                        accA[k] = fmaf(wkp[k], dvv[j][dy], accA[k]);
                        accB[k] = fmaf(wk0[k], dvv[j][dy], fmaf(-wkp[k], gnn[j][dy], accB[k]));
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kColOcg; ++k) {
        red[slice][k][lane] = accA[k];
        red[slice][kColOcg + k][lane] = accB[k];
    }
    __syncthreads();
    const int k = threadIdx.x >> 6;
    float va = 0.f, vb = 0.f;
#pragma unroll
    for (int sl = 0; sl < kColSlices; ++sl) {
        va += red[sl][k][lane];
        vb += red[sl][kColOcg + k][lane];
    }
    if (!live) return;
    const size_t nc = (size_t)(b * cout + ocg * kColOcg + k);
    if (zero_block) {
        corr0[nc * h + y] = va;
        return;
    }
    const bool a_used = d >= 1 && w >= 2 && d <= w;       // x = w - 2 exists and u = w - 2 - d >= -2
    const bool b_used = d >= 1 && d <= w + 1;             // u = w - 1 - d >= -2
    *reinterpret_cast<float2*>(corr + ((nc * h + y) * d_count + dl) * 2) =
        make_float2(a_used ? va : 0.f, b_used ? vb : 0.f);
}

int launch_l1_column_terms(const float* G, const float* G2, const float* wcol, float* corr, float* corr0,
                           size_t cstride, int rs, int co, int batch, int channels, int cout, int h, int w, int d_begin,
                           int d_count, hipStream_t s) {
    const int per_row = d_count - (d_begin == 0 ? 1 : 0);
    const int blocks = (h * per_row + 63) / 64 + (d_begin == 0 ? (h + 63) / 64 : 0);
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(l1_column_terms_kernel, dim3(blocks, batch * (cout / kColOcg)), dim3(512), 0, s, G, G2, wcol,
                       corr, corr0, cstride, rs, co, channels, cout, h, w, d_begin, d_count);
    return check_launch("l1_column_terms");
}

// weight sets [2][Cout][C][3][3] = {W1, W1} and bias sets [2][Cout] = {b1, 0} of the two-plane launch (B, H)
__global__ __launch_bounds__(256) void l1_weights2_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                                          float* __restrict__ w2, float* __restrict__ bias2, int cout,
                                                          int C) {
    const int total = cout * C * 9;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * cout; i += gridDim.x * 256) bias2[i] = i < cout ? b1[i] : 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const float v = w1[i];
        w2[i] = v;
        w2[total + i] = v;
    }
}

int launch_l1_weights2(const float* w1, const float* b1, float* w2, float* bias2, int cout, int channels,
                       hipStream_t s) {
    const int total = cout * channels * 9;
    hipLaunchKernelGGL(l1_weights2_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w1, b1, w2, bias2, cout,
                       channels);
    return check_launch("l1_weights2");
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

// COLS: y4 holds only the planes B and H ([n][C][2][h][w+2]); the special columns arrive as corrections to H
// (l1_column_terms_kernel: corr [nc][h][d_count][2], corr0 [nc][h]).
// STORE = false: statistics only (round 5: conv2d_x3's first launch forms t1 on the fly from the blocked planes below; the
// records must be the ones this kernel would have written beside t1, bit for bit)
template <bool COLS, bool STORE = true>
__global__ __launch_bounds__(256) void l1_combine_kernel(const float* __restrict__ y4, const float* __restrict__ corr,
                                                         const float* __restrict__ corr0, float* __restrict__ t1,
                                                         double* __restrict__ partials, int C, int h, int w,
                                                         int d_begin, int d_first, int d_launch, int d_count) {
    // grid: x = tile over (y, x/4), y = n*C + o ; y4 [n][C][5 or 2][h][w+2]
    const int nc = blockIdx.y, tile = blockIdx.x, tiles = gridDim.x;
    const int W2 = w + 2;
    const size_t px = (size_t)h * w;
    const float* Bp = y4 + (size_t)nc * (COLS ? 2 : kL1Planes) * h * W2;
    const float* Hp = Bp + (size_t)h * W2;
    const float* Ha = Hp + (size_t)h * W2;   // planes 2..4 exist only without COLS
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
        if (COLS) {
            if (d == 0) {  // the image border is padding at zero disparity: H0 differs from H at x = 0 only
                if (active && xb == 0) tk[0] += corr0[(size_t)nc * h + y];
            } else if (active && xb + 4 > w - 2) {
                // the two right-most columns of the image: Ha / Hb = H + correction
                const float2 cv = *reinterpret_cast<const float2*>(corr + (((size_t)nc * h + y) * d_count + dl) * 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x = xb + k;
                    if (x >= w - 2 && x < w) tk[k] += x == w - 2 ? cv.x : cv.y;
                }
            }
        } else if (d == 0) {  // the image border is padding at zero disparity: its own H plane
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
        if (STORE && active) {
            float* o = t1 + ((size_t)nc * d_count + dl) * px + (size_t)y * w + xb;
            if (vec) {
                // non-temporal: a write-only stream of 418 MB (125 -> 109 us; the consuming conv2d_x3 launch is unchanged)
                typedef float nt4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(nt4{r[0], r[1], r[2], r[3]}, reinterpret_cast<nt4*>(o));
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

// ---- the layer-1 planes, channel-blocked for conv2d_x3's staging (round 5) --------------------------------------------
// per (batch entry, channel group of 8):
//   Bc  [h][w + 2][8]                     Bc[y][x + 2] = B[c][y][x]                            (y4's own column convention)
//   Hx  [h][pad + w + 2][8]               Hx[y][u + 2 + pad] = H[c][y][u] for u >= -2, zero to the left of it
//       [d_count][h][2][8]  "edge"        what l1_combine_kernel adds a column correction to:
//                                           plane d == 0:  entry 0 = H[0] + corr0                       (x = 0)
//                                           plane d >= 1:  entry s = (u >= -2 ? H[u] : 0) + corr[s],  u = w - 2 + s - d
__host__ __device__ size_t l1_blocked_b_floats(int h, int w) { return (size_t)h * (w + 2) * 8; }
__host__ __device__ size_t l1_blocked_edge_offset_floats(int h, int w, int pad) { return (size_t)h * (pad + w + 2) * 8; }
__host__ __device__ size_t l1_blocked_h_floats(int h, int w, int pad, int d_count) {
    return l1_blocked_edge_offset_floats(h, w, pad) + (size_t)d_count * h * 2 * 8;
}

__global__ __launch_bounds__(256) void l1_blocked_kernel(const float* __restrict__ y4, const float* __restrict__ corr,
                                                         const float* __restrict__ corr0, float* __restrict__ Bc,
                                                         float* __restrict__ Hx, int C, int h, int w, int pad, int d_begin,
                                                         int d_count) {
    // grid: x = items of one (batch entry, channel group), y = batch entry * groups + group; one thread = one 32-byte slot
    const int W2 = w + 2, HW = pad + w + 2;
    const int nb = h * W2, nh = h * HW, ne = d_count * h * 2;
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= nb + nh + ne) return;
    const int ng = blockIdx.y, groups = C / 8, n = ng / groups, g = ng % groups;
    const size_t plane = (size_t)h * W2;
    const float* Bp = y4 + ((size_t)(n * C + g * 8) * 2) * plane;   // channel stride 2 planes
    const float* Hp = Bp + plane;
    float v[8];
    float* dst;
    if (item < nb) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = Bp[(size_t)c * 2 * plane + item];
        dst = Bc + (size_t)ng * l1_blocked_b_floats(h, w) + (size_t)item * 8;
    } else if (item < nb + nh) {
        const int i = item - nb, y = i / HW, j = i % HW;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = j >= pad ? Hp[(size_t)c * 2 * plane + (size_t)y * W2 + (j - pad)] : 0.f;
        dst = Hx + (size_t)ng * l1_blocked_h_floats(h, w, pad, d_count) + (size_t)i * 8;
    } else {
        const int i = item - nb - nh, slot = i & 1, y = (i >> 1) % h, dl = (i >> 1) / h, d = d_begin + dl;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const size_t nc = (size_t)n * C + g * 8 + c;
            const float* hrow = Hp + (size_t)c * 2 * plane + (size_t)y * W2;
            float t = 0.f;
            if (d == 0) {
                if (slot == 0) t = hrow[2] + corr0[nc * h + y];
            } else {
                const int u = w - 2 + slot - d;
                t = (u >= -2 ? hrow[u + 2] : 0.f) + corr[((nc * h + y) * d_count + dl) * 2 + slot];
            }
            v[c] = t;
        }
        dst = Hx + (size_t)ng * l1_blocked_h_floats(h, w, pad, d_count) + l1_blocked_edge_offset_floats(h, w, pad) +
              (size_t)i * 8;
    }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    reinterpret_cast<f32x4*>(dst)[0] = f32x4{v[0], v[1], v[2], v[3]};
    reinterpret_cast<f32x4*>(dst)[1] = f32x4{v[4], v[5], v[6], v[7]};
}

int launch_l1_blocked(const float* y4, const float* corr, const float* corr0, float* Bc, float* Hx, int batch, int channels,
                      int h, int w, int pad, int d_begin, int d_count, hipStream_t s) {
    const int items = h * (w + 2) + h * (pad + w + 2) + d_count * h * 2;
    hipLaunchKernelGGL(l1_blocked_kernel, dim3((items + 255) / 256, batch * (channels / 8)), dim3(256), 0, s, y4, corr, corr0,
                       Bc, Hx, channels, h, w, pad, d_begin, d_count);
    return check_launch("l1_blocked");
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

int launch_l1_combine(const float* y4, const float* corr, const float* corr0, float* t1, double* partials,
                      int batch, int channels, int h, int w, int d_begin, int d_count, hipStream_t s) {
    for (int first = 0; first < d_count; first += kL1MaxPlanes) {
        const int n = d_count - first < kL1MaxPlanes ? d_count - first : kL1MaxPlanes;
        const dim3 grid(l1_combine_tiles(h, w), batch * channels);
        if (corr && !t1)
            hipLaunchKernelGGL((l1_combine_kernel<true, false>), grid, dim3(256), 0, s, y4, corr, corr0, t1, partials,
                               channels, h, w, d_begin, first, n, d_count);
        else if (corr)
            hipLaunchKernelGGL(l1_combine_kernel<true>, grid, dim3(256), 0, s, y4, corr, corr0, t1, partials, channels, h,
                               w, d_begin, first, n, d_count);
        else
            hipLaunchKernelGGL(l1_combine_kernel<false>, grid, dim3(256), 0, s, y4, corr, corr0, t1, partials, channels,
                               h, w, d_begin, first, n, d_count);
    }
    return check_launch("l1_combine");
}

}  // namespace pds
