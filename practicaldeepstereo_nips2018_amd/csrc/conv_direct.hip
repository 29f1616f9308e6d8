// Generic direct (VALU) convolution / transposed convolution kernels with the fused
// prologue (deferred InstanceNorm of the producer + skip add, literal zero padding) and
// epilogue (bias, LeakyReLU, per-group partial statistics) used across the hot path.
//
// Semantics follow reference practical_deep_stereo/network_blocks.py:47-85 (conv -> LeakyReLU(0.1)
// -> InstanceNorm(affine)): a layer stores t = LeakyReLU(conv(x) + bias) and partial sums of t, t^2;
// the NEXT layer reads gamma*rstd * t + (beta - mean*gamma*rstd).  Zero padding therefore applies to
// the normalised tensor: out-of-range taps contribute literal 0.
//
// These kernels cover every channel count; the 64->64 Matching convolutions have a dedicated
// MFMA kernel (conv2d_mfma.hip).
#include "common.hpp"

namespace pds {

struct ConvArgs {
    Src a, b;
    const float* __restrict__ w;
    const float* __restrict__ bias;
    float* __restrict__ out;
    double* __restrict__ partials;
    int N, Cin, Di, Hi, Wi;
    int Cout, Do, Ho, Wo;
    int lrelu;
    int tiles;
};

__device__ __forceinline__ void src_coeffs(const Src& s, int n, int C, int c, int D, int d, float& sc,
                                           float& sh) {
    if (s.scale) {
        const int g = s.per_plane ? ((n * C + c) * D + d) : (n * C + c);
        sc = s.scale[g];
        sh = s.shift[g];
    } else {
        sc = 1.f;
        sh = 0.f;
    }
}

// Block-level reduction of per-thread (sum, sumsq) for NV values; lane 0 of wave 0 .. writes.
template <int NV>
__device__ __forceinline__ void block_stats_store(const float (&s)[NV], const float (&q)[NV],
                                                  double* __restrict__ dst /* NV records of 2 doubles, strided */,
                                                  size_t record_stride, int nvalid) {
    __shared__ double red[4][NV][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double ds = wave_sum((double)s[i]);
        double dq = wave_sum((double)q[i]);
        if (lane == 0) {
            red[wave][i][0] = ds;
            red[wave][i][1] = dq;
        }
    }
    __syncthreads();
    if (threadIdx.x < NV * 2) {
        const int i = threadIdx.x >> 1, k = threadIdx.x & 1;
        if (i < nvalid) {
            double v = red[0][i][k] + red[1][i][k] + red[2][i][k] + red[3][i][k];
            dst[(size_t)i * record_stride * 2 + k] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// conv (KD x 3 x 3), stride S in H, W (and in D when KD == 3), padding k/2.
// grid: x = tile of 256 threads over (oy, x-group); y = od; z = n * ocbs + ocb.
// ---------------------------------------------------------------------------------------------
template <int KD, int S, int OCB, int PX>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvArgs A) {
    constexpr int NIN = (PX - 1) * S + 3;
    const int tile = blockIdx.x, od = blockIdx.y;
    const int ocbs = (A.Cout + OCB - 1) / OCB;
    const int n = blockIdx.z / ocbs, oc0 = (blockIdx.z % ocbs) * OCB;
    const int xgs = (A.Wo + PX - 1) / PX;
    const int g = tile * 256 + threadIdx.x;
    const int oy = g / xgs, ox0 = (g % xgs) * PX;
    const bool active = oy < A.Ho;
    const int ix0 = ox0 * S - 1;
    const size_t plane_i = (size_t)A.Hi * A.Wi;

    float acc[OCB][PX];
#pragma unroll
    for (int o = 0; o < OCB; ++o)
#pragma unroll
        for (int p = 0; p < PX; ++p) acc[o][p] = 0.f;

    for (int ic = 0; ic < A.Cin; ++ic) {
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) {
            const int id = (KD == 1) ? od : od * S + kd - 1;
            if (id < 0 || id >= A.Di) continue;
            float sa, ha, sb = 0.f, hb = 0.f;
            src_coeffs(A.a, n, A.Cin, ic, A.Di, id, sa, ha);
            const float* pa = A.a.p + ((size_t)(n * A.Cin + ic) * A.Di + id) * plane_i;
            const float* pb = nullptr;
            if (A.b.p) {
                src_coeffs(A.b, n, A.Cin, ic, A.Di, id, sb, hb);
                pb = A.b.bcast_d ? A.b.p + (size_t)(n * A.Cin + ic) * plane_i
                                 : A.b.p + ((size_t)(n * A.Cin + ic) * A.Di + id) * plane_i;
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int iy = oy * S + kh - 1;
                const bool rowok = active && iy >= 0 && iy < A.Hi;
                float v[NIN];
#pragma unroll
                for (int j = 0; j < NIN; ++j) {
                    const int ix = ix0 + j;
                    const bool ok = rowok && ix >= 0 && ix < A.Wi;
                    float t = 0.f;
                    if (ok) {
                        const size_t off = (size_t)iy * A.Wi + ix;
                        t = fmaf(sa, pa[off], ha);
                        if (pb) t += fmaf(sb, pb[off], hb);
                    }
                    v[j] = t;
                }
#pragma unroll
                for (int o = 0; o < OCB; ++o) {
                    const int oc = min(oc0 + o, A.Cout - 1);
                    const float* wp = A.w + (((size_t)oc * A.Cin + ic) * KD + kd) * 9 + kh * 3;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float wv = wp[kw];
#pragma unroll
                        for (int p = 0; p < PX; ++p) acc[o][p] = fmaf(wv, v[p * S + kw], acc[o][p]);
                    }
                }
            }
        }
    }

    // epilogue
    float ssum[OCB], ssq[OCB];
    const size_t plane_o = (size_t)A.Ho * A.Wo;
#pragma unroll
    for (int o = 0; o < OCB; ++o) {
        ssum[o] = 0.f;
        ssq[o] = 0.f;
        const int oc = oc0 + o;
        if (oc >= A.Cout) continue;
        const float bv = A.bias ? A.bias[oc] : 0.f;
        float* po = A.out + ((size_t)(n * A.Cout + oc) * A.Do + od) * plane_o + (size_t)oy * A.Wo;
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int ox = ox0 + p;
            if (active && ox < A.Wo) {
                float t = acc[o][p] + bv;
                if (A.lrelu) t = t > 0.f ? t : t * kLeakySlope;
                po[ox] = t;
                ssum[o] += t;
                ssq[o] = fmaf(t, t, ssq[o]);
            }
        }
    }
    if (A.partials) {
        // record index ((n*Cout + oc)*Do + od)*tiles + tile
        double* dst = A.partials + ((((size_t)(n * A.Cout + oc0) * A.Do + od) * A.tiles) + tile) * 2;
        block_stats_store<OCB>(ssum, ssq, dst, (size_t)A.Do * A.tiles, min(OCB, A.Cout - oc0));
    }
}

template <int KD, int S, int OCB, int PX>
static int launch_conv_cfg(const ConvArgs& A, hipStream_t s) {
    const int xgs = (A.Wo + PX - 1) / PX;
    const int tiles = (A.Ho * xgs + 255) / 256;
    ConvArgs B = A;
    B.tiles = tiles;
    dim3 grid(tiles, A.Do, A.N * ((A.Cout + OCB - 1) / OCB));
    hipLaunchKernelGGL((conv_direct_kernel<KD, S, OCB, PX>), grid, dim3(256), 0, s, B);
    return check_launch("conv_direct");
}

// PX used for a geometry: must be the same in conv_direct_tiles() and launch_conv_direct().
static bool conv_use_small(const Geom& o) {
    // few output positions: favour parallelism over register blocking
    return (size_t)o.d * o.h * o.w * o.n * ((o.c + 7) / 8) < (size_t)256 * 1024;
}
static int conv_px(const Geom& o, int stride) {
    if (conv_use_small(o)) return 1;
    return stride == 1 ? 4 : 2;
}

int launch_conv_direct(const ConvLayer& L, hipStream_t s) {
    ConvArgs A;
    A.a = L.a;
    A.b = L.b;
    A.w = L.weight;
    A.bias = L.bias;
    A.out = L.out;
    A.partials = L.partials;
    A.N = L.in.n;
    A.Cin = L.in.c;
    A.Di = L.in.d;
    A.Hi = L.in.h;
    A.Wi = L.in.w;
    A.Cout = L.out_g.c;
    A.Do = L.out_g.d;
    A.Ho = L.out_g.h;
    A.Wo = L.out_g.w;
    A.lrelu = L.lrelu;
    A.tiles = 0;
    const int px = conv_px(L.out_g, L.stride);
    const bool small = px == 1;
    if (L.kd == 1 && L.stride == 1)
        return small ? launch_conv_cfg<1, 1, 2, 1>(A, s) : launch_conv_cfg<1, 1, 8, 4>(A, s);
    if (L.kd == 3 && L.stride == 1)
        return small ? launch_conv_cfg<3, 1, 2, 1>(A, s) : launch_conv_cfg<3, 1, 8, 4>(A, s);
    if (L.kd == 3 && L.stride == 2)
        return small ? launch_conv_cfg<3, 2, 2, 1>(A, s) : launch_conv_cfg<3, 2, 8, 2>(A, s);
    return set_error(-1, "conv_direct: unsupported kd=%d stride=%d", L.kd, L.stride);
}

// how many tiles launch_conv_direct will use (needed to size the partials buffer up front)
int conv_direct_tiles_for(const Geom& o, int stride) {
    const int px = conv_px(o, stride);
    const int xgs = (o.w + px - 1) / px;
    return (o.h * xgs + 255) / 256;
}

// ---------------------------------------------------------------------------------------------
// transposed conv, kernel (KD, 4, 4), stride (KD == 4 ? 2 : 1, 2, 2), padding 1 everywhere
// (network_blocks.py:37-44 and :124-131).  Each thread owns the 2x2 output patch of one input
// position (i, j) for OCB output channels: oy = 2i+py uses (iy, kh) in {(i,1),(i-1,3)} for py = 0
// and {(i,2),(i+1,0)} for py = 1 (oy = 2*iy - 1 + kh); likewise in x.
// grid: x = tile over (i, j); y = od; z = n * ocbs + ocb.
// ---------------------------------------------------------------------------------------------
template <int KD, int OCB>
__global__ __launch_bounds__(256) void deconv_direct_kernel(const ConvArgs A) {
    constexpr int SD = (KD == 4) ? 2 : 1;
    const int tile = blockIdx.x, od = blockIdx.y;
    const int ocbs = (A.Cout + OCB - 1) / OCB;
    const int n = blockIdx.z / ocbs, oc0 = (blockIdx.z % ocbs) * OCB;
    const int g = tile * 256 + threadIdx.x;
    const int i = g / A.Wi, j = g % A.Wi;
    const bool active = i < A.Hi;
    const size_t plane_i = (size_t)A.Hi * A.Wi;

    float acc[OCB][2][2];
#pragma unroll
    for (int o = 0; o < OCB; ++o) acc[o][0][0] = acc[o][0][1] = acc[o][1][0] = acc[o][1][1] = 0.f;

    for (int ic = 0; ic < A.Cin; ++ic) {
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) {
            const int t = od + 1 - kd;
            if (t < 0 || (t % SD) != 0) continue;
            const int id = t / SD;
            if (id >= A.Di) continue;
            float sa, ha, sb = 0.f, hb = 0.f;
            src_coeffs(A.a, n, A.Cin, ic, A.Di, id, sa, ha);
            const float* pa = A.a.p + ((size_t)(n * A.Cin + ic) * A.Di + id) * plane_i;
            const float* pb = nullptr;
            if (A.b.p) {
                src_coeffs(A.b, n, A.Cin, ic, A.Di, id, sb, hb);
                pb = A.b.bcast_d ? A.b.p + (size_t)(n * A.Cin + ic) * plane_i
                                 : A.b.p + ((size_t)(n * A.Cin + ic) * A.Di + id) * plane_i;
            }
            float v[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = i - 1 + r;
                const bool rowok = active && iy >= 0 && iy < A.Hi;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int ix = j - 1 + q;
                    float tv = 0.f;
                    if (rowok && ix >= 0 && ix < A.Wi) {
                        const size_t off = (size_t)iy * A.Wi + ix;
                        tv = fmaf(sa, pa[off], ha);
                        if (pb) tv += fmaf(sb, pb[off], hb);
                    }
                    v[r][q] = tv;
                }
            }
#pragma unroll
            for (int o = 0; o < OCB; ++o) {
                const int oc = min(oc0 + o, A.Cout - 1);
                const float* wp = A.w + (((size_t)ic * A.Cout + oc) * KD + kd) * 16;
                // (py, r, kh): (0,1,1) (0,0,3) (1,1,2) (1,2,0); same table for (px, q, kw)
#pragma unroll
                for (int py = 0; py < 2; ++py) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int r = (a == 0) ? 1 : (py == 0 ? 0 : 2);
                        const int kh = (py == 0) ? (a == 0 ? 1 : 3) : (a == 0 ? 2 : 0);
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                const int q = (b == 0) ? 1 : (px == 0 ? 0 : 2);
                                const int kw = (px == 0) ? (b == 0 ? 1 : 3) : (b == 0 ? 2 : 0);
                                acc[o][py][px] = fmaf(wp[kh * 4 + kw], v[r][q], acc[o][py][px]);
                            }
                        }
                    }
                }
            }
        }
    }

    float ssum[OCB], ssq[OCB];
    const size_t plane_o = (size_t)A.Ho * A.Wo;
#pragma unroll
    for (int o = 0; o < OCB; ++o) {
        ssum[o] = 0.f;
        ssq[o] = 0.f;
        const int oc = oc0 + o;
        if (oc >= A.Cout || !active) continue;
        const float bv = A.bias ? A.bias[oc] : 0.f;
        float* po = A.out + ((size_t)(n * A.Cout + oc) * A.Do + od) * plane_o;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            float t0 = acc[o][py][0] + bv, t1 = acc[o][py][1] + bv;
            if (A.lrelu) {
                t0 = t0 > 0.f ? t0 : t0 * kLeakySlope;
                t1 = t1 > 0.f ? t1 : t1 * kLeakySlope;
            }
            float2* dst = reinterpret_cast<float2*>(po + (size_t)(2 * i + py) * A.Wo + 2 * j);
            *dst = make_float2(t0, t1);
            ssum[o] += t0 + t1;
            ssq[o] = fmaf(t0, t0, fmaf(t1, t1, ssq[o]));
        }
    }
    if (A.partials) {
        double* dst = A.partials + ((((size_t)(n * A.Cout + oc0) * A.Do + od) * A.tiles) + tile) * 2;
        block_stats_store<OCB>(ssum, ssq, dst, (size_t)A.Do * A.tiles, min(OCB, A.Cout - oc0));
    }
}

int deconv_direct_tiles(const Geom& o) { return ((o.h / 2) * (o.w / 2) + 255) / 256; }

static int deconv_ocb(const Geom& o) {
    if (o.c <= 1) return 1;
    const size_t blocks4 = (size_t)deconv_direct_tiles(o) * o.d * o.n * ((o.c + 3) / 4);
    return blocks4 < 1024 ? 1 : 4;
}

int launch_deconv_direct(const DeconvLayer& L, hipStream_t s) {
    ConvArgs A;
    A.a = L.a;
    A.b = L.b;
    A.w = L.weight;
    A.bias = L.bias;
    A.out = L.out;
    A.partials = L.partials;
    A.N = L.in.n;
    A.Cin = L.in.c;
    A.Di = L.in.d;
    A.Hi = L.in.h;
    A.Wi = L.in.w;
    A.Cout = L.out_g.c;
    A.Do = L.out_g.d;
    A.Ho = L.out_g.h;
    A.Wo = L.out_g.w;
    A.lrelu = L.lrelu;
    A.tiles = deconv_direct_tiles(L.out_g);
    const int ocb = deconv_ocb(L.out_g);
    dim3 grid(A.tiles, A.Do, A.N * ((A.Cout + ocb - 1) / ocb));
    if (L.kd == 4) {
        if (ocb == 4)
            hipLaunchKernelGGL((deconv_direct_kernel<4, 4>), grid, dim3(256), 0, s, A);
        else
            hipLaunchKernelGGL((deconv_direct_kernel<4, 1>), grid, dim3(256), 0, s, A);
    } else if (L.kd == 3) {
        if (ocb == 4)
            hipLaunchKernelGGL((deconv_direct_kernel<3, 4>), grid, dim3(256), 0, s, A);
        else
            hipLaunchKernelGGL((deconv_direct_kernel<3, 1>), grid, dim3(256), 0, s, A);
    } else {
        return set_error(-1, "deconv_direct: unsupported kd=%d", L.kd);
    }
    return check_launch("deconv_direct");
}

// ---------------------------------------------------------------------------------------------
// InstanceNorm statistics: reduce partial records, emit folded (scale, shift) per group.
// Biased variance, eps 1e-5 (torch.nn.InstanceNorm defaults, network_blocks.py:58,72,85).
// ---------------------------------------------------------------------------------------------
// One WAVE per group (round 4; a 256-thread workgroup with two barriers took 5.4 us per launch, 22 launches per pair): the
// lanes stride over the group's records, one shuffle reduction, lane 0 finishes.
__global__ __launch_bounds__(64) void in_finalize_kernel(const double* __restrict__ partials, int per_group,
                                                         double count, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int channels,
                                                         int inner, float* __restrict__ scale,
                                                         float* __restrict__ shift, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out, float* __restrict__ bound_out,
                                                         unsigned* __restrict__ nonfinite) {
    const int g = blockIdx.x;
    if (bound_out && g == 0) in_finalize_bound(gamma, beta, channels, count, bound_out, threadIdx.x);
    in_finalize_group(partials, g, per_group, count, gamma, beta, channels, inner, scale, shift, mean_out, rstd_out,
                      nonfinite, threadIdx.x);
}

// ---- the non-finite statistics counter (include/pds_hip.h, ABI v5): one word of host-mapped memory ----------------
namespace {
std::atomic<unsigned*> g_nonfinite{nullptr};
std::atomic<int> g_nonfinite_state{0};   // 0 not tried, 1 ready, -1 unavailable
}  // namespace
unsigned* nonfinite_counter(hipStream_t s) {
    const int st = g_nonfinite_state.load(std::memory_order_acquire);
    if (st == 1) return g_nonfinite.load(std::memory_order_relaxed);
    if (st < 0) return nullptr;
    // (an allocation is not allowed while a stream is being captured into a graph: the counter then waits for the
    // next eager call)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (g_nonfinite_state.load(std::memory_order_acquire) == 1) return g_nonfinite.load(std::memory_order_relaxed);
    unsigned* p = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&p), 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess || !p) {
        (void)hipGetLastError();
        g_nonfinite_state.store(-1, std::memory_order_release);
        return nullptr;
    }
    *p = 0u;
    g_nonfinite.store(p, std::memory_order_relaxed);
    g_nonfinite_state.store(1, std::memory_order_release);
    return p;
}

long long nonfinite_statistics(int reset) {
    if (g_nonfinite_state.load(std::memory_order_acquire) != 1)
        return g_nonfinite_state.load(std::memory_order_acquire) < 0 ? -1 : 0;
    unsigned* p = g_nonfinite.load(std::memory_order_relaxed);
    volatile unsigned* vp = p;
    const unsigned v = *vp;
    if (reset) __atomic_fetch_sub(p, v, __ATOMIC_RELAXED);   // (kernels still in flight keep adding)
    return (long long)v;
}

int launch_in_finalize(const double* partials, int groups, int per_group, double count, const float* gamma,
                       const float* beta, int channels, int inner, float* scale, float* shift, float* mean,
                       float* rstd, hipStream_t s, float* bound) {
    hipLaunchKernelGGL(in_finalize_kernel, dim3(groups), dim3(64), 0, s, partials, per_group, count, gamma,
                       beta, channels, inner, scale, shift, mean, rstd, bound, nonfinite_counter(s));
    return check_launch("in_finalize");
}

// ---------------------------------------------------------------------------------------------
// out = norm(a) (+ norm(b)) as a plain tensor (module boundaries: ContractionBlock3d /
// ExpansionBlock3d / MatchingOperation outputs, residual sums)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void materialize_kernel(const Src a, const Src b, const Geom g,
                                                          float* __restrict__ out, float* __restrict__ amax) {
    __shared__ float red[16];
    float seen = 0.f;   // largest |out| of this thread (the range certificate of the plain result, Src::bound)
    const size_t vol = g.volume();
    const size_t plane = g.plane();
    const int nc = blockIdx.y;  // n*C + c
    const int n = nc / g.c, c = nc % g.c;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < vol; i += (size_t)gridDim.x * 256) {
        const int d = (int)(i / plane);
        float sa, ha;
        src_coeffs(a, n, g.c, c, g.d, d, sa, ha);
        float v = fmaf(sa, a.p[(size_t)nc * vol + i], ha);
        if (b.p) {
            float sb, hb;
            src_coeffs(b, n, g.c, c, g.d, d, sb, hb);
            const size_t off = b.bcast_d ? (size_t)nc * plane + (i - (size_t)d * plane) : (size_t)nc * vol + i;
            v += fmaf(sb, b.p[off], hb);
        }
        out[(size_t)nc * vol + i] = v;
        seen = fmaxf(seen, fabsf(v) == fabsf(v) ? fabsf(v) : __builtin_inff());
    }
    if (amax) block_amax_record(seen, amax + (size_t)blockIdx.y * gridDim.x + blockIdx.x, red);
}

// out[n,c,d,y,x] = norm(a) + x0,  x0 = A[n,c,y,x] + G[n,c,y,x-d] formed on the fly (misc.hip: l0_combine_kernel)
// One thread per 4 consecutive x: 16-byte loads of a and A, 16-byte stores; G is read unaligned from cache.
__global__ __launch_bounds__(256) void materialize_l0_kernel(const Src a, const Geom g, const float* __restrict__ A,
                                                             const float* __restrict__ G,
                                                             const float* __restrict__ G2, size_t l0_cstride,
                                                             int l0_rs, int d_begin, float* __restrict__ out,
                                                             float* __restrict__ amax) {
    __shared__ float red[16];
    float seen = 0.f, poison = 0.f;
    // grid: x = tile over (y, x/4), y = d, z = n*C + c
    const int nc = blockIdx.z, d = blockIdx.y;
    const int n = nc / g.c, c = nc % g.c;
    const int disp = d_begin + d;
    const size_t px = g.plane();
    const int xq = (g.w + 3) / 4;
    float sa, ha;
    src_coeffs(a, n, g.c, c, g.d, d, sa, ha);
    const float* pa = a.p + ((size_t)nc * g.d + d) * px;
    const float* pA = A + (size_t)nc * l0_cstride;   // row stride l0_rs, already at the column of x = 0
    const float* pG = G + (size_t)nc * l0_cstride;
    const float* pG2 = G2 + (size_t)nc * l0_cstride;
    float* po = out + ((size_t)nc * g.d + d) * px;
    const bool vec = (g.w & 3) == 0;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < g.h * xq; q += gridDim.x * 256) {
        const int y = q / xq, x0 = (q - y * xq) * 4;
        const size_t i = (size_t)y * g.w + x0;
        float av[4], lv[4], r[4];
        const size_t ia = (size_t)y * l0_rs + x0;
        if (vec) {
            const float4 t = *reinterpret_cast<const float4*>(pa + i);
            av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) av[k] = (x0 + k < g.w) ? pa[i + k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) lv[k] = (x0 + k < g.w) ? pA[ia + k] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + k;
            const int u = x - disp;
            float x0v = lv[k];
            if (u >= -1 && x < g.w) {
                const size_t off = (size_t)y * l0_rs + (u + 1);
                x0v += (x == g.w - 1 && disp >= 1) ? pG2[off] : pG[off];
            }
            r[k] = fmaf(sa, av[k], ha) + x0v;
            if (x < g.w) {
                seen = fmaxf(seen, fabsf(r[k]));
                poison = fmaf(r[k], 0.f, poison);   // NaN / inf stick (fmaxf alone drops a NaN): ADVICE r4
            }
        }
        if (vec) {
            *reinterpret_cast<float4*>(po + i) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x0 + k < g.w) po[i + k] = r[k];
        }
    }
    if (amax)
        block_amax_record(poison == poison ? seen : __builtin_inff(),
                          amax + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, red);
}

// Plane-sweeping form of the above for rows that are a multiple of 4 wide: one thread owns four consecutive x of
// one (n, c, y) for ALL disparity planes.  A is loaded once.  The four G values of a thread, G[x - d] for its x, form
// a window that slides by one column per plane: the value entering on the left is the neighbouring lane's right-most
// one (a whole-wave DPP shift), so only lane 0 of a wave and the first quad of a row load it.  The streaming traffic
// is one 16-byte load of a and one 16-byte store per plane (the per-plane G loads of the first version cost as
// much as that stream: 276 -> 157 us without them).
__global__ __launch_bounds__(256) void materialize_l0_sweep_kernel(const Src a, const Geom g, const float* __restrict__ A,
                                                                   const float* __restrict__ G,
                                                                   const float* __restrict__ G2, size_t l0_cstride,
                                                                   int l0_rs, int d_begin, float* __restrict__ out,
                                                                   float* __restrict__ amax) {
    __shared__ float red[16];
    float seen = 0.f, poison = 0.f;
    const int nc = blockIdx.y;
    const int n = nc / g.c, c = nc % g.c;
    const size_t px = g.plane();
    const int xq = g.w / 4;
    const int total = g.h * xq;
    const int q = min((int)(blockIdx.x * 256 + threadIdx.x), total - 1);  // surplus threads shadow the last quad
    const bool active = (int)(blockIdx.x * 256 + threadIdx.x) < total;
    const int y = q / xq, xi = q - y * xq, x0 = xi * 4;
    const size_t i = (size_t)y * g.w + x0;
    const size_t row = (size_t)y * l0_rs;
    const float* pa = a.p + (size_t)nc * g.d * px + i;
    float* po = out + (size_t)nc * g.d * px + i;
    const float* pA = A + (size_t)nc * l0_cstride + row + x0;   // already at the column of x = 0
    const float* pG = G + (size_t)nc * l0_cstride + row;        // column u + 1 holds G[u], u >= -1
    const float* pG2 = G2 + (size_t)nc * l0_cstride + row;
    float lv[4], gw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lv[k] = pA[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int u = x0 + k - d_begin;
        gw[k] = u >= -1 ? pG[u + 1] : 0.f;
    }
    const bool last_quad = x0 + 4 == g.w;
    const bool loads_left = xi == 0 || (threadIdx.x & 63) == 0;  // no left neighbour in this wave / this row
#ifndef PDS_MAT_UN
#define PDS_MAT_UN 8   // planes in flight per thread (round 6, Matching alone under rocprofv3: 2 -> 195 us, 4 -> 187, 8 -> 176, 12 -> 325: spills)
#endif
    constexpr int UN = PDS_MAT_UN;
    for (int d0 = 0; d0 < g.d; d0 += UN) {
        float4 t[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j)
            if (d0 + j < g.d) {
                // non-temporal: `a` is read exactly once (193 -> 175 us; a non-temporal STORE of the sum costs 10 us here
                // and the same hint on conv2d_t8w's / the fused estimator's loads or conv2d_x3's / deconv3d_cell's stores loses)
                typedef float nt4 __attribute__((ext_vector_type(4)));
                const nt4 v = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(pa + (size_t)(d0 + j) * px));
                t[j] = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int d = d0 + j;
            if (d >= g.d) break;
            float sa, ha;
            src_coeffs(a, n, g.c, c, g.d, d, sa, ha);
            const int disp = d_begin + d;
            // right-most column of the image: the dx = +1 taps of the right descriptor do not exist there (G2)
            float g3 = gw[3];
            if (last_quad && disp >= 1) g3 = disp <= g.w ? pG2[g.w - disp] : 0.f;  // u = w - 1 - disp >= -1
            const float4 r = make_float4(fmaf(sa, t[j].x, ha) + (lv[0] + gw[0]), fmaf(sa, t[j].y, ha) + (lv[1] + gw[1]),
                                         fmaf(sa, t[j].z, ha) + (lv[2] + gw[2]), fmaf(sa, t[j].w, ha) + (lv[3] + g3));
            if (active) *reinterpret_cast<float4*>(po + (size_t)d * px) = r;
            if (active) {   // (shadow threads of the last workgroup repeat the last quad: not part of the record)
                seen = fmaxf(fmaxf(seen, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
                poison = fmaf(r.x + r.y + r.z + r.w, 0.f, poison);   // a NaN / inf anywhere makes the record +inf
            }
            // slide the window to disparity disp + 1
            const float from_left = __builtin_bit_cast(
                float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gw[3]), 0x138, 0xf, 0xf, true));
            gw[3] = gw[2];
            gw[2] = gw[1];
            gw[1] = gw[0];
            gw[0] = from_left;
            if (loads_left) {
                const int u = x0 - (disp + 1);
                gw[0] = u >= -1 ? pG[u + 1] : 0.f;
            }
        }
    }
    if (amax) block_amax_record(poison == poison ? seen : __builtin_inff(), amax + (size_t)blockIdx.y * gridDim.x + blockIdx.x, red);
}

static dim3 materialize_l0_grid(const Geom& g) {
    if ((g.w & 3) == 0) return dim3((unsigned)((g.h * (g.w / 4) + 255) / 256), (unsigned)(g.n * g.c));
    const int quads = g.h * ((g.w + 3) / 4);
    unsigned bx = (unsigned)((quads + 255) / 256);
    if (bx > 64) bx = 64;
    return dim3(bx, (unsigned)g.d, (unsigned)(g.n * g.c));
}
int materialize_l0_records(const Geom& g) {
    const dim3 grid = materialize_l0_grid(g);
    return (int)(grid.x * grid.y * grid.z);
}

int launch_materialize_l0(const Src& a, const Geom& g, const float* A, const float* G, const float* G2,
                          size_t l0_cstride, int l0_rs, int d_begin, float* out, hipStream_t s, float* amax) {
    const dim3 grid = materialize_l0_grid(g);
    if ((g.w & 3) == 0)
        hipLaunchKernelGGL(materialize_l0_sweep_kernel, grid, dim3(256), 0, s, a, g, A, G, G2, l0_cstride, l0_rs, d_begin,
                           out, amax);
    else
        hipLaunchKernelGGL(materialize_l0_kernel, grid, dim3(256), 0, s, a, g, A, G, G2, l0_cstride, l0_rs, d_begin, out,
                           amax);
    return check_launch("materialize_l0");
}

// workgroups along a channel volume: enough to fill the chip, few enough that the amax records (one per workgroup)
// stay a few thousand floats for the consumer to reduce
static int materialize_bx(const Geom& g) {
    const size_t vol = g.volume();
    int bx = (int)((vol + 255) / 256);
    const int nc = g.n * g.c;
    int cap = 8192 / (nc > 0 ? nc : 1);
    if (cap < 1) cap = 1;
    if (cap > 4096) cap = 4096;
    return bx < cap ? bx : cap;
}
int materialize_records(const Geom& g) { return materialize_bx(g) * g.n * g.c; }

int launch_materialize(const Src& a, const Src& b, const Geom& g, float* out, hipStream_t s, float* amax) {
    hipLaunchKernelGGL(materialize_kernel, dim3(materialize_bx(g), g.n * g.c), dim3(256), 0, s, a, b, g, out, amax);
    return check_launch("materialize");
}

}  // namespace pds
