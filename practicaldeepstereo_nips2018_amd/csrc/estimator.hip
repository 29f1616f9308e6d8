// SubpixelMap (reference practical_deep_stereo/estimator.py:45-91) as ONE streaming pass.
//
// Per pixel: m = first arg-max over the planes; taps k = m + j, j in [lo, hi] (lo = -hw // step with
// Python floor division, hi = hw // step; estimator.py:66-68); taps outside [0, planes) get
// probability 0; disparity = sum_j softmax(s[k_j]) * step * k_j.
//
// HBM-bound: the volume is read exactly once (the reference reads it >= 6 times: max + 5 gathers).
// Each lane owns VEC consecutive pixels (16-byte loads when VEC == 4) and keeps, in registers, a
// delayed window of the last 2T+1 values; when the window's centre becomes the running maximum its
// neighbours are captured, so the taps around the arg-max are known when the sweep ends -- no second
// gather pass.
#include "common.hpp"

namespace pds {

template <int T, int VEC>
__global__ __launch_bounds__(256) void subpixel_map_kernel(const float* __restrict__ sim,
                                                           float* __restrict__ disp, int planes,
                                                           size_t plane_px, int lo, int hi, float step) {
    // lo in [-T, 0], hi in [0, T]
    const size_t b = blockIdx.y;
    const size_t p0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (p0 >= plane_px) return;
    const float* src = sim + b * planes * plane_px + p0;

    // delayed window win[0..2T] of the last planes (win[2T] newest): when its centre (plane k - T) beats the
    // running maximum the T neighbours on either side are captured -- static register indices only (a
    // formulation with conditionally indexed arrays ends up in scratch memory).
    float best[VEC], win[VEC][2 * T + 1], bprev[VEC][T], bnext[VEC][T];
    int bi[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        best[v] = -INFINITY;
        bi[v] = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) bprev[v][t] = bnext[v][t] = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2 * T + 1; ++t) win[v][t] = -INFINITY;
    }

#pragma unroll 8
    for (int k = 0; k < planes + T; ++k) {  // + T flush steps
        float x[VEC];
        if (k < planes) {
            if constexpr (VEC == 4) {
                const float4 q = *reinterpret_cast<const float4*>(src + (size_t)k * plane_px);
                x[0] = q.x;
                x[1] = q.y;
                x[2] = q.z;
                x[3] = q.w;
            } else if constexpr (VEC == 2) {
                const float2 q = *reinterpret_cast<const float2*>(src + (size_t)k * plane_px);
                x[0] = q.x;
                x[1] = q.y;
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) x[v] = src[(size_t)k * plane_px + v];
            }
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[v] = -INFINITY;
        }
        const int centre = k - T;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
#pragma unroll
            for (int t = 0; t < 2 * T; ++t) win[v][t] = win[v][t + 1];
            win[v][2 * T] = x[v];
            const bool up = centre >= 0 && win[v][T] > best[v];  // strict: first occurrence wins
            best[v] = up ? win[v][T] : best[v];
            bi[v] = up ? centre : bi[v];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                bprev[v][t] = up ? win[v][T - 1 - t] : bprev[v][t];
                bnext[v][t] = up ? win[v][T + 1 + t] : bnext[v][t];
            }
        }
    }

    float res[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        // softmax over the valid taps, shifted by the centre value (which is the maximum)
        float den = 1.f;
        float num = step * (float)bi[v];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int kb = bi[v] - (t + 1);
            if (-(t + 1) >= lo && kb >= 0) {
                const float e = expf(bprev[v][t] - best[v]);
                den += e;
                num = fmaf(e, step * (float)kb, num);
            }
            const int ka = bi[v] + (t + 1);
            if ((t + 1) <= hi && ka < planes) {
                const float e = expf(bnext[v][t] - best[v]);
                den += e;
                num = fmaf(e, step * (float)ka, num);
            }
        }
        res[v] = num / den;
    }
    float* dst = disp + b * plane_px + p0;
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(res[0], res[1], res[2], res[3]);
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2*>(dst) = make_float2(res[0], res[1]);
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) dst[v] = res[v];
    }
}

// Fallback for very wide windows (more than 4 taps per side): arg-max sweep + direct gather.
__global__ __launch_bounds__(256) void subpixel_map_wide_kernel(const float* __restrict__ sim,
                                                                float* __restrict__ disp, int planes,
                                                                size_t plane_px, int lo, int hi,
                                                                float step) {
    const size_t b = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= plane_px) return;
    const float* src = sim + b * planes * plane_px + p;
    float best = -INFINITY;
    int bi = 0;
    for (int k = 0; k < planes; ++k) {
        const float x = src[(size_t)k * plane_px];
        if (x > best) {
            best = x;
            bi = k;
        }
    }
    float den = 0.f, num = 0.f;
    for (int j = lo; j <= hi; ++j) {
        const int k = bi + j;
        if (k < 0 || k >= planes) continue;
        const float e = expf(src[(size_t)k * plane_px] - best);
        den += e;
        num = fmaf(e, step * (float)k, num);
    }
    disp[b * plane_px + p] = num / den;
}

template <int T>
static void launch_t(const float* sim, float* disp, int batch, int planes, size_t px, int lo, int hi, int step,
                     hipStream_t s) {
    // pixels per lane: 16-byte loads are the widest, but the sweep is a chain of dependent steps per lane, so what fills
    // the memory pipes is the number of waves: two pixels per lane (8-byte loads, twice the waves) measured fastest in
    // rounds 3-5; the four-pixel form and its switch (PDS_SUBPIXEL_VEC) were retired in round 6.  One pixel per lane
    // serves odd pixel counts.
    if (px % 2 == 0) {
        dim3 grid((unsigned)((px / 2 + 255) / 256), batch);
        hipLaunchKernelGGL((subpixel_map_kernel<T, 2>), grid, dim3(256), 0, s, sim, disp, planes, px, lo, hi,
                           (float)step);
    } else {
        dim3 grid((unsigned)((px + 255) / 256), batch);
        hipLaunchKernelGGL((subpixel_map_kernel<T, 1>), grid, dim3(256), 0, s, sim, disp, planes, px, lo, hi,
                           (float)step);
    }
}

int launch_subpixel_map(const float* sim, float* disp, int batch, int planes, int height, int width, int lo,
                        int hi, int step, hipStream_t s) {
    const size_t px = (size_t)height * width;
    const int t = (-lo > hi) ? -lo : hi;
    if (t <= 1)
        launch_t<1>(sim, disp, batch, planes, px, lo, hi, step, s);
    else if (t <= 2)
        launch_t<2>(sim, disp, batch, planes, px, lo, hi, step, s);
    else if (t <= 4)
        launch_t<4>(sim, disp, batch, planes, px, lo, hi, step, s);
    else {
        dim3 grid((unsigned)((px + 255) / 256), batch);
        hipLaunchKernelGGL(subpixel_map_wide_kernel, grid, dim3(256), 0, s, sim, disp, planes, px, lo, hi,
                           (float)step);
    }
    return check_launch("subpixel_map");
}

}  // namespace pds
