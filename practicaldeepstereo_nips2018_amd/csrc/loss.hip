// SubpixelCrossEntropy (reference practical_deep_stereo/loss.py:16-78) as two streaming kernels.
//
//   log P_k = sim_k - lse(sim);   T_k = exp(-|gt - k*step| / diversity) / (2*diversity)   (unnormalised Laplace)
//   entropy = -sum_k T_k log P_k / sum_k T_k  =  lse - A / S,   A = sum_k T_k sim_k,  S = sum_k T_k
//   loss = mean of entropy over pixels with known ground truth (gt != inf), or, with weights,
//          sum(w * entropy) / (sum(w) + 1e-15) over those pixels.
//   d loss / d sim_k = coef * (softmax_k - T_k / S),  coef = grad * (w or 1) / (sum(w) + 1e-15 or count)
//   d loss / d w_i   = grad * (entropy_i - loss) / (sum(w) + 1e-15)   for known pixels, 0 otherwise (loss.py:74-77)
//
// Forward: ONE pass over the similarity volume with an online log-sum-exp (the reference makes a log-softmax
// copy of the volume and then loops over the planes in Python: > 4 full passes); it keeps lse per pixel.
// Backward: one read of the volume + one write of the gradient.  Reductions are two-stage and deterministic.
#include "common.hpp"

namespace pds {

__device__ __forceinline__ float laplace_target(float gt, int k, float step, float inv_div, float norm) {
    return expf(-fabsf(gt - step * (float)k) * inv_div) * norm;
}

// one thread per pixel; partial records [block] x {sum w*entropy, sum w}
__global__ __launch_bounds__(256) void sce_fwd_kernel(const float* __restrict__ sim, const float* __restrict__ gt,
                                                      const float* __restrict__ weights, float* __restrict__ lse_out,
                                                      double* __restrict__ partials, int planes, size_t plane_px,
                                                      size_t total_px, float step, float diversity) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    double num = 0.0, den = 0.0;
    if (p < total_px) {
        const size_t b = p / plane_px, i = p - b * plane_px;
        const float* src = sim + b * planes * plane_px + i;
        const float g = gt[p];
        const float inv_div = 1.f / diversity, norm = 0.5f / diversity;
        float m = -INFINITY, se = 0.f, S = 0.f, A = 0.f;
        for (int k = 0; k < planes; ++k) {
            const float x = src[(size_t)k * plane_px];
            const float mn = fmaxf(m, x);
            se = se * expf(m - mn) + expf(x - mn);
            m = mn;
            const float t = laplace_target(g, k, step, inv_div, norm);
            S += t;
            A = fmaf(t, x, A);
        }
        const float lse = m + logf(se);
        lse_out[p] = lse;
        if (g != INFINITY) {
            const float entropy = lse - A / S;
            const float w = weights ? weights[p] : 1.f;
            num = (double)w * (double)entropy;
            den = (double)w;
        }
    }
    __shared__ double red[4][2];
    num = wave_sum(num);
    den = wave_sum(den);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave][0] = num;
        red[wave][1] = den;
    }
    __syncthreads();
    if (threadIdx.x < 2)
        partials[(size_t)blockIdx.x * 2 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void sce_finalize_kernel(const double* __restrict__ partials, int blocks,
                                                           int has_weights, float* __restrict__ loss,
                                                           float* __restrict__ stats) {
    double num = 0.0, den = 0.0;
    for (int i = threadIdx.x; i < blocks; i += 256) {
        num += partials[2 * (size_t)i];
        den += partials[2 * (size_t)i + 1];
    }
    __shared__ double red[4][2];
    num = wave_sum(num);
    den = wave_sum(den);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave][0] = num;
        red[wave][1] = den;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        num = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        den = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        const double d = has_weights ? den + 1e-15 : den;  // loss.py:74-78
        stats[0] = (float)num;
        stats[1] = (float)d;
        loss[0] = (float)(num / d);
    }
}

__global__ __launch_bounds__(256) void sce_bwd_kernel(const float* __restrict__ sim, const float* __restrict__ gt,
                                                      const float* __restrict__ weights,
                                                      const float* __restrict__ lse_in,
                                                      const float* __restrict__ stats,
                                                      const float* __restrict__ grad_loss, float* __restrict__ gsim,
                                                      int planes, size_t plane_px, size_t total_px, float step,
                                                      float diversity) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= total_px) return;
    const size_t b = p / plane_px, i = p - b * plane_px;
    const float* src = sim + b * planes * plane_px + i;
    float* dst = gsim + b * planes * plane_px + i;
    const float g = gt[p];
    if (g == INFINITY) {
        for (int k = 0; k < planes; ++k) dst[(size_t)k * plane_px] = 0.f;
        return;
    }
    const float inv_div = 1.f / diversity, norm = 0.5f / diversity;
    float S = 0.f;
    for (int k = 0; k < planes; ++k) S += laplace_target(g, k, step, inv_div, norm);
    const float coef = grad_loss[0] * (weights ? weights[p] : 1.f) / stats[1];
    const float lse = lse_in[p], inv_s = 1.f / S;
    for (int k = 0; k < planes; ++k) {
        const float soft = expf(src[(size_t)k * plane_px] - lse);
        dst[(size_t)k * plane_px] = coef * (soft - laplace_target(g, k, step, inv_div, norm) * inv_s);
    }
}

// Gradient of the weighted loss with respect to the per-pixel weights (loss.py:74-77 under autograd): one more read of the
// volume, entropy re-formed from the kept log-sum-exp.
__global__ __launch_bounds__(256) void sce_weights_bwd_kernel(const float* __restrict__ sim, const float* __restrict__ gt,
                                                              const float* __restrict__ lse_in,
                                                              const float* __restrict__ stats,
                                                              const float* __restrict__ grad_loss,
                                                              float* __restrict__ gweights, int planes, size_t plane_px,
                                                              size_t total_px, float step, float diversity) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= total_px) return;
    const float g = gt[p];
    if (g == INFINITY) {
        gweights[p] = 0.f;
        return;
    }
    const size_t b = p / plane_px, i = p - b * plane_px;
    const float* src = sim + b * planes * plane_px + i;
    const float inv_div = 1.f / diversity, norm = 0.5f / diversity;
    float S = 0.f, A = 0.f;
    for (int k = 0; k < planes; ++k) {
        const float t = laplace_target(g, k, step, inv_div, norm);
        S += t;
        A = fmaf(t, src[(size_t)k * plane_px], A);
    }
    const float entropy = lse_in[p] - A / S;
    gweights[p] = grad_loss[0] * (entropy - stats[0] / stats[1]) / stats[1];
}

size_t sce_partial_doubles(size_t total_px) { return ((total_px + 255) / 256) * 2; }

int launch_sce_fwd(const float* sim, const float* gt, const float* weights, float* loss, float* lse, float* stats,
                   double* partials, int n, int planes, int h, int w, float diversity, int step, hipStream_t s) {
    const size_t plane_px = (size_t)h * w, total = plane_px * n;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(sce_fwd_kernel, dim3(blocks), dim3(256), 0, s, sim, gt, weights, lse, partials, planes, plane_px,
                       total, (float)step, diversity);
    hipLaunchKernelGGL(sce_finalize_kernel, dim3(1), dim3(256), 0, s, partials, blocks, weights ? 1 : 0, loss, stats);
    return check_launch("subpixel_cross_entropy_fwd");
}

int launch_sce_bwd(const float* sim, const float* gt, const float* weights, const float* lse, const float* stats,
                   const float* grad_loss, float* gsim, int n, int planes, int h, int w, float diversity, int step,
                   hipStream_t s) {
    const size_t plane_px = (size_t)h * w, total = plane_px * n;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(sce_bwd_kernel, dim3(blocks), dim3(256), 0, s, sim, gt, weights, lse, stats, grad_loss, gsim,
                       planes, plane_px, total, (float)step, diversity);
    return check_launch("subpixel_cross_entropy_bwd");
}

int launch_sce_weights_bwd(const float* sim, const float* gt, const float* lse, const float* stats,
                           const float* grad_loss, float* gweights, int n, int planes, int h, int w, float diversity,
                           int step, hipStream_t s) {
    const size_t plane_px = (size_t)h * w, total = plane_px * n;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(sce_weights_bwd_kernel, dim3(blocks), dim3(256), 0, s, sim, gt, lse, stats, grad_loss, gweights,
                       planes, plane_px, total, (float)step, diversity);
    return check_launch("subpixel_cross_entropy_weights_bwd");
}

}  // namespace pds
