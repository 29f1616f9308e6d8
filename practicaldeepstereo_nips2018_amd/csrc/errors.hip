// Evaluation metrics of reference practical_deep_stereo/errors.py:9-74 (called per example from
// pds_trainer.py:48-58) as ONE streaming pass: pixel-wise absolute error, pixel-wise n-pixels error and the
// three sums their averages need, so that evaluation never copies a disparity map to the host.
//
//   known      = ground truth is finite (errors.py:28,59: unknown pixels carry +-inf)
//   abs[p]     = known ? |est - gt| : 0                                   (errors.py:27-30)
//   bad[p]     = known && |est - gt| > n ? 1 : 0                           (errors.py:60-63)
//   stats      = { sum of abs over known, number of known, number of bad }   fp64, two-stage, deterministic
#include "common.hpp"

namespace pds {

namespace {

constexpr int kPerThread = 4;
constexpr int kPerBlock = 256 * kPerThread;

__global__ __launch_bounds__(256) void disparity_errors_kernel(const float* __restrict__ est,
                                                               const float* __restrict__ gt, size_t total, float n,
                                                               float* __restrict__ abs_out, float* __restrict__ bad_out,
                                                               double* __restrict__ partials) {
    double sum = 0.0, known = 0.0, bad = 0.0;
    const size_t base = (size_t)blockIdx.x * kPerBlock + threadIdx.x;
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
        const size_t p = base + (size_t)j * 256;
        if (p < total) {
            const float g = gt[p];
            const bool has = !isinf(g);
            const float diff = fabsf(est[p] - g);
            const float a = has ? diff : 0.f;
            const float b = (has && diff > n) ? 1.f : 0.f;
            if (abs_out) abs_out[p] = a;
            if (bad_out) bad_out[p] = b;
            sum += (double)a;
            known += has ? 1.0 : 0.0;
            bad += (double)b;
        }
    }
    __shared__ double red[4][3];
    sum = wave_sum(sum);
    known = wave_sum(known);
    bad = wave_sum(bad);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave][0] = sum;
        red[wave][1] = known;
        red[wave][2] = bad;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        partials[(size_t)blockIdx.x * 3 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void disparity_errors_finalize_kernel(const double* __restrict__ partials, int blocks,
                                                                        double* __restrict__ stats) {
    double v[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < blocks; i += 256)
        for (int k = 0; k < 3; ++k) v[k] += partials[(size_t)i * 3 + k];
    __shared__ double red[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < 3; ++k) {
        v[k] = wave_sum(v[k]);
        if (lane == 0) red[wave][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) stats[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

}  // namespace

size_t disparity_errors_partial_doubles(size_t total) { return ((total + kPerBlock - 1) / kPerBlock) * 3; }

int launch_disparity_errors(const float* est, const float* gt, size_t total, float n, float* abs_out, float* bad_out,
                            double* stats, double* partials, hipStream_t s) {
    const int blocks = (int)((total + kPerBlock - 1) / kPerBlock);
    hipLaunchKernelGGL(disparity_errors_kernel, dim3(blocks), dim3(256), 0, s, est, gt, total, n, abs_out, bad_out,
                       partials);
    hipLaunchKernelGGL(disparity_errors_finalize_kernel, dim3(1), dim3(256), 0, s, partials, blocks, stats);
    return check_launch("disparity_errors");
}

}  // namespace pds
