// Issue rate of the 16-bit MFMA forms on gfx950: cycles per instruction for one wave per SIMD issuing back to back into
// four independent accumulators.  hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o build/ubench/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void rate_kernel(float* out, long long* cycles, int iters) {
    f16x4 a4, b4;
    f16x8 a8, b8;
    for (int i = 0; i < 4; ++i) a4[i] = b4[i] = (_Float16)(threadIdx.x * 0.001f + i);
    for (int i = 0; i < 8; ++i) a8[i] = b8[i] = (_Float16)(threadIdx.x * 0.001f + i);
    f32x4 c[NACC] = {};
    f32x16 d[KIND == 2 || KIND == 3 ? NACC : 1] = {};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            if (KIND == 0) c[k] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[k], 0, 0, 0);
            if (KIND == 1) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[k], 0, 0, 0);
            if (KIND == 2) d[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d[k], 0, 0, 0);
            if (KIND == 3) d[k] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, d[k], 0, 0, 0);
            if (KIND == 4) c[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0], b4[0], c[k], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) s += c[k][0] + d[KIND == 2 || KIND == 3 ? k : 0][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND, int NACC>
void run(const char* name, double flops_per, int blocks) {
    const int iters = 20000;
    float* out;
    long long* cyc;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate_kernel<KIND, NACC><<<blocks, 256>>>(out, cyc, 100);
    hipEventRecord(e0);
    rate_kernel<KIND, NACC><<<blocks, 256>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[1];
    hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC;
    printf("%-26s %2d accumulators, %d waves/SIMD: %7.2f counter ticks per MFMA per wave; chip: %8.1f TFLOP/s\n", name, NACC,
           blocks / 256, h[0] / n, flops_per * n * 4 * blocks / (ms * 1e-3) / 1e12);
}

int main() {
    for (int blocks : {256, 512}) {
        run<0, 4>("v_mfma_f32_16x16x16_f16", 2.0 * 16 * 16 * 16, blocks);
        run<0, 12>("v_mfma_f32_16x16x16_f16", 2.0 * 16 * 16 * 16, blocks);
        run<1, 4>("v_mfma_f32_16x16x32_f16", 2.0 * 16 * 16 * 32, blocks);
        run<1, 12>("v_mfma_f32_16x16x32_f16", 2.0 * 16 * 16 * 32, blocks);
        run<2, 4>("v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16, blocks);
        run<3, 4>("v_mfma_f32_32x32x8_f16", 2.0 * 32 * 32 * 8, blocks);
        run<4, 4>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4, blocks);
        run<4, 12>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4, blocks);
    }
    return 0;
}
