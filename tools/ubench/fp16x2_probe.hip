// Probe: a TWO-way fp16 split of both fp32 operands (11 + 11 significand bits), three of the four partial products
// (hi*hi + hi*lo + lo*hi) on v_mfma_f32_32x32x16_f16 with fp32 accumulation, against the fp32 fmaf chain and the
// six-product bf16 form of bf16x3_probe.hip.  fp16 has a 5-bit exponent: the low parts of small operands fall into the
// subnormal range (spacing 6e-8), so the probe also runs scaled operands (power-of-two scale, exact) and small inputs.
//   hipcc --offload-arch=gfx950 -O3 -o fp16x2_probe fp16x2_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// products: 3 = hh + hl + lh, 4 = all
template <int PRODUCTS>
__global__ void emulated(const float* A, const float* B, float* D, int K, float sa, float sb) {
    const int lane = threadIdx.x, m = lane & 31, kg = lane >> 5;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 a[2], b[2];
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + kg * 8 + i;
            const float va = A[m * K + k] * sa, vb = B[k * 32 + m] * sb;
            a[0][i] = (_Float16)va;
            a[1][i] = (_Float16)(va - (float)a[0][i]);
            b[0][i] = (_Float16)vb;
            b[1][i] = (_Float16)(vb - (float)b[0][i]);
        }
        if (PRODUCTS == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
    }
    const float unscale = 1.f / (sa * sb);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
        D[row * 32 + m] = acc[r] * unscale;
    }
}

__global__ void exact(const float* A, const float* B, float* D, int K) {   // fmaf chain == fp32 MFMA
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const int row = t / 32, col = t % 32;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(A[row * K + k], B[k * 32 + col], acc);
    D[row * 32 + col] = acc;
}

// issue-rate probe: REPS x (3 products x 4 accumulator tiles) per wave
__global__ __launch_bounds__(512) void rate(float* out, int reps) {
    f16x8 a[2][2], b[2][2];
    for (int p = 0; p < 2; ++p)
        for (int j = 0; j < 2; ++j)
            for (int i = 0; i < 8; ++i) {
                a[p][j][i] = (_Float16)(1.f + 0.001f * (threadIdx.x + p + j + i));
                b[p][j][i] = (_Float16)(0.5f + 0.002f * (threadIdx.x * 3 + p + j + i));
            }
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n)
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c == 2][m], b[c == 1][n], acc[m][n], 0, 0, 0);
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n)
            for (int i = 0; i < 16; ++i) s += acc[m][n][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int K = 576;
    float *dA, *dB, *dD;
    hipMalloc(&dA, 32 * K * 4);
    hipMalloc(&dB, K * 32 * 4);
    hipMalloc(&dD, 32 * 32 * 4);
    for (double xscale : {1.0, 0.01, 30.0}) {
        std::mt19937 gen(1);
        std::normal_distribution<float> nw(0.f, 1.f / 24.f), nx(0.f, (float)xscale);
        std::vector<float> A(32 * K), B(K * 32);
        for (auto& v : A) v = nw(gen);
        for (auto& v : B) v = nx(gen);
        std::vector<double> ref(32 * 32, 0.0), mag(32 * 32, 0.0);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j)
                for (int k = 0; k < K; ++k) {
                    ref[i * 32 + j] += (double)A[i * K + k] * (double)B[k * 32 + j];
                    mag[i * 32 + j] += std::fabs((double)A[i * K + k] * (double)B[k * 32 + j]);
                }
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> D(32 * 32);
        printf("-- activations ~ N(0, %g), weights ~ N(0, 1/24), K = %d\n", xscale, K);
        auto report = [&](const char* name) {
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            double mx = 0, mean = 0, rel = 0;
            for (int i = 0; i < 32 * 32; ++i) {
                const double e = std::fabs((double)D[i] - ref[i]);
                mx = e > mx ? e : mx;
                mean += e;
                rel += e / mag[i];
            }
            printf("%-52s max abs %.3e  mean abs %.3e  mean err / sum|ab| %.3e\n", name, mx, mean / 1024, rel / 1024);
        };
        hipLaunchKernelGGL(exact, dim3(16), dim3(64), 0, 0, dA, dB, dD, K);
        report("fp32 fmaf chain (== v_mfma_f32_16x16x4_f32)");
        hipLaunchKernelGGL(emulated<3>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, 1.f, 1.f);
        report("fp16 x2, 3 products, unscaled");
        hipLaunchKernelGGL(emulated<3>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, 1024.f, 1.f);
        report("fp16 x2, 3 products, weights x 2^10");
        hipLaunchKernelGGL(emulated<3>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, 1024.f, 16.f);
        report("fp16 x2, 3 products, weights x 2^10, act x 2^4");
        hipLaunchKernelGGL(emulated<4>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, 1024.f, 16.f);
        report("fp16 x2, 4 products, weights x 2^10, act x 2^4");
    }
    float* dO;
    hipMalloc(&dO, 256 * 512 * 4);
    for (int waves : {4, 8}) {
        const int reps = 4000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(rate, dim3(256), dim3(64 * waves), 0, 0, dO, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate, dim3(256), dim3(64 * waves), 0, 0, dO, reps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = (double)reps * 12 * waves * 256;
        const double flops = mfmas * 2.0 * 32 * 32 * 16;
        printf("rate: %d waves/CU x 256 CUs: %.3f ms, %.1f TF f16 (%.1f TF fp32-equivalent at 3 products), %.1f cycles/MFMA/SIMD at 2.4 GHz\n",
               waves, ms, flops / ms / 1e9, flops / 3 / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)reps * 12 * waves / 4));
    }
    return 0;
}
