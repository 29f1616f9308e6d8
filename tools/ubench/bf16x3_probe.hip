// Probe: is a 3-way bf16 split of BOTH fp32 operands, 6 of the 9 partial products on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation, as accurate as the exact-fp32 MFMA chain (v_mfma_f32_16x16x4_f32 == fmaf chain)?  And how fast
// does one wave issue the six products?  Stand-alone: hipcc --offload-arch=gfx950 -O3 -o bf16x3_probe bf16x3_probe.hip
//
//   D[32 x 32] = A[32 x K] * B[K x 32], K = 576 (one 3x3 x 64-channel output of Matching's 64 -> 64 layers),
//   A ~ N(0, 1/24) (weights), B ~ N(0, 1) (normalised activations); reference in fp64 on the host.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned short bf16_rne(float v) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// mode 0: RNE split, 6 products; 1: truncation split, 6 products; 2: RNE split, 3 products (2 parts each);
// 3: RNE, all 9 products
template <int MODE>
__global__ void emulated(const float* A, const float* B, float* D, int K) {
    const int lane = threadIdx.x, m = lane & 31, kg = lane >> 5;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 a[3], b[3];
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + kg * 8 + i;
            float va = A[m * K + k], vb = B[k * 32 + m];
            for (int s = 0; s < 2; ++s) {
                float v = s ? vb : va;
                unsigned short h[3];
                for (int p = 0; p < 3; ++p) {
                    h[p] = MODE == 1 ? (unsigned short)(__builtin_bit_cast(unsigned, v) >> 16) : bf16_rne(v);
                    v -= bf16_f32(h[p]);
                }
                for (int p = 0; p < 3; ++p) (s ? b : a)[p][i] = (short)h[p];
            }
        }
        // small terms first
        if (MODE == 3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], acc, 0, 0, 0);
        }
        if (MODE != 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
        D[row * 32 + m] = acc[r];
    }
}

__global__ void exact(const float* A, const float* B, float* D, int K) {   // fmaf chain == fp32 MFMA
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const int row = t / 32, col = t % 32;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(A[row * K + k], B[k * 32 + col], acc);
    D[row * 32 + col] = acc;
}

// issue-rate probe: REPS x (6 products x 4 accumulator tiles) per wave, WAVES waves per workgroup, one workgroup per CU
__global__ __launch_bounds__(512) void rate(float* out, int reps) {
    bf16x8 a[3][2], b[3][2];
    for (int p = 0; p < 3; ++p)
        for (int j = 0; j < 2; ++j)
            for (int i = 0; i < 8; ++i) {
                a[p][j][i] = (short)(0x3f80 + threadIdx.x + p + j + i);
                b[p][j][i] = (short)(0x3f00 + threadIdx.x * 3 + p + j + i);
            }
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n)
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int pa = 0; pa < 3; ++pa)
#pragma unroll
            for (int pb = 0; pb < 3 - pa; ++pb)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][m], b[pb][n], acc[m][n], 0, 0, 0);
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n)
            for (int i = 0; i < 16; ++i) s += acc[m][n][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int K = 576;
    std::mt19937 gen(1);
    std::normal_distribution<float> nw(0.f, 1.f / 24.f), nx(0.f, 1.f);
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
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4);
    hipMalloc(&dB, B.size() * 4);
    hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> D(32 * 32);
    auto report = [&](const char* name) {
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        double mx = 0, mean = 0, rel = 0;
        for (int i = 0; i < 32 * 32; ++i) {
            const double e = std::fabs((double)D[i] - ref[i]);
            mx = e > mx ? e : mx;
            mean += e;
            rel += e / mag[i];
        }
        printf("%-44s max abs %.3e  mean abs %.3e  mean err / sum|ab| %.3e\n", name, mx, mean / 1024, rel / 1024);
    };
    hipLaunchKernelGGL(exact, dim3(16), dim3(64), 0, 0, dA, dB, dD, K);
    report("fp32 fmaf chain (== v_mfma_f32_16x16x4_f32)");
    hipLaunchKernelGGL(emulated<0>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    report("bf16 x3 split RNE, 6 products");
    hipLaunchKernelGGL(emulated<1>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    report("bf16 x3 split truncation, 6 products");
    hipLaunchKernelGGL(emulated<3>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    report("bf16 x3 split RNE, 9 products");
    hipLaunchKernelGGL(emulated<2>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    report("bf16 x2 split RNE, 3 products");

    // issue rate
    float* dO;
    hipMalloc(&dO, 256 * 512 * 4);
    for (int waves : {4, 8}) {
        const int reps = 2000;
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
        const double mfmas = (double)reps * 24 * waves * 256;
        const double flops = mfmas * 2.0 * 32 * 32 * 16;
        printf("rate: %d waves/CU x 256 CUs: %.3f ms, %.1f TF bf16 (%.1f TF fp32-equivalent at 6 products), %.1f cycles/MFMA/SIMD at 2.4 GHz\n",
               waves, ms, flops / ms / 1e9, flops / 6 / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)reps * 24 * waves / 4));
    }
    return 0;
}
