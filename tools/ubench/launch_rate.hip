// Microbenchmarks behind DESIGN.md's kernel-structure choices (gfx950):
//  (1) workgroup launch throughput as a function of dynamic LDS per workgroup and VGPR allocation: an (almost) empty
//      kernel over many workgroups;
//  (2) fp32 VALU throughput: v_fma_f32 with an SGPR operand vs v_pk_fma_f32.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/launch_rate tools/ubench/launch_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int VG>
__global__ __launch_bounds__(256) void empty_kernel(float* out, int spin) {
    extern __shared__ float lds[];
    if (VG >= 128) asm volatile("v_mov_b32 v120, 0" ::: "v120");
    if (VG >= 200) asm volatile("v_mov_b32 v190, 0" ::: "v190");
    lds[threadIdx.x] = (float)blockIdx.x;
    __syncthreads();
    float v = lds[(threadIdx.x + 1) & 255];
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 1.f;
    if (v == 12345.678f) out[0] = v;
}

__global__ __launch_bounds__(256) void fma_kernel(float* out, const float* w, int iters) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i);
    const float s0 = w[0], s1 = w[1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], s0, s1);
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    if (r == 12345.678f) out[0] = r;
}

typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void pkfma_kernel(float* out, const float* w, int iters) {
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)(threadIdx.x + i), (float)i};
    const f2 s0 = f2{w[0], w[0]}, s1 = f2{w[1], w[1]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], s0, s1);
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    if (r == 12345.678f) out[0] = r;
}

template <class F>
float time_ms(F f, int reps = 10) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        f();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    float* out;
    hipMalloc(&out, 1024);
    float* w;
    hipMalloc(&w, 1024);
    hipMemset(w, 0, 1024);
    const int grids[] = {512, 2592, 10368};
    const int ldss[] = {1024, 16384, 40960, 76800, 153600};
    hipFuncSetAttribute((const void*)empty_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)empty_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)empty_kernel<200>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int spin : {0, 2000}) {
        for (int g : grids)
            for (int l : ldss) {
                float t0 = time_ms([&] { hipLaunchKernelGGL(empty_kernel<64>, dim3(g), dim3(256), l, 0, out, spin); });
                float t1 = time_ms([&] { hipLaunchKernelGGL(empty_kernel<128>, dim3(g), dim3(256), l, 0, out, spin); });
                float t2 = time_ms([&] { hipLaunchKernelGGL(empty_kernel<200>, dim3(g), dim3(256), l, 0, out, spin); });
                printf("spin %5d grid %6d lds %6d B: vgpr<=64 %8.1f us | ~128 %8.1f us | ~200 %8.1f us\n", spin, g, l,
                       t0 * 1e3, t1 * 1e3, t2 * 1e3);
            }
    }
    const int iters = 4096;
    for (int blocks : {1024, 2048, 4096}) {
        float t = time_ms([&] { hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(256), 0, 0, out, w, iters); });
        double fl = 2.0 * 16 * iters * 256.0 * blocks;
        float tp = time_ms([&] { hipLaunchKernelGGL(pkfma_kernel, dim3(blocks), dim3(256), 0, 0, out, w, iters); });
        printf("blocks %d: v_fma_f32 %.1f TFLOP/s, v_pk_fma_f32 %.1f TFLOP/s\n", blocks, fl / (t * 1e-3) / 1e12,
               fl / (tp * 1e-3) / 1e12);
    }
    return 0;
}
