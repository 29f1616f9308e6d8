// Issue rate of v_fma_f32 / v_pk_fma_f32 / v_cndmask / SALU on gfx950, W waves per SIMD (tools/ubench/valu_rate.hip):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o build/ubench/valu_rate && build/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    float a[8];
    f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 1.f}; }
    const float w = 1.0001f;
    const f32x2 w2 = {1.0001f, 0.9999f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(a[i]) : "v"(w));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(p[i]) : "v"(w2));
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(p[i]) : "s"(w2));
                if (MODE == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(w));
                if (MODE == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(w));
                if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel:[0,1,0]" : "+v"(p[i]) : "s"(w2));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char* name, float* out, int wgs_per_cu) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE><<<256 * wgs_per_cu, 256>>>(out, 10);
    hipEventRecord(e0);
    rate_kernel<MODE><<<256 * wgs_per_cu, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 32 * wgs_per_cu;   // one wave of every workgroup per SIMD
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s %d waves/SIMD: %.3f ms, %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, wgs_per_cu, ms,
           cycles / insts_per_simd);
}

int main() {
    float* out; hipMalloc(&out, 4);
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", out, w);
        run<1>("v_pk_fma_f32 (vgpr w)", out, w);
        run<2>("v_pk_fma_f32 (sgpr w)", out, w);
        run<5>("v_pk_fma_f32 (sgpr, op_sel)", out, w);
        run<3>("v_cndmask_b32", out, w);
        run<4>("v_mov_b32", out, w);
    }
    return 0;
}
