// follow-up of valu_rate.hip: what does a select cost?  (v_cndmask_b32 measured 23 cycles there)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float thr) {
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = a[i] * 0.5f; }
    const float w = 1.0001f;
    unsigned long long m = __ballot(threadIdx.x & 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(w));
                if (MODE == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "s"(m));
                if (MODE == 2) asm volatile("v_cndmask_b32_e64 %0, %2, %1, %3" : "=v"(a[i]) : "v"(w), "v"(b[i]), "s"(m));
                if (MODE == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
                if (MODE == 4) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(w) : "vcc");
                if (MODE == 5) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(w) : "vcc");
                if (MODE == 6) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(w) : "vcc");
                if (MODE == 7) { a[i] = (b[i] > thr) ? a[i] * w : a[i]; }
                if (MODE == 8) asm volatile("v_cmp_gt_f32 %2, %0, %1" : : "v"(a[i]), "v"(w), "s"(m));
                if (MODE == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
                if (MODE == 10) { unsigned long long t; asm volatile("v_cmp_gt_f32_e64 %2, %0, %3\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %3, %2\n\tv_cndmask_b32_e64 %1, %1, %3, %2" : "+v"(a[i]), "+v"(b[i]), "=&s"(t) : "v"(w)); }
                if (MODE == 11) { unsigned long long t; asm volatile("v_cmp_gt_f32_e64 %2, %0, %3\n\ts_and_b64 %2, %2, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %3, %2\n\tv_cndmask_b32_e64 %1, %1, %3, %2\n\tv_cndmask_b32_e64 %0, %0, %1, %2\n\tv_cndmask_b32_e64 %1, %1, %0, %2" : "+v"(a[i]), "+v"(b[i]), "=&s"(t) : "v"(w), "s"(m)); }
                if (MODE == 12) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %1, %1, %0, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(w) : "vcc");
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + b[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char* name, float* out, int wgs_per_cu, int per_iter) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE><<<256 * wgs_per_cu, 256>>>(out, 10, 0.5f);
    hipEventRecord(e0);
    rate_kernel<MODE><<<256 * wgs_per_cu, 256>>>(out, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)iters * 32 * wgs_per_cu;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-44s %d waves/SIMD: %.3f ms, %.2f cycles per group (%d instr) per SIMD at 2.4 GHz\n", name, wgs_per_cu, ms,
           cycles / groups, per_iter);
}

int main() {
    float* out; (void)hipMalloc(&out, 4);
    for (int w : {1, 4}) {
        run<9>("v_add_f32", out, w, 1);
        run<0>("v_cndmask_b32 (vcc, e32)", out, w, 1);
        run<1>("v_cndmask_b32_e64 (sgpr mask, dst=src)", out, w, 1);
        run<2>("v_cndmask_b32_e64 (sgpr mask, dst!=src)", out, w, 1);
        run<3>("v_max_f32", out, w, 1);
        run<4>("v_cmp_gt_f32 vcc", out, w, 1);
        run<8>("v_cmp_gt_f32 sgpr", out, w, 1);
        run<5>("v_cmp + v_cndmask", out, w, 2);
        run<6>("v_cmp + 2 v_cndmask", out, w, 3);
        run<7>("compiler select", out, w, 3);
        run<10>("v_cmp_e64 s + nop + 2 v_cndmask_e64 s", out, w, 3);
        run<11>("v_cmp_e64 s + s_and + nop + 4 cndmask_e64 s", out, w, 5);
        run<12>("v_cmp vcc + nop + 4 v_cndmask vcc", out, w, 5);
    }
    return 0;
}
