// Probe: how long do the stores of one conv2d_x3 output tile (4 waves x 32 KB = 128 KB per CU: 16 x 32 pixels x 64
// channels of a [64][48][144][240] fp32 tensor) take for different lane -> address mappings, with nothing else running?
//   A  16 bytes per lane, lane & 31 = channel (32 planes 6.6 MB apart per instruction), lane >> 5 = +4 pixels  (round 3a)
//   B  4 bytes per lane, lane & 31 = pixel of a row (one 128-byte line per half wave), lane >> 5 = channel + 4
//   C  16 bytes per lane, 8 consecutive lanes = 32 pixels of a row (128 bytes), lane >> 3 = channel (8 per instruction)
//   hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int C = 64, D = 48, H = 144, W = 240;
constexpr size_t PLANE = (size_t)H * W, CSTRIDE = (size_t)D * PLANE;

template <int MODE>
__global__ __launch_bounds__(256) void stores(float* out, int tiles_per_wg, long long* cycles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < tiles_per_wg; ++it) {
        const int v = blockIdx.x + it * gridDim.x;        // tile id: plane-major like the kernel's queues
        const int d = (v / 72) % D, tile = v % 72, ty = tile / 8, tx = tile % 8;
        const int y0 = ty * 16 + 4 * wave, x0 = tx * 32 > W - 32 ? W - 32 : tx * 32;
        float* base = out + (size_t)d * PLANE;
        if (MODE == 0) {
            const int m32 = lane & 31, kgl = lane >> 5;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        float* p = base + (size_t)(nb * 32 + m32) * CSTRIDE + (size_t)(y0 + mb) * W + x0 + 8 * g + 4 * kgl;
                        *reinterpret_cast<f32x4*>(p) = f32x4{1.f * it, 2.f, 3.f, 4.f};
                    }
        } else if (MODE == 1) {
            const int m32 = lane & 31, kgl = lane >> 5;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const int ch = (r >> 4) * 32 + 8 * ((r >> 2) & 3) + 4 * kgl + (r & 3);
                    base[(size_t)ch * CSTRIDE + (size_t)(y0 + mb) * W + x0 + m32] = 1.f * it;
                }
        } else {
            const int q = lane & 7, cg = lane >> 3;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float* p = base + (size_t)(r * 8 + cg) * CSTRIDE + (size_t)(y0 + mb) * W + x0 + 4 * q;
                    *reinterpret_cast<f32x4*>(p) = f32x4{1.f * it, 2.f, 3.f, 4.f};
                }
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = __builtin_readcyclecounter() - t0;
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, C * CSTRIDE * 4);
    hipMalloc(&cyc, 8);
    const int tiles = 13;   // 256 x 13 = 3328 of the 3456 tiles
    for (int wgs : {256, 64, 8})
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(stores<0>, dim3(wgs), dim3(256), 0, 0, out, tiles, cyc);
            if (mode == 1) hipLaunchKernelGGL(stores<1>, dim3(wgs), dim3(256), 0, 0, out, tiles, cyc);
            if (mode == 2) hipLaunchKernelGGL(stores<2>, dim3(wgs), dim3(256), 0, 0, out, tiles, cyc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%3d CUs, pattern %c: %.3f ms for %d tiles per CU (%.1f MB, %.2f TB/s), issue loop %lld cycles = %lld per tile\n",
                   wgs, 'A' + mode, ms, tiles, 1.0 * wgs * tiles * 131072 / 1e6, 1.0 * wgs * tiles * 131072 / ms / 1e9, c, c / tiles);
        }
    return 0;
}
