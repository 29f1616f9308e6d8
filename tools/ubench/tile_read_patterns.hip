// Probe: HBM read rate of the tile-staging access patterns of conv2d_t8 / conv2d_x3 over two [64][48][144][240] fp32
// tensors (850 MB), nothing else running.  512 persistent workgroups of 256 threads, 4 channels x 2 tensors per step.
//   A  16 x 32-pixel tiles with their 18 x 34 halo, one dword per lane (136-byte row segments, rows 960 bytes apart)
//   B  8-row x 240-column tiles (10 x 240 halo: one contiguous 9.6 KB block per channel), 16 bytes per lane
//   C  like A with 16-byte loads of the aligned 40-column span that covers the 34 columns
//   hipcc --offload-arch=gfx950 -O3 -o tile_read_patterns tile_read_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int C = 64, D = 48, H = 144, W = 240;
constexpr size_t PLANE = (size_t)H * W, CSTRIDE = (size_t)D * PLANE;

template <int MODE>
__global__ __launch_bounds__(256) void reads(const float* a, const float* b, float* out) {
    const int tid = threadIdx.x;
    float acc = 0.f;
    if (MODE == 0 || MODE == 2) {
        const int tiles = D * 9 * 8;
        for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
            const int d = t / 72, r = t % 72, y0 = (r / 8) * 16, x0 = (r % 8) * 32;
            for (int c0 = 0; c0 < C; c0 += 4) {
                if (MODE == 0) {
                    for (int k = 0; k < 3; ++k) {
                        const int p = min(tid + 256 * k, 611), yy = min(max(y0 + p / 34 - 1, 0), H - 1),
                                  xx = min(max(x0 + p % 34 - 1, 0), W - 1);
                        for (int ch = 0; ch < 4; ++ch) {
                            const size_t o = (size_t)(c0 + ch) * CSTRIDE + (size_t)d * PLANE + (size_t)yy * W + xx;
                            acc += a[o] + b[o];
                        }
                    }
                } else {
                    // 18 rows x 10 aligned quads = 180 items per channel, 720 per chunk: 3 per thread
                    for (int k = 0; k < 3; ++k) {
                        const int it = min(tid + 256 * k, 719), ch = it / 180, rem = it % 180;
                        const int yy = min(max(y0 + rem / 10 - 1, 0), H - 1), xq = min(max(x0 / 4 - 1 + rem % 10, 0), W / 4 - 1);
                        const size_t o = (size_t)(c0 + ch) * CSTRIDE + (size_t)d * PLANE + (size_t)yy * W + 4 * xq;
                        const f32x4 va = *reinterpret_cast<const f32x4*>(a + o), vb = *reinterpret_cast<const f32x4*>(b + o);
                        acc += va[0] + va[3] + vb[1] + vb[2];
                    }
                }
            }
        }
    } else {
        const int tiles = D * 18;
        for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
            const int d = t / 18, y0 = (t % 18) * 8;
            for (int c0 = 0; c0 < C; c0 += 4) {
                // 10 rows x 60 quads = 600 items per channel, 2400 per chunk: 10 per thread (9.4)
                for (int k = 0; k < 10; ++k) {
                    const int it = min(tid + 256 * k, 2399), ch = it / 600, rem = it % 600;
                    const int yy = min(max(y0 + rem / 60 - 1, 0), H - 1);
                    const size_t o = (size_t)(c0 + ch) * CSTRIDE + (size_t)d * PLANE + (size_t)yy * W + 4 * (rem % 60);
                    const f32x4 va = *reinterpret_cast<const f32x4*>(a + o), vb = *reinterpret_cast<const f32x4*>(b + o);
                    acc += va[0] + va[3] + vb[1] + vb[2];
                }
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    float *a, *b, *o;
    hipMalloc(&a, C * CSTRIDE * 4);
    hipMalloc(&b, C * CSTRIDE * 4);
    hipMalloc(&o, 64);
    hipMemset(a, 0, C * CSTRIDE * 4);
    hipMemset(b, 0, C * CSTRIDE * 4);
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(reads<0>, dim3(512), dim3(256), 0, 0, a, b, o);
            if (mode == 1) hipLaunchKernelGGL(reads<1>, dim3(512), dim3(256), 0, 0, a, b, o);
            if (mode == 2) hipLaunchKernelGGL(reads<2>, dim3(512), dim3(256), 0, 0, a, b, o);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("pattern %c: %.3f ms  (%.2f TB/s of the 850 MB)\n", 'A' + mode, ms, 2.0 * C * CSTRIDE * 4 / ms / 1e9);
        }
    return 0;
}
