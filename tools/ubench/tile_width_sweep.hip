// Probe: read rate of halo-tile staging as a function of the tile width (rows of TW + 2 floats, TH + 2 rows, 16 channels
// per step, one tensor [64][48][144][240]) -- the conv2d_x3 staging pattern.  256 persistent workgroups of 256 threads.
//   hipcc --offload-arch=gfx950 -O3 -o tile_width_sweep tile_width_sweep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int C = 64, D = 48, H = 144, W = 240;
constexpr size_t PLANE = (size_t)H * W, CSTRIDE = (size_t)D * PLANE;

__global__ __launch_bounds__(256) void reads(const float* a, float* out, int TH, int TW) {
    const int tid = threadIdx.x;
    const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH, per_plane = tx * ty, tiles = D * per_plane;
    const int hr = TH + 2, hc = TW + 2, npos = hr * hc;
    float acc = 0.f;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int d = t / per_plane, r = t % per_plane, y0 = (r / tx) * TH, x0 = (r % tx) * TW;
        for (int c0 = 0; c0 < C; c0 += 16)
            for (int p = tid; p < npos; p += 256) {
                const int yy = min(max(y0 + p / hc - 1, 0), H - 1), xx = min(max(x0 + p % hc - 1, 0), W - 1);
                const size_t o = (size_t)c0 * CSTRIDE + (size_t)d * PLANE + (size_t)yy * W + xx;
#pragma unroll
                for (int ch = 0; ch < 16; ++ch) acc += a[o + ch * CSTRIDE];
            }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    float *a, *o;
    hipMalloc(&a, C * CSTRIDE * 4);
    hipMalloc(&o, 64);
    hipMemset(a, 0, C * CSTRIDE * 4);
    const int shapes[][2] = {{16, 32}, {8, 64}, {4, 128}, {8, 120}, {4, 240}, {2, 240}, {16, 64}};
    for (auto& s : shapes)
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(reads, dim3(256), dim3(256), 0, 0, a, o, s[0], s[1]);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double halo = (double)(s[0] + 2) * (s[1] + 2) / (s[0] * s[1]);
            printf("tile %2d x %3d (halo x%.2f): %.3f ms  (%.2f TB/s of the 425 MB, %.2f TB/s incl. halo)\n", s[0], s[1], halo, ms,
                   C * CSTRIDE * 4 / ms / 1e9, halo * C * CSTRIDE * 4 / ms / 1e9);
        }
    return 0;
}
