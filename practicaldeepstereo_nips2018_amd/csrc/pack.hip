// Weight re-layout into MFMA fragment order for every MFMA kernel, batched: all layers of a module are
// packed by ONE launch (grid.y = job) instead of one tiny launch per layer.
//
//   wpk[chunk][tap][ks][mb][k][i]   (i = lane & 15, k = lane >> 4: exactly the A-fragment of
//                                    v_mfma_f32_16x16x4_f32, so the kernels read weights lane-linearly)
//   conv   (taps 9 or 27): W[oc = mb*16 + i][c = chunk*kc + ks*4 + k][tap]          PyTorch [Cout, Cin, k...]
//   deconv (taps 27 on the input grid): virtual channel v = class*Cout + oc, class = output parity;
//           value = Wt[c][oc][kd][kh][kw] for the transposed-conv tap that parity uses at that input
//           offset, else 0; plus one 27-bit tap mask per 16-channel block (see conv3d_mfma.hip).
//   cell   (mode 5, taps 8): the dense per-cell form of the k4 s2 transposed convolution (conv3d_ks.hip).
//   x3     (mode 6): conv 3x3 weights split three ways into bf16 (round-to-nearest: w = w1 + w2 + w3 exactly) in the
//           A-fragment order of v_mfma_f32_32x32x16_bf16, for conv2d_x3.hip:
//           [k-step of 16 channels][dy][dx][part][32-channel block][lane][8 bf16]; `total` counts dwords (2 bf16).
//   x3 fp16 (mode 7): the same order with TWO fp16 parts of ws * w (round-to-nearest), for v_mfma_f32_32x32x16_f16.
//           ws is the largest power of two with ws * max|w| <= 2^14, derived from the layer's own weights by a
//           one-workgroup-per-job launch ahead of the packing (pack_wscale_kernel: at packing time only), so any
//           finite weights are in range and the low parts of all but vanishing weights stay normal; ws and 1 / ws sit
//           behind the tile-queue counters (dwords 12 and 13 of the 16-dword tail) for the kernel's epilogue.
#include "common.hpp"

namespace pds {

// transposed-conv tap used by output parity `par` at input offset index `o` (0,1,2 <-> -1,0,+1); -1: none
// out = 2*i - 1 + k: even outputs use (i, k=1), (i-1, k=3); odd outputs (i, k=2), (i+1, k=0)
__host__ __device__ inline int tconv_tap(int par, int o) {
    if (o == 1) return par == 0 ? 1 : 2;
    if (par == 0) return o == 0 ? 3 : -1;
    return o == 2 ? 0 : -1;
}

constexpr int kMaxJobs = 24;
struct PackTable {
    PackJob j[kMaxJobs];
};

__device__ __forceinline__ bool deconv_tap_valid(int mode, int cls, int tap, int& kd, int& kh, int& kw) {
    const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
    const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
    kd = mode == 1 ? tconv_tap(pd, dz) : 2 - dz;  // (3,.,.) stride 1: id = od + 1 - kd
    kh = tconv_tap(ph, dy);
    kw = tconv_tap(pw, dx);
    return kd >= 0 && kh >= 0 && kw >= 0;
}

__device__ __forceinline__ unsigned bf16_rne_bits(float v) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
// part 0, 1 of the two-way fp16 split of v (as the 16 bits of an fp16), round to nearest
__device__ __forceinline__ unsigned fp16_split_part(float v, int part) {
    _Float16 h = (_Float16)v;
    if (part) h = (_Float16)(v - (float)h);
    return __builtin_bit_cast(unsigned short, h);
}
// part 0, 1, 2 of the three-way bf16 split of v (as the 16 bits of a bf16)
__device__ __forceinline__ unsigned bf16_split_part(float v, int part) {
    unsigned h = bf16_rne_bits(v);
    for (int p = 0; p < part; ++p) {
        v -= __builtin_bit_cast(float, h << 16);
        h = bf16_rne_bits(v);
    }
    return h;
}

__global__ __launch_bounds__(256) void multi_pack_kernel(const PackTable T) {
    const PackJob J = T.j[blockIdx.y];
    const int ks_n = J.kc / 4;
    // mode 7: the weight scale of the job, left in the tail by pack_wscale_kernel (launched ahead of this kernel)
    const float wscale = J.mode >= 7 ? J.dst[J.total - 16 + 12] : 1.f;
    const int ncls = J.mode == 1 ? 8 : (J.mode == 2 ? 4 : 1);  // modes 0 and 3: plain convolutions
    const int kdn = J.mode == 1 ? 4 : 3;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < J.total; e += gridDim.x * 256) {
        int r = e;
        if (J.mode == 8 || J.mode == 9) {
            // conv3d_ks.hip, X form: two-way fp16 split of ws * w in the A-fragment order of v_mfma_f32_16x16x16_f16,
            // [ic / 4][K-step][block][part][64 lanes][2 dwords]; lane = (k group q) * 16 + row, k = 4 q + channel.
            //   mode 8 (3x3x3 convolution): K-step = (dz, dy), k group = dx (the fourth group is zero)
            //   mode 9 (k4 s2 transposed convolution, cell form of mode 5): K-step = zi, k group = (yi, xi)
            if (e >= J.total - 16) continue;   // (the tail: ws, 1 / ws from pack_wscale_kernel)
            const int i2 = r % 2;
            r /= 2;
            const int ln = r % 64;
            r /= 64;
            const int part = r % 2;
            r /= 2;
            const int mb = r % J.mblocks;
            r /= J.mblocks;
            const int tsteps = J.mode == 8 ? 9 : 2;
            const int t = r % tsteps;
            const int g = r / tsteps;
            const int v = mb * 16 + (ln & 15), q = ln >> 4;
            unsigned h2[2] = {0u, 0u};
            for (int k = 0; k < 2; ++k) {
                const int c = 4 * g + 2 * i2 + k;
                if (c >= J.cin) continue;
                float val = 0.f;
                if (J.mode == 8) {
                    if (v < J.cout && q < 3) val = J.src[((size_t)v * J.cin + c) * 27 + t * 3 + q];
                } else if (v < 8 * J.cout) {
                    const int cls = v / J.cout, oc = v % J.cout;
                    const int kz = 2 + ((cls >> 2) & 1) - 2 * t, ky = 2 + ((cls >> 1) & 1) - 2 * (q >> 1),
                              kx = 2 + (cls & 1) - 2 * (q & 1);
                    val = J.src[(((size_t)c * J.cout + oc) * 4 + kz) * 16 + ky * 4 + kx];
                }
                h2[k] = fp16_split_part(val * wscale, part);
            }
            reinterpret_cast<unsigned*>(J.dst)[e] = h2[0] | (h2[1] << 16);
            continue;
        }
        if (J.mode == 6 || J.mode == 7) {
            if (e >= J.total - 16) {   // the tile-queue counters of conv2d_x3 sit behind its weights: zeroed with them
                if (J.mode == 6 || e < J.total - 16 + 12) reinterpret_cast<unsigned*>(J.dst)[e] = 0u;   // (12, 13: ws, 1 / ws)
                continue;
            }
            const int nparts = J.mode == 6 ? 3 : 2;
            const int i2 = r % 4;
            r /= 4;
            const int ln = r % 64;
            r /= 64;
            const int mb = r % J.mblocks;
            r /= J.mblocks;
            const int part = r % nparts;
            r /= nparts;
            const int dx = r % 3;
            r /= 3;
            const int dy = r % 3;
            const int kstep = r / 3;
            const int oc = mb * 32 + (ln & 31), ic = kstep * 16 + (ln >> 5) * 8 + 2 * i2;
            unsigned lo = 0, hi = 0;
            if (oc < J.cout && ic < J.cin) {
                const float v = J.src[((size_t)oc * J.cin + ic) * 9 + dy * 3 + dx];
                lo = J.mode == 6 ? bf16_split_part(v, part) : fp16_split_part(v * wscale, part);
            }
            if (oc < J.cout && ic + 1 < J.cin) {
                const float v = J.src[((size_t)oc * J.cin + ic + 1) * 9 + dy * 3 + dx];
                hi = J.mode == 6 ? bf16_split_part(v, part) : fp16_split_part(v * wscale, part);
            }
            reinterpret_cast<unsigned*>(J.dst)[e] = lo | (hi << 16);
            continue;
        }
        const int i = r % 16;
        r /= 16;
        const int k = r % 4;
        r /= 4;
        const int mb = r % J.mblocks;
        r /= J.mblocks;
        const int ks = r % ks_n;
        r /= ks_n;
        const int tap = r % J.taps;
        const int chunk = r / J.taps;
        const int v = mb * 16 + i, c = chunk * J.kc + ks * 4 + k;
        float val = 0.f;
        if (J.mode == 0) {
            if (v < J.cout && c < J.cin) val = J.src[((size_t)v * J.cin + c) * J.taps + tap];
        } else if (J.mode == 3) {
            // F(2,3) filter transform along x of a 3x3 kernel (conv2d_wino.hip): tap = dy * 4 + position
            if (v < J.cout && c < J.cin) {
                const float* g = J.src + ((size_t)v * J.cin + c) * 9 + (tap >> 2) * 3;
                const double g0 = g[0], g1 = g[1], g2 = g[2];
                const int pos = tap & 3;
                val = pos == 0 ? g[0] : pos == 3 ? g[2] : (float)(pos == 1 ? 0.5 * (g0 + g1 + g2) : 0.5 * (g0 - g1 + g2));
            }
        } else if (J.mode == 5) {
            // dense "cell" form of the k4 s2 p1 transposed convolution (deconv3d_cell.hip, conv3d_ks.hip): 8 taps =
            // input corners (zi, yi, xi); virtual channel v = class * Cout + oc, class = (pz, py, px);
            // weight = Wt[c][oc][2 + pz - 2 zi][2 + py - 2 yi][2 + px - 2 xi]
            if (v < 8 * J.cout && c < J.cin) {
                const int cls = v / J.cout, oc = v % J.cout;
                const int kz = 2 + ((cls >> 2) & 1) - 2 * ((tap >> 2) & 1), ky = 2 + ((cls >> 1) & 1) - 2 * ((tap >> 1) & 1),
                          kx = 2 + (cls & 1) - 2 * (tap & 1);
                val = J.src[(((size_t)c * J.cout + oc) * 4 + kz) * 16 + ky * 4 + kx];
            }
        } else if (v < ncls * J.cout && c < J.cin) {
            const int cls = v / J.cout, oc = v % J.cout;
            int kd, kh, kw;
            if (deconv_tap_valid(J.mode, cls, tap, kd, kh, kw))
                val = J.src[(((size_t)c * J.cout + oc) * kdn + kd) * 16 + kh * 4 + kw];
        }
        J.dst[e] = val;
    }
    // tap masks of the transposed convolutions: one 27-bit word per 16-channel block
    if ((J.mode == 1 || J.mode == 2) && blockIdx.x == 0) {
        for (int mb = threadIdx.x; mb < J.mblocks; mb += 256) {
            unsigned m = 0;
            for (int i = 0; i < 16; ++i) {
                const int v = mb * 16 + i;
                if (v >= ncls * J.cout) break;
                for (int tap = 0; tap < 27; ++tap) {
                    int kd, kh, kw;
                    if (deconv_tap_valid(J.mode, v / J.cout, tap, kd, kh, kw)) m |= 1u << tap;
                }
            }
            J.mask[mb] = m;
        }
    }
}

// ws and 1 / ws of every fp16-split job (modes 7, 8, 9) of the table (one workgroup per job; the others leave at once)
__global__ __launch_bounds__(1024) void pack_wscale_kernel(const PackTable T) {
    const PackJob J = T.j[blockIdx.x];
    if (J.mode < 7) return;
    __shared__ float red[16];
    float m = 0.f;
    const int nw = J.cout * J.cin * (J.mode == 7 ? 9 : (J.mode == 8 ? 27 : 64));
    // (a tensor from PyTorch's allocator is 16-byte aligned; the tail covers counts that are not a multiple of 4)
    const float4* src4 = reinterpret_cast<const float4*>(J.src);
    const bool vec = (reinterpret_cast<uintptr_t>(J.src) & 15) == 0;
    const int n4 = vec ? nw / 4 : 0;
#pragma unroll 4
    for (int i = threadIdx.x; i < n4; i += 1024) {
        const float4 v = src4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (int i = 4 * n4 + threadIdx.x; i < nw; i += 1024) m = fmaxf(m, fabsf(J.src[i]));
    m = block_max(m, red);
    if (threadIdx.x == 0) {
        const float ws = pow2_scale(m, kHalfTarget);
        J.dst[J.total - 16 + 12] = ws;
        J.dst[J.total - 16 + 13] = 1.f / ws;
    }
}

int launch_multi_pack(const PackJob* jobs, int count, hipStream_t s) {
    for (int first = 0; first < count; first += kMaxJobs) {
        PackTable T;
        const int n = count - first < kMaxJobs ? count - first : kMaxJobs;
        int biggest = 0;
        for (int i = 0; i < n; ++i) {
            T.j[i] = jobs[first + i];
            if (T.j[i].total > biggest) biggest = T.j[i].total;
        }
        for (int i = n; i < kMaxJobs; ++i) T.j[i] = PackJob{};
        int bx = (biggest + 255) / 256;
        if (bx > 128) bx = 128;
        if (bx < 1) bx = 1;
        bool any_fp16 = false;
        for (int i = 0; i < n; ++i) any_fp16 |= T.j[i].mode >= 7;
        if (any_fp16) {
            hipLaunchKernelGGL(pack_wscale_kernel, dim3(n), dim3(1024), 0, s, T);
            if (int rc = check_launch("pack_wscale")) return rc;
        }
        hipLaunchKernelGGL(multi_pack_kernel, dim3(bx, n), dim3(256), 0, s, T);
        if (int rc = check_launch("multi_pack")) return rc;
    }
    return 0;
}

}  // namespace pds
