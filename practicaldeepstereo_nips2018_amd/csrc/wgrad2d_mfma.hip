// Weight gradient of the 2-D 3x3 convolutions of MatchingOperation on the fp32 MFMA units:
//   dW[oc][c][tap] = sum over (n, d, y, x) of dz[oc][p] * xhat[c][p + tap]
// GEMM view: M = output channels (16 per block), N = (tap, input channel) columns (16 channels per block),
// K = positions, walked 4 at a time with v_mfma_f32_16x16x4_f32.
//   work item   one row segment of 32 positions of one (n, d) plane; persistent workgroups stride over the items
//               and keep their partial dW in registers (fp32 over a few thousand positions), then write ONE
//               partial per workgroup; a second kernel sums the partials in fp64.
//   workgroup   4 waves, 64 input channels (grid.y walks further groups of 64).  Cout = 64: wave = one 16-channel
//               output block x all 36 column blocks (144 accumulator registers); Cout <= 16: the 4 waves split the
//               36 column blocks.
//   LDS         xhat tile [64 ch][3 rows][34] (deferred InstanceNorm, skip sum and zero padding applied while
//               staging) and dz tile [64][32]; row strides chosen so that both fragment reads (16 channels x 4
//               consecutive positions per instruction) are bank-conflict free: channel stride == 2 (mod 32).
#include "common.hpp"

namespace pds {

namespace {

constexpr int THREADS = 256;
constexpr int TWG = 32;        // positions per work item
constexpr int RSX = 54;        // xhat row stride: >= TWG + 2 and == 22 (mod 32), so 3 * RSX == 2 (mod 32)
constexpr int XS = 3 * RSX;    // xhat channel stride
constexpr int DS = 34;         // dz row stride, == 2 (mod 32)
constexpr int CG = 64;         // input channels per workgroup
constexpr int NBLK = CG * 9 / 16;
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WArgs {
    Src a, b;
    const float* __restrict__ dz;
    float* __restrict__ partial;  // [workgroup][Cout][Cin][9]
    int N, Cin, D, H, W, Cout;
    int items, segs;
};

}  // namespace

template <int MBW, bool HAS_B, bool PARTIAL = false>
__global__ __launch_bounds__(THREADS, 2) void wgrad2d_mfma_kernel(const WArgs A) {
    constexpr int PARTS = 4 / MBW;          // waves sharing one output block split the column blocks
    constexpr int NB = NBLK / PARTS;        // column blocks per wave
    __shared__ __attribute__((aligned(16))) float xl[CG * XS];
    __shared__ __attribute__((aligned(16))) float dzl[MBW * 16 * DS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb = wave % MBW, part = wave / MBW;
    const int cg0 = blockIdx.y * CG;
    const size_t plane = (size_t)A.H * A.W;

    f32x4 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // a partial channel group (the 12-channel space-to-depth layer of the embedding): the 16-channel blocks beyond Cin are
    // neither staged nor multiplied (their accumulators stay zero and are never written)
    // (a separate instantiation: the checks cost the full groups 134 -> 172 us per single-plane layer)
    const int cblocks = PARTIAL ? min(CG / 16, (A.Cin - cg0 + 15) / 16) : CG / 16;
    // 16-byte staging needs rows that start 16-byte aligned
    const bool vec_ok = (A.W & 3) == 0 && ((reinterpret_cast<uintptr_t>(A.a.p) | reinterpret_cast<uintptr_t>(A.b.p)) & 15) == 0;
    for (int item = blockIdx.x; item < A.items; item += gridDim.x) {
        int r = item;
        const int seg = r % A.segs;
        r /= A.segs;
        const int y = r % A.H;
        r /= A.H;
        const int d = r % A.D;
        const int n = r / A.D;
        const int x0 = seg * TWG;

        // ---- stage xhat: 64 channels x 3 rows x 34 columns (x0-1 .. x0+32) -------------------------------
        // One wave stages one (channel, row) run of 34 contiguous columns per step: the channel, the row and the
        // InstanceNorm coefficients are wave-uniform (scalar registers), the per-lane work is one load, one fma, one
        // LDS write.  (The first version walked a flat element index: two divisions and two coefficient loads per
        // element cost as much issue time as a third of the MFMAs.)
        if (vec_ok) {
            // Rows that are a multiple of 4 wide: a (channel, row) run is 8 aligned 16-byte loads plus its two halo
            // columns, i.e. 10 lanes; a wave covers 6 runs per load instruction and its 48 runs in 8 instructions,
            // all issued before the first one is needed.  (With one 4-byte load per lane and run -- 34 of 64 lanes,
            // 136 bytes per instruction -- the 48 instructions per wave went out in six dependent batches and staging
            // took twice as long as the MFMAs of the item.)
            constexpr int RPI = 6, ITER = 8;          // runs per instruction, instructions per wave (6 * 8 = 48 runs)
            static_assert(CG * 3 == 4 * RPI * ITER, "a wave stages a quarter of the runs");
            const int rs = lane / 10, part = lane - rs * 10;
            const bool lane_on = rs < RPI;
            const int xq = x0 + 4 * min(part, 7);     // interior quad (parts 0..7)
            const int xh = part == 8 ? x0 - 1 : x0 + TWG;   // halo column (parts 8, 9)
            const bool halo = part >= 8;
            const bool colok = halo ? (xh >= 0 && xh < A.W) : xq < A.W;
            const int xcol = halo ? min(max(xh, 0), A.W - 1) : min(xq, A.W - 4);
            constexpr int FLY = MBW == 4 ? 4 : 8;     // instructions in flight (144 accumulator registers leave room for 4)
            for (int it0 = 0; it0 < ITER; it0 += FLY) {
            f32x4 qa[FLY], qb[HAS_B ? FLY : 1];
#pragma unroll
            for (int itl = 0; itl < FLY; ++itl) {
                const int it = it0 + itl;
                const int cr = min(wave * (RPI * ITER) + it * RPI + rs, CG * 3 - 1), c = cr / 3, rr = cr - c * 3;
                const int yc = min(max(y - 1 + rr, 0), A.H - 1);
                const int chc = min(cg0 + c, A.Cin - 1);
                const size_t off = ((size_t)(n * A.Cin + chc) * A.D + d) * plane + (size_t)yc * A.W + xcol;
                if (halo) {
                    qa[itl] = f32x4{A.a.p[off], 0.f, 0.f, 0.f};
                    if (HAS_B) qb[itl] = f32x4{A.b.p[off], 0.f, 0.f, 0.f};
                } else {
                    qa[itl] = *reinterpret_cast<const f32x4*>(A.a.p + off);
                    if (HAS_B) qb[itl] = *reinterpret_cast<const f32x4*>(A.b.p + off);
                }
            }
#pragma unroll
            for (int itl = 0; itl < FLY; ++itl) {
                const int it = it0 + itl;
                const int cr = min(wave * (RPI * ITER) + it * RPI + rs, CG * 3 - 1), c = cr / 3, rr = cr - c * 3;
                if (PARTIAL && c >= 16 * cblocks) continue;
                const int yy = y - 1 + rr;
                const bool chok = cg0 + c < A.Cin;
                const int ch = min(cg0 + c, A.Cin - 1);
                const int g = A.a.per_plane ? ((n * A.Cin + ch) * A.D + d) : (n * A.Cin + ch);
                float sa = 1.f, ha = 0.f, sb2 = 1.f, hb2 = 0.f;
                if (A.a.scale) {
                    sa = A.a.scale[g];
                    ha = A.a.shift[g];
                }
                if (HAS_B && A.b.scale) {
                    const int gb = A.b.per_plane ? ((n * A.Cin + ch) * A.D + d) : (n * A.Cin + ch);
                    sb2 = A.b.scale[gb];
                    hb2 = A.b.shift[gb];
                }
                const bool ok = chok && colok && yy >= 0 && yy < A.H;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = fmaf(sa, qa[itl][j], ha);
                    if (HAS_B) t += fmaf(sb2, qb[itl][j], hb2);
                    v[j] = ok ? t : 0.f;
                }
                float* dst = xl + c * XS + rr * RSX;
                if (lane_on) {
                    if (halo) {
                        dst[part == 8 ? 0 : TWG + 1] = v[0];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) dst[1 + 4 * part + j] = v[j];
                    }
                }
            }
            }
        } else {
            const int xx = min(lane, TWG + 1);        // lanes 0..33 active, the rest shadow lane 33
            const int x = x0 - 1 + xx;
            const bool colok = x >= 0 && x < A.W;
            const int xc = min(max(x, 0), A.W - 1);
            constexpr int BATCH = 8;                  // (channel, row) runs in flight per wave
            static_assert((CG * 3) % (4 * BATCH) == 0, "runs split evenly over waves and batches");
            for (int cr0 = wave * BATCH; cr0 < CG * 3; cr0 += 4 * BATCH) {
                float va[BATCH], vb[BATCH];
#pragma unroll
                for (int j = 0; j < BATCH; ++j) {     // loads first: unconditional, clamped addresses
                    const int cr = cr0 + j, c = cr / 3, rr = cr - c * 3;
                    const int yc = min(max(y - 1 + rr, 0), A.H - 1);
                    const int chc = min(cg0 + c, A.Cin - 1);   // (a last group may hold fewer than 64 channels)
                    const size_t off = ((size_t)(n * A.Cin + chc) * A.D + d) * plane + (size_t)yc * A.W + xc;
                    va[j] = A.a.p[off];
                    vb[j] = A.b.p ? A.b.p[off] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < BATCH; ++j) {
                    const int cr = cr0 + j, c = cr / 3, rr = cr - c * 3;
                    const int yy = y - 1 + rr;
                    const bool chok = cg0 + c < A.Cin;
                    const int ch = min(cg0 + c, A.Cin - 1);
                    const int g = A.a.per_plane ? ((n * A.Cin + ch) * A.D + d) : (n * A.Cin + ch);
                    float v = A.a.scale ? fmaf(A.a.scale[g], va[j], A.a.shift[g]) : va[j];
                    if (A.b.p) {
                        const int gb = A.b.per_plane ? ((n * A.Cin + ch) * A.D + d) : (n * A.Cin + ch);
                        v += A.b.scale ? fmaf(A.b.scale[gb], vb[j], A.b.shift[gb]) : vb[j];
                    }
                    v = (chok && colok && yy >= 0 && yy < A.H) ? v : 0.f;
                    if (lane < TWG + 2) xl[c * XS + rr * RSX + xx] = v;
                }
            }
        }
        // ---- stage dz: (padded) output channels x 32 positions ---------------------------------------------------
        if (vec_ok && (reinterpret_cast<uintptr_t>(A.dz) & 15) == 0) {
            // 16-byte loads: 8 lanes per channel row, two rows of loads per thread for 64 channels
#pragma unroll
            for (int k = 0; k < (MBW * 16 * 8 + THREADS - 1) / THREADS; ++k) {
                const int e = tid + k * THREADS;
                const int oc = e >> 3, q = e & 7;
                if (oc < MBW * 16) {
                    const int x = x0 + 4 * q;
                    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (oc < A.Cout && x < A.W)
                        v = *reinterpret_cast<const f32x4*>(
                            A.dz + ((size_t)(n * A.Cout + oc) * A.D + d) * plane + (size_t)y * A.W + x);
                    float* dst = dzl + oc * DS + 4 * q;
                    *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);        // DS is even: 8-byte aligned
                    *reinterpret_cast<float2*>(dst + 2) = make_float2(v[2], v[3]);
                }
            }
        } else {
            const int px = lane & 31, x = x0 + px;
            for (int o2 = wave; o2 < MBW * 8; o2 += 4) {
                const int oc = o2 * 2 + (lane >> 5);
                float v = 0.f;
                if (oc < A.Cout && x < A.W)
                    v = A.dz[((size_t)(n * A.Cout + oc) * A.D + d) * plane + (size_t)y * A.W + x];
                dzl[oc * DS + px] = v;
            }
        }
        __syncthreads();

        const float* arow = dzl + (mb * 16 + (lane & 15)) * DS + (lane >> 4);
        const float* brow = xl + (lane & 15) * XS + (lane >> 4);
#pragma unroll 2
        for (int ks = 0; ks < TWG / 4; ++ks) {
            const float af = arow[ks * 4];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int nb = part * NB + i;           // column block: tap = nb / 4, channel block = nb % 4
                const int t = nb / (CG / 16), cb = nb % (CG / 16);
                const int dy = t / 3, dx = t % 3;
                if (!PARTIAL || cb < cblocks) {   // workgroup-uniform
                    const float bf = brow[cb * 16 * XS + dy * RSX + ks * 4 + dx];
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- one partial per workgroup: [Cout][Cin][9] -----------------------------------------------------------
    float* dst = A.partial + (size_t)blockIdx.x * A.Cout * A.Cin * 9;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int nb = part * NB + i;
        const int t = nb / (CG / 16), cb = nb % (CG / 16);
        const int c = cg0 + cb * 16 + (lane & 15);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int oc = mb * 16 + 4 * (lane >> 4) + rr;
            if (oc < A.Cout && c < A.Cin) dst[((size_t)oc * A.Cin + c) * 9 + t] = acc[i][rr];
        }
    }
}

// dw[i] = sum over the workgroup partials, in fp64 and in a fixed order: a workgroup owns 64 consecutive weights,
// its four waves each walk a quarter of the partials (four independent loads in flight per lane), LDS combines them.
__global__ __launch_bounds__(256) void wgrad_reduce_f32_kernel(const float* __restrict__ partial, size_t wcount,
                                                               int parts, float* __restrict__ dw, int accumulate) {
    __shared__ double red[4][64];
    const int col = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + col;
    double s = 0.0;
    if (i < wcount) {
        // sixteen independent loads in flight per lane: a 64 -> 64 layer has 512 partials and only 576 workgroups, so the
        // kernel is a chain of load round trips (four in flight: 43 us for 75 MB)
        int k = pg;
        for (; k + 60 < parts; k += 64) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = partial[(size_t)(k + 4 * j) * wcount + i];
#pragma unroll
            for (int j = 0; j < 16; ++j) s += (double)v[j];
        }
        for (; k < parts; k += 4) s += (double)partial[(size_t)k * wcount + i];
    }
    red[pg][col] = s;
    __syncthreads();
    if (pg == 0 && i < wcount) {
        const double total = ((red[0][col] + red[1][col]) + red[2][col]) + red[3][col];
        dw[i] = (accumulate ? dw[i] : 0.f) + (float)total;
    }
}

int launch_wgrad_reduce_f32(const float* partial, size_t wcount, int parts, float* dw, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(wgrad_reduce_f32_kernel, dim3((unsigned)((wcount + 63) / 64)), dim3(256), 0, s, partial, wcount,
                       parts, dw, accumulate);
    return check_launch("wgrad_reduce");
}

bool wgrad2d_mfma_supported(int transposed, int kd, int stride, const Src& b, const Geom& in, const Geom& out) {
    if (transposed || kd != 1 || stride != 1) return false;
    if (in.c % CG != 0 && in.c > CG) return false;   // whole groups of 64 channels, or one partial group
    if (!(out.c == 64 || out.c <= 16)) return false;
    if (b.p && b.bcast_d) return false;
    return true;
}

static int wgrad2d_workgroups(const Geom& in) {
    const size_t items = (size_t)in.n * in.d * in.h * ((in.w + TWG - 1) / TWG);
    size_t wgs = items / 8;  // at least ~8 items per workgroup so the partial write amortises ...
    if (wgs < 256) wgs = items / 2 < 256 ? items / 2 : 256;   // ... but one workgroup per CU first (single-plane layers)
    if (wgs > 512) wgs = 512;
    return wgs < 1 ? 1 : (int)wgs;
}

size_t wgrad2d_mfma_scratch_floats(const Geom& in, const Geom& out) {
    return (size_t)wgrad2d_workgroups(in) * out.c * in.c * 9;
}

int launch_wgrad2d_mfma(const Src& a, const Src& b, const Src& dzs, float* dw, const Geom& in, const Geom& out,
                        int accumulate, float* scratch, hipStream_t s) {
    if (wgrad2d_x3_supported(a, b, dzs, in, out)) {   // fp16-split form on the 16-bit matrix pipe (wgrad2d_x3.hip)
        const int wgs = wgrad2d_workgroups(in);
        if (int rc = launch_wgrad2d_x3(a, b, dzs, scratch, wgs, in, out, s)) return rc;
        return launch_wgrad_reduce_f32(scratch, (size_t)out.c * in.c * 9, wgs, dw, accumulate, s);
    }
    const float* dz = dzs.p;
    WArgs A;
    A.a = a;
    A.b = b;
    A.dz = dz;
    A.partial = scratch;
    A.N = in.n;
    A.Cin = in.c;
    A.D = in.d;
    A.H = in.h;
    A.W = in.w;
    A.Cout = out.c;
    A.segs = (in.w + TWG - 1) / TWG;
    A.items = in.n * in.d * in.h * A.segs;
    const int wgs = wgrad2d_workgroups(in);
    dim3 grid(wgs, (in.c + CG - 1) / CG);
    if (out.c == 64 && in.c < CG && !b.p) {   // one partial channel group (the embedding's 12-channel layer)
        hipLaunchKernelGGL((wgrad2d_mfma_kernel<4, false, true>), grid, dim3(THREADS), 0, s, A);
    } else if (out.c == 64) {
        if (b.p) hipLaunchKernelGGL((wgrad2d_mfma_kernel<4, true>), grid, dim3(THREADS), 0, s, A);
        else hipLaunchKernelGGL((wgrad2d_mfma_kernel<4, false>), grid, dim3(THREADS), 0, s, A);
    } else {
        if (b.p) hipLaunchKernelGGL((wgrad2d_mfma_kernel<1, true>), grid, dim3(THREADS), 0, s, A);
        else hipLaunchKernelGGL((wgrad2d_mfma_kernel<1, false>), grid, dim3(THREADS), 0, s, A);
    }
    if (int rc = check_launch("wgrad2d_mfma")) return rc;
    return launch_wgrad_reduce_f32(scratch, (size_t)out.c * in.c * 9, wgs, dw, accumulate, s);
}

}  // namespace pds
