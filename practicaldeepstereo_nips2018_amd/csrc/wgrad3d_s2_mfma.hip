// Weight gradients of the stride-2 layers of the hourglass on the fp32 MFMA units (reference regularization.py:28-31,
// 54-57, 88-92 and network_blocks.py:37-44, 61-72 under loss.backward(), pds_trainer.py:40-46):
//   convolution k3 s2 p1          dW[oc][c][k]  = sum_o dz[oc][o]  * xhat[c][2o - 1 + k]      (K = 3)
//   transposed convolution k4 s2  dW[c][oc][k]  = sum_i xhat[c][i] * dz[oc][2i - 1 + k]       (K = 4)
// Both are  R[s][b][k] = sum_p S[s][p] * B[b][2p - 1 + k]  per axis, with a "small-grid" tensor S (dz of the strided
// convolution / the input of the transposed one) and a "big-grid" tensor B (the convolution's input / the transposed
// convolution's dz); R is laid out exactly like the PyTorch weight of either layer.
// GEMM view: M = 16 small-grid channels, N = 16 big-grid channels per tap (K^3 column blocks), K = positions, 4 at a time
// with v_mfma_f32_16x16x4_f32.
//   workgroup   one (small block, big block) channel pair (grid.y); the K^3 taps are split over the 4 waves; persistent
//               over work items with the partial R in registers, ONE partial per workgroup (summed by
//               wgrad_reduce_f32_kernel, wgrad2d_mfma.hip).
//   work item   a 32-position x-segment of one (n, z, y) row of the small grid.
//   LDS         S tile [16 ch][32]; B tile [16 ch][K x K rows][odd columns | even columns]: the stride-2 read
//               B[2p - 1 + kx] becomes a unit-stride read of the odd (kx = 0, 2) or even (kx = 1, 3) half, so both
//               fragment reads are bank-conflict free (channel strides == 2 mod 32).  The deferred InstanceNorm, the skip
//               sum (possibly broadcast along D, regularization.py:115) and the zero padding are applied while staging.
#include "common.hpp"

namespace pds {

namespace {

constexpr int THREADS = 256;
constexpr int TWG = 32;           // small-grid positions per work item
constexpr int HS = 34;            // floats per parity half of a big row (33 used)
constexpr int RS = 2 * HS;        // big row stride
constexpr int DS = 34;            // small row stride, == 2 (mod 32)
constexpr int BCOLS = 2 * TWG + 2;  // big columns staged per row: 2 x0 - 1 .. 2 x0 + 64
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K>
struct Cfg {
    static constexpr int ROWS = K * K;
    static constexpr int RAW = ROWS * RS;
    static constexpr int XS = RAW + ((2 - RAW % 32) + 32) % 32;   // channel stride == 2 (mod 32)
    static constexpr int TAPS = K * K * K;
    static constexpr int TPW = (TAPS + 3) / 4;                    // taps per wave
    static_assert(XS % 32 == 2, "bank layout");
};
static_assert(DS % 32 == 2, "bank layout");

struct WS2Args {
    Src a, b;                      // the normalised tensor (the layer's input)
    const float* __restrict__ dz;  // the plain one (gradient of the layer's raw output)
    float* __restrict__ partial;   // [workgroup][Cs][Cb][K^3]
    int N, Cs, Cb;
    int Ds, Hs, Ws;                // small grid
    int Db, Hb, Wb;                // big grid
    int items, segs, sblocks;
};

// value of the layer's (normalised, summed) input at an in-range position
__device__ __forceinline__ float xhat_value(const Src& a, const Src& b, int n, int C, int c, int D, int H, int W, int z,
                                            int y, int x) {
    const size_t plane = (size_t)H * W, inplane = (size_t)y * W + x;
    const int g = n * C + c;
    float sa = 1.f, ha = 0.f;
    if (a.scale) {
        sa = a.scale[g];
        ha = a.shift[g];
    }
    float v = fmaf(sa, a.p[((size_t)g * D + z) * plane + inplane], ha);
    if (b.p) {
        float sb = 1.f, hb = 0.f;
        if (b.scale) {
            sb = b.scale[g];
            hb = b.shift[g];
        }
        v += fmaf(sb, b.p[(b.bcast_d ? (size_t)g : (size_t)g * D + z) * plane + inplane], hb);
    }
    return v;
}

}  // namespace

// SMALL_NORM: the small-grid tensor is the normalised one (transposed convolution); otherwise the big-grid one is.
// NT: taps per column group.  A big-grid tensor of 8 (4) channels would fill 8 (4) of the 16 MFMA columns; instead the
// columns hold 8 channels x 2 taps (4 channels x 4 taps): half (a quarter) of the MFMAs for the full-resolution layers.
template <int K, bool SMALL_NORM, int NT>
__global__ __launch_bounds__(THREADS) void wgrad3d_s2_mfma_kernel(const WS2Args A) {
    using C = Cfg<K>;
    constexpr int CBP = 16 / NT;                         // channels per column group (== Cb when NT > 1)
    constexpr int GROUPS = (C::TAPS + NT - 1) / NT;      // column groups: taps g NT .. g NT + NT - 1
    constexpr int GPW = (GROUPS + 3) / 4;                // groups per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // big tile: [rows][XS] with rows = min(16, Cb) (+ one all-zero row that the lanes beyond Cb read): few-channel layers
    // (8 -> 4 at full half-resolution) need 20 KB instead of 70, so more workgroups fit on a CU
    const int nbm = min(16, A.Cb), brows = nbm < 16 ? nbm + 1 : 16;
    float* bl = smem;                     // [brows][XS]
    float* sl = smem + brows * C::XS;     // [16][DS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sb = blockIdx.y % A.sblocks, bb = blockIdx.y / A.sblocks;
    const int s0 = sb * 16, b0 = bb * 16;
    const size_t plane_s = (size_t)A.Hs * A.Ws, vol_s = (size_t)A.Ds * plane_s;
    const size_t plane_b = (size_t)A.Hb * A.Wb, vol_b = (size_t)A.Db * plane_b;

    // tap -> LDS offset of its B fragment row: row (kz, ky), parity half and shift of kx (wave-uniform for NT == 1; with
    // NT > 1 the lane's column selects one of the group's taps)
    const int tj = (lane & 15) / CBP;
    int toff[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int tap = min((wave + 4 * i) * NT + tj, C::TAPS - 1);
        const int rr = tap / K, kx = tap % K;
        toff[i] = rr * RS + ((kx & 1) ? HS : 0) + ((kx + 1) >> 1) - ((kx & 1) ? 1 : 0);
    }
    // kx = 0: odd[j], 1: even[j], 2: odd[j + 1], 3: even[j + 1]   (odd[j] = B[2(x0+j) - 1], even[j] = B[2(x0+j)])

    f32x4 acc[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // channel rows beyond the tensors' channel counts are zeroed once and never staged (few-channel layers: 8 -> 4)
    const int nb = min(16, A.Cb - b0), ns = min(16, A.Cs - s0);
    for (int e = tid; e < brows * C::XS + 16 * DS; e += THREADS) smem[e] = 0.f;
    __syncthreads();

    // 16-byte staging of the big tile needs rows that start 16-byte aligned
    const bool vec_ok = (A.Wb & 3) == 0 && ((reinterpret_cast<uintptr_t>(A.dz) | reinterpret_cast<uintptr_t>(A.a.p) |
                                              reinterpret_cast<uintptr_t>(A.b.p)) & 15) == 0;
    for (int item = blockIdx.x; item < A.items; item += gridDim.x) {
        int r = item;
        const int seg = r % A.segs;
        r /= A.segs;
        const int y = r % A.Hs;
        r /= A.Hs;
        const int z = r % A.Ds;
        const int n = r / A.Ds;
        const int x0 = seg * TWG;

        // ---- stage the big tile: 16 channels x K*K rows x 66 columns (2 x0 - 1 .. 2 x0 + 64), split by column parity
        if (vec_ok) {
            // a (channel, row) run = 16 aligned 16-byte loads (columns 2 x0 .. 2 x0 + 63) + its two halo columns = 18
            // lanes; a wave covers 3 runs per load instruction and up to 8 instructions are in flight (the flat
            // element loop below, one conditional 4-byte load per element, took 8 x longer than the MFMAs of an item)
            constexpr int LPR = 18, RPI = 3, FLY = SMALL_NORM ? 6 : 8;
            const int rs = lane / LPR, part = lane - rs * LPR;
            const bool lane_on = rs < RPI;
            const bool halo = part >= 16;
            const int X = halo ? (part == 16 ? 2 * x0 - 1 : 2 * x0 + 2 * TWG) : 2 * x0 + 4 * part;
            const bool colok = X >= 0 && X < A.Wb;
            const int Xc = halo ? min(max(X, 0), A.Wb - 1) : min(X, A.Wb - 4);
            const int runs = nb * C::ROWS;
            const int iters = (runs + 4 * RPI - 1) / (4 * RPI);
            for (int it0 = 0; it0 < iters; it0 += FLY) {
                f32x4 qa[FLY], qb[SMALL_NORM ? 1 : FLY];
#pragma unroll
                for (int itl = 0; itl < FLY; ++itl) {
                    const int run = min(((it0 + itl) * 4 + wave) * RPI + rs, runs - 1);
                    const int c = run / C::ROWS, rr = run - c * C::ROWS;
                    const int zc = min(max(2 * z - 1 + rr / K, 0), A.Db - 1), yc = min(max(2 * y - 1 + rr % K, 0), A.Hb - 1);
                    const int g = n * A.Cb + b0 + c;
                    const size_t inplane = (size_t)yc * A.Wb + Xc;
                    const float* pa = (SMALL_NORM ? A.dz : A.a.p) + (size_t)g * vol_b + (size_t)zc * plane_b + inplane;
                    const float* pb = (!SMALL_NORM && A.b.p)
                                          ? A.b.p + (A.b.bcast_d ? (size_t)g * plane_b : (size_t)g * vol_b + (size_t)zc * plane_b) + inplane
                                          : pa;
                    if (halo) {
                        qa[itl] = f32x4{*pa, 0.f, 0.f, 0.f};
                        if (!SMALL_NORM) qb[itl] = f32x4{*pb, 0.f, 0.f, 0.f};
                    } else {
                        qa[itl] = *reinterpret_cast<const f32x4*>(pa);
                        if (!SMALL_NORM) qb[itl] = *reinterpret_cast<const f32x4*>(pb);
                    }
                }
#pragma unroll
                for (int itl = 0; itl < FLY; ++itl) {
                    const int run_raw = ((it0 + itl) * 4 + wave) * RPI + rs;
                    const int run = min(run_raw, runs - 1);
                    const int c = run / C::ROWS, rr = run - c * C::ROWS;
                    const int zz = 2 * z - 1 + rr / K, yy = 2 * y - 1 + rr % K;
                    const bool ok = colok && zz >= 0 && zz < A.Db && yy >= 0 && yy < A.Hb;
                    float sa = 1.f, ha = 0.f, sb2 = 1.f, hb2 = 0.f;
                    const bool two = !SMALL_NORM && A.b.p;
                    if (!SMALL_NORM) {
                        const int g = n * A.Cb + b0 + c;
                        if (A.a.scale) {
                            sa = A.a.scale[g];
                            ha = A.a.shift[g];
                        }
                        if (two && A.b.scale) {
                            sb2 = A.b.scale[g];
                            hb2 = A.b.shift[g];
                        }
                    }
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = SMALL_NORM ? qa[itl][j] : fmaf(sa, qa[itl][j], ha);
                        if (!SMALL_NORM && two) t += fmaf(sb2, qb[SMALL_NORM ? 0 : itl][j], hb2);
                        v[j] = ok ? t : 0.f;
                    }
                    if (lane_on && run_raw < runs) {
                        float* dst = bl + c * C::XS + rr * RS;   // [odd columns (HS) | even columns (HS)]
                        if (halo) {
                            dst[part == 16 ? 0 : HS + TWG] = v[0];          // odd[0] = B[2 x0 - 1], even[32] = B[2 x0 + 64]
                        } else {
                            // columns 2 x0 + 4 q + (0..3): even[2q], odd[2q+1], even[2q+1], odd[2q+2]
                            *reinterpret_cast<float2*>(dst + HS + 2 * part) = make_float2(v[0], v[2]);
                            dst[2 * part + 1] = v[1];
                            dst[2 * part + 2] = v[3];
                        }
                    }
                }
            }
        } else
        for (int c = 0; c < nb; ++c) {
            const int ch = b0 + c;
            for (int e = tid; e < C::ROWS * BCOLS; e += THREADS) {
                const int rr = e / BCOLS, xx = e - rr * BCOLS;
                const int zz = 2 * z - 1 + rr / K, yy = 2 * y - 1 + rr % K, x = 2 * x0 - 1 + xx;
                float v = 0.f;
                if (zz >= 0 && zz < A.Db && yy >= 0 && yy < A.Hb && x >= 0 && x < A.Wb) {
                    if (SMALL_NORM)
                        v = A.dz[((size_t)(n * A.Cb + ch)) * vol_b + (size_t)zz * plane_b + (size_t)yy * A.Wb + x];
                    else
                        v = xhat_value(A.a, A.b, n, A.Cb, ch, A.Db, A.Hb, A.Wb, zz, yy, x);
                }
                bl[c * C::XS + rr * RS + (xx & 1) * HS + (xx >> 1)] = v;
            }
        }
        // ---- stage the small tile: 16 channels x 32 positions ------------------------------------------------
        for (int e = tid; e < ns * TWG; e += THREADS) {
            const int o = e / TWG, px = e - o * TWG;
            const int x = x0 + px, ch = s0 + o;
            float v = 0.f;
            if (x < A.Ws) {
                if (SMALL_NORM)
                    v = xhat_value(A.a, A.b, n, A.Cs, ch, A.Ds, A.Hs, A.Ws, z, y, x);
                else
                    v = A.dz[((size_t)(n * A.Cs + ch)) * vol_s + (size_t)z * plane_s + (size_t)y * A.Ws + x];
            }
            sl[o * DS + px] = v;
        }
        __syncthreads();

        const float* arow = sl + (lane & 15) * DS + (lane >> 4);
        const float* brow = bl + min((lane & 15) & (CBP - 1), brows - 1) * C::XS + (lane >> 4);
#pragma unroll 2
        for (int ks = 0; ks < TWG / 4; ++ks) {
            const float af = arow[ks * 4];
#pragma unroll
            for (int i = 0; i < GPW; ++i) {
                if (wave + 4 * i < GROUPS) {   // wave-uniform
                    const float bf = brow[toff[i] + ks * 4];
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- one partial per workgroup: [Cs][Cb][K^3] ------------------------------------------------------------
    float* dst = A.partial + (size_t)blockIdx.x * A.Cs * A.Cb * C::TAPS;
    const int bc = b0 + ((lane & 15) & (CBP - 1));
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int tap = (wave + 4 * i) * NT + tj;
        if (wave + 4 * i >= GROUPS || tap >= C::TAPS) continue;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int scn = s0 + 4 * (lane >> 4) + rr;
            if (scn < A.Cs && bc < A.Cb) dst[((size_t)scn * A.Cb + bc) * C::TAPS + tap] = acc[i][rr];
        }
    }
}

int launch_wgrad_reduce_f32(const float* partial, size_t wcount, int parts, float* dw, int accumulate, hipStream_t s);
bool wgrad3d_s2_rolling_supported(int transposed, const Src& b, const Geom& small, const Geom& big, const float* dz,
                                  const Src& a);   // wgrad3d_s2r.hip
int launch_wgrad3d_s2_rolling(int transposed, const Src& a, const Src& b, const float* dz, float* dw, const Geom& small,
                              const Geom& big, int accumulate, float* scratch, int max_wgs, hipStream_t s);

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the full-resolution transposed convolution k(3,4,4) s(1,2,2) p1, Cin <= 5 -> Cout = 1 (reference
// regularization.py:90-92, network_blocks.py:124-131):
//   dW[c][0][kd][kh][kw] = sum_{oz, iy, ix} xhat[c][oz + 1 - kd][iy][ix] * dz[oz][2 iy - 1 + kh][2 ix - 1 + kw]
// With one output channel the channel-pair tiling above would use 4 x 1 of a 16 x 16 MFMA block, so the TAPS go on the
// matrix sides instead: M = (c, kd) (12 of 16 rows for 4 channels), N = (kh, kw) (all 16 columns), K = positions along x:
// ONE MFMA per four positions yields all 48 taps of all channels.  Every wave works alone (wave-private LDS tiles, no
// workgroup barrier) on items (n, oz, iy, 32-position x segment) and keeps one partial in registers.
// ---------------------------------------------------------------------------------------------------------------
namespace {
constexpr int UF_HS = 36, UF_RS = 2 * UF_HS;   // dz rows: parity halves 36 apart, rows 72 apart: banks 8 kh + 4 (kw & 1) + 0..2
constexpr int UF_WAVE_FLOATS = 16 * DS + 4 * UF_RS;
struct UFArgs {
    Src a;
    const float* __restrict__ dz;
    float* __restrict__ partial;   // [wave][Cin][48]
    int N, Cin, Di, Hi, Wi;        // input grid; output grid is (Di, 2 Hi, 2 Wi)
    int items, segs;
};
}  // namespace

__global__ __launch_bounds__(256) void wgrad_up_full_mfma_kernel(const UFArgs A) {
    __shared__ __attribute__((aligned(16))) float smem_uf[4 * UF_WAVE_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sl = smem_uf + wave * UF_WAVE_FLOATS;   // [16 rows m = c * 4 + kd][DS]
    float* bl = sl + 16 * DS;                      // [4 kh][odd | even][UF_HS]
    const int Ho = 2 * A.Hi, Wo = 2 * A.Wi;
    const size_t plane_i = (size_t)A.Hi * A.Wi, plane_o = (size_t)Ho * Wo;
    for (int e = lane; e < UF_WAVE_FLOATS; e += 64) sl[e] = 0.f;   // rows kd = 3 and c >= Cin stay zero

    const int gwave = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    const bool vec_ok = (A.Wi & 3) == 0 && A.Cin * 3 * (TWG / 4) <= 128 &&
                        ((reinterpret_cast<uintptr_t>(A.dz) | reinterpret_cast<uintptr_t>(A.a.p)) & 15) == 0;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nn = lane & 15;
    const float* arow = sl + nn * DS + (lane >> 4);
    const float* brow = bl + (nn >> 2) * UF_RS + (nn & 1) * UF_HS + ((nn & 3) >> 1) + (lane >> 4);
    if (vec_ok) {
        // 16-byte loads, and the NEXT item's operands requested before this item's MFMAs: a wave works alone on its items
        // (no workgroup barrier), so without the prefetch every item waited out a full global-load round trip -- 5.4 us
        // per item for 8 MFMAs
        f32x4 vz, vx[2];
        float hv = 0.f;
        bool okz = false, okx[2] = {false, false};
        auto load_item = [&](int item) __attribute__((always_inline)) {
            int r = item;
            const int seg = r % A.segs;
            r /= A.segs;
            const int iy = r % A.Hi;
            r /= A.Hi;
            const int oz = r % A.Di;
            const int n = r / A.Di;
            const int x0 = seg * TWG;
            const float* pz = A.dz + ((size_t)n * A.Di + oz) * plane_o;
            {
                const int kh = lane >> 4, q = lane & 15;
                const int oy = 2 * iy - 1 + kh, ox = 2 * x0 + 4 * q;
                okz = oy >= 0 && oy < Ho && ox < Wo;
                vz = *reinterpret_cast<const f32x4*>(pz + (size_t)min(max(oy, 0), Ho - 1) * Wo + min(ox, Wo - 4));
                hv = 0.f;
                if (lane < 8) {   // halo columns 2 x0 - 1 and 2 x0 + 64
                    const int hk = lane >> 1, side = lane & 1;
                    const int hy = 2 * iy - 1 + hk, hx = side ? 2 * x0 + 64 : 2 * x0 - 1;
                    if (hy >= 0 && hy < Ho && hx >= 0 && hx < Wo) hv = pz[(size_t)hy * Wo + hx];
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int e = lane + 64 * j;
                okx[j] = false;
                vx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (e < A.Cin * 3 * (TWG / 4)) {
                    const int q = e % (TWG / 4), ck = e / (TWG / 4);
                    const int c = ck / 3, kd = ck - c * 3;
                    const int iz = oz + 1 - kd, ix = x0 + 4 * q;
                    okx[j] = iz >= 0 && iz < A.Di && ix < A.Wi;
                    const int g = n * A.Cin + c;
                    const float sa = A.a.scale ? A.a.scale[g] : 1.f, ha = A.a.scale ? A.a.shift[g] : 0.f;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(
                        A.a.p + ((size_t)g * A.Di + min(max(iz, 0), A.Di - 1)) * plane_i + (size_t)iy * A.Wi + min(ix, A.Wi - 4));
#pragma unroll
                    for (int k = 0; k < 4; ++k) vx[j][k] = fmaf(sa, v[k], ha);
                }
            }
        };
        auto store_item = [&]() __attribute__((always_inline)) {
            const int kh = lane >> 4, q = lane & 15;
            float* row = bl + kh * UF_RS;
            // columns xx = 4 q + 1 .. 4 q + 4 of the staged row: odd half [2 q], even [2 q + 1], odd [2 q + 1], even [2 q + 2]
            *reinterpret_cast<float2*>(row + UF_HS + 2 * q) = make_float2(okz ? vz[0] : 0.f, okz ? vz[2] : 0.f);
            row[2 * q + 1] = okz ? vz[1] : 0.f;
            row[2 * q + 2] = okz ? vz[3] : 0.f;
            if (lane < 8) bl[(lane >> 1) * UF_RS + ((lane & 1) ? UF_HS + TWG : 0)] = hv;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int e = lane + 64 * j;
                if (e < A.Cin * 3 * (TWG / 4)) {
                    const int q2 = e % (TWG / 4), ck = e / (TWG / 4);
                    const int c = ck / 3, kd = ck - c * 3;
                    float* dst = sl + (c * 4 + kd) * DS + 4 * q2;
                    *reinterpret_cast<float2*>(dst) = make_float2(okx[j] ? vx[j][0] : 0.f, okx[j] ? vx[j][1] : 0.f);
                    *reinterpret_cast<float2*>(dst + 2) = make_float2(okx[j] ? vx[j][2] : 0.f, okx[j] ? vx[j][3] : 0.f);
                }
            }
        };
        if (gwave < A.items) load_item(gwave);
        for (int item = gwave; item < A.items; item += nwaves) {
            store_item();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (item + nwaves < A.items) load_item(item + nwaves);
#pragma unroll
            for (int ks = 0; ks < TWG / 4; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[ks * 4], brow[ks * 4], acc, 0, 0, 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    } else
    for (int item = gwave; item < A.items; item += nwaves) {
        int r = item;
        const int seg = r % A.segs;
        r /= A.segs;
        const int iy = r % A.Hi;
        r /= A.Hi;
        const int oz = r % A.Di;
        const int n = r / A.Di;
        const int x0 = seg * TWG;
        // dz rows 2 iy - 1 .. 2 iy + 2, columns 2 x0 - 1 .. 2 x0 + 64, split by column parity
        const float* pz = A.dz + ((size_t)n * A.Di + oz) * plane_o;
        for (int e = lane; e < 4 * BCOLS; e += 64) {
            const int kh = e / BCOLS, xx = e - kh * BCOLS;
            const int oy = 2 * iy - 1 + kh, ox = 2 * x0 - 1 + xx;
            float v = 0.f;
            if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) v = pz[(size_t)oy * Wo + ox];
            bl[kh * UF_RS + (xx & 1) * UF_HS + (xx >> 1)] = v;
        }
        // xhat rows (c, kd): plane oz + 1 - kd of channel c
        for (int e = lane; e < A.Cin * 3 * TWG; e += 64) {
            const int px = e % TWG, ck = e / TWG;
            const int c = ck / 3, kd = ck - c * 3;
            const int iz = oz + 1 - kd, ix = x0 + px;
            float v = 0.f;
            if (iz >= 0 && iz < A.Di && ix < A.Wi) {
                const int g = n * A.Cin + c;
                const float sa = A.a.scale ? A.a.scale[g] : 1.f, ha = A.a.scale ? A.a.shift[g] : 0.f;
                v = fmaf(sa, A.a.p[((size_t)g * A.Di + iz) * plane_i + (size_t)iy * A.Wi + ix], ha);
            }
            sl[(c * 4 + kd) * DS + px] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ks = 0; ks < TWG / 4; ++ks)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[ks * 4], brow[ks * 4], acc, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // D[m = 4 * (lane >> 4) + r][n = lane & 15]: m = c * 4 + kd, n = kh * 4 + kw
    float* dst = A.partial + (size_t)gwave * A.Cin * 48;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int m = 4 * (lane >> 4) + rr;
        const int c = m >> 2, kd = m & 3;
        if (c < A.Cin && kd < 3) dst[c * 48 + kd * 16 + nn] = acc[rr];
    }
}

static int up_full_workgroups(const Geom& in) {
    const size_t items = (size_t)in.n * in.d * in.h * ((in.w + TWG - 1) / TWG);
    size_t wgs = items / 16;
    if (wgs > 1024) wgs = 1024;
    return wgs < 1 ? 1 : (int)wgs;
}

bool wgrad_up_full_mfma_supported(int transposed, int kd, const Src& b, const Geom& in, const Geom& out) {
    static const bool enabled = []() {
        const char* e = debug_switch("PDS_WGRAD3D_S2_MFMA");
        return !(e && e[0] == '0');
    }();
    return enabled && transposed && kd == 3 && !b.p && in.c <= 4 && out.c == 1 && out.d == in.d && out.h == 2 * in.h &&
           out.w == 2 * in.w;
}

size_t wgrad_up_full_mfma_scratch_floats(const Geom& in) { return (size_t)up_full_workgroups(in) * 4 * in.c * 48; }


// type: 0 convolution (kd 3, stride 2), 1 transposed convolution (kd 4: k4 s2 p1 on all three axes)
bool wgrad3d_s2_mfma_supported(int transposed, int kd, int stride, const Geom& in, const Geom& out) {
    static const bool enabled = []() {  // PDS_WGRAD3D_S2_MFMA=0 selects the VALU kernels (A/B, debugging)
        const char* e = debug_switch("PDS_WGRAD3D_S2_MFMA");
        return !(e && e[0] == '0');
    }();
    if (!enabled) return false;
    if (transposed) return kd == 4 && out.d == 2 * in.d && out.h == 2 * in.h && out.w == 2 * in.w;
    return kd == 3 && stride == 2 && out.d == (in.d - 1) / 2 + 1 && out.h == (in.h - 1) / 2 + 1 &&
           out.w == (in.w - 1) / 2 + 1;
}

static void s2_roles(int transposed, const Geom& in, const Geom& out, Geom& small, Geom& big) {
    small = transposed ? in : out;
    big = transposed ? out : in;
}

static int wgrad3d_s2_workgroups(const Geom& small, int pairs) {
    const size_t items = (size_t)small.n * small.d * small.h * ((small.w + TWG - 1) / TWG);
    size_t wgs = items / 4;                       // >= ~4 items per workgroup so the partial write amortises ...
    const size_t cap = (size_t)(2048 / pairs) > 0 ? (size_t)(2048 / pairs) : 1;
    if (wgs > cap) wgs = cap;
    // ... unless that leaves a quarter of the CUs idle: the deepest level of the hourglass (27 items, 32 channel pairs)
    // is a chain of 4-5 items on 6 workgroups per pair -- one item per workgroup there (measured: 166 -> 92 us and
    // 139 -> 124 us; the same rule applied to the levels above it costs more in partial sums than it gains)
    if (wgs * pairs < 200) {
        const size_t fill = (size_t)(1024 / pairs) > 0 ? (size_t)(1024 / pairs) : 1;
        wgs = items < fill ? items : fill;
    }
    return wgs < 1 ? 1 : (int)wgs;
}

size_t wgrad3d_s2_mfma_scratch_floats(int transposed, const Geom& in, const Geom& out) {
    Geom small, big;
    s2_roles(transposed, in, out, small, big);
    const int pairs = ((small.c + 15) / 16) * ((big.c + 15) / 16);
    const int taps = transposed ? 64 : 27;
    return (size_t)wgrad3d_s2_workgroups(small, pairs) * small.c * big.c * taps;
}

int launch_wgrad3d_s2_mfma(int transposed, const Src& a, const Src& b, const float* dz, float* dw, const Geom& in,
                           const Geom& out, int accumulate, float* scratch, hipStream_t s) {
    Geom small, big;
    s2_roles(transposed, in, out, small, big);
    {   // the full-resolution layers (4 or 8 big-grid channels): rolling form, wgrad3d_s2r.hip
        const int pairs0 = ((small.c + 15) / 16) * ((big.c + 15) / 16);
        if (wgrad3d_s2_rolling_supported(transposed, b, small, big, dz, a))
            return launch_wgrad3d_s2_rolling(transposed, a, b, dz, dw, small, big, accumulate, scratch,
                                             wgrad3d_s2_workgroups(small, pairs0), s);
    }
    WS2Args A;
    A.a = a;
    A.b = b;
    A.dz = dz;
    A.partial = scratch;
    A.N = in.n;
    A.Cs = small.c;
    A.Cb = big.c;
    A.Ds = small.d;
    A.Hs = small.h;
    A.Ws = small.w;
    A.Db = big.d;
    A.Hb = big.h;
    A.Wb = big.w;
    A.segs = (small.w + TWG - 1) / TWG;
    A.items = small.n * small.d * small.h * A.segs;
    A.sblocks = (small.c + 15) / 16;
    const int pairs = A.sblocks * ((big.c + 15) / 16);
    const int wgs = wgrad3d_s2_workgroups(small, pairs);
    const int taps = transposed ? 64 : 27;
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3d_s2_mfma_kernel<4, true, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3d_s2_mfma_kernel<3, false, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    }
    const int nbm = big.c < 16 ? big.c + 1 : 16;
    const int nt = big.c == 8 ? 2 : big.c == 4 ? 4 : 1;   // taps per column group (kernel comment)
    const dim3 grid(wgs, pairs);
    if (transposed) {
        const size_t lds = (size_t)(nbm * Cfg<4>::XS + 16 * DS) * sizeof(float);
        if (nt == 4) hipLaunchKernelGGL((wgrad3d_s2_mfma_kernel<4, true, 4>), grid, dim3(THREADS), lds, s, A);
        else if (nt == 2) hipLaunchKernelGGL((wgrad3d_s2_mfma_kernel<4, true, 2>), grid, dim3(THREADS), lds, s, A);
        else hipLaunchKernelGGL((wgrad3d_s2_mfma_kernel<4, true, 1>), grid, dim3(THREADS), lds, s, A);
    } else {
        const size_t lds = (size_t)(nbm * Cfg<3>::XS + 16 * DS) * sizeof(float);
        if (nt == 4) hipLaunchKernelGGL((wgrad3d_s2_mfma_kernel<3, false, 4>), grid, dim3(THREADS), lds, s, A);
        else if (nt == 2) hipLaunchKernelGGL((wgrad3d_s2_mfma_kernel<3, false, 2>), grid, dim3(THREADS), lds, s, A);
        else hipLaunchKernelGGL((wgrad3d_s2_mfma_kernel<3, false, 1>), grid, dim3(THREADS), lds, s, A);
    }
    if (int rc = check_launch("wgrad3d_s2_mfma")) return rc;
    return launch_wgrad_reduce_f32(scratch, (size_t)small.c * big.c * taps, wgs, dw, accumulate, s);
}

int launch_wgrad_up_full_mfma(const Src& a, const float* dz, float* dw, const Geom& in, int accumulate, float* scratch,
                              hipStream_t s) {
    UFArgs A;
    A.a = a;
    A.dz = dz;
    A.partial = scratch;
    A.N = in.n;
    A.Cin = in.c;
    A.Di = in.d;
    A.Hi = in.h;
    A.Wi = in.w;
    A.segs = (in.w + TWG - 1) / TWG;
    A.items = in.n * in.d * in.h * A.segs;
    const int wgs = up_full_workgroups(in);
    hipLaunchKernelGGL(wgrad_up_full_mfma_kernel, dim3(wgs), dim3(256), 0, s, A);
    if (int rc = check_launch("wgrad_up_full_mfma")) return rc;
    return launch_wgrad_reduce_f32(scratch, (size_t)in.c * 48, wgs * 4, dw, accumulate, s);
}

}  // namespace pds
