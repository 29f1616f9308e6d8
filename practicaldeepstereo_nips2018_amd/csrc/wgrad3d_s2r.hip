// Weight gradients of the full-resolution stride-2 layers of the hourglass, rolling form (round 4; reference
// regularization.py:28-31, 54-57, 88-89 under loss.backward(), pds_trainer.py:40-46).  Same contraction as
// wgrad3d_s2_mfma.hip,
//   R[s][b][kz][ky][kx] = sum over small-grid positions p of S[s][p] * B[b][2 p - 1 + k]      (k < K = 3 or 4 per axis)
// for the layers whose big-grid tensor has 4 or 8 channels (the 8 -> 16 contraction, the 16 -> 8 and 8 -> 4 expansions):
// 70 % of the stride-2 weight-gradient time.  That kernel stages K x K big rows of 66 columns for EVERY row of 32 small
// positions (a big row is read by 2 x 2 items and staged each time) and is bound by exactly that.  Here:
//   unit      2 small rows x 32 small columns, walked along z over a chunk of small planes.  The K big planes a step needs
//             live in an LDS ring of four (plane Z in slot (Z + 1) & 3); a step stages only its TWO new big planes
//             (2 TY + K - 2 rows of 66 columns each), requested before the MFMAs of the step before and written behind
//             its barrier into the two slots that step freed: 2.7 x fewer staged elements per position.
//   staging   16-byte loads: a (plane, channel, row) run is 16 aligned quads + 2 halo columns = 18 lanes; what a thread
//             stages is the same in every step, so offsets, bounds and LDS addresses are computed once per unit.  Rows
//             are stored split by column parity (the stride-2 read becomes a unit-stride one, as in wgrad3d_s2_mfma).
//   MFMA      v_mfma_f32_16x16x4_f32 (exact); the 16 columns hold the big tensor's 8 (4) channels x 2 (4) taps.
//   workgroup 4 waves, persistent over units, ONE partial per workgroup (summed in fp64 by wgrad_reduce_f32_kernel).
#include <atomic>

#include "common.hpp"

namespace pds {

namespace {

constexpr int R_THREADS = 256;
constexpr int R_TY = 2, R_TWG = 32;       // small rows x columns per step
constexpr int R_DS = R_TY * R_TWG + 2;    // small channel stride: 66 == 2 (mod 32)
constexpr int R_LPR = 18;                 // lanes per staged run: 16 quads + 2 halo columns
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int CB>
struct RCfg {
    // Bank layout of the B fragment read: the 32 lanes of a half-wave are (channel c, tap tj of the group, position k < 2) at
    // c XS + t(tj) + k with t = {0, HS} or {1, HS + 1} (two taps, K = 4) resp. {0, HS, 1, HS + 1} (four taps).  4 channels
    // x 4 taps: HS == 4, XS == 8 (mod 32) gives 8 c + {0, 1, 2 | 4, 5, 6} (the coinciding pairs are the SAME address: a
    // broadcast); 8 channels x 2 taps: HS == 2, XS == 4 gives 4 c + {0 .. 3}.  (With XS == 2 for both, every read was a
    // 2- to 4-way conflict and the kernel was bound by them.)
    static constexpr int HS = CB == 4 ? 36 : 34, RS = 2 * HS;    // parity half / big row stride (floats)
    static constexpr int RB = 2 * R_TY + K - 2;                  // big rows per plane
    static constexpr int PS = RB * RS;                            // plane (ring slot) stride
    static constexpr int RAW = 4 * PS;
    static constexpr int XMOD = CB == 4 ? 8 : 4;
    static constexpr int XS = RAW + ((XMOD - RAW % 32) + 32) % 32;
    static constexpr int TAPS = K * K * K;
    static constexpr int NT = 16 / CB;                            // taps per column group
    static constexpr int GROUPS = (TAPS + NT - 1) / NT;
    static constexpr int GPW = (GROUPS + 3) / 4;
    static constexpr int RUNS = 2 * CB * RB;                      // (plane of the pair, channel, row) runs per step
    static constexpr int NSTG = (RUNS * R_LPR + R_THREADS - 1) / R_THREADS;
    static constexpr int LDS_FLOATS = CB * XS + 16 * R_DS;
    static_assert(XS % 32 == XMOD && XS % 2 == 0, "bank layout");
};

struct RArgs {
    Src a, b;                      // the normalised tensor (the layer's input)
    const float* __restrict__ dz;  // the plain one (gradient of the layer's raw output)
    float* __restrict__ partial;   // [workgroup][Cs][Cb][K^3]
    int N, Cs, Cb;
    int Ds, Hs, Ws;                // small grid
    int Db, Hb, Wb;                // big grid
    int units, segs, yblocks, zchunks, zc;
};

}  // namespace

// SMALL_NORM: the small-grid tensor is the normalised one (transposed convolution); otherwise the big-grid one is
template <int K, int CB, bool SMALL_NORM, bool HAS_B>
__global__ __launch_bounds__(R_THREADS, 2) void wgrad3d_s2r_kernel(const RArgs A) {
    using C = RCfg<K, CB>;
    extern __shared__ __attribute__((aligned(16))) float smem_r[];
    float* bl = smem_r;               // [CB][4 slots][RB][odd | even]
    float* sl = smem_r + CB * C::XS;  // [16][R_DS]
    __shared__ f32x4 coef[16];        // per channel of the normalised tensor: scale, shift of the two sources

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int plane_s = A.Hs * A.Ws, plane_b = A.Hb * A.Wb;
    const unsigned vol_b = (unsigned)A.Db * plane_b;

    // lane's tap of column group i: taps g NT + tj, g = wave + 4 i
    const int tj = (lane & 15) / CB;
    int tkz[C::GPW], tbase[C::GPW];
#pragma unroll
    for (int i = 0; i < C::GPW; ++i) {
        const int tap = min((wave + 4 * i) * C::NT + tj, C::TAPS - 1);
        const int kz = tap / (K * K), ky = (tap / K) % K, kx = tap % K;
        tkz[i] = kz;
        // row 2 r + ky of the plane, parity half of 2 xx + kx, shift (kx + 1) >> 1 minus one for the odd half (cf.
        // wgrad3d_s2_mfma.hip: kx = 0: odd[j], 1: even[j], 2: odd[j + 1], 3: even[j + 1])
        tbase[i] = ky * C::RS + ((kx & 1) ? C::HS : 0) + ((kx + 1) >> 1) - ((kx & 1) ? 1 : 0);
    }
    f32x4 acc[C::GPW];
#pragma unroll
    for (int i = 0; i < C::GPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* arow = sl + (lane & 15) * R_DS + (lane >> 4);
    const float* brow = bl + ((lane & 15) & (CB - 1)) * C::XS + (lane >> 4);

    for (int e = tid; e < 16 * R_DS; e += R_THREADS) sl[e] = 0.f;   // rows beyond Cs stay zero
    __syncthreads();

    for (int unit = blockIdx.x; unit < A.units; unit += gridDim.x) {
        int r = unit;
        const int seg = r % A.segs;
        r /= A.segs;
        const int yb = r % A.yblocks;
        r /= A.yblocks;
        const int zci = r % A.zchunks;
        const int n = r / A.zchunks;
        const int x0 = seg * R_TWG, y0 = yb * R_TY;
        const int z0 = zci * A.zc, z1 = min(z0 + A.zc, A.Ds);
        const int rows = min(R_TY, A.Hs - y0);

        if (tid < 16) {   // deferred InstanceNorm of the normalised tensor (small: Cs channels, big: Cb)
            const int nch = SMALL_NORM ? A.Cs : A.Cb;
            const int g = n * nch + min(tid, nch - 1);
            f32x4 cf{1.f, 0.f, 1.f, 0.f};
            if (A.a.scale) {
                cf[0] = A.a.scale[g];
                cf[1] = A.a.shift[g];
            }
            if (HAS_B && A.b.scale) {
                cf[2] = A.b.scale[g];
                cf[3] = A.b.shift[g];
            }
            coef[tid] = cf;
        }

        // ---- what this thread stages of a PAIR of big planes (Z, Z + 1): runs of 18 lanes ----------------------------
        // boff: offset inside (channel block, plane 0) or -1; bdst: (LDS row address << 3) | halo kind << 1 | plane of the pair, or -1
        int boff[C::NSTG], bdst[C::NSTG];
#pragma unroll
        for (int j = 0; j < C::NSTG; ++j) {
            const int e = tid + j * R_THREADS;
            const int run = e / R_LPR, part = e - run * R_LPR;
            const bool slot = run < C::RUNS;
            const int pl = run / (CB * C::RB), cr = run - pl * (CB * C::RB);
            const int c = cr / C::RB, row = cr - c * C::RB;
            const int Y = 2 * y0 - 1 + row;
            const int X = part < 16 ? 2 * x0 + 4 * part : (part == 16 ? 2 * x0 - 1 : 2 * x0 + 2 * R_TWG);
            const bool ok = slot && c < A.Cb && Y >= 0 && Y < A.Hb && X >= 0 && X < A.Wb;
            boff[j] = ok ? c * (int)vol_b + Y * A.Wb + X : -1;
            // low bits: which of the pair's planes (bit 0), halo kind in bits 1-2 (0 quad, 1 left, 2 right)
            bdst[j] = slot ? ((c * C::XS + row * C::RS) << 3) | pl | ((part < 16 ? 0 : part == 16 ? 1 : 2) << 1) : -1;
            if (slot && part < 16) bdst[j] += (2 * part) << 3;
        }
        const float* bsrc = (SMALL_NORM ? A.dz : A.a.p) + (size_t)n * A.Cb * vol_b;
        const unsigned bshrink = (!SMALL_NORM && HAS_B && A.b.bcast_d) ? vol_b - (unsigned)plane_b : 0u;
        const float* bsrc2 = (!SMALL_NORM && HAS_B) ? A.b.p + (size_t)n * A.Cb * (vol_b - bshrink) : nullptr;

        f32x4 qa[C::NSTG], qb[(!SMALL_NORM && HAS_B) ? C::NSTG : 1];
        // big planes Z0, Z0 + 1 -> registers (zero outside the volume)
        auto load_big = [&](int Z0) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < C::NSTG; ++j) {
                const int pl = bdst[j] & 1, kind = (bdst[j] >> 1) & 3;
                const int Z = Z0 + pl;
                const bool ok = boff[j] >= 0 && Z >= 0 && Z < A.Db;
                const unsigned off = (unsigned)max(boff[j], 0) + (unsigned)(min(max(Z, 0), A.Db - 1) * plane_b);
                f32x4 v{0.f, 0.f, 0.f, 0.f}, w{0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    if (kind == 0) v = *reinterpret_cast<const f32x4*>(bsrc + off);
                    else v[0] = bsrc[off];
                    if (!SMALL_NORM && HAS_B) {
                        const unsigned c = (unsigned)(tid + j * R_THREADS) / R_LPR % (CB * C::RB) / C::RB;
                        const unsigned off2 = A.b.bcast_d ? (unsigned)max(boff[j], 0) - c * bshrink : off - c * bshrink;
                        if (kind == 0) w = *reinterpret_cast<const f32x4*>(bsrc2 + off2);
                        else w[0] = bsrc2[off2];
                    }
                }
                qa[j] = v;
                if (!SMALL_NORM && HAS_B) qb[j] = w;
            }
        };
        auto store_big = [&](int Z0) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < C::NSTG; ++j) {
                if (bdst[j] < 0) continue;
                const int pl = bdst[j] & 1, kind = (bdst[j] >> 1) & 3;
                const int Z = Z0 + pl;
                const bool ok = boff[j] >= 0 && Z >= 0 && Z < A.Db;
                f32x4 v = qa[j];
                if (!SMALL_NORM) {
                    const int c = (tid + j * R_THREADS) / R_LPR % (CB * C::RB) / C::RB;
                    const f32x4 cf = coef[c];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = fmaf(cf[0], v[e], cf[1]);
                        if (HAS_B) t += fmaf(cf[2], qb[j][e], cf[3]);
                        v[e] = ok ? t : 0.f;   // the literal zero padding, not the normalised zero
                    }
                }
                float* row = bl + (bdst[j] >> 3) + ((Z + 1) & 3) * C::PS;
                if (kind == 0) {
                    // columns xx = 4 q + 1 .. 4 q + 4 of the staged row: odd half [2 q], even [2 q + 1], odd [2 q + 1], even [2 q + 2]
                    // (row already points at entry 2 q of the odd half)
                    *reinterpret_cast<float2*>(row + C::HS) = make_float2(v[0], v[2]);
                    row[1] = v[1];
                    row[2] = v[3];
                } else if (kind == 1) {
                    row[0] = v[0];                 // column 2 x0 - 1: even half [0]
                } else {
                    row[C::HS + R_TWG] = v[0];      // column 2 x0 + 64: odd half [32]
                }
            }
        };
        // small rows of plane z: Cs channels x 2 rows x 8 quads = one quad per thread
        f32x4 qs;
        bool qs_ok = false;
        auto load_small = [&](int z) __attribute__((always_inline)) {
            const int q = tid & 7, rr = (tid >> 3) & 1, ch = tid >> 4;
            const int x = x0 + 4 * q, y = y0 + rr;
            qs_ok = ch < A.Cs && x < A.Ws && y < A.Hs;
            const float* src = SMALL_NORM ? A.a.p : A.dz;
            qs = f32x4{0.f, 0.f, 0.f, 0.f};
            if (qs_ok)
                qs = *reinterpret_cast<const f32x4*>(src + ((size_t)(n * A.Cs + ch) * A.Ds + z) * plane_s + (size_t)y * A.Ws + x);
        };
        auto store_small = [&]() __attribute__((always_inline)) {
            const int q = tid & 7, rr = (tid >> 3) & 1, ch = tid >> 4;
            f32x4 v = qs;
            if (SMALL_NORM) {
                const f32x4 cf = coef[ch];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = qs_ok ? fmaf(cf[0], v[e], cf[1]) : 0.f;
            }
            float* dst = sl + ch * R_DS + rr * R_TWG + 4 * q;
            *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
            *reinterpret_cast<float2*>(dst + 2) = make_float2(v[2], v[3]);
        };

        // ---- prologue: planes 2 z0 - 1 .. 2 z0 + 2 in the ring (K = 3 uses three of them), step z0's small rows ----------
        __syncthreads();   // coef; the previous unit's last MFMAs are done with LDS
        load_big(2 * z0 - 1);
        store_big(2 * z0 - 1);
        load_big(2 * z0 + 1);
        store_big(2 * z0 + 1);
        load_small(z0);
        store_small();
        __syncthreads();

        for (int z = z0; z < z1; ++z) {
            const bool more = z + 1 < z1;
            if (more) {   // the next step's two new planes and its small rows: in flight during the MFMAs
                load_big(2 * z + K - 1);
                load_small(z + 1);
            }
            int toff[C::GPW];
#pragma unroll
            for (int i = 0; i < C::GPW; ++i) toff[i] = ((2 * z + tkz[i]) & 3) * C::PS + tbase[i];
            for (int rr = 0; rr < rows; ++rr) {
#pragma unroll 2
                for (int ks = 0; ks < R_TWG / 4; ++ks) {
                    const float af = arow[rr * R_TWG + ks * 4];
#pragma unroll
                    for (int i = 0; i < C::GPW; ++i) {
                        if (wave + 4 * i < C::GROUPS) {   // wave-uniform
                            const float bf = brow[toff[i] + 2 * rr * C::RS + ks * 4];
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();
            if (more) {
                store_big(2 * z + K - 1);   // into the slots of planes 2 z - 1 and 2 z (K = 3: one free slot and 2 z - 1)
                store_small();
                __syncthreads();
            }
        }
    }

    // ---- one partial per workgroup: [Cs][Cb][K^3] ------------------------------------------------------------
    float* dst = A.partial + (size_t)blockIdx.x * A.Cs * A.Cb * C::TAPS;
    const int bc = (lane & 15) & (CB - 1);
#pragma unroll
    for (int i = 0; i < C::GPW; ++i) {
        const int tap = (wave + 4 * i) * C::NT + tj;
        if (wave + 4 * i >= C::GROUPS || tap >= C::TAPS) continue;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int scn = 4 * (lane >> 4) + rr;
            if (scn < A.Cs && bc < A.Cb) dst[((size_t)scn * A.Cb + bc) * C::TAPS + tap] = acc[i][rr];
        }
    }
}

int launch_wgrad_reduce_f32(const float* partial, size_t wcount, int parts, float* dw, int accumulate, hipStream_t s);

namespace {

template <int K, int CB, bool SMALL_NORM, bool HAS_B>
void launch_s2r(const RArgs& A, int wgs, hipStream_t s) {
    using C = RCfg<K, CB>;
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done})
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3d_s2r_kernel<K, CB, SMALL_NORM, HAS_B>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C::LDS_FLOATS * sizeof(float)));
    hipLaunchKernelGGL((wgrad3d_s2r_kernel<K, CB, SMALL_NORM, HAS_B>), dim3(wgs), dim3(R_THREADS),
                       C::LDS_FLOATS * sizeof(float), s, A);
}

}  // namespace

// small / big: the grids of the contraction (wgrad3d_s2_mfma.hip: s2_roles); max_wgs: partial slots the scratch holds
bool wgrad3d_s2_rolling_supported(int transposed, const Src& b, const Geom& small, const Geom& big, const float* dz,
                                  const Src& a) {
    // PDS_WGRAD3D_S2_ROLLING=0 keeps the one-row-per-item kernel (A/B, debugging); =2 takes this one for every layer of
    // the right shape, however small (tests)
    static const int mode = []() {
        const char* e = debug_switch("PDS_WGRAD3D_S2_ROLLING");
        return e && e[0] == '0' ? 0 : e && e[0] == '2' ? 2 : 1;
    }();
    if (!mode || !(big.c == 4 || big.c == 8) || small.c > 16) return false;
    if ((big.w & 3) != 0 || (small.w & 3) != 0) return false;
    if (transposed && b.p) return false;
    if ((size_t)big.d * big.h * big.w * 8 >= ((size_t)1 << 31)) return false;   // 32-bit offsets inside a channel block
    if ((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(a.p) | reinterpret_cast<uintptr_t>(b.p)) & 15) return false;
    // the full-resolution layers only: a small level has too few units to fill the chip with 4-plane chunks
    return mode == 2 || small.d * ((small.h + R_TY - 1) / R_TY) * ((small.w + R_TWG - 1) / R_TWG) * small.n >= 512;
}

int launch_wgrad3d_s2_rolling(int transposed, const Src& a, const Src& b, const float* dz, float* dw, const Geom& small,
                              const Geom& big, int accumulate, float* scratch, int max_wgs, hipStream_t s) {
    RArgs A;
    A.a = a;
    A.b = b;
    A.dz = dz;
    A.partial = scratch;
    A.N = small.n;
    A.Cs = small.c;
    A.Cb = big.c;
    A.Ds = small.d;
    A.Hs = small.h;
    A.Ws = small.w;
    A.Db = big.d;
    A.Hb = big.h;
    A.Wb = big.w;
    A.segs = (small.w + R_TWG - 1) / R_TWG;
    A.yblocks = (small.h + R_TY - 1) / R_TY;
    const int columns = small.n * A.yblocks * A.segs;
    // chunks of small planes: a chunk stages K - 2 extra planes once; ~3 units per persistent workgroup
    const int resident = big.c == 4 ? 1280 : 768;   // 30 KB / 56 KB of LDS per workgroup: 5 / 3 (2 for K = 4) per CU
    int wgs = max_wgs < resident ? max_wgs : resident;
    int zc = 8;
    while (zc > 2 && columns * ((small.d + zc - 1) / zc) < 3 * wgs) zc -= 2;
    A.zc = zc;
    A.zchunks = (small.d + zc - 1) / zc;
    A.units = columns * A.zchunks;
    if (wgs > A.units) wgs = A.units;
    const int taps = transposed ? 64 : 27;
    if (transposed) {
        if (big.c == 4) launch_s2r<4, 4, true, false>(A, wgs, s);
        else launch_s2r<4, 8, true, false>(A, wgs, s);
    } else if (b.p) {
        if (big.c == 4) launch_s2r<3, 4, false, true>(A, wgs, s);
        else launch_s2r<3, 8, false, true>(A, wgs, s);
    } else {
        if (big.c == 4) launch_s2r<3, 4, false, false>(A, wgs, s);
        else launch_s2r<3, 8, false, false>(A, wgs, s);
    }
    if (int rc = check_launch("wgrad3d_s2_rolling")) return rc;
    return launch_wgrad_reduce_f32(scratch, (size_t)small.c * big.c * taps, wgs, dw, accumulate, s);
}

}  // namespace pds
