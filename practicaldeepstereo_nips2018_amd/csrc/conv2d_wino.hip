// Winograd-domain variant of the dominant kernel: 3x3 stride-1 convolution Cin -> 64 over a stack of 2-D
// planes (the 64 -> 64 layers of reference practical_deep_stereo/matching.py:85-88, network_blocks.py:47-58,
// 97-103, 134-144), exact-fp32 MFMA, F(2,3) along x.
//
// Why: the direct kernel (conv2d_mfma.hip) sits on the chip's power/clock plateau -- issue
// efficiency gained is returned as clock -- so the lever left is fewer MFMAs per output.  With the 1-D minimal
// filtering algorithm F(2,3) two neighbouring outputs of a row cost 4 multiplies per (dy, ic, oc) instead of 6:
//
//   input  d0..d3 = x[2t-1 .. 2t+2]         V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3
//   filter g0..g2 = w[dy][-1, 0, +1]        U0 = g0   U1 = (g0 + g1 + g2) / 2   U2 = (g0 - g1 + g2) / 2   U3 = g2
//   M_p = sum over (dy, ic) of U_p * V_p    out[2t] = M0 + M1 + M2      out[2t + 1] = M1 - M2 - M3
//
//   GEMM view   per position p and row offset dy: M = output channels (16 per MFMA block, 4 blocks),
//               N = 16 consecutive Winograd tiles (32 pixels) of a row, K = Cin; v_mfma_f32_16x16x4_f32.
//               12 (p, dy) products per 2 pixels instead of 9 taps per pixel: 2/3 of the MFMAs.
//   workgroup   4 waves; output tile 4 rows x 64 columns (32 tiles) of one (n, d) plane, all 64 channels;
//               wave r owns row r: accumulators [4 positions][4 channel blocks][2 tile blocks] = 128 VGPRs.
//   LDS         double-buffered [V of 4 channels: [ic][6 rows][4 positions][32 tiles] | U fragments of the chunk];
//               the input transform (with the producer's deferred InstanceNorm and the literal zero padding)
//               is applied while staging; the filter transform is part of the weight packing (pack.hip mode 3).
//   pipeline    two chunks deep: chunk c + 2 is being loaded, chunk c + 1 is transformed and written to the idle
//               LDS buffer in the shadow of the MFMAs of chunk c (one basic block per chunk, sched_group_barrier
//               interleave); one barrier per chunk.
//   epilogue    output transform, + bias, LeakyReLU(0.1), 8-byte stores, per-(plane, channel) sum / sum of
//               squares partials in fp64 (one deterministic record per tile), as in conv2d_mfma.hip.
// Rounding: transforms are additions and one exact halving; measured against the fp64 oracle the layer is as
// close as the direct kernel to within a factor ~2 (tests/test_gpu_parity.py).
#include "common.hpp"

namespace pds {

namespace {

constexpr int TWX = 64, NT = TWX / 2, NBT = NT / 16, KC = 4, MB = 4;
constexpr int PS = NT;                    // floats between the 4 positions of one row
constexpr int RSV = 4 * PS;               // row stride
constexpr int W_CHUNK = 12 * MB * 64;     // 12 (dy, p) products x 4 channel blocks x 64 lanes
// TH rows per tile = row-waves per workgroup: 4, or 6 for launches of a few hundred tiles (round 5, launch_conv2d_wino)
template <int TH>
struct WinoGeom {
    static constexpr int CS = (TH + 2) * RSV + 16;   // channel stride, == 16 (mod 32): the two k-halves of a 32-lane group
                                                     // read disjoint banks
    static constexpr int IN_CHUNK = KC * CS;
    static constexpr int BUF = IN_CHUNK + W_CHUNK;
    static constexpr int ITEMS = KC * (TH + 2) * NT;  // (channel, row, tile) items per chunk: 768 / 1 024
    static_assert(ITEMS % 64 == 0 && ((TH + 2) * NT) % 64 == 0, "a wave stages whole rows of one channel");
    static_assert(CS % 32 == 16, "bank layout");
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// timing-decomposition hooks (tools/build_variant.sh): any of them set makes the results wrong
#ifdef PDS_X_NOLOADIN
#define PDS_X_LOADIN(x) (float)(tid)
#define PDS_X_LOADIN2(x) make_float2((float)tid, 1.f)
#else
#define PDS_X_LOADIN(x) (x)
#define PDS_X_LOADIN2(x) (x)
#endif
#ifdef PDS_X_NOLOADW
#define PDS_X_LOADW(x) f32x4{(float)tid, 1.f, 2.f, 3.f}
#else
#define PDS_X_LOADW(x) (x)
#endif
#ifdef PDS_X_NOMFMA
#define PDS_X_MFMA(c, a, b) (c)[0] += (a) * (b)
#else
#define PDS_X_MFMA(c, a, b) (c) = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif

struct WinoArgs {
    Src a;
    const float* __restrict__ wpk;   // [chunk][dy*4 + p][mb][64 lanes]
    const float* __restrict__ bias;
    float* __restrict__ out;
    double* __restrict__ partials;
    int N, Cin, D, H, W, Cout;
    int CoutStride;                  // channels per batch entry of the output tensor (>= Cout)
    int lrelu;
    int tiles_x, tiles;
    size_t w_set_stride;             // floats between the packed weight sets of consecutive planes (0: shared)
    int bias_set_stride;             // ditto for the bias
};

__device__ __forceinline__ float row16_sum_w(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

// lane i receives lane i - 1 (DPP wave_shr:1) / lane i + 1 (DPP wave_shl:1): whole-wave shifts of GFX9, no LDS
__device__ __forceinline__ float wave_shift_up(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_shift_down(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

}  // namespace

// NORM: the source carries a deferred InstanceNorm (scale / shift per channel or per (channel, plane)).
// HALVES: 1 = 4 waves, wave r owns row r and all 64 output channels (128 accumulator registers, 2 waves/SIMD);
//         2 = 8 waves, wave (r, half) owns row r and 32 output channels (64 accumulator registers, 4 waves/SIMD).
template <bool NORM, int HALVES, int TH = 4>
__global__ __launch_bounds__(64 * TH * HALVES, TH == 4 ? 2 * HALVES : (TH * HALVES) / 4)
void conv2d_wino_kernel(const WinoArgs A) {
    using WG = WinoGeom<TH>;
    constexpr int CS = WG::CS, IN_CHUNK = WG::IN_CHUNK, BUF = WG::BUF, ITEMS = WG::ITEMS;
    constexpr int THREADS = 64 * TH * HALVES;
    constexpr int MBW = MB / HALVES;                              // channel blocks per wave
    constexpr int IPT = (ITEMS + THREADS - 1) / THREADS;          // 3 or 2 (the last one only on waves 0-3)
    constexpr int W_ITERS = (W_CHUNK / 4 + THREADS - 1) / THREADS;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2 buffers][V chunk | U chunk]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_id % TH, half = wave_id / TH;
    // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in launch order, so
    // the launch index is re-mapped to make every XCD work on whole planes: neighbouring tiles (shared halo rows)
    // and the plane's statistics stay in one L2.  Needs D % 8 == 0; otherwise the identity mapping.
    int tile = blockIdx.x, d = blockIdx.y;
    const int n = blockIdx.z;
#ifndef PDS_WINO_NO_XCD_MAP
    if ((A.D & 7) == 0) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = lin & 7, slot = lin >> 3;
        d = (slot / A.tiles) * 8 + xcd;
        tile = slot % A.tiles;
    }
#endif
    const int ty = tile / A.tiles_x, tx = tile % A.tiles_x;
    const int y0 = ty * TH, x0 = tx * TWX;
    const size_t plane = (size_t)A.H * A.W;
    const size_t cstride = (size_t)A.D * plane;
    const int nchunks = A.Cin / KC;

    // ---- staging map: thread -> IPT items (channel c of the chunk, halo row r, tile t).  64 | 6 * 32, so a wave's
    // 64 consecutive items are two full rows of one channel and t == lane & 31 for every item.  An item loads the
    // aligned pair (d1, d2) = x[x0 + 2t, x0 + 2t + 1] with ONE 8-byte load (fully coalesced rows); d0 and d3 are the
    // neighbouring lanes' d2 / d1, except at the ends of the row segment (t == 0 / 31), whose halo value comes from
    // one extra 4-byte load.  Addresses are clamped, padding is a select, offsets are 32-bit relative to a pointer
    // that is uniform per chunk.
    unsigned offp[IPT], offe[IPT], goff[IPT];
    int l_off[IPT];
    bool in1[IPT], in2[IPT], ine[IPT];
    const unsigned gstride = NORM ? (A.a.per_plane ? (unsigned)A.D : 1u) : 0u;
    const int t = lane & 31;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int e = min(tid + k * THREADS, ITEMS - 1);  // surplus items (8-wave form) are skipped below
        const int c = e / ((TH + 2) * NT), r = (e % ((TH + 2) * NT)) / NT;
        const int y = y0 - 1 + r;
        const bool rowok = y >= 0 && y < A.H;
        const int yc = min(max(y, 0), A.H - 1);
        const int x = x0 + 2 * t;                       // d1; d2 = x + 1 (W is even: both or neither inside)
        const int xe = t == 0 ? x0 - 1 : x0 + TWX;      // halo value of the row segment (used by t == 0 / 31)
        in1[k] = rowok && x < A.W;
        in2[k] = rowok && x + 1 < A.W;
        ine[k] = rowok && xe >= 0 && xe < A.W && (t == 0 || t == NT - 1);
        const unsigned rowbase = (unsigned)c * (unsigned)cstride + (unsigned)(yc * A.W);
        offp[k] = rowbase + (unsigned)min(x, A.W - 2);
        offe[k] = rowbase + (unsigned)((t == 0 || t == NT - 1) ? min(max(xe, 0), A.W - 1) : min(x, A.W - 2));
        // (a wave's 64 consecutive items lie in one channel: the offset of its scale / shift is wave-uniform, so these
        // become scalar loads)
        goff[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)c * gstride));
        l_off[k] = c * CS + r * RSV + t;
    }
    const float* pa = A.a.p + ((size_t)n * A.Cin * A.D + d) * plane;
    const size_t chunk_stride = (size_t)KC * cstride;
    const float* ps = NORM ? A.a.scale + (A.a.per_plane ? ((size_t)n * A.Cin * A.D + d) : (size_t)n * A.Cin) : nullptr;
    const float* ph = NORM ? A.a.shift + (A.a.per_plane ? ((size_t)n * A.Cin * A.D + d) : (size_t)n * A.Cin) : nullptr;
    const int wlast = W_CHUNK / 4 - 1;
    const float* wbase = A.wpk + (size_t)d * A.w_set_stride;
    const float* bias = A.bias ? A.bias + d * A.bias_set_stride : nullptr;

    float2 vp[IPT];
    float ve[IPT], vs[IPT], vh[IPT];
    f32x4 vw[W_ITERS];

#define PDS_WFETCH(chunk_)                                                                           \
    {                                                                                                \
        const float* src = pa + (size_t)(chunk_) * chunk_stride;          /* uniform */              \
        const float* ssrc = NORM ? ps + (size_t)(chunk_) * KC * gstride : nullptr;                   \
        const float* hsrc = NORM ? ph + (size_t)(chunk_) * KC * gstride : nullptr;                   \
        _Pragma("unroll") for (int k = 0; k < IPT; ++k) {                                            \
            if ((k + 1) * THREADS > ITEMS && tid + k * THREADS >= ITEMS) continue;                   \
            vp[k] = PDS_X_LOADIN2(*reinterpret_cast<const float2*>(src + offp[k]));                  \
            ve[k] = PDS_X_LOADIN(src[offe[k]]);                                                      \
            if (NORM) {                                                                              \
                vs[k] = ssrc[goff[k]];                                                               \
                vh[k] = hsrc[goff[k]];                                                               \
            }                                                                                        \
        }                                                                                            \
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(wbase + (size_t)(chunk_) * W_CHUNK);       \
        _Pragma("unroll") for (int it = 0; it < W_ITERS; ++it)                                       \
            vw[it] = PDS_X_LOADW(wsrc[min(it * THREADS + tid, wlast)]);                               \
    }

#define PDS_WSTASH(buf_)                                                                             \
    {                                                                                                \
        _Pragma("unroll") for (int k = 0; k < IPT; ++k) {                                            \
            if ((k + 1) * THREADS > ITEMS && tid + k * THREADS >= ITEMS) continue;                   \
            const float d1 = in1[k] ? (NORM ? fmaf(vs[k], vp[k].x, vh[k]) : vp[k].x) : 0.f;          \
            const float d2 = in2[k] ? (NORM ? fmaf(vs[k], vp[k].y, vh[k]) : vp[k].y) : 0.f;          \
            const float de = ine[k] ? (NORM ? fmaf(vs[k], ve[k], vh[k]) : ve[k]) : 0.f;              \
            const float up = wave_shift_up(d2), dn = wave_shift_down(d1);                            \
            const float d0 = t == 0 ? de : up;                                                       \
            const float d3 = t == NT - 1 ? de : dn;                                                  \
            float* dst = (buf_) + l_off[k];                                                          \
            dst[0 * PS] = d0 - d2;                                                                   \
            dst[1 * PS] = d1 + d2;                                                                   \
            dst[2 * PS] = d2 - d1;                                                                   \
            dst[3 * PS] = d1 - d3;                                                                   \
        }                                                                                            \
        f32x4* wdst = reinterpret_cast<f32x4*>((buf_) + IN_CHUNK);                                   \
        _Pragma("unroll") for (int it = 0; it < W_ITERS; ++it)                                       \
            wdst[min(it * THREADS + tid, wlast)] = vw[it];                                           \
    }

    f32x4 acc[4][MBW][NBT];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int m = 0; m < MBW; ++m)
#pragma unroll
            for (int j = 0; j < NBT; ++j) acc[p][m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Software pipeline, two chunks deep: while the MFMAs of chunk c run out of one LDS buffer, the registers fetched
    // during chunk c - 1 (chunk c + 1) are transformed and written to the other buffer IN THEIR SHADOW (the body is one
    // basic block; sched_group_barrier spreads the riders between the MFMAs), and the global loads of chunk c + 2 are
    // issued.  Chunks past the end re-stage the last one into the idle buffer instead of branching.
    const int last_chunk = nchunks - 1;
    PDS_WFETCH(0)
    PDS_WSTASH(lds)
    PDS_WFETCH(min(1, last_chunk))
    __syncthreads();

    const int b_lane = (lane >> 4) * CS + wave * RSV + (lane & 15);

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        float* buf = lds + (chunk & 1) * BUF;
        float* nxt = lds + ((chunk + 1) & 1) * BUF;
        const float* xin = buf + b_lane;
        const float* win = buf + IN_CHUNK + half * MBW * 64 + lane;
        PDS_WSTASH(nxt)                               // chunk + 1, fetched one iteration ago
        PDS_WFETCH(min(chunk + 2, last_chunk))        // lands during the next iteration
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float af[MBW], bf[NBT];
#pragma unroll
                for (int m = 0; m < MBW; ++m) af[m] = win[((dy * 4 + p) * MB + m) * 64];
#pragma unroll
                for (int j = 0; j < NBT; ++j) bf[j] = xin[dy * RSV + p * PS + j * 16];
#pragma unroll
                for (int m = 0; m < MBW; ++m)
#pragma unroll
                    for (int j = 0; j < NBT; ++j)
                        PDS_X_MFMA(acc[p][m][j], af[m], bf[j]);
            }
        }
#pragma unroll
        for (int i = 0; i < 12 * MBW * NBT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU (input transform of the next chunk)
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
        }
        __syncthreads();
    }
#undef PDS_WFETCH
#undef PDS_WSTASH

    // ---- epilogue: output transform, bias, LeakyReLU, store, statistics ----------------------------------
    const int y = y0 + wave;
    const bool rowok = y < A.H;
    const int jx = lane & 15, q = lane >> 4;
    const bool pairs = (A.W & 1) == 0;  // rows start 8-byte aligned
    float* red = lds;  // [TH rows][64 channels][2]
#pragma unroll
    for (int m = 0; m < MBW; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = (half * MBW + m) * 16 + q * 4 + r;
            const float bv = bias ? bias[oc] : 0.f;
            float* po = A.out + (((size_t)n * A.CoutStride + oc) * A.D + d) * plane + (size_t)y * A.W;
            float s = 0.f, sq = 0.f;
#pragma unroll
            for (int j = 0; j < NBT; ++j) {
                const int x = x0 + 2 * (j * 16 + jx);
                const float m0 = acc[0][m][j][r], m1 = acc[1][m][j][r], m2 = acc[2][m][j][r], m3 = acc[3][m][j][r];
                float t0 = (m0 + m1) + m2 + bv;
                float t1 = (m1 - m2) - m3 + bv;
                if (A.lrelu) {
                    t0 = t0 > 0.f ? t0 : t0 * kLeakySlope;
                    t1 = t1 > 0.f ? t1 : t1 * kLeakySlope;
                }
                if (rowok && x + 1 < A.W && pairs) {
                    *reinterpret_cast<float2*>(po + x) = make_float2(t0, t1);
                    s += t0 + t1;
                    sq = fmaf(t0, t0, fmaf(t1, t1, sq));
                } else if (rowok) {
                    if (x < A.W) {
                        po[x] = t0;
                        s += t0;
                        sq = fmaf(t0, t0, sq);
                    }
                    if (x + 1 < A.W) {
                        po[x + 1] = t1;
                        s += t1;
                        sq = fmaf(t1, t1, sq);
                    }
                }
            }
            if (A.partials) {
                s = row16_sum_w(s);
                sq = row16_sum_w(sq);
                if (jx == 15) {
                    red[((wave * MB * 16) + oc) * 2 + 0] = s;
                    red[((wave * MB * 16) + oc) * 2 + 1] = sq;
                }
            }
        }
    }
    if (A.partials) {
        __syncthreads();
        if (tid < MB * 16 * 2) {
            const int oc = tid >> 1, k = tid & 1;
            double v = 0.0;
#pragma unroll
            for (int wv = 0; wv < TH; ++wv) v += (double)red[((wv * MB * 16) + oc) * 2 + k];
            A.partials[((((size_t)n * A.Cout + oc) * A.D + d) * A.tiles + tile) * 2 + k] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
bool conv2d_wino_eligible(const ConvLayer& L) {
    static const bool enabled = []() {  // PDS_WINOGRAD=0 selects the direct kernel (A/B, debugging)
        const char* e = debug_switch("PDS_WINOGRAD");
        return !(e && e[0] == '0');
    }();
    if (!enabled) return false;
    if (L.kd != 1 || L.stride != 1 || L.out_g.c != 64) return false;
    if (L.in.c % KC != 0 || L.in.c > 256) return false;
    if (L.b.p || L.l0A || L.side_out) return false;
    if (L.plane_weight_sets > 0 && (L.plane_weight_sets != L.in.d || L.plane_weight_sets > 8)) return false;
    if (L.in.w % 2 != 0 || L.in.w < 2) return false;  // rows are read as aligned 8-byte pairs
    if ((size_t)L.in.d * L.in.h * L.in.w * KC >= ((size_t)1 << 31)) return false;
    if (L.in.d > 65535 || L.in.n > 65535) return false;
    return true;
}

// conv2d_wino16.hip: the 16 x 16-tile form of this kernel (same packed weights), preferred where it covers the plane
// with fewer workgroups
int conv2d_wino16_tiles(int h, int w);
bool conv2d_wino16_preferred(int h, int w);
int launch_conv2d_wino16(const ConvLayer& L, size_t w_set_stride, int bias_set_stride, hipStream_t s);

// Rows per tile.  A workgroup of the 4-row form keeps the fp32 matrix pipe of its CU busy for ~22 us per 64 input channels,
// and two of them on one CU share that pipe: the 2-plane launches of Matching's front end (288 workgroups on 256 CUs) took
// 66 us because 32 CUs held two.  With 6-row tiles the same launch is 192 workgroups of 1.5 x the work, one per CU.  The
// rule compares (rounds of workgroups over the CUs) x (rows per tile); large launches keep the 4-row form.
static int wino_rows(const Geom& o) {
    static const bool enabled = []() {  // PDS_WINO_ROWS6=0: 4-row tiles everywhere (A/B)
        const char* e = debug_switch("PDS_WINO_ROWS6");
        return !(e && e[0] == '0');
    }();
    if (!enabled) return 4;
    const long long planes = (long long)o.n * o.d, tx = (o.w + TWX - 1) / TWX;
    const long long w4 = planes * tx * ((o.h + 3) / 4), w6 = planes * tx * ((o.h + 5) / 6);
    const long long cus = 256;
    const long long t4 = ((w4 + cus - 1) / cus) * 4, t6 = ((w6 + cus - 1) / cus) * 6;
    return t6 < t4 ? 6 : 4;
}

int conv2d_wino_tiles(const Geom& o) {
    if (conv2d_wino16_preferred(o.h, o.w)) return conv2d_wino16_tiles(o.h, o.w);
    const int th = wino_rows(o);
    return ((o.h + th - 1) / th) * ((o.w + TWX - 1) / TWX);
}

size_t conv2d_wino_packed_floats(int cin, int cout) { return (size_t)(cin / KC) * W_CHUNK; }

int launch_conv2d_wino(const ConvLayer& L, hipStream_t s) {
    if (!L.packed) return set_error(-1, "conv2d_wino: packed weights missing");
    const int total = (int)conv2d_wino_packed_floats(L.in.c, L.out_g.c);
    const int sets = L.plane_weight_sets > 0 ? L.plane_weight_sets : 1;
    const PackPhase phase = L.sink ? L.sink->phase : kPackInline;
    if (phase != kPackDone) {
        PackJob jobs[8];
        for (int i = 0; i < sets; ++i) {
            PackJob& j = jobs[i];
            j.src = L.weight + (size_t)i * L.out_g.c * L.in.c * 9;
            j.dst = L.packed + (size_t)i * total;
            j.cout = L.out_g.c;
            j.cin = L.in.c;
            j.mblocks = MB;
            j.kc = KC;
            j.taps = 12;
            j.mode = 3;  // F(2,3) filter transform along x
            j.total = total;
            if (phase == kPackCollect && !L.sink->push(j)) return set_error(-1, "pack job table full");
        }
        if (phase == kPackCollect) return 0;
        if (int rc = launch_multi_pack(jobs, sets, s)) return rc;
    }
    if (conv2d_wino16_preferred(L.in.h, L.in.w))
        return launch_conv2d_wino16(L, L.plane_weight_sets > 0 ? (size_t)total : 0,
                                    L.plane_weight_sets > 0 ? L.out_g.c : 0, s);
    WinoArgs A;
    A.a = L.a;
    A.wpk = L.packed;
    A.bias = L.bias;
    A.out = L.out;
    A.partials = L.partials;
    A.N = L.in.n;
    A.Cin = L.in.c;
    A.D = L.in.d;
    A.H = L.in.h;
    A.W = L.in.w;
    A.Cout = L.out_g.c;
    A.CoutStride = L.out_batch_channels > 0 ? L.out_batch_channels : L.out_g.c;
    A.lrelu = L.lrelu;
    A.tiles_x = (A.W + TWX - 1) / TWX;
    A.tiles = conv2d_wino_tiles(L.out_g);
    A.w_set_stride = L.plane_weight_sets > 0 ? (size_t)total : 0;
    A.bias_set_stride = L.plane_weight_sets > 0 ? L.out_g.c : 0;
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done}) {
        const int bytes = (int)(160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_wino_kernel<true, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_wino_kernel<false, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_wino_kernel<true, 2, 6>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_wino_kernel<false, 2, 6>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    }
    const dim3 grid(A.tiles, A.D, A.N);
    if (wino_rows(L.out_g) == 6) {
        const size_t lds_bytes = (size_t)2 * WinoGeom<6>::BUF * sizeof(float);
        if (L.a.scale) hipLaunchKernelGGL((conv2d_wino_kernel<true, 2, 6>), grid, dim3(768), lds_bytes, s, A);
        else hipLaunchKernelGGL((conv2d_wino_kernel<false, 2, 6>), grid, dim3(768), lds_bytes, s, A);
        return check_launch("conv2d_wino");
    }
    const size_t lds_bytes = (size_t)2 * WinoGeom<4>::BUF * sizeof(float);
    if (L.a.scale) hipLaunchKernelGGL((conv2d_wino_kernel<true, 2>), grid, dim3(512), lds_bytes, s, A);
    else hipLaunchKernelGGL((conv2d_wino_kernel<false, 2>), grid, dim3(512), lds_bytes, s, A);
    return check_launch("conv2d_wino");
}

}  // namespace pds
