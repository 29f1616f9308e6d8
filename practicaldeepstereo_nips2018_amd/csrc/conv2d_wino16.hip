// Square-tile form of the Winograd-domain 64-channel convolution (conv2d_wino.hip: same algorithm F(2,3) along x, same
// packed filter transform, same pipeline), for images whose width is not a multiple of 64: reference
// practical_deep_stereo/matching.py:85-88 at 960x540 works on 144 x 240 planes, where 4 x 64 tiles compute 256 columns
// for 240 (6.25 % of the MFMAs of the dominant kernel are spent outside the image).  Here a workgroup owns 16 rows x
// 16 columns -- 9 x 15 tiles cover the plane exactly -- and the N side of an MFMA block is 2 rows x 8 Winograd tiles:
//
//   workgroup   8 waves; wave (r, half) owns rows 4r .. 4r+3 (two N blocks) and 32 output channels: the same
//               [4 positions][2 channel blocks][2 N blocks] accumulators as the 4 x 64 form, 4 waves per SIMD.
//   LDS         V[ic][18 rows][4 positions][8 tiles], row stride 40, channel stride 720 == 16 (mod 32): the fragment
//               read of a 32-lane group (2 rows x 8 tiles x 2 channels) and the staging writes are conflict free.
//   staging     one item = (channel, halo row, PAIR of Winograd tiles) = one aligned 16-byte load; the inner halo values
//               of the two tiles are the lane's own, the outer ones come from the neighbouring lanes (DPP) or, at the two
//               ends of the 16-column segment, from one extra 4-byte load.  288 items per 4-channel chunk: waves 0-4
//               stage, waves 5-7 go straight to their MFMAs.  Halo overhead (18 x 18) / (16 x 16) = 1.27 against
//               (6 x 66) / (4 x 64) = 1.55 of the wide form.
// Needs W % 4 == 0 (aligned 16-byte loads); chosen by conv2d_wino16_preferred() when it computes fewer pixels.
#include "common.hpp"

namespace pds {

namespace {

constexpr int TH = 16, TWX = 16, NT = TWX / 2, KC = 4, MB = 4;
constexpr int ROWS = TH + 2;
constexpr int PS = NT;                    // floats between the 4 positions of one row
constexpr int RSV = 4 * PS + 8;           // row stride 40: rows r and r + 1 of an N block land in disjoint banks
constexpr int CS = ROWS * RSV;            // channel stride 720 == 16 (mod 32)
constexpr int IN_CHUNK = KC * CS;
constexpr int W_CHUNK = 12 * MB * 64;     // 12 (dy, p) products x 4 channel blocks x 64 lanes
constexpr int BUF = IN_CHUNK + W_CHUNK;
constexpr int ITEMS = KC * ROWS * (NT / 2);   // 288 (channel, row, tile pair) items per chunk
constexpr int THREADS = 512;
constexpr int MBW = MB / 2;               // channel blocks per wave
constexpr int NBT = 2;                    // N blocks per wave
constexpr int W_ITERS = (W_CHUNK / 4 + THREADS - 1) / THREADS;
static_assert(CS % 32 == 16, "bank layout");
static_assert(ITEMS <= THREADS, "one staging item per thread");

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Wino16Args {
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
    int bias_set_stride;
};

__device__ __forceinline__ float row16_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ float lane_above(float v) {   // lane i receives lane i - 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_below(float v) {   // lane i receives lane i + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

}  // namespace

template <bool NORM>
__global__ __launch_bounds__(THREADS, 2) void conv2d_wino16_kernel(const Wino16Args A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2 buffers][V chunk | U chunk]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_id & 3, half = wave_id >> 2;
    // every XCD works on whole planes (see conv2d_wino.hip)
    int tile = blockIdx.x, d = blockIdx.y;
    const int n = blockIdx.z;
    if ((A.D & 7) == 0) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = lin & 7, slot = lin >> 3;
        d = (slot / A.tiles) * 8 + xcd;
        tile = slot % A.tiles;
    }
    const int ty = tile / A.tiles_x, tx = tile % A.tiles_x;
    const int y0 = ty * TH, x0 = tx * TWX;
    const size_t plane = (size_t)A.H * A.W;
    const size_t cstride = (size_t)A.D * plane;
    const int nchunks = A.Cin / KC;

    // ---- staging map: thread -> (channel c of the chunk, halo row r, tile pair tp): 4 consecutive lanes are one row
    const bool stager = tid < ITEMS;
    const int e = stager ? tid : 0;
    const int sc = e / (ROWS * 4), srem = e % (ROWS * 4);
    const int sr = srem >> 2, tp = srem & 3;
    const int sy = y0 - 1 + sr;
    const bool rowok = sy >= 0 && sy < A.H;
    const int syc = min(max(sy, 0), A.H - 1);
    const int sx = x0 + 4 * tp;                        // 4 inputs x .. x + 3 (W % 4 == 0: all inside or all outside)
    const int xe = tp == 0 ? x0 - 1 : x0 + TWX;        // outer halo column (used by tp == 0 / 3)
    const bool inx = rowok && sx < A.W;
    const bool ine = rowok && xe >= 0 && xe < A.W && (tp == 0 || tp == 3);
    const unsigned rowbase = (unsigned)sc * (unsigned)cstride + (unsigned)(syc * A.W);
    const unsigned offp = rowbase + (unsigned)min(sx, A.W - 4);
    const unsigned offe = rowbase + (unsigned)min(max(xe, 0), A.W - 1);
    const unsigned gstride = NORM ? (A.a.per_plane ? (unsigned)A.D : 1u) : 0u;
    const unsigned goff = (unsigned)sc * gstride;
    const int l_off = sc * CS + sr * RSV + 2 * tp;

    const float* pa = A.a.p + ((size_t)n * A.Cin * A.D + d) * plane;
    const size_t chunk_stride = (size_t)KC * cstride;
    const float* ps = NORM ? A.a.scale + (A.a.per_plane ? ((size_t)n * A.Cin * A.D + d) : (size_t)n * A.Cin) : nullptr;
    const float* ph = NORM ? A.a.shift + (A.a.per_plane ? ((size_t)n * A.Cin * A.D + d) : (size_t)n * A.Cin) : nullptr;
    const int wlast = W_CHUNK / 4 - 1;
    const float* wbase = A.wpk + (size_t)d * A.w_set_stride;
    const float* bias = A.bias ? A.bias + d * A.bias_set_stride : nullptr;

    // one register set (SETS = 2 with the request of chunk c + 3 issued two MFMA phases ahead was measured: no
    // difference, 0.78 ms either way -- the loads are not what the loop waits for)
    constexpr int SETS = 1;
    f32x4 vp[SETS];
    float ve[SETS] = {0.f}, vs[SETS] = {1.f}, vh[SETS] = {0.f};
    f32x4 vw[SETS][W_ITERS];

#define PDS_W16_FETCH(chunk_, S)                                                                     \
    {                                                                                                \
        const float* src = pa + (size_t)(chunk_) * chunk_stride;          /* uniform */              \
        if (stager) {                                                                                \
            vp[S] = *reinterpret_cast<const f32x4*>(src + offp);                                     \
            ve[S] = src[offe];                                                                       \
            if (NORM) {                                                                              \
                vs[S] = ps[(size_t)(chunk_) * KC * gstride + goff];                                  \
                vh[S] = ph[(size_t)(chunk_) * KC * gstride + goff];                                  \
            }                                                                                        \
        }                                                                                            \
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(wbase + (size_t)(chunk_) * W_CHUNK);       \
        _Pragma("unroll") for (int it = 0; it < W_ITERS; ++it)                                       \
            vw[S][it] = wsrc[min(it * THREADS + tid, wlast)];                                         \
    }

#define PDS_W16_STASH(buf_, S)                                                                       \
    {                                                                                                \
        if (stager) {                                                                                \
            const float a0 = inx ? (NORM ? fmaf(vs[S], vp[S][0], vh[S]) : vp[S][0]) : 0.f;           \
            const float a1 = inx ? (NORM ? fmaf(vs[S], vp[S][1], vh[S]) : vp[S][1]) : 0.f;           \
            const float a2 = inx ? (NORM ? fmaf(vs[S], vp[S][2], vh[S]) : vp[S][2]) : 0.f;           \
            const float a3 = inx ? (NORM ? fmaf(vs[S], vp[S][3], vh[S]) : vp[S][3]) : 0.f;           \
            const float de = ine ? (NORM ? fmaf(vs[S], ve[S], vh[S]) : ve[S]) : 0.f;                 \
            const float up = lane_above(a3), dn = lane_below(a0);                                    \
            const float left = tp == 0 ? de : up;       /* x - 1 */                                  \
            const float right = tp == 3 ? de : dn;      /* x + 4 */                                  \
            float* dst = (buf_) + l_off;                                                             \
            /* tile A: d0..d3 = left, a0, a1, a2;  tile B: d0..d3 = a1, a2, a3, right */             \
            *reinterpret_cast<float2*>(dst + 0 * PS) = make_float2(left - a1, a1 - a3);              \
            *reinterpret_cast<float2*>(dst + 1 * PS) = make_float2(a0 + a1, a2 + a3);                \
            *reinterpret_cast<float2*>(dst + 2 * PS) = make_float2(a1 - a0, a3 - a2);                \
            *reinterpret_cast<float2*>(dst + 3 * PS) = make_float2(a0 - a2, a2 - right);             \
        }                                                                                            \
        f32x4* wdst = reinterpret_cast<f32x4*>((buf_) + IN_CHUNK);                                   \
        _Pragma("unroll") for (int it = 0; it < W_ITERS; ++it)                                       \
            wdst[min(it * THREADS + tid, wlast)] = vw[S][it];                                        \
    }

    f32x4 acc[4][MBW][NBT];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int m = 0; m < MBW; ++m)
#pragma unroll
            for (int j = 0; j < NBT; ++j) acc[p][m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Software pipeline, as conv2d_wino.hip: chunk c + 1 (requested one iteration ago) is transformed into the idle
    // LDS buffer and chunk c + 2 is requested before the MFMAs of chunk c.  Chunks past the end re-stage the last one
    // instead of branching.
    const int last_chunk = nchunks - 1;
    const int nn = lane & 15;
    const int b_lane = (lane >> 4) * CS + (4 * wave + (nn >> 3)) * RSV + (nn & 7);

#define PDS_W16_MFMAS(buf)                                                                           \
    {                                                                                                \
        const float* xin = (buf) + b_lane;                                                           \
        const float* win = (buf) + IN_CHUNK + half * MBW * 64 + lane;                                \
        _Pragma("unroll") for (int dy = 0; dy < 3; ++dy) {                                           \
            _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                          \
                float af[MBW], bf[NBT];                                                              \
                _Pragma("unroll") for (int m = 0; m < MBW; ++m) af[m] = win[((dy * 4 + p) * MB + m) * 64]; \
                _Pragma("unroll") for (int j = 0; j < NBT; ++j) bf[j] = xin[(dy + 2 * j) * RSV + p * PS]; \
                _Pragma("unroll") for (int m = 0; m < MBW; ++m)                                      \
                    _Pragma("unroll") for (int j = 0; j < NBT; ++j)                                  \
                        acc[p][m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[j], acc[p][m][j], 0, 0, 0); \
            }                                                                                        \
        }                                                                                            \
    }

    PDS_W16_FETCH(0, 0)
    PDS_W16_STASH(lds, 0)
    PDS_W16_FETCH(min(1, last_chunk), 0)
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        float* buf = lds + (chunk & 1) * BUF;
        float* nxt = lds + ((chunk + 1) & 1) * BUF;
        PDS_W16_STASH(nxt, 0)                               // chunk + 1, fetched one iteration ago
        PDS_W16_FETCH(min(chunk + 2, last_chunk), 0)        // lands during the next iteration
        PDS_W16_MFMAS(buf)
        __syncthreads();
    }
#undef PDS_W16_MFMAS
#undef PDS_W16_FETCH
#undef PDS_W16_STASH

    // ---- epilogue: output transform, bias, LeakyReLU, store, statistics ----------------------------------
    const int jx = nn & 7, q = lane >> 4;
    const int x = x0 + 2 * jx;
    const bool colok = x + 1 < A.W;   // W is even
    float* red = lds;  // [4 row groups][64 channels][2]
#pragma unroll
    for (int m = 0; m < MBW; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = (half * MBW + m) * 16 + q * 4 + r;
            const float bv = bias ? bias[oc] : 0.f;
            float* po = A.out + (((size_t)n * A.CoutStride + oc) * A.D + d) * plane;
            float s = 0.f, sq = 0.f;
#pragma unroll
            for (int j = 0; j < NBT; ++j) {
                const int y = y0 + 4 * wave + 2 * j + (nn >> 3);
                const float m0 = acc[0][m][j][r], m1 = acc[1][m][j][r], m2 = acc[2][m][j][r], m3 = acc[3][m][j][r];
                float t0 = (m0 + m1) + m2 + bv;
                float t1 = (m1 - m2) - m3 + bv;
                if (A.lrelu) {
                    t0 = t0 > 0.f ? t0 : t0 * kLeakySlope;
                    t1 = t1 > 0.f ? t1 : t1 * kLeakySlope;
                }
                if (y < A.H && colok) {
                    *reinterpret_cast<float2*>(po + (size_t)y * A.W + x) = make_float2(t0, t1);
                    s += t0 + t1;
                    sq = fmaf(t0, t0, fmaf(t1, t1, sq));
                }
            }
            if (A.partials) {
                s = row16_sum16(s);
                sq = row16_sum16(sq);
                if (nn == 15) {
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
            for (int wv = 0; wv < 4; ++wv) v += (double)red[((wv * MB * 16) + oc) * 2 + k];
            A.partials[((((size_t)n * A.Cout + oc) * A.D + d) * A.tiles + tile) * 2 + k] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
int conv2d_wino16_tiles(int h, int w) { return ((h + TH - 1) / TH) * ((w + TWX - 1) / TWX); }

// The square tiles win when they compute fewer pixels than the 4 x 64 ones (both are 256 pixels per workgroup).
bool conv2d_wino16_preferred(int h, int w) {
    static const bool enabled = []() {  // PDS_WINO_TILE16=0 keeps the 4 x 64 tiles everywhere (A/B, debugging)
        const char* e = debug_switch("PDS_WINO_TILE16");
        return !(e && e[0] == '0');
    }();
    if (!enabled || w % 4 != 0 || w < 4) return false;
    const int wide = ((h + 3) / 4) * ((w + 63) / 64);
    return conv2d_wino16_tiles(h, w) < wide;
}

// L.packed holds the filter transform in the layout of conv2d_wino.hip (its launcher packs, then hands over here)
int launch_conv2d_wino16(const ConvLayer& L, size_t w_set_stride, int bias_set_stride, hipStream_t s) {
    Wino16Args A;
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
    A.tiles = conv2d_wino16_tiles(A.H, A.W);
    A.w_set_stride = w_set_stride;
    A.bias_set_stride = bias_set_stride;
    const size_t lds_bytes = (size_t)2 * BUF * sizeof(float);
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_wino16_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_wino16_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    }
    const dim3 grid(A.tiles, A.D, A.N);
    if (L.a.scale) hipLaunchKernelGGL((conv2d_wino16_kernel<true>), grid, dim3(THREADS), lds_bytes, s, A);
    else hipLaunchKernelGGL((conv2d_wino16_kernel<false>), grid, dim3(THREADS), lds_bytes, s, A);
    return check_launch("conv2d_wino16");
}

}  // namespace pds
