// 2-D Winograd-domain variant of the dominant kernel: 3x3 stride-1 convolution Cin -> 64 over a stack of 2-D planes
// (the 64 -> 64 layers of reference practical_deep_stereo/matching.py:85-88, network_blocks.py:47-58, 97-103,
// 134-144), exact-fp32 MFMA, F(2x2, 3x3): 16 products per 2 x 2 outputs instead of 36 -- 0.44x the MFMAs of the direct
// form (conv2d_mfma.hip), 0.67x those of the F(2,3)-along-x kernel (conv2d_wino.hip).
//
//   V = B^T d B   (d: 4 x 4 input patch)     U = G g G^T   (g: 3 x 3 filter)     Y = A^T (sum_ic U . V) A
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
//   GEMM view    per position p = (py, px): M = output channels (16 per MFMA block, 4 blocks), N = 16 tiles,
//                K = Cin; v_mfma_f32_16x16x4_f32.
//   workgroup    output tile 4 rows x 48 columns = 2 x 24 tiles = 3 blocks of 16 tiles, all 64 channels;
//                16 waves with two ROLES (the MFMA work per staged byte is so small that staging must run beside it,
//                not between it):
//                  waves 0..11   MFMA waves (channel block ocb, tile block nb): 16 accumulators (one per position);
//                                their A fragments (U) come straight from global memory, packed per lane, two chunks
//                                deep in registers -- weights never touch LDS;
//                  waves 12..15  staging waves, one per channel of the chunk: lane = (tile row, tile column) owns the
//                                4 x 4 input patch of ONE tile: 8-byte pair loads (a chunk ahead), row transform
//                                with whole-wave DPP shifts for the horizontal neighbours, column transform in
//                                registers, 16 LDS writes (one thread per tile column for both tile rows was the
//                                first version: its ~250 instructions per chunk outlasted the 16 MFMAs of a wave).
//   LDS          V double-buffered: [4 ic][16 positions][48 tiles] (channel stride == 16 mod 32), 25 KB.
//   sync         one barrier per 4-channel chunk; staging runs one chunk ahead of the MFMA waves.
//   epilogue     output transform, + bias, LeakyReLU(0.1), 8-byte stores, per-(plane, channel) statistics partials.
#include "common.hpp"

namespace pds {

namespace {

constexpr int TH = 4, TWX = 48, TC = TWX / 2, TR = TH / 2, NT = TR * TC, NBT = NT / 16, KC = 4;
constexpr int NPOS = 16;
constexpr int VCS = NPOS * NT + 16;        // V channel stride: 784, == 16 (mod 32)
constexpr int KS = 2;                      // MFMA k-steps (4 channels each) per barrier step
constexpr int KCB = KC * KS;               // channels staged per barrier step
constexpr int V_CHUNK = KCB * VCS;         // floats per V buffer
constexpr int MFMA_WAVES = 4 * NBT;        // 12
constexpr int STAGE_WAVES = 4;
constexpr int THREADS = 64 * (MFMA_WAVES + STAGE_WAVES);
constexpr int W_CHUNK = 4 * 64 * NPOS;     // packed U floats per chunk: [ocb][lane][position]
static_assert(NT % 16 == 0 && VCS % 32 == 16, "tile blocks / bank layout");

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Wino2Args {
    Src a;
    const float* __restrict__ wpk;   // [chunk][ocb][lane][16 positions]
    const float* __restrict__ bias;
    float* __restrict__ out;
    double* __restrict__ partials;
    int N, Cin, D, H, W, Cout;
    int lrelu;
    int tiles_x, tiles;
};

__device__ __forceinline__ float row16_sum_2d(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}
// lane i receives lane i - 1 (DPP wave_shr:1) / lane i + 1 (DPP wave_shl:1)
__device__ __forceinline__ float shift_up_2d(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float shift_down_2d(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

}  // namespace

// Work distribution: PERSISTENT workgroups (one per CU).  With one 16-wave workgroup per CU nothing else hides a
// tile's prologue, so a workgroup walks a list of (batch, plane, tile) items and its staging waves run straight on into
// the first chunks of the next item while the MFMA waves finish the current one.  Items are dealt so that every XCD
// (workgroup index mod 8) works on whole planes (shared halo rows and statistics stay in one L2).
__device__ __forceinline__ int w2_item_count(const Wino2Args& A) {
    const int G = gridDim.x;
    if ((A.D & 7) == 0 && (G & 7) == 0) {
        const int r = blockIdx.x >> 3, R = G >> 3, per_xcd = A.N * (A.D >> 3) * A.tiles;
        return r < per_xcd ? (per_xcd - r + R - 1) / R : 0;
    }
    const int total = A.N * A.D * A.tiles;
    return (int)blockIdx.x < total ? (total - (int)blockIdx.x + G - 1) / G : 0;
}
__device__ __forceinline__ void w2_item(const Wino2Args& A, int k, int& n, int& d, int& tile) {
    const int G = gridDim.x;
    if ((A.D & 7) == 0 && (G & 7) == 0) {
        const int j = (int)(blockIdx.x >> 3) + k * (G >> 3);
        tile = j % A.tiles;
        const int pl = j / A.tiles, dpx = A.D >> 3;
        n = pl / dpx;
        d = (pl % dpx) * 8 + (int)(blockIdx.x & 7);
    } else {
        const int j = (int)blockIdx.x + k * G;
        tile = j % A.tiles;
        const int pl = j / A.tiles;
        n = pl / A.D;
        d = pl % A.D;
    }
}

template <bool NORM>
__global__ __launch_bounds__(THREADS) void conv2d_wino2d_kernel(const Wino2Args A) {
    __shared__ __attribute__((aligned(16))) float lds[2 * V_CHUNK];   // two V buffers
    __shared__ float red[NBT * 64 * 2];                               // statistics of one tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t plane = (size_t)A.H * A.W;
    const size_t cstride = (size_t)A.D * plane;
    const int nchunks = A.Cin / KCB;      // barrier steps per item
    const int nitems = w2_item_count(A);
    if (nitems == 0) return;
    const int total = nitems * nchunks;   // (item, chunk) steps of this workgroup, one barrier each

    if (wave >= MFMA_WAVES) {
        // ===================================== staging role =====================================
        const int ic = wave - MFMA_WAVES;          // channel inside the chunk: one staging wave each
        const int tr = lane >> 5, tc = lane & 31;  // tile of this lane (tc < 24 active)
        const bool active = tc < TC;
        const int tcc = min(tc, TC - 1);
        const bool edge = tc == 0 || tc == TC - 1;
        const unsigned gstride = NORM ? (A.a.per_plane ? (unsigned)A.D : 1u) : 0u;
        const size_t chunk_stride = (size_t)KCB * cstride;
        const int l_off = ic * VCS + tr * TC + tcc;

        // context of the item whose chunks are being FETCHED (offsets, masks, base pointers)
        unsigned offp[4], offe[4], fmask = 0;      // fmask: bits 0-3 pair inside, bits 4-7 halo value inside
        const float *pa = nullptr, *ps = nullptr, *ph = nullptr;
        auto set_item = [&](int k) {
            int n, d, tile;
            w2_item(A, k, n, d, tile);
            const int y0 = (tile / A.tiles_x) * TH, x0 = (tile % A.tiles_x) * TWX;
            const int x = x0 + 2 * tcc;                    // d1 column; d2 = x + 1 (W even: both or neither inside)
            const int xe = tc == 0 ? x0 - 1 : x0 + TWX;    // halo column of the row segment (tc == 0 / TC - 1)
            fmask = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = y0 - 1 + 2 * tr + r;
                const bool rowok = y >= 0 && y < A.H;
                const int yc = min(max(y, 0), A.H - 1);
                if (rowok && x + 1 < A.W) fmask |= 1u << r;
                if (rowok && edge && xe >= 0 && xe < A.W) fmask |= 16u << r;
                const unsigned rowbase = (unsigned)ic * (unsigned)cstride + (unsigned)(yc * A.W);
                offp[r] = rowbase + (unsigned)min(x, A.W - 2);
                offe[r] = rowbase + (unsigned)(edge ? min(max(xe, 0), A.W - 1) : min(x, A.W - 2));
            }
            pa = A.a.p + ((size_t)n * A.Cin * A.D + d) * plane;
            if (NORM) {
                const size_t g0 = (A.a.per_plane ? ((size_t)n * A.Cin * A.D + d) : (size_t)n * A.Cin) + (size_t)ic * gstride;
                ps = A.a.scale + g0;
                ph = A.a.shift + g0;
            }
        };

        // two register sets: the loads of step g + 2 are in flight while step g + 1 is transformed (one step, ~1.6 us,
        // does not cover a load round trip under load)
        float2 vp[2][KS][4];
        float ve[2][KS][4], vs[2][KS], vh[2][KS];
        unsigned smask[2] = {0, 0};                // masks of the step held in each set
#define PDS_W2FETCH(set_, chunk_)                                                                     \
    {                                                                                                 \
        _Pragma("unroll") for (int h = 0; h < KS; ++h) {                                              \
            const float* src = pa + (size_t)(chunk_) * chunk_stride + (size_t)h * KC * cstride;       \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                           \
                vp[set_][h][r] = *reinterpret_cast<const float2*>(src + offp[r]);                     \
                ve[set_][h][r] = src[offe[r]];                                                        \
            }                                                                                         \
            vs[set_][h] = NORM ? ps[((size_t)(chunk_) * KCB + h * KC) * gstride] : 1.f;               \
            vh[set_][h] = NORM ? ph[((size_t)(chunk_) * KCB + h * KC) * gstride] : 0.f;               \
        }                                                                                             \
        smask[set_] = fmask;                                                                          \
    }
#ifdef PDS_X2D_NOSTAGE
#define PDS_W2STASH(set_, buf_) { if (smask[set_] == 0xffffffffu) (buf_)[l_off] = vp[set_][0][0].x; }
#else
#define PDS_W2STASH(set_, buf_)                                                                       \
    {                                                                                                 \
        _Pragma("unroll") for (int h = 0; h < KS; ++h) {                                              \
            float R[4][4];                                                                            \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                           \
                const bool i1 = (smask[set_] >> r) & 1u, ie = (smask[set_] >> (4 + r)) & 1u;          \
                const float sc = vs[set_][h], sh = vh[set_][h];                                       \
                const float d1 = i1 ? (NORM ? fmaf(sc, vp[set_][h][r].x, sh) : vp[set_][h][r].x) : 0.f; \
                const float d2 = i1 ? (NORM ? fmaf(sc, vp[set_][h][r].y, sh) : vp[set_][h][r].y) : 0.f; \
                const float de = ie ? (NORM ? fmaf(sc, ve[set_][h][r], sh) : ve[set_][h][r]) : 0.f;   \
                const float up = shift_up_2d(d2), dn = shift_down_2d(d1);                             \
                const float d0 = tc == 0 ? de : up;                                                   \
                const float d3 = tc == TC - 1 ? de : dn;                                              \
                R[r][0] = d0 - d2;                                                                    \
                R[r][1] = d1 + d2;                                                                    \
                R[r][2] = d2 - d1;                                                                    \
                R[r][3] = d1 - d3;                                                                    \
            }                                                                                         \
            if (active) {                                                                             \
                float* dst = (buf_) + l_off + h * KC * VCS;                                           \
                _Pragma("unroll") for (int px = 0; px < 4; ++px) {                                    \
                    dst[(0 * 4 + px) * NT] = R[0][px] - R[2][px];                                     \
                    dst[(1 * 4 + px) * NT] = R[1][px] + R[2][px];                                     \
                    dst[(2 * 4 + px) * NT] = R[2][px] - R[1][px];                                     \
                    dst[(3 * 4 + px) * NT] = R[1][px] - R[3][px];                                     \
                }                                                                                     \
            }                                                                                         \
        }                                                                                             \
    }
#endif
        // fetch cursor: (item, chunk) of the step fetched last
        int fitem = 0, fchunk = 0;
#define PDS_W2ADVANCE()               \
    if (++fchunk == nchunks) {        \
        fchunk = 0;                   \
        set_item(++fitem);            \
    }
        set_item(0);
        PDS_W2FETCH(0, 0)                    // step 0 -> set 0
        if (total > 1) {                     // step 1 -> set 1
            PDS_W2ADVANCE()
            PDS_W2FETCH(1, fchunk)
        }
        PDS_W2STASH(0, lds)                  // step 0 into buffer 0
        if (total > 2) {                     // step 2 -> set 0
            PDS_W2ADVANCE()
            PDS_W2FETCH(0, fchunk)
        }
        __syncthreads();
        int chunk = 0;                       // chunk index of step g (for the statistics barrier)
        // step g: the MFMA waves consume buffer g & 1; step g + 1 (set (g + 1) & 1) is transformed into the other
        // buffer, then the loads of step g + 3 refill that set.  Unrolled by two for static register sets.
        for (int g = 0; g < total; g += 2) {
            if (g + 1 < total) {
                PDS_W2STASH(1, lds + V_CHUNK)
                if (g + 3 < total) {
                    PDS_W2ADVANCE()
                    PDS_W2FETCH(1, fchunk)
                }
            }
            __syncthreads();
            if (++chunk == nchunks) {
                chunk = 0;
                if (A.partials) __syncthreads();   // the barrier of the MFMA waves' statistics reduction
            }
            if (g + 1 >= total) break;
            if (g + 2 < total) {
                PDS_W2STASH(0, lds)
                if (g + 4 < total) {
                    PDS_W2ADVANCE()
                    PDS_W2FETCH(0, fchunk)
                }
            }
            __syncthreads();
            if (++chunk == nchunks) {
                chunk = 0;
                if (A.partials) __syncthreads();
            }
        }
#undef PDS_W2FETCH
#undef PDS_W2STASH
#undef PDS_W2ADVANCE
        return;
    }

    // ========================================= MFMA role =========================================
    const int ocb = wave & 3, nb = wave >> 2;
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(A.wpk) + ((size_t)ocb * 64 + lane) * (NPOS / 4);
    constexpr int W_F4 = W_CHUNK / 4;  // float4 per chunk
    f32x4 wa[NPOS / 4], wn[NPOS / 4];
#pragma unroll
    for (int i = 0; i < NPOS / 4; ++i) wa[i] = wsrc[i];
    f32x4 acc[NPOS];
    const int b_lane = (lane >> 4) * VCS + nb * 16 + (lane & 15);
    const int jx = lane & 15, q = lane >> 4;
    const int tl = nb * 16 + jx, ttr = tl / TC, tcl = tl % TC;
    const bool pairs = (A.W & 1) == 0;
    __syncthreads();  // V of step 0 is staged

    int g = 0;
    for (int k = 0; k < nitems; ++k) {
        int n, d, tile;
        w2_item(A, k, n, d, tile);
#pragma unroll
        for (int p = 0; p < NPOS; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int chunk = 0; chunk < nchunks; ++chunk, ++g) {
#pragma unroll
            for (int h = 0; h < KS; ++h) {
                const int kq = chunk * KS + h;                                  // 4-channel quarter of the item
                const int nxt = kq + 1 < nchunks * KS ? kq + 1 : 0;             // the weights repeat for every item
#pragma unroll
#ifdef PDS_X2D_NOWLOAD
                for (int i = 0; i < NPOS / 4; ++i) wn[i] = wa[i];
#else
                for (int i = 0; i < NPOS / 4; ++i) wn[i] = wsrc[(size_t)nxt * W_F4 + i];
#endif
                const float* xin = lds + (g & 1) * V_CHUNK + h * KC * VCS + b_lane;
#pragma unroll
                for (int p = 0; p < NPOS; ++p) {
                    const float bf = xin[p * NT];
#ifdef PDS_X2D_NOMFMA
                    acc[p][0] += wa[p >> 2][p & 3] * bf;
#else
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[p >> 2][p & 3], bf, acc[p], 0, 0, 0);
#endif
                }
#pragma unroll
                for (int i = 0; i < NPOS / 4; ++i) wa[i] = wn[i];
            }
            __syncthreads();
        }

        // ---- epilogue of the item: output transform, bias, LeakyReLU, store, statistics ---------------------
        const int y = (tile / A.tiles_x) * TH + 2 * ttr, x = (tile % A.tiles_x) * TWX + 2 * tcl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = ocb * 16 + q * 4 + r;
            const float bv = A.bias ? A.bias[oc] : 0.f;
            float T[4][2];
#pragma unroll
            for (int py = 0; py < 4; ++py) {
                const float m0 = acc[py * 4 + 0][r], m1 = acc[py * 4 + 1][r], m2 = acc[py * 4 + 2][r],
                            m3 = acc[py * 4 + 3][r];
                T[py][0] = (m0 + m1) + m2;
                T[py][1] = (m1 - m2) - m3;
            }
            float o[2][2];
#pragma unroll
            for (int ox = 0; ox < 2; ++ox) {
                o[0][ox] = (T[0][ox] + T[1][ox]) + T[2][ox] + bv;
                o[1][ox] = (T[1][ox] - T[2][ox]) - T[3][ox] + bv;
            }
            float s = 0.f, sq = 0.f;
            float* po = A.out + (((size_t)n * A.Cout + oc) * A.D + d) * plane;
#pragma unroll
            for (int oy = 0; oy < 2; ++oy) {
                float t0 = o[oy][0], t1 = o[oy][1];
                if (A.lrelu) {
                    t0 = t0 > 0.f ? t0 : t0 * kLeakySlope;
                    t1 = t1 > 0.f ? t1 : t1 * kLeakySlope;
                }
                const int yy = y + oy;
                if (yy < A.H && x + 1 < A.W && pairs) {
                    *reinterpret_cast<float2*>(po + (size_t)yy * A.W + x) = make_float2(t0, t1);
                    s += t0 + t1;
                    sq = fmaf(t0, t0, fmaf(t1, t1, sq));
                } else if (yy < A.H) {
                    if (x < A.W) {
                        po[(size_t)yy * A.W + x] = t0;
                        s += t0;
                        sq = fmaf(t0, t0, sq);
                    }
                    if (x + 1 < A.W) {
                        po[(size_t)yy * A.W + x + 1] = t1;
                        s += t1;
                        sq = fmaf(t1, t1, sq);
                    }
                }
            }
            if (A.partials) {
                s = row16_sum_2d(s);
                sq = row16_sum_2d(sq);
                if (jx == 15) {
                    red[(nb * 64 + oc) * 2 + 0] = s;
                    red[(nb * 64 + oc) * 2 + 1] = sq;
                }
            }
        }
        if (A.partials) {
            __syncthreads();
            if (tid < 128) {
                const int oc = tid >> 1, kk = tid & 1;
                double v = 0.0;
#pragma unroll
                for (int b = 0; b < NBT; ++b) v += (double)red[(b * 64 + oc) * 2 + kk];
                A.partials[((((size_t)n * A.Cout + oc) * A.D + d) * A.tiles + tile) * 2 + kk] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
bool conv2d_wino2d_eligible(const ConvLayer& L) {
    static const bool enabled = []() {  // opt-in while it is being measured: PDS_WINO2D=1
        const char* e = getenv("PDS_WINO2D");
        return e && e[0] == '1';
    }();
    if (!enabled) return false;
    if (L.kd != 1 || L.stride != 1 || L.out_g.c != 64) return false;
    if (L.in.c % KCB != 0 || L.in.c > 256) return false;
    if (L.b.p || L.l0A || L.side_out || L.plane_weight_sets > 0) return false;
    if (L.in.w % 2 != 0 || L.in.w < 2) return false;
    if ((size_t)L.in.d * L.in.h * L.in.w * KCB >= ((size_t)1 << 31)) return false;
    if (L.in.d > 65535 || L.in.n > 65535) return false;
    return true;
}

int conv2d_wino2d_tiles(const Geom& o) { return ((o.h + TH - 1) / TH) * ((o.w + TWX - 1) / TWX); }

size_t conv2d_wino2d_packed_floats(int cin, int cout) { return (size_t)(cin / KC) * W_CHUNK; }

int launch_conv2d_wino2d(const ConvLayer& L, hipStream_t s) {
    if (!L.packed) return set_error(-1, "conv2d_wino2d: packed weights missing");
    const int total = (int)conv2d_wino2d_packed_floats(L.in.c, L.out_g.c);
    const PackPhase phase = L.sink ? L.sink->phase : kPackInline;
    if (phase != kPackDone) {
        PackJob j;
        j.src = L.weight;
        j.dst = L.packed;
        j.cout = L.out_g.c;
        j.cin = L.in.c;
        j.mblocks = 4;
        j.kc = KC;
        j.taps = 16;
        j.mode = 4;  // F(2x2, 3x3) filter transform, per-lane fragment order
        j.total = total;
        if (phase == kPackCollect) return L.sink->push(j) ? 0 : set_error(-1, "pack job table full");
        if (int rc = launch_multi_pack(&j, 1, s)) return rc;
    }
    Wino2Args A;
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
    A.lrelu = L.lrelu;
    A.tiles_x = (A.W + TWX - 1) / TWX;
    A.tiles = conv2d_wino2d_tiles(L.out_g);
    // persistent workgroups: one per CU (16 waves fill a CU), a multiple of 8 for the XCD-aware item order
    static const int cus = []() {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 8 ? n & ~7 : 8;
    }();
    const long long items = (long long)A.tiles * A.D * A.N;
    int wgs = (int)(items < cus ? items : cus);
    if (wgs >= 8) wgs &= ~7;
    const dim3 grid(wgs, 1, 1);
    if (L.a.scale)
        hipLaunchKernelGGL(conv2d_wino2d_kernel<true>, grid, dim3(THREADS), 0, s, A);
    else
        hipLaunchKernelGGL(conv2d_wino2d_kernel<false>, grid, dim3(THREADS), 0, s, A);
    return check_launch("conv2d_wino2d");
}

}  // namespace pds
