// Exact-fp32 MFMA implicit-GEMM 3x3 convolution over a stack of 2-D planes (kd = 1, stride 1, pad 1):
// the MatchingOperation convolutions of reference practical_deep_stereo/matching.py:84-93
// (network_blocks.py:19-24, 47-58, 97-103) evaluated for ALL disparity planes in one launch.
//
//   GEMM view      M = output channels (16 per MFMA block), N = 16 consecutive pixels of a row,
//                  K = Cin * 9, walked as 8-channel chunks x 9 taps x 2 k-steps of 4 channels.
//   instruction    v_mfma_f32_16x16x4_f32 (exact fp32: bitwise an fmaf chain; 32-cycle issue).
//                  A (weights): lane l -> A[i = l & 15][k = l >> 4];  B (pixels): B[k = l >> 4][j = l & 15];
//                  D: column j = l & 15 (pixel), row = 4 * (l >> 4) + reg (channel) -> stores are
//                  contiguous along x.
//   workgroup      256 threads = 4 waves; output tile TH=4 rows x TW=80 columns of one (n, d) plane, all
//                  output channels.  Wave r owns row r: NB=5 pixel blocks x MB channel blocks accumulators.
//   LDS            per buffer: input chunk [8 ch][TH+2 rows][84] (channel stride == 16 mod 32 banks, so the two
//                  k-halves of a 32-lane group never collide) + weight chunk in exact fragment order
//                  [tap][k-step][MB][64 lanes] (pre-packed in HBM, so staging is a straight copy and
//                  fragment reads are lane-linear).  Two buffers: chunk c+1 is fetched to registers
//                  before the MFMAs of chunk c and written after them -> one barrier per chunk.
//   prologue       the InstanceNorm of the producing layer (scale * raw + shift), an optional second
//                  source (residual sum) and literal zero padding are applied while staging; every thread
//                  owns <= 2 fixed halo positions for all 8 channels, loads are unconditional (clamped).
//   epilogue       + bias, LeakyReLU(0.1), store, and per-(plane, channel) sum / sum-of-squares partials in
//                  fp64 for the deferred InstanceNorm of THIS layer (deterministic: one record per tile).
#include "common.hpp"

namespace pds {

namespace {

#ifndef PDS_TW
#define PDS_TW 80
#endif
#ifndef PDS_WAVES
#define PDS_WAVES 3
#endif
constexpr int TH = 4, TW = PDS_TW, NB = TW / 16, KC = 4;
constexpr int RS = TW + 4;                           // LDS row stride (>= TW + 2)
constexpr int CS = ((TH + 2) * RS + 31) / 32 * 32 + 16;  // channel stride, == 16 (mod 32)
constexpr int IN_CHUNK = KC * CS;                    // floats
constexpr int THREADS = 256;
constexpr int IN_ELEMS = KC * (TH + 2) * (TW + 2);
constexpr int IN_ITERS = (IN_ELEMS + THREADS - 1) / THREADS;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// timing-decomposition hooks (tools/build_variant.sh): any of them set makes the results wrong
#ifdef PDS_X2_NOLOAD
#define PDS_X2_LOAD(v) (float)(tid)
#else
#define PDS_X2_LOAD(v) (v)
#endif
#ifdef PDS_X2_NOMFMA
#define PDS_X2_MFMA(c, a, b) (c)[0] += (a) * (b)
#else
#define PDS_X2_MFMA(c, a, b) (c) = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif

struct MfmaArgs {
    Src a, b;
    const float* __restrict__ wpk;   // packed weights, see pack kernel
    const float* __restrict__ bias;
    float* __restrict__ out;
    double* __restrict__ partials;
    int N, Cin, D, H, W, Cout;
    int lrelu;
    int tiles_x, tiles;              // tiles per row of tiles, tiles per plane
    // layer-0 terms formed on the fly (SRC 2, 3): x0[c,d,y,x] = A[c,y,x] + G[c,y,x-d] (+ right-edge fix)
    // all three: row stride l0_rs, channel stride l0_cstride
    const float* __restrict__ l0A;   // conv_L(left) + bias, pointing at the column of x = 0 of its rows
    const float* __restrict__ l0G;   // conv_R(right), column u + 1 for u = x - d
    const float* __restrict__ l0G2;  // same without the dx = +1 taps (used at x = W-1, d >= 1)
    size_t l0_cstride;
    int l0_rs;
    int d_begin;                     // disparity of plane 0
    size_t w_set_stride;             // floats between the packed weight sets of consecutive planes (0: shared)
    int bias_set_stride;             // ditto for the bias
    float* __restrict__ side_out;    // optional: the staged (summed, normalised) input is also written here
};

// KCT: input channels staged per chunk.  4 for the 64-channel-block kernels (keeps 3 workgroups per CU); 8 for the
// single-block kernels (8 / 16 output channels), whose 45 MFMAs per 4-channel chunk are far shorter than a load round
// trip: twice the bytes in flight per barrier.
template <int MB, int KCT = KC>
struct Cfg {
    static constexpr int IN_CHUNK_T = KCT * CS;
    static constexpr int W_CHUNK = 9 * (KCT / 4) * MB * 64;     // floats per weight chunk
    static constexpr int BUF = IN_CHUNK_T + W_CHUNK;            // floats per LDS buffer
    static constexpr int W_ITERS = (W_CHUNK / 4 + THREADS - 1) / THREADS;  // float4 per thread
};

}  // namespace

// sum over the 16 lanes of a DPP row; the total lands in lane 15 of each row
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

// SRC: 0 = source a;  1 = a + b;  2 = layer-0 terms (A + shifted G);  3 = a + layer-0 terms
// PAIR (even widths, SRC 0 / 1): every thread stages ONE aligned pair of columns per channel with an 8-byte load
// (6 rows x 42 pairs cover columns x0 - 2 .. x0 + 81) instead of two scalar positions: half the load instructions
// and fully used cache lines.
template <int MB, int SRC, bool PAIR = false, int KCT = KC>
__global__ __launch_bounds__(THREADS, (SRC == 0 || SRC == 2 || MB == 1) ? PDS_WAVES : 2) void conv2d_mfma_kernel(const MfmaArgs A) {
    constexpr bool HAS_A = SRC != 2;
    constexpr bool HAS_B = SRC == 1;
    constexpr bool HAS_L0 = SRC >= 2;
    static_assert(!PAIR || (!HAS_L0 && TW % 2 == 0 && RS >= TW + 4), "pair staging: plain sources, even tiles");
    using C = Cfg<MB, KCT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2 buffers][input chunk | weight chunk]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int d = blockIdx.y, n = blockIdx.z;
    const int ty = tile / A.tiles_x, tx = tile % A.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const size_t plane = (size_t)A.H * A.W;
    const size_t cstride = (size_t)A.D * plane;  // channel stride of NCDHW
    const int nchunks = A.Cin / KCT;

    // ---- staging map: thread -> up to POS positions (row r, column xx) of the (TH+2) x (TW+2) halo tile,
    // the same positions for each of the 8 channels of a chunk.  Loads are unconditional from clamped
    // coordinates; padding is applied as a select when the value is written to LDS.
    constexpr int NPOS = (TH + 2) * (TW + 2);
    constexpr int POS = (NPOS + THREADS - 1) / THREADS;
    int g_off[POS], l_off[POS], gg_off[HAS_L0 ? POS : 1];
    bool inside[POS], gvalid[HAS_L0 ? POS : 1], interior[POS];
    const float* gsel[HAS_L0 ? POS : 1];
    const int disp = A.d_begin + d;
#pragma unroll
    for (int k = 0; k < POS; ++k) {
        int r, xx, p;
        if (PAIR) {  // k = 0 / 1: the two columns of this thread's pair; LDS column = x - (x0 - 2)
            constexpr int NPAIR = (TH + 2) * ((TW + 4) / 2);
            static_assert(NPAIR <= THREADS && POS == 2, "one pair per thread");
            p = min(tid, NPAIR - 1);
            r = p / ((TW + 4) / 2);
            xx = 2 * (p % ((TW + 4) / 2)) + k - 1;   // relative to x0 - 1, like the scalar map
        } else {
            p = min(tid + k * THREADS, NPOS - 1);  // surplus threads duplicate the last position
            r = p / (TW + 2);
            xx = p % (TW + 2);
        }
        const int y = y0 - 1 + r, x = x0 - 1 + xx;
        inside[k] = y >= 0 && y < A.H && x >= 0 && x < A.W;
        interior[k] = inside[k] && r >= 1 && r <= TH && xx >= 1 && xx <= TW && (tid + k * THREADS) < NPOS;
        const int yc = min(max(y, 0), A.H - 1), xc = min(max(x, 0), A.W - 1);
        g_off[k] = yc * A.W + (PAIR ? min(max(x - k, 0), A.W - 2) + k : xc);  // pairs stay aligned when clamped
        l_off[k] = r * RS + xx + (PAIR ? 1 : 0);
        if (HAS_L0) {
            const int u = xc - disp;  // column of the un-shifted right descriptor
            gvalid[k] = u >= -1;
            gg_off[k] = yc * A.l0_rs + max(u, -1) + 1;
            gsel[k] = ((xc == A.W - 1 && disp >= 1) ? A.l0G2 : A.l0G) + (size_t)n * A.Cin * A.l0_cstride;
        }
    }
    const float* pa = HAS_A ? A.a.p + ((size_t)n * A.Cin * A.D + d) * plane : nullptr;
    const float* pb = HAS_B ? A.b.p + ((size_t)n * A.Cin * A.D + d) * plane : nullptr;
    const float* pl = HAS_L0 ? A.l0A + (size_t)n * A.Cin * A.l0_cstride : nullptr;
    const size_t gplane = A.l0_cstride;
    int la_off[HAS_L0 ? POS : 1];
    if (HAS_L0) {
#pragma unroll
        for (int k = 0; k < POS; ++k) la_off[k] = (g_off[k] / A.W) * A.l0_rs + g_off[k] % A.W;
    }
    const float* wbase = A.wpk + (size_t)d * A.w_set_stride;
    const float* bias = A.bias ? A.bias + d * A.bias_set_stride : nullptr;
    const int wlast = C::W_CHUNK / 4 - 1;

    float va[HAS_A ? KCT : 1][POS], vb[(HAS_B || HAS_L0) ? KCT : 1][POS], vg[HAS_L0 ? KCT : 1][POS];
    f32x4 vw[C::W_ITERS];  // ext-vector type: HIP's float4 struct keeps the array in scratch

#define PDS_FETCH(chunk_)                                                                          \
    {                                                                                              \
        const float* ca = HAS_A ? pa + (size_t)(chunk_) * KCT * cstride : nullptr;                  \
        const float* cb = HAS_B ? pb + (size_t)(chunk_) * KCT * cstride : nullptr;                  \
        _Pragma("unroll") for (int c = 0; c < KCT; ++c) {                                           \
            if (PAIR) {                                                                            \
                if (HAS_A) {                                                                       \
                    const float2 t = *reinterpret_cast<const float2*>(ca + c * cstride + g_off[0]); \
                    va[c][0] = t.x;                                                                \
                    va[c][1] = t.y;                                                                \
                }                                                                                  \
                if (HAS_B) {                                                                       \
                    const float2 t = *reinterpret_cast<const float2*>(cb + c * cstride + g_off[0]); \
                    vb[c][0] = t.x;                                                                \
                    vb[c][1] = t.y;                                                                \
                }                                                                                  \
                continue;                                                                          \
            }                                                                                      \
            _Pragma("unroll") for (int k = 0; k < POS; ++k) {                                      \
                if (HAS_A) va[c][k] = PDS_X2_LOAD(ca[c * cstride + g_off[k]]);                     \
                if (HAS_B) vb[c][k] = PDS_X2_LOAD(cb[c * cstride + g_off[k]]);                     \
                if (HAS_L0) {                                                                      \
                    const int ch = (chunk_) * KCT + c;                                              \
                    vb[c][k] = pl[(size_t)ch * gplane + la_off[k]];                                \
                    vg[c][k] = gsel[k][(size_t)ch * gplane + gg_off[k]];                           \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(wbase + (size_t)(chunk_) * C::W_CHUNK);  \
        _Pragma("unroll") for (int it = 0; it < C::W_ITERS; ++it)                                  \
            vw[it] = wsrc[min(it * THREADS + tid, wlast)];                                         \
    }

#define PDS_STASH(chunk_, buf_)                                                                    \
    {                                                                                              \
        _Pragma("unroll") for (int c = 0; c < KCT; ++c) {                                           \
            const int ch = (chunk_) * KCT + c;                                                      \
            float sa = 1.f, ha = 0.f, sb = 1.f, hb = 0.f;                                          \
            if (HAS_A && A.a.scale) {                                                              \
                const int g = A.a.per_plane ? ((n * A.Cin + ch) * A.D + d) : (n * A.Cin + ch);      \
                sa = A.a.scale[g];                                                                 \
                ha = A.a.shift[g];                                                                 \
            }                                                                                      \
            if (HAS_B && A.b.scale) {                                                              \
                const int g = A.b.per_plane ? ((n * A.Cin + ch) * A.D + d) : (n * A.Cin + ch);      \
                sb = A.b.scale[g];                                                                 \
                hb = A.b.shift[g];                                                                 \
            }                                                                                      \
            _Pragma("unroll") for (int k = 0; k < POS; ++k) {                                      \
                float v = 0.f;                                                                     \
                if (HAS_A) v = fmaf(sa, va[c][k], ha);                                             \
                if (HAS_B) v += fmaf(sb, vb[c][k], hb);                                            \
                if (HAS_L0) {                                                                      \
                    const float x0v = vb[c][k] + (gvalid[k] ? vg[c][k] : 0.f);                     \
                    v = HAS_A ? v + x0v : x0v;                                                     \
                }                                                                                  \
                v = inside[k] ? v : 0.f;                                                           \
                (buf_)[c * CS + l_off[k]] = v;                                                     \
                if (A.side_out && interior[k])                                                     \
                    A.side_out[(((size_t)n * A.Cin + ch) * A.D + d) * plane + g_off[k]] = v;       \
            }                                                                                      \
        }                                                                                          \
        f32x4* wdst = reinterpret_cast<f32x4*>((buf_) + C::IN_CHUNK_T);                                 \
        _Pragma("unroll") for (int it = 0; it < C::W_ITERS; ++it)                                  \
            wdst[min(it * THREADS + tid, wlast)] = vw[it];                                         \
    }

    f32x4 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    PDS_FETCH(0)
    PDS_STASH(0, lds)
    __syncthreads();

    // lane-constant part of the fragment addresses
    const int b_lane = (lane >> 4) * CS + wave * RS + (lane & 15) + (PAIR ? 1 : 0);

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        float* buf = lds + (chunk & 1) * C::BUF;
        float* nxt = lds + ((chunk + 1) & 1) * C::BUF;
        const bool more = chunk + 1 < nchunks;
        if (more) PDS_FETCH(chunk + 1)
        const float* xin = buf + b_lane;
        const float* win = buf + C::IN_CHUNK_T + lane;
#ifdef PDS_SETPRIO
        __builtin_amdgcn_s_setprio(PDS_SETPRIO);
#endif
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int ks = 0; ks < KCT / 4; ++ks) {
                float af[MB], bf[NB];
#pragma unroll
                for (int m = 0; m < MB; ++m) af[m] = win[((tap * (KCT / 4) + ks) * MB + m) * 64];
#pragma unroll
                for (int j = 0; j < NB; ++j) bf[j] = xin[ks * 4 * CS + dy * RS + j * 16 + dx];
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        PDS_X2_MFMA(acc[m][j], af[m], bf[j]);
            }
        }
#ifdef PDS_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if (more) PDS_STASH(chunk + 1, nxt)
        __syncthreads();
    }
#undef PDS_FETCH
#undef PDS_STASH

    // ---- epilogue ----------------------------------------------------------------------------
    const int y = y0 + wave;
    const bool rowok = y < A.H;
    const int jx = lane & 15, q = lane >> 4;
    float* red = lds;  // [4 waves][MB*16 channels][2]; the staging buffers are free after the last barrier
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = m * 16 + q * 4 + r;
            const bool chok = oc < A.Cout;
            const float bv = (chok && bias) ? bias[oc] : 0.f;
            float* po = A.out + (((size_t)n * A.Cout + (chok ? oc : 0)) * A.D + d) * plane + (size_t)y * A.W;
            float s = 0.f, sq = 0.f;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int x = x0 + j * 16 + jx;
                float t = acc[m][j][r] + bv;
                if (A.lrelu) t = t > 0.f ? t : t * kLeakySlope;
                if (rowok && chok && x < A.W) {
                    po[x] = t;
                    s += t;
                    sq = fmaf(t, t, sq);
                }
            }
            if (A.partials) {
                s = row16_sum(s);
                sq = row16_sum(sq);
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
            if (oc < A.Cout) {
                double v = 0.0;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) v += (double)red[((wv * MB * 16) + oc) * 2 + k];
                A.partials[((((size_t)n * A.Cout + oc) * A.D + d) * A.tiles + tile) * 2 + k] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
static int mfma_blocks(int cout) { return cout == 64 ? 4 : (cout <= 16 ? 1 : 0); }

bool conv2d_mfma_supported(const ConvLayer& L) {
    if (L.kd != 1 || L.stride != 1) return false;
    if (L.in.c % KC != 0 || L.in.c > 256) return false;
    if (mfma_blocks(L.out_g.c) == 0) return false;
    if (L.b.p && L.b.bcast_d) return false;
    if (L.l0A && L.in.h * (L.in.w + 2) * (size_t)L.in.c >= ((size_t)1 << 31)) return false;
    // offsets inside an 8-channel chunk are kept in 32-bit registers
    if ((size_t)KC * L.in.d * L.in.h * L.in.w >= ((size_t)1 << 31)) return false;
    if (L.in.d > 65535 || L.in.n > 65535) return false;
    return true;
}

int conv2d_mfma_tiles(const Geom& o) { return ((o.h + TH - 1) / TH) * ((o.w + TW - 1) / TW); }

size_t conv2d_mfma_packed_floats(int cin, int cout) {
    return (size_t)(cin / KC) * 9 * (KC / 4) * mfma_blocks(cout) * 64;
}

template <int MB, int SRC, bool PAIR = false, int KCT = KC>
static int launch_cfg(const MfmaArgs& A, hipStream_t s) {
    using C = Cfg<MB, KCT>;
    const size_t lds_bytes = (size_t)(2 * C::BUF) * sizeof(float);
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_mfma_kernel<MB, SRC, PAIR, KCT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    }
    dim3 grid(A.tiles, A.D, A.N);
    hipLaunchKernelGGL((conv2d_mfma_kernel<MB, SRC, PAIR, KCT>), grid, dim3(THREADS), lds_bytes, s, A);
    return check_launch("conv2d_mfma");
}

int launch_conv2d_mfma(const ConvLayer& L, hipStream_t s) {
    if (!L.packed) return set_error(-1, "conv2d_mfma: packed weights missing");
    const int mb = mfma_blocks(L.out_g.c);
    const int kc = KC;
    const int sets = L.plane_weight_sets > 0 ? L.plane_weight_sets : 1;
    if (L.plane_weight_sets > 0 && L.plane_weight_sets != L.in.d)
        return set_error(-1, "conv2d_mfma: %d weight sets for %d planes", L.plane_weight_sets, L.in.d);
    const int total = (int)conv2d_mfma_packed_floats(L.in.c, L.out_g.c);
    const PackPhase phase = L.sink ? L.sink->phase : kPackInline;
    if (phase != kPackDone) {
        PackJob jobs[8];
        if (sets > 8) return set_error(-1, "conv2d_mfma: too many weight sets");
        for (int i = 0; i < sets; ++i) {
            PackJob& j = jobs[i];
            j.src = L.weight + (size_t)i * L.out_g.c * L.in.c * 9;
            j.dst = L.packed + (size_t)i * total;
            j.cout = L.out_g.c;
            j.cin = L.in.c;
            j.mblocks = mb;
            j.kc = kc;
            j.taps = 9;
            j.mode = 0;
            j.total = total;
            if (phase == kPackCollect && !L.sink->push(j)) return set_error(-1, "pack job table full");
        }
        if (phase == kPackCollect) return 0;
        if (int rc = launch_multi_pack(jobs, sets, s)) return rc;
    }
    MfmaArgs A;
    A.a = L.a;
    A.b = L.b;
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
    A.tiles_x = (A.W + TW - 1) / TW;
    A.tiles = conv2d_mfma_tiles(L.out_g);
    A.l0A = L.l0A;
    A.l0G = L.l0G;
    A.l0G2 = L.l0G2;
    A.l0_cstride = L.l0_cstride;
    A.l0_rs = L.l0_rs;
    A.w_set_stride = L.plane_weight_sets > 0 ? (size_t)total : 0;
    A.bias_set_stride = L.plane_weight_sets > 0 ? L.out_g.c : 0;
    A.d_begin = L.d_begin;
    A.side_out = L.side_out;
    const bool has_b = L.b.p != nullptr, has_l0 = L.l0A != nullptr, has_a = L.a.p != nullptr;
    if (has_l0 && has_b) return set_error(-1, "conv2d_mfma: layer-0 terms and a second source together");
    const int src = has_l0 ? (has_a ? 3 : 2) : (has_b ? 1 : 0);
    if (mb == 4) {
        switch (src) {
            case 0: return launch_cfg<4, 0>(A, s);
            case 1: return launch_cfg<4, 1>(A, s);
            case 2: return launch_cfg<4, 2>(A, s);
            default: return launch_cfg<4, 3>(A, s);
        }
    }
    switch (src) {
        case 0: return launch_cfg<1, 0>(A, s);
        case 1: return launch_cfg<1, 1>(A, s);
        case 2: return launch_cfg<1, 2>(A, s);
        default: return launch_cfg<1, 3>(A, s);
    }
}

}  // namespace pds
