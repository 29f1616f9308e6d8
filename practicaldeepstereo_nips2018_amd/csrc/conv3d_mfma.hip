// Exact-fp32 MFMA implicit-GEMM 3x3x3 convolution (stride 1 or 2, pad 1) for the hourglass of
// reference practical_deep_stereo/regularization.py:77-82,85-86 (network_blocks.py:61-72, 106-121).
//
// Same scheme as conv2d_mfma.hip, generalised to volumes:
//   GEMM view   M = output channels (16 per block), N = 16 consecutive output x, K = Cin * 27 walked as
//               KC-channel chunks x 27 taps x KC/4 k-steps; v_mfma_f32_16x16x4_f32 (exact fp32).
//   workgroup   4 waves; output tile TZ x TY rows of 16*NB columns, MB channel blocks (grid.y walks the
//               remaining channel blocks).  Each wave owns RW = TZ*TY/4 rows.
//   LDS         double-buffered [input halo tile of KC channels | weight fragments of the chunk]; the input
//               channel stride is padded so the two k-halves of a 32-lane group hit disjoint banks
//               (stride 1: == 16 mod 32; stride 2, where lanes step 2 floats: odd).
//   prologue    deferred InstanceNorm of the producer(s), skip sum (second source, optionally broadcast
//               along D: regularization.py:115,119) and zero padding folded into the staging.
//   epilogue    bias, LeakyReLU(0.1), store, per-(n, channel) partial sums for InstanceNorm3d, one
//               deterministic record per tile.
// Transposed convolutions (network_blocks.py:37-44, 75-85, 124-131: kernel 4 / stride 2 / pad 1, and the
// final (3,4,4) / (1,2,2) layer) run on the same kernel (MODE 1 / 2): a stride-2 transposed convolution is a
// stride-1 3x3x3 convolution on the INPUT grid with one group of "virtual" output channels per output
// parity class (8 resp. 4 groups), v = class * Cout + oc, whose weights are the transposed-conv taps that
// parity uses (out = 2*i - 1 + k: even outputs use (i, k=1), (i-1, k=3); odd outputs (i, k=2), (i+1, k=0))
// and zero elsewhere; taps that are structurally zero for a whole 16-channel block are skipped through a
// per-block tap mask, so the MFMA count equals the class-wise optimum.  The epilogue scatters virtual
// channel (class, oc) at input position (z, y, x) to output (2z+pd, 2y+ph, 2x+pw).
// The small deep layers of the hourglass are latency-bound, so their configurations use 16-channel
// chunks (fewer global round trips) and a single channel block per workgroup (more workgroups).
#include "common.hpp"

namespace pds {

namespace {

constexpr int THREADS = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// timing-decomposition hooks (tools/build_variant.sh): any of them set makes the results wrong
#ifdef PDS_X3_NOLOAD
#define PDS_X3_LOAD(x) (float)(tid)
#else
#define PDS_X3_LOAD(x) (x)
#endif
#ifdef PDS_X3_NOMFMA
#define PDS_X3_MFMA(c, a, b) (c)[0] += (a) * (b)
#else
#define PDS_X3_MFMA(c, a, b) (c) = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif

struct Args3 {
    Src a, b;
    const float* __restrict__ wpk;
    const float* __restrict__ bias;
    float* __restrict__ out;
    double* __restrict__ partials;
    int N, Cin, Di, Hi, Wi;
    int Cout, Do, Ho, Wo;
    int lrelu;
    int tiles_x, tiles_y, tiles;  // tiles per batch element = tiles_x * tiles_y * tiles_z
    int mblocks;                  // ceil(virtual Cout / 16)
    int Creal;                    // real output channels (== Cout for a convolution)
    const unsigned* __restrict__ tapmask;  // [mblocks] 27-bit masks of the taps a channel block uses
    int xcd_run;                  // tiles per XCD (ceil(tiles / 8)); 0: identity mapping
};

template <int S, int MB, int TZ, int TY, int NB, int KC>
struct Cfg3 {
    static constexpr int RW = TZ * TY / 4;
    static constexpr int ZT = (TZ - 1) * S + 3, YT = (TY - 1) * S + 3, XT = (16 * NB - 1) * S + 3;
    static constexpr int RS = XT;
    static constexpr int CS_RAW = ZT * YT * RS;
    static constexpr int CS = (S == 1) ? ((CS_RAW + 15) / 32 * 32 + 16) : (CS_RAW | 1);
    static constexpr int IN_CHUNK = (KC * CS + 3) / 4 * 4;
    static constexpr int KS = KC / 4;
    static constexpr int W_CHUNK = 27 * KS * MB * 64;
    static constexpr int BUF = IN_CHUNK + W_CHUNK;
    static constexpr int NPOS = ZT * YT * XT;
    static constexpr int POS = (NPOS + THREADS - 1) / THREADS;
    static constexpr int W_ITERS = (W_CHUNK / 4 + THREADS - 1) / THREADS;
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF * sizeof(float);
    static_assert(TZ * TY % 4 == 0, "rows must split over 4 waves");
    static_assert(CS >= CS_RAW, "bad padding");
};

__device__ __forceinline__ float row16_sum3(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

}  // namespace

// MODE 0: convolution; 1: transposed k4 s2 p1 (8 classes); 2: transposed k(3,4,4) s(1,2,2) p1 (4 classes)
template <int MODE, int S, int MB, int TZ, int TY, int NB, int KC>
__global__ __launch_bounds__(THREADS) void conv3d_mfma_kernel(const Args3 A) {
    static_assert(MODE == 0 || (S == 1 && MB == 1), "transposed convolutions use stride-1 tiles, one block");
    using C = Cfg3<S, MB, TZ, TY, NB, KC>;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs (one L2 each) in launch order; with the
    // launch index re-mapped every XCD works on a contiguous run of tiles (a slab of z), so the z / y halo planes that
    // neighbouring tiles share are re-read from the same L2 instead of being fetched through eight of them.
    int tile = blockIdx.x;
    if (A.xcd_run > 0) {
        tile = (int)(blockIdx.x & 7) * A.xcd_run + (int)(blockIdx.x >> 3);
        if (tile >= A.tiles) return;  // grid.x is padded to a multiple of 8
    }
    const int mb0 = blockIdx.y * MB;  // first channel block of this workgroup
    const int n = blockIdx.z;
    const int tx = tile % A.tiles_x;
    const int ty = (tile / A.tiles_x) % A.tiles_y;
    const int tz = tile / (A.tiles_x * A.tiles_y);
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * 16 * NB;  // output coordinates
    const size_t plane_i = (size_t)A.Hi * A.Wi;
    const size_t cstride_a = (size_t)A.Di * plane_i;
    const bool hasb = A.b.p != nullptr;
    const size_t cstride_b = A.b.bcast_d ? plane_i : cstride_a;
    const int nchunks = (A.Cin + KC - 1) / KC;
    const unsigned tapmask = MODE != 0 ? A.tapmask[mb0] : 0x7ffffffu;

    unsigned ga[C::POS], gb[C::POS];   // 32-bit offsets inside one channel (conv3d_mfma_supported bounds the volume)
    int lo[C::POS];
    bool inside[C::POS];
#pragma unroll
    for (int k = 0; k < C::POS; ++k) {
        const int p = min(tid + k * THREADS, C::NPOS - 1);
        const int xx = p % C::XT, yy = (p / C::XT) % C::YT, zz = p / (C::XT * C::YT);
        const int z = z0 * S - 1 + zz, y = y0 * S - 1 + yy, x = x0 * S - 1 + xx;
        inside[k] = z >= 0 && z < A.Di && y >= 0 && y < A.Hi && x >= 0 && x < A.Wi;
        const int zc = min(max(z, 0), A.Di - 1), yc = min(max(y, 0), A.Hi - 1), xc = min(max(x, 0), A.Wi - 1);
        ga[k] = (unsigned)((zc * A.Hi + yc) * A.Wi + xc);
        gb[k] = A.b.bcast_d ? (unsigned)(yc * A.Wi + xc) : ga[k];
        lo[k] = (zz * C::YT + yy) * C::RS + xx;
    }
    const float* pa = A.a.p + (size_t)n * A.Cin * cstride_a;
    const float* pb = hasb ? A.b.p + (size_t)n * A.Cin * cstride_b : nullptr;
    const int wlast = C::W_CHUNK / 4 - 1;
    const int wrow = A.mblocks * 64;  // floats per (chunk, tap, ks) row of the packed weights

    float va[KC][C::POS], vb[KC][C::POS];
    f32x4 vw[C::W_ITERS];

#define PDS_FETCH3(chunk_)                                                                              \
    {                                                                                                   \
        /* the second source is tested ONCE per chunk, not per load: straight-line bursts of loads from */ \
        /* wave-uniform channel pointers plus 32-bit lane offsets                                         */ \
        _Pragma("unroll") for (int c = 0; c < KC; ++c) {                                                \
            const float* ca = pa + (size_t)min((chunk_) * KC + c, A.Cin - 1) * cstride_a;               \
            _Pragma("unroll") for (int k = 0; k < C::POS; ++k) va[c][k] = PDS_X3_LOAD(ca[ga[k]]);       \
        }                                                                                               \
        if (hasb) {                                                                                     \
            _Pragma("unroll") for (int c = 0; c < KC; ++c) {                                            \
                const float* cb = pb + (size_t)min((chunk_) * KC + c, A.Cin - 1) * cstride_b;           \
                _Pragma("unroll") for (int k = 0; k < C::POS; ++k) vb[c][k] = PDS_X3_LOAD(cb[gb[k]]);   \
            }                                                                                           \
        }                                                                                               \
        const float* wsrc = A.wpk + (size_t)(chunk_) * 27 * C::KS * wrow + mb0 * 64;                    \
        _Pragma("unroll") for (int it = 0; it < C::W_ITERS; ++it) {                                     \
            const int e = min(it * THREADS + tid, wlast);       /* float4 index inside the LDS image */ \
            const int row = e / (MB * 16), col = e % (MB * 16); /* row = tap*KS + ks, col in float4 */  \
            vw[it] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)row * wrow + col * 4);              \
        }                                                                                               \
    }

#define PDS_STASH3(chunk_, buf_)                                                                        \
    {                                                                                                   \
        _Pragma("unroll") for (int c = 0; c < KC; ++c) {                                                \
            const int chr = (chunk_) * KC + c;                                                          \
            const bool chv = chr < A.Cin;                                                               \
            const int ch = min(chr, A.Cin - 1);                                                         \
            float sa = 1.f, ha = 0.f, sb = 1.f, hb = 0.f;                                               \
            if (A.a.scale) {                                                                            \
                sa = A.a.scale[n * A.Cin + ch];                                                         \
                ha = A.a.shift[n * A.Cin + ch];                                                         \
            }                                                                                           \
            if (hasb && A.b.scale) {                                                                    \
                sb = A.b.scale[n * A.Cin + ch];                                                         \
                hb = A.b.shift[n * A.Cin + ch];                                                         \
            }                                                                                           \
            if (hasb) {                                                                                 \
                _Pragma("unroll") for (int k = 0; k < C::POS; ++k) {                                    \
                    const float v = fmaf(sa, va[c][k], ha) + fmaf(sb, vb[c][k], hb);                    \
                    (buf_)[c * C::CS + lo[k]] = (inside[k] && chv) ? v : 0.f;                           \
                }                                                                                       \
            } else {                                                                                    \
                _Pragma("unroll") for (int k = 0; k < C::POS; ++k)                                      \
                    (buf_)[c * C::CS + lo[k]] = (inside[k] && chv) ? fmaf(sa, va[c][k], ha) : 0.f;      \
            }                                                                                           \
        }                                                                                               \
        f32x4* wdst = reinterpret_cast<f32x4*>((buf_) + C::IN_CHUNK);                                   \
        _Pragma("unroll") for (int it = 0; it < C::W_ITERS; ++it)                                       \
            wdst[min(it * THREADS + tid, wlast)] = vw[it];                                              \
    }

    f32x4 acc[MB][C::RW][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < C::RW; ++r)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[m][r][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    PDS_FETCH3(0)
    PDS_STASH3(0, lds)
    __syncthreads();

    // lane-constant fragment address: channel (lane >> 4), column (lane & 15) * S
    int b_row[C::RW];
#pragma unroll
    for (int r = 0; r < C::RW; ++r) {
        const int rho = wave * C::RW + r;
        const int zr = rho / TY, yr = rho % TY;
        b_row[r] = (lane >> 4) * C::CS + ((zr * S) * C::YT + yr * S) * C::RS + (lane & 15) * S;
    }

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        float* buf = lds + (chunk & 1) * C::BUF;
        float* nxt = lds + ((chunk + 1) & 1) * C::BUF;
        const bool more = chunk + 1 < nchunks;
        if (more) PDS_FETCH3(chunk + 1)
        const float* win = buf + C::IN_CHUNK + lane;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            if (MODE != 0 && !((tapmask >> tap) & 1u)) continue;  // wave-uniform: structurally zero weights
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                float af[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) af[m] = win[((tap * C::KS + ks) * MB + m) * 64];
#pragma unroll
                for (int r = 0; r < C::RW; ++r) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const float bf =
                            buf[b_row[r] + ks * 4 * C::CS + (dz * C::YT + dy) * C::RS + j * 16 * S + dx];
#pragma unroll
                        for (int m = 0; m < MB; ++m)
                            PDS_X3_MFMA(acc[m][r][j], af[m], bf);
                    }
                }
            }
        }
        if (more) PDS_STASH3(chunk + 1, nxt)
        __syncthreads();
    }
#undef PDS_FETCH3
#undef PDS_STASH3

    // ---- epilogue ---------------------------------------------------------------------------------
    const int jx = lane & 15, q = lane >> 4;
    const size_t plane_o = (size_t)A.Ho * A.Wo;
    constexpr int NCLS = MODE == 1 ? 8 : (MODE == 2 ? 4 : 1);
    float* red = lds;  // [4 waves][MB*16][2]
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int v = (mb0 + m) * 16 + q * 4 + rr;       // (virtual) output channel
            const bool chok = v < A.Cout;
            const int cls = MODE == 0 ? 0 : v / A.Creal;
            const int oc = MODE == 0 ? v : v - cls * A.Creal;
            const int pd = MODE == 1 ? (cls >> 2) & 1 : 0, ph = (cls >> 1) & 1, pw = cls & 1;
            const float bv = (chok && A.bias) ? A.bias[oc] : 0.f;
            float s = 0.f, sq = 0.f;
#pragma unroll
            for (int r = 0; r < C::RW; ++r) {
                const int rho = wave * C::RW + r;
                const int z = z0 + rho / TY, y = y0 + rho % TY;   // tile coordinates (input grid when MODE != 0)
                int oz = z, oy = y;
                bool rowok;
                if (MODE == 0) {
                    rowok = chok && z < A.Do && y < A.Ho;
                } else {
                    rowok = chok && z < A.Di && y < A.Hi;
                    oz = MODE == 1 ? 2 * z + pd : z;
                    oy = 2 * y + ph;
                }
                float* po = A.out + (((size_t)n * A.Creal + (chok ? oc : 0)) * A.Do + min(oz, A.Do - 1)) * plane_o +
                            (size_t)min(oy, A.Ho - 1) * A.Wo;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int x = x0 + j * 16 + jx;
                    float t = acc[m][r][j][rr] + bv;
                    if (A.lrelu) t = t > 0.f ? t : t * kLeakySlope;
                    const bool ok = rowok && (MODE == 0 ? x < A.Wo : x < A.Wi);
                    if (ok) {
                        po[MODE == 0 ? x : 2 * x + pw] = t;
                        s += t;
                        sq = fmaf(t, t, sq);
                    }
                }
            }
            if (A.partials) {
                s = row16_sum3(s);
                sq = row16_sum3(sq);
                if (jx == 15) {
                    red[((wave * MB + m) * 16 + q * 4 + rr) * 2 + 0] = s;
                    red[((wave * MB + m) * 16 + q * 4 + rr) * 2 + 1] = sq;
                }
            }
        }
    }
    if (A.partials) {
        __syncthreads();
        if (tid < MB * 16 * 2) {
            const int ocl = tid >> 1, k = tid & 1;
            const int v = mb0 * 16 + ocl;
            if (v < A.Cout) {
                double sum = 0.0;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) sum += (double)red[((wv * MB * 16) + ocl) * 2 + k];
                const int cls = MODE == 0 ? 0 : v / A.Creal;
                const int oc = MODE == 0 ? v : v - cls * A.Creal;
                // records of one real channel: [tile][class]
                A.partials[((((size_t)n * A.Creal + oc) * A.tiles + tile) * NCLS + cls) * 2 + k] = sum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
namespace {

struct Plan3 {
    int id;   // configuration index
    int mb, tz, ty, nb, kc;
};

// Configuration choice by output geometry and channel counts.
Plan3 choose_plan(const Geom& o, int cin, int stride) {
    const int mblocks = (o.c + 15) / 16;
    if (stride == 1) {
        if (mblocks == 1 && (size_t)o.d * o.h * o.w >= 100000) {
            if (o.w % 80 == 0) return Plan3{0, 1, 2, 4, 5, 4};
            return Plan3{1, 1, 2, 4, 4, 4};
        }
        if (o.w > 32 && mblocks >= 2 && cin <= 32) return Plan3{2, 2, 2, 2, 4, 4};
        if (o.w > 32) return Plan3{3, 1, 2, 2, 4, 4};
        if (o.w > 16) return Plan3{4, 1, 2, 2, 2, 16};
        return Plan3{5, 1, 2, 2, 1, 16};
    }
    // large stride-2 layers (c0.down: 8 -> 16 from the full-resolution volume): the 129-column halo tile of the 64-wide
    // plan takes 117 KB of LDS -- one workgroup, one wave per SIMD on a CU, nothing to hide the staging behind.  16-wide
    // tiles (40 KB, three workgroups per CU): 64 -> 45 us at 24 x 72 x 120 (32-wide: 48)
    if (o.w > 32 && (size_t)o.d * o.h * o.w >= 100000) return Plan3{9, 1, 2, 2, 1, 4};
    if (o.w > 32) return Plan3{6, 1, 2, 2, 4, 4};
    if (o.w > 16) return Plan3{7, 1, 2, 2, 2, 8};
    return Plan3{8, 1, 2, 2, 1, 8};
}

template <int MODE, int S, int MB, int TZ, int TY, int NB, int KC>
int launch3(const Args3& A0, hipStream_t s) {
    using C = Cfg3<S, MB, TZ, TY, NB, KC>;
    Args3 A = A0;
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_mfma_kernel<MODE, S, MB, TZ, TY, NB, KC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    }
    static const bool xcd_map = []() {  // PDS_CONV3D_XCD_MAP=0: launch order = tile order (A/B)
        const char* e = debug_switch("PDS_CONV3D_XCD_MAP");
        return !(e && e[0] == '0');
    }();
    A.xcd_run = (xcd_map && A.tiles >= 64) ? (A.tiles + 7) / 8 : 0;
    dim3 grid(A.xcd_run > 0 ? 8 * A.xcd_run : A.tiles, (A.mblocks + MB - 1) / MB, A.N);
    hipLaunchKernelGGL((conv3d_mfma_kernel<MODE, S, MB, TZ, TY, NB, KC>), grid, dim3(THREADS), C::LDS_BYTES, s, A);
    return check_launch("conv3d_mfma");
}

}  // namespace

bool conv3d_mfma_supported(const ConvLayer& L) {
    if (L.kd != 3) return false;
    if (L.in.c % 4 != 0 || L.in.c < 8) return false;
    if (L.stat_per_plane) return false;
    if ((L.a.scale && L.a.per_plane) || (L.b.scale && L.b.per_plane)) return false;
    if ((size_t)L.in.d * L.in.h * L.in.w >= ((size_t)1 << 31)) return false;
    if (L.in.n > 65535) return false;
    const Plan3 p = choose_plan(L.out_g, L.in.c, L.stride);
    if (L.in.c % p.kc != 0 && p.kc > 4) {
        // chunks may overhang Cin (zero-filled), but keep it to the 4-channel granularity
        if (L.in.c % 4 != 0) return false;
    }
    return true;
}

int conv3d_mfma_tiles(const Geom& o, int cin, int stride) {
    const Plan3 p = choose_plan(o, cin, stride);
    return ((o.w + 16 * p.nb - 1) / (16 * p.nb)) * ((o.h + p.ty - 1) / p.ty) * ((o.d + p.tz - 1) / p.tz);
}

size_t conv3d_mfma_packed_floats(const Geom& o, int cin, int stride) {
    const Plan3 p = choose_plan(o, cin, stride);
    const int chunks = (cin + p.kc - 1) / p.kc;
    return (size_t)chunks * 27 * (p.kc / 4) * ((o.c + 15) / 16) * 64;
}

int launch_conv3d_mfma(const ConvLayer& L, hipStream_t s) {
    if (!L.packed) return set_error(-1, "conv3d_mfma: packed weights missing");
    const Plan3 p = choose_plan(L.out_g, L.in.c, L.stride);
    Args3 A;
    A.a = L.a;
    A.b = L.b;
    A.wpk = L.packed;
    A.bias = L.bias;
    A.out = L.out;
    A.partials = L.partials;
    A.N = L.in.n;
    A.Cin = L.in.c;
    A.Di = L.in.d;
    A.Hi = L.in.h;
    A.Wi = L.in.w;
    A.Cout = L.out_g.c;
    A.Do = L.out_g.d;
    A.Ho = L.out_g.h;
    A.Wo = L.out_g.w;
    A.lrelu = L.lrelu;
    A.tiles_x = (A.Wo + 16 * p.nb - 1) / (16 * p.nb);
    A.tiles_y = (A.Ho + p.ty - 1) / p.ty;
    A.tiles = conv3d_mfma_tiles(L.out_g, L.in.c, L.stride);
    A.mblocks = (A.Cout + 15) / 16;
    A.Creal = A.Cout;
    A.tapmask = nullptr;
    {
        const PackPhase phase = L.sink ? L.sink->phase : kPackInline;
        if (phase != kPackDone) {
            PackJob j;
            j.src = L.weight;
            j.dst = L.packed;
            j.cout = A.Cout;
            j.cin = A.Cin;
            j.mblocks = A.mblocks;
            j.kc = p.kc;
            j.taps = 27;
            j.mode = 0;
            j.total = (int)conv3d_mfma_packed_floats(L.out_g, L.in.c, L.stride);
            if (phase == kPackCollect) return L.sink->push(j) ? 0 : set_error(-1, "pack job table full");
            if (int rc = launch_multi_pack(&j, 1, s)) return rc;
        }
    }
    switch (p.id) {
        case 0: return launch3<0, 1, 1, 2, 4, 5, 4>(A, s);
        case 1: return launch3<0, 1, 1, 2, 4, 4, 4>(A, s);
        case 2: return launch3<0, 1, 2, 2, 2, 4, 4>(A, s);
        case 3: return launch3<0, 1, 1, 2, 2, 4, 4>(A, s);
        case 4: return launch3<0, 1, 1, 2, 2, 2, 16>(A, s);
        case 5: return launch3<0, 1, 1, 2, 2, 1, 16>(A, s);
        case 6: return launch3<0, 2, 1, 2, 2, 4, 4>(A, s);
        case 7: return launch3<0, 2, 1, 2, 2, 2, 8>(A, s);
        case 8: return launch3<0, 2, 1, 2, 2, 1, 8>(A, s);
        case 9: return launch3<0, 2, 1, 2, 2, 1, 4>(A, s);
    }
    return set_error(-1, "conv3d_mfma: no configuration");
}

// ---------------------------------------------------------------------------------------------------
// transposed convolutions
// ---------------------------------------------------------------------------------------------------
namespace {

Plan3 choose_plan_deconv(const Geom& in) {
    if ((size_t)in.d * in.h * in.w >= 100000) {
        if (in.w % 80 == 0) return Plan3{0, 1, 2, 4, 5, 4};
        return Plan3{1, 1, 2, 4, 4, 4};
    }
    if (in.w > 32) return Plan3{3, 1, 2, 2, 4, 4};
    if (in.w > 16) return Plan3{4, 1, 2, 2, 2, 16};
    return Plan3{5, 1, 2, 2, 1, 16};
}

}  // namespace

bool deconv3d_mfma_supported(const DeconvLayer& L) {
    if (L.kd != 4 && L.kd != 3) return false;
    if (L.in.c % 4 != 0) return false;
    if ((L.a.scale && L.a.per_plane) || (L.b.scale && L.b.per_plane)) return false;
    if ((size_t)L.out_g.d * L.out_g.h * L.out_g.w >= ((size_t)1 << 31)) return false;
    if (L.in.n > 65535) return false;
    return true;
}

static int deconv_classes(int kd) { return kd == 4 ? 8 : 4; }

int deconv3d_mfma_tiles(const Geom& in) {
    const Plan3 p = choose_plan_deconv(in);
    return ((in.w + 16 * p.nb - 1) / (16 * p.nb)) * ((in.h + p.ty - 1) / p.ty) * ((in.d + p.tz - 1) / p.tz);
}

// floats of scratch: packed virtual weights followed by the tap masks
size_t deconv3d_mfma_packed_floats(const Geom& in, int cout, int kd) {
    const Plan3 p = choose_plan_deconv(in);
    const int chunks = (in.c + p.kc - 1) / p.kc;
    const int mblocks = (deconv_classes(kd) * cout + 15) / 16;
    return (size_t)chunks * 27 * (p.kc / 4) * mblocks * 64 + mblocks + 64;
}

int launch_deconv3d_mfma(const DeconvLayer& L, hipStream_t s) {
    if (!L.packed) return set_error(-1, "deconv3d_mfma: packed weights missing");
    const Plan3 p = choose_plan_deconv(L.in);
    const int mode = L.kd == 4 ? 1 : 2;
    const int ncls = deconv_classes(L.kd);
    Args3 A;
    A.a = L.a;
    A.b = L.b;
    A.wpk = L.packed;
    A.bias = L.bias;
    A.out = L.out;
    A.partials = L.partials;
    A.N = L.in.n;
    A.Cin = L.in.c;
    A.Di = L.in.d;
    A.Hi = L.in.h;
    A.Wi = L.in.w;
    A.Creal = L.out_g.c;
    A.Cout = ncls * L.out_g.c;  // virtual channels
    A.Do = L.out_g.d;
    A.Ho = L.out_g.h;
    A.Wo = L.out_g.w;
    A.lrelu = L.lrelu;
    A.tiles_x = (A.Wi + 16 * p.nb - 1) / (16 * p.nb);
    A.tiles_y = (A.Hi + p.ty - 1) / p.ty;
    A.tiles = deconv3d_mfma_tiles(L.in);
    A.mblocks = (A.Cout + 15) / 16;
    const int chunks = (A.Cin + p.kc - 1) / p.kc;
    const size_t wfloats = (size_t)chunks * 27 * (p.kc / 4) * A.mblocks * 64;
    unsigned* mask = reinterpret_cast<unsigned*>(L.packed + wfloats);
    A.tapmask = mask;
    {
        const PackPhase phase = L.sink ? L.sink->phase : kPackInline;
        if (phase != kPackDone) {
            PackJob j;
            j.src = L.weight;
            j.dst = L.packed;
            j.mask = mask;
            j.cout = A.Creal;
            j.cin = A.Cin;
            j.mblocks = A.mblocks;
            j.kc = p.kc;
            j.taps = 27;
            j.mode = mode;
            j.total = (int)wfloats;
            if (phase == kPackCollect) return L.sink->push(j) ? 0 : set_error(-1, "pack job table full");
            if (int rc = launch_multi_pack(&j, 1, s)) return rc;
        }
    }
    if (mode == 1) {
        switch (p.id) {
            case 0: return launch3<1, 1, 1, 2, 4, 5, 4>(A, s);
            case 1: return launch3<1, 1, 1, 2, 4, 4, 4>(A, s);
            case 3: return launch3<1, 1, 1, 2, 2, 4, 4>(A, s);
            case 4: return launch3<1, 1, 1, 2, 2, 2, 16>(A, s);
            case 5: return launch3<1, 1, 1, 2, 2, 1, 16>(A, s);
        }
    } else {
        switch (p.id) {
            case 0: return launch3<2, 1, 1, 2, 4, 5, 4>(A, s);
            case 1: return launch3<2, 1, 1, 2, 4, 4, 4>(A, s);
            case 3: return launch3<2, 1, 1, 2, 2, 4, 4>(A, s);
            case 4: return launch3<2, 1, 1, 2, 2, 2, 16>(A, s);
            case 5: return launch3<2, 1, 1, 2, 2, 1, 16>(A, s);
        }
    }
    return set_error(-1, "deconv3d_mfma: no configuration");
}

}  // namespace pds
