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
//                  source (residual sum) and literal zero padding are applied while staging.
//   epilogue       + bias, LeakyReLU(0.1), store, and per-(plane, channel) sum / sum-of-squares partials in
//                  fp64 for the deferred InstanceNorm of THIS layer (deterministic: one record per tile).
#include "common.hpp"

namespace pds {

namespace {

constexpr int TH = 4, TW = 80, NB = TW / 16, KC = 8;
constexpr int RS = 84;                               // LDS row stride (>= TW + 2)
constexpr int CS = ((TH + 2) * RS + 31) / 32 * 32 + 16;  // channel stride, == 16 (mod 32)
constexpr int IN_CHUNK = KC * CS;                    // floats
constexpr int THREADS = 256;
constexpr int IN_ELEMS = KC * (TH + 2) * (TW + 2);
constexpr int IN_ITERS = (IN_ELEMS + THREADS - 1) / THREADS;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MfmaArgs {
    Src a, b;
    const float* __restrict__ wpk;   // packed weights, see pack kernel
    const float* __restrict__ bias;
    float* __restrict__ out;
    double* __restrict__ partials;
    int N, Cin, D, H, W, Cout;
    int lrelu;
    int tiles_x, tiles;              // tiles per row of tiles, tiles per plane
};

template <int MB>
struct Cfg {
    static constexpr int W_CHUNK = 9 * (KC / 4) * MB * 64;      // floats per weight chunk
    static constexpr int BUF = IN_CHUNK + W_CHUNK;              // floats per LDS buffer
    static constexpr int W_ITERS = (W_CHUNK / 4 + THREADS - 1) / THREADS;  // float4 per thread
};

}  // namespace

// wpk[chunk][tap][ks][mb][k][i] = W[oc = mb*16 + i][c = chunk*8 + ks*4 + k][tap]   (0 when oc >= Cout)
__global__ __launch_bounds__(256) void pack_conv2d_weights_kernel(const float* __restrict__ w,
                                                                  float* __restrict__ wpk, int Cout, int Cin,
                                                                  int MBn) {
    const int total = (Cin / KC) * 9 * (KC / 4) * MBn * 64;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        int r = e;
        const int i = r % 16;
        r /= 16;
        const int k = r % 4;
        r /= 4;
        const int mb = r % MBn;
        r /= MBn;
        const int ks = r % (KC / 4);
        r /= (KC / 4);
        const int tap = r % 9;
        const int chunk = r / 9;
        const int oc = mb * 16 + i, c = chunk * KC + ks * 4 + k;
        wpk[e] = oc < Cout ? w[((size_t)oc * Cin + c) * 9 + tap] : 0.f;
    }
}

template <int MB, bool HAS_B>
__global__ __launch_bounds__(THREADS, 2) void conv2d_mfma_kernel(const MfmaArgs A) {
    using C = Cfg<MB>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // [2 buffers][input chunk | weight chunk] then coefficient table [2 src][2][Cin]
    float* coef = lds + 2 * C::BUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int d = blockIdx.y, n = blockIdx.z;
    const int ty = tile / A.tiles_x, tx = tile % A.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const size_t plane = (size_t)A.H * A.W;
    const size_t cstride = (size_t)A.D * plane;  // channel stride of NCDHW
    const int nchunks = A.Cin / KC;

    // ---- per-channel affine coefficients of the two sources into LDS ------------------------
    for (int c = tid; c < A.Cin; c += THREADS) {
        float sa = 1.f, ha = 0.f, sb = 1.f, hb = 0.f;
        if (A.a.scale) {
            const int g = A.a.per_plane ? ((n * A.Cin + c) * A.D + d) : (n * A.Cin + c);
            sa = A.a.scale[g];
            ha = A.a.shift[g];
        }
        if (HAS_B && A.b.scale) {
            const int g = A.b.per_plane ? ((n * A.Cin + c) * A.D + d) : (n * A.Cin + c);
            sb = A.b.scale[g];
            hb = A.b.shift[g];
        }
        coef[c] = sa;
        coef[A.Cin + c] = ha;
        coef[2 * A.Cin + c] = sb;
        coef[3 * A.Cin + c] = hb;
    }

    // ---- staging bookkeeping: element e of the input chunk -> (channel, row, column) ------------
    const float* pa = A.a.p + ((size_t)n * A.Cin * A.D + d) * plane;
    const float* pb = HAS_B ? A.b.p + ((size_t)n * A.Cin * A.D + d) * plane : nullptr;
    int g_off[IN_ITERS];   // offset inside a chunk of 8 channels, or -1 when padding / unused
    int l_off[IN_ITERS];   // LDS offset (floats), -1 when the slot does not exist
    int e_ch[IN_ITERS];
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int e = it * THREADS + tid;
        const int c = e / ((TH + 2) * (TW + 2));
        const int rem = e % ((TH + 2) * (TW + 2));
        const int r = rem / (TW + 2), xx = rem % (TW + 2);
        const int y = y0 - 1 + r, x = x0 - 1 + xx;
        const bool exists = e < IN_ELEMS;
        const bool inside = exists && y >= 0 && y < A.H && x >= 0 && x < A.W;
        l_off[it] = exists ? c * CS + r * RS + xx : -1;
        g_off[it] = inside ? (int)(c * cstride + (size_t)y * A.W + x) : -1;
        e_ch[it] = c;
    }

    float va[IN_ITERS], vb[IN_ITERS];
    float4 vw[C::W_ITERS];

    auto fetch = [&](int chunk) {
        const size_t cbase = (size_t)chunk * KC * cstride;
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            va[it] = g_off[it] >= 0 ? pa[cbase + g_off[it]] : 0.f;
            if (HAS_B) vb[it] = g_off[it] >= 0 ? pb[cbase + g_off[it]] : 0.f;
        }
        const float4* wsrc = reinterpret_cast<const float4*>(A.wpk + (size_t)chunk * C::W_CHUNK);
#pragma unroll
        for (int it = 0; it < C::W_ITERS; ++it) {
            const int e = it * THREADS + tid;
            if (e < C::W_CHUNK / 4) vw[it] = wsrc[e];
        }
    };
    auto stash = [&](int chunk, float* buf) {
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            if (l_off[it] >= 0) {
                const int c = chunk * KC + e_ch[it];
                float v = 0.f;
                if (g_off[it] >= 0) {
                    v = fmaf(coef[c], va[it], coef[A.Cin + c]);
                    if (HAS_B) v += fmaf(coef[2 * A.Cin + c], vb[it], coef[3 * A.Cin + c]);
                }
                buf[l_off[it]] = v;
            }
        }
        float4* wdst = reinterpret_cast<float4*>(buf + IN_CHUNK);
#pragma unroll
        for (int it = 0; it < C::W_ITERS; ++it) {
            const int e = it * THREADS + tid;
            if (e < C::W_CHUNK / 4) wdst[e] = vw[it];
        }
    };

    f32x4 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    __syncthreads();  // coefficient table visible
    stash(0, lds);
    __syncthreads();

    // lane-constant part of the fragment addresses
    const int b_lane = (lane >> 4) * CS + wave * RS + (lane & 15);

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        float* buf = lds + (chunk & 1) * C::BUF;
        if (chunk + 1 < nchunks) fetch(chunk + 1);
        const float* xin = buf + b_lane;
        const float* win = buf + IN_CHUNK + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                float af[MB], bf[NB];
#pragma unroll
                for (int m = 0; m < MB; ++m) af[m] = win[((tap * (KC / 4) + ks) * MB + m) * 64];
#pragma unroll
                for (int j = 0; j < NB; ++j) bf[j] = xin[ks * 4 * CS + dy * RS + j * 16 + dx];
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[j], acc[m][j], 0, 0, 0);
            }
        }
        if (chunk + 1 < nchunks) stash(chunk + 1, lds + ((chunk + 1) & 1) * C::BUF);
        __syncthreads();
    }

    // ---- epilogue ----------------------------------------------------------------------------
    const int y = y0 + wave;
    const bool rowok = y < A.H;
    const int jx = lane & 15, q = lane >> 4;
    double* red = reinterpret_cast<double*>(lds);  // [4 waves][MB*16 channels][2]; staging LDS is free now
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = m * 16 + q * 4 + r;
            const bool chok = oc < A.Cout;
            const float bv = (chok && A.bias) ? A.bias[oc] : 0.f;
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
                double ds = (double)s, dq = (double)sq;
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) {
                    ds += __shfl_xor(ds, off, 64);
                    dq += __shfl_xor(dq, off, 64);
                }
                if (jx == 0) {
                    red[((wave * MB * 16) + oc) * 2 + 0] = ds;
                    red[((wave * MB * 16) + oc) * 2 + 1] = dq;
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
                for (int wv = 0; wv < 4; ++wv) v += red[((wv * MB * 16) + oc) * 2 + k];
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
    // offsets inside an 8-channel chunk are kept in 32-bit registers
    if ((size_t)KC * L.in.d * L.in.h * L.in.w >= ((size_t)1 << 31)) return false;
    if (L.in.d > 65535 || L.in.n > 65535) return false;
    return true;
}

int conv2d_mfma_tiles(const Geom& o) { return ((o.h + TH - 1) / TH) * ((o.w + TW - 1) / TW); }

size_t conv2d_mfma_packed_floats(int cin, int cout) {
    return (size_t)(cin / KC) * 9 * (KC / 4) * mfma_blocks(cout) * 64;
}

template <int MB, bool HAS_B>
static int launch_cfg(const MfmaArgs& A, hipStream_t s) {
    using C = Cfg<MB>;
    const size_t lds_bytes = (size_t)(2 * C::BUF + 4 * A.Cin) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_mfma_kernel<MB, HAS_B>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        attr_done = true;
    }
    dim3 grid(A.tiles, A.D, A.N);
    hipLaunchKernelGGL((conv2d_mfma_kernel<MB, HAS_B>), grid, dim3(THREADS), lds_bytes, s, A);
    return check_launch("conv2d_mfma");
}

int launch_conv2d_mfma(const ConvLayer& L, hipStream_t s) {
    if (!L.packed) return set_error(-1, "conv2d_mfma: packed weights missing");
    const int mb = mfma_blocks(L.out_g.c);
    {
        const int total = (int)conv2d_mfma_packed_floats(L.in.c, L.out_g.c);
        hipLaunchKernelGGL(pack_conv2d_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, L.weight,
                           L.packed, L.out_g.c, L.in.c, mb);
        if (int rc = check_launch("pack_conv2d_weights")) return rc;
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
    const bool has_b = L.b.p != nullptr;
    if (mb == 4) return has_b ? launch_cfg<4, true>(A, s) : launch_cfg<4, false>(A, s);
    return has_b ? launch_cfg<1, true>(A, s) : launch_cfg<1, false>(A, s);
}

}  // namespace pds
