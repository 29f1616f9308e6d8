// conv2d 3x3 (pad 1, stride 1), Cin -> 64 channels over all (batch, disparity) planes in one launch, with fp32
// operands emulated on the 16-bit matrix pipe: the dominant layers of MatchingOperation (reference
// practical_deep_stereo/matching.py:85-88, network_blocks.py:134-144; also the 64-channel layers of embedding.py).
//
// Arithmetic.  gfx950 runs v_mfma_f32_16x16x4_f32 at 1/16 of the 16-bit MFMA rate, so the exact-fp32 Winograd kernel
// (conv2d_wino16.hip) is pipe-bound at ~106 executed TFLOP/s.  Here every fp32 operand is split into 16-bit parts and the
// product is the sum of the leading partial products, each an exact 16-bit x 16-bit product accumulated in fp32 by
// v_mfma_f32_32x32x16_{f16,bf16}.  Two forms (template parameter P = parts per operand):
//   P = 2, fp16 (round 3b)  a = a1 + a2 (2 x 11 significand bits; |a - a1 - a2| <= 2^-22 |a|),
//       a*b ~= a1*b2 + a2*b1 + a1*b1: THREE products.  fp16 has a 5-bit exponent, so both operands are pre-scaled by
//       exact powers of two that are DERIVED FROM THE DATA (round 4; rounds 3's constants 2^10 / 2^4 overflowed for
//       |w| >= 64 or large gamma): weights by ws = the largest power of two with ws max|w| <= 2^14, found when they are
//       packed (pack.hip mode 7); activations by as = the largest power of two with as * bound <= 2^14, where `bound` is
//       the range certificate that travels with the source (common.hpp Src::bound: |gamma| sqrt(count) + |beta| behind
//       an InstanceNorm, the producer's own per-workgroup maxima for a plain tensor); the epilogue multiplies the sums
//       by 1 / (ws as).  Any finite operands are therefore in range, low parts stay in or near the normal range (the
//       format has 30 binades for 22 bits, a bound may be loose by 2^8 at no cost) and MFMA inputs keep their
//       subnormals.  A source WITHOUT a certificate (a caller's tensor of unknown scale) takes the bf16 form below.
//       Measured on an MI355X (tools/ubench/fp16x2_probe.hip, K = 576 as in this layer): mean |error| 2.0e-7 against
//       3.1e-7 of the fp32 fmaf chain, 1 800 TFLOP/s of fp16 = 600 fp32-equivalent TFLOP/s (power-limited clock):
//       0.20 ms of bare MFMA time for this layer's 122.3 GFLOP at config 2.
//   P = 3, bf16 (round 3a)  a = a1 + a2 + a3 exactly (3 x 8 bits), SIX products of order <= 2^-16 (dropped:
//       a2*b3 + a3*b2 + a3*b3 <= 2^-23 |a*b|); bf16 has the fp32 exponent, so this form is range-safe: it takes inputs of
//       unknown scale (data gradients, plain tensors of the generic entry point).  Mean |error| 2.4e-7; 320
//       fp32-equivalent TFLOP/s.
// Weights are split once (round-to-nearest, pack.hip modes 6 / 7), activations while they are staged.
// Shape.  One persistent 512-thread workgroup per CU (all of its LDS) pulls 16 x 32-pixel tiles of one plane from eight
// per-XCD queues (atomic counters; planes stay on one XCD's L2, idle workgroups steal).  A tile is a GEMM
// [512 px] x [64 oc] x [K = Cin * 9] with the PIXELS on the M side, and the waves are specialised:
//   waves 0-3   MFMA waves, one per SIMD: wave w owns rows 4w .. 4w+3 of the tile (four M blocks of one row x 32
//               columns) and all 64 output channels (two N blocks): 4 x 2 accumulator tiles of 32 x 32 = 128 VGPRs.
//               They only read fragments (ds_read_b128, double-buffered by half-taps) and issue MFMAs: per tap
//               4P pixel + 2P weight fragments feed 8 x (3 or 6) MFMAs.  In the D fragment a lane holds one output channel
//               and FOUR CONSECUTIVE pixels per register quad, so the epilogue stores 16 bytes per lane and the
//               InstanceNorm statistics are sums over a lane's own registers (plus one lane exchange).
//               Tiles whose right half lies outside the image (240 = 7.5 x 32) map their M blocks to 2 rows x 16
//               columns instead: two M blocks per wave, half the MFMAs, none spent off the plane.
//   waves 4-7   staging waves, one per SIMD: global loads, the deferred InstanceNorm of the producer, the bf16 split
//               and the LDS writes run beside the MFMA waves' matrix work (separate pipes, the hardware interleaves
//               the waves); they also draw the next tile and fold the statistics records.
//   K-step      16 input channels; stage = (K-step, dy) = 3 taps, one barrier per stage (72 / 144 MFMAs per MFMA wave).
//   LDS         inputs  IN[2][part P][channel group 2][18 rows][34 columns][8 x 16 bit]   2 x P x 19 584 B, written once per
//               K-step and read by all nine taps; a lane's fragment is one 16-byte slot and the 32 lanes of an M block
//               read 32 consecutive slots (conflict free for any row);
//               weights W[2][dx 3][part P][N block 2][64 lanes][16 B]                  2 x P x 6 144 B per stage, in
//               fragment order (lane-linear reads).
//   staging     one stage ahead: what was requested during stage s-1 is converted and written during stage s (inputs
//               of the next K-step by thirds, weights of stage s+1), then the next requests are issued; the sequence
//               runs across tile boundaries (the next tile is drawn at stage 0), so a CU never drains between tiles.
//   epilogue    bias, LeakyReLU, per-(plane, channel) sum / sum of squares -> one fp64 record per tile (the deferred
//               InstanceNorm of common.hpp), folded by the staging waves during the next tile.  Stores: the fp16 form
//               transposes every M block through a wave-private LDS area so that a store instruction writes 8 channels x
//               128 contiguous bytes (the bf16 form and border tiles: 16 bytes per lane straight from the D fragment,
//               i.e. 32-byte pieces in 32 channel planes per instruction -- 10 400 against 7 100 cycles per tile).
#include <type_traits>

#include "common.hpp"

namespace pds {

namespace {

constexpr int TH = 16, TW = 32, ROWS = TH + 2, COLS = TW + 2, PIX = ROWS * COLS;   // halo tile 18 x 34 = 612 pixels
constexpr int THREADS = 512, STAGERS = 256;   // waves 0-3 MFMA, waves 4-7 staging
constexpr int IN_PART = 2 * PIX * 16;          // bytes of one split part: [channel group][row][column][8 x 16 bit]
constexpr int W_FRAG = 64 * 16;                // one B fragment
constexpr int CMAX = 256;                      // most input channels with a deferred InstanceNorm on the input
constexpr int THIRD_ROWS = ROWS / 3, THIRD_PIX = THIRD_ROWS * COLS;   // 204 staging items per channel group and third
static_assert(THIRD_PIX <= STAGERS, "one pixel per staging thread and third");

// LDS map and arithmetic constants of the two forms: P parts per operand (3: bf16, six products; 2: fp16, three products)
template <int P>
struct X3Cfg {
    static constexpr int IN_BUF = P * IN_PART;             // 58 752 / 39 168
    static constexpr int W_STAGE = 3 * P * 2 * W_FRAG;     // [dx][part][N block]: 18 432 / 12 288
    static constexpr int LDS_W = 2 * IN_BUF;
    static constexpr int LDS_RED = LDS_W + 2 * W_STAGE;    // [4 MFMA waves][64 channels][2] floats
    static constexpr int LDS_NEXT = LDS_RED + 4 * 64 * 2 * 4;
    static constexpr int LDS_COEF = LDS_NEXT + 16;         // [tile parity 2][scale | shift][CMAX] floats
    // fp16 form only (the bf16 form fills the LDS): per MFMA wave a [64 channels][32 pixels] fp32 staging area of one M block,
    // rows padded to 144 bytes (conflict-free 16-byte writes of 8 consecutive channels), for the epilogue's transposition
    static constexpr int EPI_ROW = 144, EPI_WAVE = 64 * EPI_ROW;
    static constexpr int LDS_EPI = LDS_COEF + 2 * 2 * CMAX * 4;
    static constexpr int LDS_BYTES = LDS_EPI + (P == 2 ? 4 * EPI_WAVE : 0);
    static constexpr int W_ITERS = (W_STAGE / 16 + STAGERS - 1) / STAGERS;   // 16-byte pieces of a weight stage per thread
    static constexpr int PRODUCTS = P == 3 ? 6 : 3;
};
static_assert(X3Cfg<3>::LDS_BYTES <= 160 * 1024 && X3Cfg<2>::LDS_BYTES <= 160 * 1024, "one workgroup per CU");

typedef short bf16x8 __attribute__((ext_vector_type(8)));   // eight 16-bit operands (bf16 or fp16 bits)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct X3Args {
    Src a;
    const unsigned char* __restrict__ wpk;   // [stage = kstep * 3 + dy][W_STAGE bytes]
    const float* __restrict__ bias;
    float* __restrict__ out;
    double* __restrict__ partials;
    int* __restrict__ queue;                  // 8 counters + a count of finished workgroups; zero at launch: written by the
                                              // weight packing, then re-zeroed by the last workgroup of every launch
    int N, Cin, D, H, W, Cout;
    int CoutStride;                           // channels per batch entry of the output tensor (>= Cout)
    int lrelu;
    int tiles_x, tiles_y, tiles;              // per plane
    int tiles_x_full;                         // tile columns whose right half is inside the image
    int planes;                               // N * D
    int nks;                                  // Cin / 16
    int epi_lds;                              // fp16 form: stores of interior tiles go through the LDS transposition
    // Channel-blocked activations (round 5, fp16 form only): [N][D][C / 8][H][W][8] -- the eight channels of a group are
    // the 32 contiguous bytes of a pixel, i.e. exactly one 16-byte LDS slot per split part, and a halo row of a group is
    // ONE run of 34 x 32 bytes instead of eight 136-byte segments in eight channel planes.  Private to the fused
    // Matching chain (l1_combine -> conv2d_x3 x 3 -> materialize_l0 -> conv2d_t8w), see matching_pipeline (api.hip).
    int in_cb8, out_cb8;
    // Input formed on the fly from the layer-1 planes of the fused Matching path (misc.hip: l1_blocked_kernel; replaces the
    // 425 MB round trip through l1_combine's t1): B [n][C / 8]{[H][W + 2][8]} and H [n][C / 8]{[H][l1_P + W + 2][8] zero-padded on
    // the left | edge columns [D][H][2][8]}; group strides and the edge block's offset in bytes; l1_d0: disparity of plane 0
    const float* __restrict__ l1B;
    const float* __restrict__ l1H;
    unsigned l1_bstride, l1_hstride, l1_edge;
    int l1_P, l1_d0;
    // fp16 form: the power-of-two operand scales (header comment).  ascale is computed by every workgroup from the
    // source's range certificate, 1 / ws was left behind the tile-queue counters by the weight packing.
    const float* __restrict__ bound;
    int bound_n;
    float ascale, unscale;                    // device side only: filled in by the kernel before the roles split
};

struct Tile {
    int n, d, tile, y0, x0;
};

// the sixteen channels of a K-step that a staging thread requests for its pixel: sixteen dword loads from sixteen channel
// planes, or (channel-blocked input) four 16-byte loads from the two groups of eight
template <int CBI>
struct X3In {
    float v[16];
    __device__ __forceinline__ float get(int c) const { return v[c]; }
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = 0.f;
    }
};
template <>
struct X3In<1> {
    f32x4 q[4];
    __device__ __forceinline__ float get(int c) const { return q[c >> 2][c & 3]; }
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
};
// layer-1 planes (round 5, X3Args::l1B): t1 = LeakyReLU(B[x] + H[x - d]) is formed while it is staged -- the B and the H
// values of the pixel's sixteen channels, channel-blocked, eight 16-byte loads
template <>
struct X3In<2> {
    f32x4 q[4], h[4];
    __device__ __forceinline__ float get(int c) const {   // exactly l1_combine_kernel's arithmetic (misc.hip)
        float v, m;
        asm("v_add_f32 %0, %1, %2" : "=v"(v) : "v"(q[c >> 2][c & 3]), "v"(h[c >> 2][c & 3]));
        asm("v_mul_f32 %0, %1, %2" : "=v"(m) : "v"(v), "v"(kLeakySlope));
        asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(m));   // == v > 0 ? v : slope * v for 0 < slope < 1
        return v;
    }
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = h[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
};

// One lane draws the next tile: its home queue first, then the others (work stealing).  Virtual tile id =
// plane * tiles + tile, or -1 when every queue is empty.  Queue q holds the planes p = q (mod 8): all their full tiles
// (plane-major, row-major), then the tiles whose right half lies outside the image.
__device__ __forceinline__ int draw_tile(int* queue, int home, int planes, int tiles, int tiles_x, int tiles_x_full,
                                         int n_full) {
    const int n_half = tiles - n_full;
#pragma unroll 1
    for (int attempt = 0; attempt < 8; ++attempt) {
        const int q = (home + attempt) & 7;
        const int planes_q = (planes - q + 7) >> 3;
        if (planes_q <= 0) continue;
        const int j = atomicAdd(queue + q, 1);
        if (j >= planes_q * tiles) continue;
        int pl, tile;
        // the half-cost tiles (right half outside the image) come FIRST: workgroups that start on one run half a tile
        // out of phase with the others for the rest of the launch, so the chip-wide store bursts of the epilogues
        // (every CU finishes its tile at the same time otherwise) come in two halves
        if (j >= planes_q * n_half) {
            const int jj = j - planes_q * n_half;
            pl = jj / n_full;
            const int r = jj - pl * n_full;
            const int ty = r / tiles_x_full;
            tile = ty * tiles_x + (r - ty * tiles_x_full);
        } else {
            pl = j / n_half;
            const int rr = j - pl * n_half;
            const int cols_half = tiles_x - tiles_x_full;
            const int ty = rr / cols_half;
            tile = ty * tiles_x + tiles_x_full + (rr - ty * cols_half);
        }
        return (q + 8 * pl) * tiles + tile;
    }
    return -1;
}

// The MFMAs of one stage (K-step, kernel row): three taps x NMB M blocks = steps of 12 MFMAs (two N blocks x six partial
// products).  Fragments are double-buffered at two rates: the six weight fragments of a tap are requested, a pair per
// step, during the previous tap; the three pixel fragments of an M block during the previous step.  A scheduling
// barrier closes every step, so the requests of step k + 1 are in flight while the MFMAs of step k issue and the
// compiler cannot pile up more fragments than the two sets (128 accumulator + 72 fragment registers).
// xb: the lane's pixel slot of M block 0 at (dy, dx = 0) in part 0; wb: the lane's slot in the stage's weights.
// NARROW: M blocks are 2 rows x 16 columns and there are two of them.
// scheduling pattern of a step: N x (one MFMA, one fragment request), then the remaining MFMAs
template <int N>
__device__ __forceinline__ void x3_interleave() {
    if constexpr (N > 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        x3_interleave<N - 1>();
    }
}

template <int P>
__device__ __forceinline__ f32x16 x3_mma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr (P == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int P, bool NARROW, int T, int I>
__device__ __forceinline__ void x3_step(f32x16 (&acc)[4][2], bf16x8 (&fw)[2][P][2], bf16x8 (&fp)[2][P],
                                        const unsigned char* xb, const unsigned char* wb) {
    constexpr int NMB = NARROW ? 2 : 4;
    constexpr int K = T * NMB + I;
    constexpr int ROWSTEP = NARROW ? 2 : 1;
    // requests: pixel fragments of the next step ...
    if constexpr (I + 1 < NMB || T < 2) {
        constexpr int tn = I + 1 < NMB ? T : T + 1, in = I + 1 < NMB ? I + 1 : 0;
#pragma unroll
        for (int p = 0; p < P; ++p)
            fp[(K + 1) & 1][p] =
                *reinterpret_cast<const bf16x8*>(xb + p * IN_PART + (in * ROWSTEP * COLS + tn) * 16);
    }
    // ... and this step's share of the next tap's weight fragments (the parts go out over the first steps of the tap)
    constexpr int first = (NARROW && P == 3) ? (I == 0 ? 0 : 2) : I;
    constexpr int last = (NARROW && P == 3) ? (I == 0 ? 1 : 2) : (I < P ? I : -1);
    if constexpr (T < 2) {
#pragma unroll
        for (int p = first; p <= last; ++p)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                fw[(T + 1) & 1][p][nb] =
                    *reinterpret_cast<const bf16x8*>(wb + (((T + 1) * P + p) * 2 + nb) * W_FRAG);
    }
    // small partial products first
    constexpr int NPROD = X3Cfg<P>::PRODUCTS;
#pragma unroll
    for (int c = 0; c < NPROD; ++c) {
        // (pixel part, weight part): P = 3: (0,2) (2,0) (1,1) (0,1) (1,0) (0,0);  P = 2: (0,1) (1,0) (0,0)
        const int pa = P == 3 ? (c == 0 ? 0 : c == 1 ? 2 : c == 2 ? 1 : c == 3 ? 0 : c == 4 ? 1 : 0) : (c == 1 ? 1 : 0);
        const int pb = P == 3 ? (c == 0 ? 2 : c == 1 ? 0 : c == 2 ? 1 : c == 3 ? 1 : c == 4 ? 0 : 0) : (c == 0 ? 1 : 0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[I][nb] = x3_mma<P>(fp[K & 1][pa], fw[T & 1][pb][nb], acc[I][nb]);
    }
    // the requests go out in the shadow of the first MFMAs (each on registers of the idle set), not after the last use
    // of the registers they would otherwise recycle
    constexpr int NP = (I + 1 < NMB || T < 2) ? P : 0;
    constexpr int NW = (T < 2 && last >= first) ? 2 * (last - first + 1) : 0;
    x3_interleave<NP + NW>();
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NPROD - NP - NW, 0);
    __builtin_amdgcn_sched_barrier(0);
}

template <int P, bool NARROW>
__device__ __forceinline__ void x3_mfma_stage(f32x16 (&acc)[4][2], const unsigned char* xb, const unsigned char* wb) {
    bf16x8 fw[2][P][2], fp[2][P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        fp[0][p] = *reinterpret_cast<const bf16x8*>(xb + p * IN_PART);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) fw[0][p][nb] = *reinterpret_cast<const bf16x8*>(wb + (p * 2 + nb) * W_FRAG);
    }
    __builtin_amdgcn_sched_barrier(0);
#define PDS_X3_TAP(T)                                            \
    x3_step<P, NARROW, T, 0>(acc, fw, fp, xb, wb);               \
    x3_step<P, NARROW, T, 1>(acc, fw, fp, xb, wb);               \
    if constexpr (!NARROW) {                                     \
        x3_step<P, NARROW, T, 2>(acc, fw, fp, xb, wb);           \
        x3_step<P, NARROW, T, 3>(acc, fw, fp, xb, wb);           \
    }
    PDS_X3_TAP(0)
    PDS_X3_TAP(1)
    PDS_X3_TAP(2)
#undef PDS_X3_TAP
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global access of
// the wave (vmcnt(0)): for the staging waves that would drain the requests they have just issued, for the MFMA waves
// the epilogue's stores -- both are meant to stay in flight across the barrier.
__device__ __forceinline__ void x3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// a * b + c as ONE v_fma_f32: left to the compiler, pairs of these become v_pk_fma_f32, which issues to the matrix pipe's
// side of the SIMD and starves behind the MFMA wave that shares it (measured: +190 us per 48-plane launch)
__device__ __forceinline__ float x3_fma(float a, float b, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float x3_mul(float a, float b) {   // (same reason)
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ Tile decode_tile(const X3Args& A, int v) {
    Tile t;
    const int p = v / A.tiles;
    t.tile = v - p * A.tiles;
    t.n = p / A.D;
    t.d = p - t.n * A.D;
    const int ty = t.tile / A.tiles_x;
    t.y0 = ty * TH;
    t.x0 = (t.tile - ty * A.tiles_x) * TW;
    return t;
}
__device__ __forceinline__ Tile pick_tile(bool second, const Tile& b, const Tile& a) {   // uniform selects, no branch
    Tile t;
    t.n = second ? b.n : a.n;
    t.d = second ? b.d : a.d;
    t.tile = second ? b.tile : a.tile;
    t.y0 = second ? b.y0 : a.y0;
    t.x0 = second ? b.x0 : a.x0;
    return t;
}

// Both roles walk the same sequence of tiles and stages and meet at the same barriers (three in the prologue, one per
// stage, one before the last statistics record); each is a loop nest of its own so that the accumulators of the MFMA
// waves and the staging registers of the others never share a live range.

// ---- waves 0-3 -----------------------------------------------------------------------------------------------------
struct MfmaLane {
#ifdef PDS_X3_TIMING   // debugging: cycles in the MFMA stream, at the barriers, in the epilogue; stage count
    long long tm[4] = {0, 0, 0, 0};
#endif
    int wave, m32, kgl;
    int x_lane, w_lane;      // byte offsets of the lane's pixel slot (M block 0, dy = dx = 0, part 0) and weight slot
    float bias0, bias1;
    unsigned cstride;
    size_t plane;
};

// One tile on an MFMA wave: all stages, then the epilogue.  The accumulators live and die inside this function, per
// variant, so they never cross a control-flow merge (a phi of 128 registers costs copies and their live ranges).
// Returns the id of the next tile (read at stage 1).
template <int P, bool NORM, bool NARROW, bool CBO>
__device__ __forceinline__ int x3_mfma_tile(const X3Args& A, unsigned char* lds, const MfmaLane& L, const Tile& cur,
                                            int& upar, int& wpar) {
    using C = X3Cfg<P>;
    const int nstages = 3 * A.nks;
    const int* next_slot = reinterpret_cast<const int*>(lds + C::LDS_NEXT);
    int nxt_id = -1;
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
#pragma unroll 1
    for (int rs = 0; rs < nstages; ++rs) {
        const int dy = rs % 3;
        if (rs == 1) nxt_id = __builtin_amdgcn_readfirstlane(next_slot[0]);
        const unsigned char* wb = lds + C::LDS_W + wpar * C::W_STAGE + L.w_lane;
        const unsigned char* xb = lds + upar * C::IN_BUF + dy * (COLS * 16) + L.x_lane;
#ifdef PDS_X3_TIMING
        const long long t_a = __builtin_readcyclecounter();
#endif
#ifndef PDS_X3_NOMFMA   // (PDS_X3_NO*: timing ablations, never defined in the product build)
        x3_mfma_stage<P, NARROW>(acc, xb, wb);
#endif
        if (dy == 2) upar ^= 1;
        wpar ^= 1;
#ifdef PDS_X3_TIMING
        const long long t_b = __builtin_readcyclecounter();
#endif
        x3_barrier();
#ifdef PDS_X3_TIMING
        const long long t_c = __builtin_readcyclecounter();
        const_cast<MfmaLane&>(L).tm[0] += t_b - t_a;
        const_cast<MfmaLane&>(L).tm[1] += t_c - t_b;
        const_cast<MfmaLane&>(L).tm[3] += 1;
#endif
    }
#ifdef PDS_X3_TIMING
    const long long t_e0 = __builtin_readcyclecounter();
#endif
    // ---- epilogue of the tile: bias, LeakyReLU, 16-byte stores, statistics
#ifdef PDS_X3_NOEPI
    if (acc[0][0][0] != 12345.f) return nxt_id;
#endif
    // The epilogue takes ~11 000 cycles per tile whatever its instruction count (measured with PDS_X3_TIMING: the scalar
    // form with ~2 500 instructions and this packed form with ~900 take the same time): all CUs run the same schedule
    // and store their 128 KB tiles together, so the wave waits for the memory pipeline, not for the VALU.  On the
    // six-product form nothing recovered that time (half-cost tiles first, a start-up stagger of 4-48 us over the CUs
    // of an XCD: the launch is power-limited, any gap is returned as clock).  Packed fp32 arithmetic (v_pk_fma /
    // v_pk_mul / v_pk_add: two values per instruction and lane) on register pairs that stay where the MFMA left them:
    // 3.5 instructions per value, no moves to assemble the 16-byte store operands.  Tiles that touch the image border
    // (or rows that are not 16-byte aligned) take the masked form.
    f32x2 s2[2][2], q2[2][2];   // [N block][register pair of the quad]: four independent chains per statistic
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int k = 0; k < 2; ++k) s2[nb][k] = q2[nb][k] = f32x2{0.f, 0.f};
    float* obase = A.out + (((size_t)cur.n * A.CoutStride) * A.D + cur.d) * L.plane;   // uniform
    const float slope = A.lrelu ? kLeakySlope : 1.f;
    const f32x2 slope2 = {slope, slope};
    // the fp16 form accumulates (ws w) * (as x): one exact power of two back, in the same fma as the bias
    const f32x2 unscale2 = {A.unscale, A.unscale};
    constexpr int MBLOCKS = NARROW ? 2 : 4;
    const bool interior = (A.W & 3) == 0 && cur.y0 + TH <= A.H && cur.x0 + (NARROW ? 16 : TW) <= A.W;   // uniform
#define PDS_X3_EPILOGUE(MASKED)                                                                                       \
    _Pragma("unroll") for (int mb = 0; mb < MBLOCKS; ++mb) {                                                          \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                               \
            const int m0 = 8 * g + 4 * L.kgl;   /* first of the lane's four consecutive pixels of the M block */      \
            const int y = cur.y0 + 4 * L.wave + (NARROW ? 2 * mb + (m0 >> 4) : mb);                                   \
            const int x = cur.x0 + (NARROW ? (m0 & 15) : m0);                                                         \
            _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                                        \
                const float bv = nb ? L.bias1 : L.bias0;                                                              \
                const f32x2 bv2 = {bv, bv};                                                                           \
                f32x2 ta = __builtin_elementwise_fma(f32x2{acc[mb][nb][4 * g], acc[mb][nb][4 * g + 1]}, unscale2, bv2); \
                f32x2 tb = __builtin_elementwise_fma(f32x2{acc[mb][nb][4 * g + 2], acc[mb][nb][4 * g + 3]}, unscale2, \
                                                     bv2);                                                            \
                ta = __builtin_elementwise_max(ta, ta * slope2);                                                      \
                tb = __builtin_elementwise_max(tb, tb * slope2);                                                      \
                float* po = obase + (size_t)(nb * 32 + L.m32) * L.cstride + (size_t)y * A.W + x;                      \
                if (!(MASKED)) {                                                                                      \
                    *reinterpret_cast<f32x4*>(po) = f32x4{ta[0], ta[1], tb[0], tb[1]};                                \
                } else {                                                                                              \
                    const bool row = y < A.H;                                                                         \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
                        const bool in = row && x + e < A.W;                                                           \
                        const float v = e < 2 ? ta[e] : tb[e - 2];                                                    \
                        if (in) po[e] = v;                                                                            \
                        if (e < 2) ta[e] = in ? v : 0.f;                                                              \
                        else tb[e - 2] = in ? v : 0.f;                                                                \
                    }                                                                                                 \
                }                                                                                                     \
                s2[nb][0] += ta;                                                                                      \
                s2[nb][1] += tb;                                                                                      \
                q2[nb][0] = __builtin_elementwise_fma(ta, ta, q2[nb][0]);                                             \
                q2[nb][1] = __builtin_elementwise_fma(tb, tb, q2[nb][1]);                                             \
            }                                                                                                         \
        }                                                                                                             \
    }
    bool done = false;
    if constexpr (P == 2) {
        if (interior && A.epi_lds) {
            // Transposed stores.  In the D fragment a lane holds ONE channel and four consecutive pixels, so a direct
            // 16-byte store scatters 32-byte pieces over 32 channel planes per instruction: the per-CU store path
            // sustains ~12 bytes per cycle for that (9 000 cycles per 128 KB tile, tools/ubench/store_patterns.hip)
            // and the MFMA wave sits in its epilogue for 11 000.  Through a wave-private LDS area every store
            // instruction writes 8 channels x 128 contiguous bytes instead (2 500 cycles per tile in the probe):
            // lane -> (quad q = lane & 7 of the 32 pixels, channel 8 j + (lane >> 3)).
            unsigned char* epi = lds + C::LDS_EPI + L.wave * C::EPI_WAVE;
            unsigned char* wr = epi + L.m32 * C::EPI_ROW + L.kgl * 16;       // + nb * 32 rows, + g * 32 bytes
            const int q = (L.m32 & 7), cg = ((L.kgl << 5) | L.m32) >> 3;      // lane = kgl * 32 + m32
            const unsigned char* rd = epi + cg * C::EPI_ROW + q * 16;        // + j * 8 rows
#pragma unroll
            for (int mb = 0; mb < MBLOCKS; ++mb) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const float bv = nb ? L.bias1 : L.bias0;
                        const f32x2 bv2 = {bv, bv};
                        f32x2 ta = __builtin_elementwise_fma(f32x2{acc[mb][nb][4 * g], acc[mb][nb][4 * g + 1]}, unscale2, bv2);
                        f32x2 tb = __builtin_elementwise_fma(f32x2{acc[mb][nb][4 * g + 2], acc[mb][nb][4 * g + 3]}, unscale2, bv2);
                        ta = __builtin_elementwise_max(ta, ta * slope2);
                        tb = __builtin_elementwise_max(tb, tb * slope2);
#ifdef PDS_X3_EPI_DUMPONLY   // timing ablation (wrong results): the MFMA wave only parks its raw accumulators in LDS
                        *reinterpret_cast<f32x4*>(wr + nb * 32 * C::EPI_ROW + g * 32) =
                            f32x4{acc[mb][nb][4 * g], acc[mb][nb][4 * g + 1], acc[mb][nb][4 * g + 2], acc[mb][nb][4 * g + 3]};
                        continue;
#endif
                        *reinterpret_cast<f32x4*>(wr + nb * 32 * C::EPI_ROW + g * 32) = f32x4{ta[0], ta[1], tb[0], tb[1]};
                        s2[nb][0] += ta;
                        s2[nb][1] += tb;
                        q2[nb][0] = __builtin_elementwise_fma(ta, ta, q2[nb][0]);
                        q2[nb][1] = __builtin_elementwise_fma(tb, tb, q2[nb][1]);
                    }
#if defined(PDS_X3_EPI_DUMPONLY) || defined(PDS_X3_EPI_NOSTORE)   // timing ablations (wrong results)
                if (acc[0][0][0] != 12345.f) continue;
#endif
                if constexpr (CBO) {
                    // channel-blocked output: lane -> (pixel p of the M block, channel quad hq); store j writes channels
                    // 8 j + 4 hq .. + 3 of that pixel, i.e. one instruction covers the 32 pixels x 32 bytes of group j:
                    // 1 024 contiguous bytes (wide block) or two 512-byte runs (narrow block: two rows of 16)
                    const int lane = (L.kgl << 5) | L.m32;
                    const int p = lane >> 1, hq = lane & 1;
                    const int y = cur.y0 + 4 * L.wave + (NARROW ? 2 * mb + (p >> 4) : mb);
                    const int x = cur.x0 + (NARROW ? (p & 15) : p);
                    float* po = A.out + (((size_t)(cur.n * A.D + cur.d) * 8) * L.plane + (size_t)y * A.W + x) * 8 + 4 * hq;
                    const unsigned char* rdc = epi + (4 * hq) * C::EPI_ROW + p * 4;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = *reinterpret_cast<const float*>(rdc + (8 * j + e) * C::EPI_ROW);
                        *reinterpret_cast<f32x4*>(po + (size_t)j * L.plane * 8) = v;
                    }
                } else {
                // pixel quad q of the M block: wide = one row of 32 columns; narrow = two rows of 16
                const int y = cur.y0 + 4 * L.wave + (NARROW ? 2 * mb + (q >> 2) : mb);
                const int x = cur.x0 + (NARROW ? 4 * (q & 3) : 4 * q);
                float* po = obase + (size_t)cg * L.cstride + (size_t)y * A.W + x;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *reinterpret_cast<f32x4*>(po + (size_t)(8 * j) * L.cstride) =
                        *reinterpret_cast<const f32x4*>(rd + j * 8 * C::EPI_ROW);
                }
            }
            done = true;
        }
    }
    if (!done) {
        if (interior) {
            PDS_X3_EPILOGUE(false)
        } else {
            PDS_X3_EPILOGUE(true)
        }
    }
#undef PDS_X3_EPILOGUE
    float s0 = (s2[0][0][0] + s2[0][0][1]) + (s2[0][1][0] + s2[0][1][1]);
    float q0 = (q2[0][0][0] + q2[0][0][1]) + (q2[0][1][0] + q2[0][1][1]);
    float s1 = (s2[1][0][0] + s2[1][0][1]) + (s2[1][1][0] + s2[1][1][1]);
    float q1 = (q2[1][0][0] + q2[1][0][1]) + (q2[1][1][0] + q2[1][1][1]);
    if (A.partials) {
        // the two halves of the wave hold the same channels (pixels 4 apart): one exchange, then lanes 0-31 write
        s0 += __shfl_xor(s0, 32, 64);
        q0 += __shfl_xor(q0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        q1 += __shfl_xor(q1, 32, 64);
        if (L.kgl == 0) {
            float* red = reinterpret_cast<float*>(lds + C::LDS_RED);
            *reinterpret_cast<float2*>(red + (L.wave * 64 + L.m32) * 2) = make_float2(s0, q0);
            *reinterpret_cast<float2*>(red + (L.wave * 64 + 32 + L.m32) * 2) = make_float2(s1, q1);
        }
    }
#ifdef PDS_X3_TIMING
    const_cast<MfmaLane&>(L).tm[2] += __builtin_readcyclecounter() - t_e0;
#endif
    return nxt_id;
}

template <int P, bool NORM, bool CBO>
__device__ __forceinline__ void x3_mfma_waves(const X3Args& A, unsigned char* lds, int wave, int lane, int cur_id) {
    MfmaLane L;
    L.wave = wave;
    L.m32 = lane & 31;
    L.kgl = lane >> 5;
    L.w_lane = lane * 16;
    L.plane = (size_t)A.H * A.W;
    L.cstride = (unsigned)(A.D * L.plane);
    L.bias0 = A.bias ? A.bias[L.m32] : 0.f;
    L.bias1 = A.bias ? A.bias[32 + L.m32] : 0.f;
    const int x_full = (L.kgl * PIX + (4 * wave) * COLS + L.m32) * 16;                            // M block = row, 32 columns
    const int x_narrow = (L.kgl * PIX + (4 * wave + (L.m32 >> 4)) * COLS + (L.m32 & 15)) * 16;    // M block = 2 rows x 16
    int upar = 0, wpar = 0;   // LDS buffer of the K-step / weight stage being consumed
    x3_barrier();
    x3_barrier();
#ifndef PDS_X3_NOPRIO
    __builtin_amdgcn_s_setprio(3);   // the matrix pipe's feeders issue ahead of the staging wave of their SIMD
#endif
    for (;;) {
        const Tile cur = decode_tile(A, cur_id);
        if (cur.x0 + 16 >= A.W) {   // the right half of the tile is outside the image (uniform)
            L.x_lane = x_narrow;
            cur_id = x3_mfma_tile<P, NORM, true, CBO>(A, lds, L, cur, upar, wpar);
        } else {
            L.x_lane = x_full;
            cur_id = x3_mfma_tile<P, NORM, false, CBO>(A, lds, L, cur, upar, wpar);
        }
        if (cur_id < 0) break;
    }
#ifdef PDS_X3_TIMING
    if ((blockIdx.x == 0 || blockIdx.x == 131) && lane == 0)
        printf("[x3] wg %d wave %d: mfma %lld  barrier %lld  epilogue %lld cycles over %lld stages\n", (int)blockIdx.x, wave,
               L.tm[0], L.tm[1], L.tm[2], L.tm[3]);
#endif
    x3_barrier();
}

// ---- waves 4-7 -----------------------------------------------------------------------------------------------------
// All four waves work at every stage, on a two-deep ring of request registers: at stage s a thread writes what it
// requested at stage s - 2 (set s & 1) and re-uses that set for the requests of stage s + 2.  A request therefore has
// two whole stages to land (global latency under load is ~2 us, about one stage; with a one-stage lag every stage
// began by waiting for the loads issued at the end of the previous one).  The stage loop is unrolled by two so that
// the set index is a compile-time constant; a stage body is one function that also does the end-of-tile bookkeeping.
template <int P, bool NORM, int CBI>
__device__ __forceinline__ void x3_staging_waves(const X3Args& A, unsigned char* lds, int st, int cur_id) {
    using C = X3Cfg<P>;
    constexpr int W_ITERS = C::W_ITERS, W_STAGE = C::W_STAGE, IN_BUF = C::IN_BUF;
    constexpr int LDS_W = C::LDS_W, LDS_RED = C::LDS_RED, LDS_NEXT = C::LDS_NEXT, LDS_COEF = C::LDS_COEF;
    const size_t plane = (size_t)A.H * A.W;
    const unsigned cstride = (unsigned)(A.D * plane);          // floats between channels
    const int nks = A.nks, nstages = 3 * nks;
    // a thread stages one pixel (of the 204 of a third) x 16 channels (the two groups of 8)
    const bool valid = st < THIRD_PIX;
    const int prow = valid ? st / COLS : 0, pcol = valid ? st % COLS : 0;
    const int lds_item = (prow * COLS + pcol) * 16;   // + (part * 2 + channel group) * PIX * 16 + third * THIRD_PIX * 16
    int* next_slot = reinterpret_cast<int*>(lds + LDS_NEXT);
    const int home = blockIdx.x & 7;
    const int n_full = A.tiles_y * A.tiles_x_full;

    X3In<CBI> xin[2];         // [set]: the channels of the K-step
    u32x4 win[2][W_ITERS];    // [set][this thread's pieces of a weight stage]
    float coef_s[2] = {1.f, 1.f}, coef_h[2] = {0.f, 0.f};   // [set]: the thread's channel of the NEXT tile's table
    const int wlast = W_STAGE / 16 - 1;
    const int coef_c = min(st, A.Cin - 1);
    float* coef_tab = reinterpret_cast<float*>(lds + LDS_COEF);

    // The staged loads are issued from inline assembly and waited for with an explicit count.  Left to the compiler,
    // the waits of a stage count down to vmcnt(0): it cannot order the two request sets across the loop's merge points
    // and waits for the YOUNGER set as well -- the loads issued one stage ago -- so a request had one stage to land, not
    // two, and the MFMA waves waited 1 186 cycles per stage at the barrier (PDS_X3_TIMING).  With the explicit count:
    // 670 cycles, and +2.0 % pairs/s in a same-box A/B of the whole benchmark (384.6 / 387.3 / 386.1 against 378.4 /
    // 378.6 / 378.8; the isolated launch is power-limited and barely moves: the cycles saved come back as clock).
    // kSetLoads = loads every stage issues: 2 coefficients (NORM), 16 inputs, W_ITERS weight pieces; a wait for "at most
    // kSetLoads outstanding" therefore covers the whole older set (vector-memory results return in order; other
    // memory operations in between only make the wait stricter).
    constexpr int kInLoads = CBI == 2 ? 8 : (CBI == 1 ? 4 : 16);
    constexpr int kSetLoads = kInLoads + W_ITERS + (NORM ? 2 : 0);
    auto request_inputs = [&](X3In<CBI>& x, const Tile& tl, int ks, int third) {
        const int y = tl.y0 - 1 + third * THIRD_ROWS + prow, xx = tl.x0 - 1 + pcol;
        const int yc = min(max(y, 0), A.H - 1), xc = min(max(xx, 0), A.W - 1);
        if constexpr (CBI == 2) {
            // B at column x + 2 of its row; H[x - d] at column x - d + 2 + l1_P of the zero-padded row (0 for x - d < -2), or,
            // where l1_combine adds a column correction (x = 0 at d = 0; x = w - 2, w - 1 at d >= 1), the edge entry that
            // holds H + correction: a per-lane choice of the offset, the same eight loads for every lane
            const int dd = A.l1_d0 + tl.d;
            const bool edge = dd == 0 ? xc == 0 : xc >= A.W - 2;
            const int slot = dd == 0 ? 0 : xc - (A.W - 2);
            const unsigned boffB = (unsigned)(yc * (A.W + 2) + xc + 2) * 32u;
            const unsigned boffH = edge ? A.l1_edge + (unsigned)((tl.d * A.H + yc) * 2 + slot) * 32u
                                        : (unsigned)(yc * (A.l1_P + A.W + 2) + xc - dd + 2 + A.l1_P) * 32u;
            const unsigned long long bb = reinterpret_cast<unsigned long long>(A.l1B) +
                                          (unsigned long long)(tl.n * (A.Cin >> 3) + 2 * ks) * A.l1_bstride;   // uniform
            const unsigned long long hb = reinterpret_cast<unsigned long long>(A.l1H) +
                                          (unsigned long long)(tl.n * (A.Cin >> 3) + 2 * ks) * A.l1_hstride;
            const unsigned b_lo = (unsigned)bb, b_hi = (unsigned)(bb >> 32), h_lo = (unsigned)hb, h_hi = (unsigned)(hb >> 32);
            asm volatile("s_mov_b32 s60, %10\n\ts_mov_b32 s61, %11\n\t"
                         "global_load_dwordx4 %0, %8, s[60:61]\n\tglobal_load_dwordx4 %1, %8, s[60:61] offset:16\n\t"
                         "s_add_u32 s60, s60, %14\n\ts_addc_u32 s61, s61, 0\n\t"
                         "global_load_dwordx4 %2, %8, s[60:61]\n\tglobal_load_dwordx4 %3, %8, s[60:61] offset:16\n\t"
                         "s_mov_b32 s60, %12\n\ts_mov_b32 s61, %13\n\t"
                         "global_load_dwordx4 %4, %9, s[60:61]\n\tglobal_load_dwordx4 %5, %9, s[60:61] offset:16\n\t"
                         "s_add_u32 s60, s60, %15\n\ts_addc_u32 s61, s61, 0\n\t"
                         "global_load_dwordx4 %6, %9, s[60:61]\n\tglobal_load_dwordx4 %7, %9, s[60:61] offset:16"
                         : "=&v"(x.q[0]), "=&v"(x.q[1]), "=&v"(x.q[2]), "=&v"(x.q[3]), "=&v"(x.h[0]), "=&v"(x.h[1]),
                           "=&v"(x.h[2]), "=&v"(x.h[3])
                         : "v"(boffB), "v"(boffH), "s"(b_lo), "s"(b_hi), "s"(h_lo), "s"(h_hi), "s"(A.l1_bstride),
                           "s"(A.l1_hstride)
                         : "memory", "s60", "s61", "scc");
        } else if constexpr (CBI == 1) {
            // groups 2 ks and 2 ks + 1 of plane (n, d) of [N][D][C / 8][H][W][8]: the pixel's 32 bytes in each, two 16-byte
            // loads per group -- the base walks in s[60:61] as below
            const float* src = A.a.p + (((size_t)(tl.n * A.D + tl.d) * (A.Cin >> 3) + 2 * ks) * plane) * 8;   // uniform
            const unsigned boff = (unsigned)(yc * A.W + xc) * 32u;
            const unsigned long long base = reinterpret_cast<unsigned long long>(src);
            const unsigned base_lo = (unsigned)base, base_hi = (unsigned)(base >> 32), step = (unsigned)(plane * 32);
            asm volatile("s_mov_b32 s60, %5\n\ts_mov_b32 s61, %6\n\t"
                         "global_load_dwordx4 %0, %4, s[60:61]\n\tglobal_load_dwordx4 %1, %4, s[60:61] offset:16\n\t"
                         "s_add_u32 s60, s60, %7\n\ts_addc_u32 s61, s61, 0\n\t"
                         "global_load_dwordx4 %2, %4, s[60:61]\n\tglobal_load_dwordx4 %3, %4, s[60:61] offset:16"
                         : "=&v"(x.q[0]), "=&v"(x.q[1]), "=&v"(x.q[2]), "=&v"(x.q[3])
                         : "v"(boff), "s"(base_lo), "s"(base_hi), "s"(step)
                         : "memory", "s60", "s61", "scc");
        } else {
        const float* src = A.a.p + ((size_t)(tl.n * A.Cin + ks * 16) * A.D + tl.d) * plane;   // uniform
        const unsigned boff = (unsigned)(yc * A.W + xc) * 4u;   // (a channel plane is far below 4 GB)
        // One statement for the sixteen loads: the channel base walks in s[60:61] (scalar adds) and the lane offset is
        // one VGPR.  The base arrives through s_mov: an SGPR that the compiler has just re-loaded from a spill lane
        // (v_readlane) must not be read by a VMEM instruction within five wait states -- a hazard its recogniser does
        // not see through inline assembly (it produced wild base addresses, i.e. memory faults, in some builds).
        const unsigned long long base = reinterpret_cast<unsigned long long>(src);
        const unsigned base_lo = (unsigned)base, base_hi = (unsigned)(base >> 32), step = cstride * 4u;
#define PDS_X3_LD(I) "global_load_dword %" #I ", %16, s[60:61]\n\ts_add_u32 s60, s60, %19\n\ts_addc_u32 s61, s61, 0\n\t"
        asm volatile("s_mov_b32 s60, %17\n\ts_mov_b32 s61, %18\n\t" PDS_X3_LD(0) PDS_X3_LD(1) PDS_X3_LD(2) PDS_X3_LD(3)
                         PDS_X3_LD(4) PDS_X3_LD(5) PDS_X3_LD(6) PDS_X3_LD(7) PDS_X3_LD(8) PDS_X3_LD(9) PDS_X3_LD(10)
                             PDS_X3_LD(11) PDS_X3_LD(12) PDS_X3_LD(13) PDS_X3_LD(14) "global_load_dword %15, %16, s[60:61]"
                     : "=&v"(x.v[0]), "=&v"(x.v[1]), "=&v"(x.v[2]), "=&v"(x.v[3]), "=&v"(x.v[4]), "=&v"(x.v[5]), "=&v"(x.v[6]),
                       "=&v"(x.v[7]), "=&v"(x.v[8]), "=&v"(x.v[9]), "=&v"(x.v[10]), "=&v"(x.v[11]), "=&v"(x.v[12]), "=&v"(x.v[13]),
                       "=&v"(x.v[14]), "=&v"(x.v[15])
                     : "v"(boff), "s"(base_lo), "s"(base_hi), "s"(step)
                     : "memory", "s60", "s61", "scc");
#undef PDS_X3_LD
        }
    };
    // every value of a set passes through this statement before its first use: the wait cannot be scheduled after a
    // consumer, and no consumer before it
    auto await_set = [&](X3In<CBI>& x, u32x4 (&w)[W_ITERS], float& cs, float& ch) {
        if constexpr (CBI == 2) {
            asm volatile("s_waitcnt vmcnt(%8)"
                         : "+v"(x.q[0]), "+v"(x.q[1]), "+v"(x.q[2]), "+v"(x.q[3]), "+v"(x.h[0]), "+v"(x.h[1]), "+v"(x.h[2]), "+v"(x.h[3])
                         : "n"(kSetLoads)
                         : "memory");
        } else if constexpr (CBI == 1) {
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(x.q[0]), "+v"(x.q[1]), "+v"(x.q[2]), "+v"(x.q[3]) : "n"(kSetLoads) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%16)"
                         : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]), "+v"(x.v[5]), "+v"(x.v[6]),
                           "+v"(x.v[7]), "+v"(x.v[8]), "+v"(x.v[9]), "+v"(x.v[10]), "+v"(x.v[11]), "+v"(x.v[12]), "+v"(x.v[13]),
                           "+v"(x.v[14]), "+v"(x.v[15])
                         : "n"(kSetLoads)
                         : "memory");
        }
#pragma unroll
        for (int it = 0; it < W_ITERS; ++it) asm volatile("" : "+v"(w[it]) : : "memory");
        asm volatile("" : "+v"(cs), "+v"(ch) : : "memory");
    };
    f32x4 cs[4], ch[4];   // the sixteen (scale, shift) pairs of the K-step being written: read at its first third only
    auto write_inputs = [&](const X3In<CBI>& xs, const Tile& tl, int ks, int third, unsigned char* buf,
                            const float* coef) {
        float x[16];   // (names for the set's registers: no copies survive)
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = xs.get(c);
        // the producer's folded InstanceNorm of this (batch entry, plane): table of the tile in LDS, one address per
        // wave (broadcast reads), all sixteen channels up front; the three thirds of a K-step share them
        if (NORM && third == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cs[j] = *reinterpret_cast<const f32x4*>(coef + ks * 16 + 4 * j);
                ch[j] = *reinterpret_cast<const f32x4*>(coef + CMAX + ks * 16 + 4 * j);
            }
        }
        const int y = tl.y0 - 1 + third * THIRD_ROWS + prow, xx = tl.x0 - 1 + pcol;
        const bool inimg = y >= 0 && y < A.H && xx >= 0 && xx < A.W;
#if !defined(PDS_X3_NOCONVERT) && !defined(PDS_X3_NOPKCVT)
        if constexpr (P == 2) {
            // fp16 split of a PAIR of channels with the packed conversion of gfx950 (round to nearest even, like the
            // scalar one): hi pair = cvt_pk(r0, r1) lands in slot order (the odd channel above the even one), the halves
            // are widened again (the upper one through SDWA), the remainders (exact in fp32) converted as a pair --
            // 6 instructions per pair instead of 4 conversions + 2 widenings + 2 subtractions + 2 byte permutes.
            u32x4 whi[2], wlo[2];
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {
                const int c = 2 * pr;
                float r0 = NORM ? x3_fma(cs[c >> 2][c & 3], x[c], ch[c >> 2][c & 3]) : x3_mul(x[c], A.ascale);
                float r1 = NORM ? x3_fma(cs[c >> 2][(c & 3) + 1], x[c + 1], ch[c >> 2][(c & 3) + 1]) : x3_mul(x[c + 1], A.ascale);
                r0 = inimg ? r0 : 0.f;
                r1 = inimg ? r1 : 0.f;
                unsigned hi, lo;
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(r0), "v"(r1));
#ifdef PDS_X3_SPLIT_CVT   // (round-4 form: widen the halves again, subtract, convert the pair -- 5 instructions for the low parts)
                float f0, f1;
                asm("v_cvt_f32_f16 %0, %1" : "=v"(f0) : "v"(hi));
                asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f1) : "v"(hi));
                float d0, d1;
                asm("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(r0), "v"(f0));
                asm("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(r1), "v"(f1));
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(d0), "v"(d1));
#else
                // round 5: the low part straight from the mixed-precision fma -- lo = f16(r - hi) with hi read as the fp16
                // half it is (r - hi is exact in fp32, so the one rounding is the conversion: bit-identical), 2 instructions.
                // Same-box A/B against the five-instruction form and against a select-free copy of the loop for tiles whose
                // halo lies inside the image: all three within noise (the launch does not care about staging VALU counts)
                asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(r0));
                asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(r1));
#endif
                whi[pr >> 2][pr & 3] = hi;
                wlo[pr >> 2][pr & 3] = lo;
            }
            if (valid) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    *reinterpret_cast<u32x4*>(buf + (0 * 2 + g) * (PIX * 16) + third * (THIRD_PIX * 16) + lds_item) = whi[g];
                    *reinterpret_cast<u32x4*>(buf + (1 * 2 + g) * (PIX * 16) + third * (THIRD_PIX * 16) + lds_item) = wlo[g];
                }
            }
            return;
        }
#endif
        unsigned h[P][16];   // part p of channel c: P = 3 in the high half, P = 2 in the low half of the dword
        constexpr unsigned pack_sel = P == 3 ? 0x07060302u : 0x05040100u;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
#ifdef PDS_X3_NOCONVERT   // timing ablation (wrong results by design): staging as a pure copy -- no InstanceNorm, no split
            h[0][c] = __builtin_bit_cast(unsigned, x[c]);
            if (P > 1) h[1][c] = __builtin_bit_cast(unsigned, x[c]) >> 16;
            if (P > 2) h[P - 1][c] = 0u;
            continue;
#endif
            float r = NORM ? x3_fma(cs[c >> 2][c & 3], x[c], ch[c >> 2][c & 3]) : (P == 2 ? x3_mul(x[c], A.ascale) : x[c]);
            r = inimg ? r : 0.f;
            if constexpr (P == 3) {
                // truncation split: hi = top 16 bits, remainder exact; three parts carry all 24 significand bits
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const unsigned u = __builtin_bit_cast(unsigned, r);
                    h[q][c] = u;   // (the bf16 is the high half)
                    if (q < 2) r -= __builtin_bit_cast(float, u & 0xffff0000u);
                }
            } else {
                // fp16, round to nearest: hi carries 11 bits, the remainder (exact in fp32) is rounded to 11 more
                const _Float16 hi = (_Float16)r;
                const _Float16 lo = (_Float16)(r - (float)hi);
                h[0][c] = __builtin_bit_cast(unsigned short, hi);
                h[1][c] = __builtin_bit_cast(unsigned short, lo);
            }
        }
        if (valid) {
#pragma unroll
            for (int q = 0; q < P; ++q)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    u32x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j)   // the odd channel above the even one
                        w[j] = __builtin_amdgcn_perm(h[q][8 * g + 2 * j + 1], h[q][8 * g + 2 * j], pack_sel);
                    *reinterpret_cast<u32x4*>(buf + (q * 2 + g) * (PIX * 16) + third * (THIRD_PIX * 16) + lds_item) = w;
                }
        }
    };
    unsigned woff[W_ITERS];   // byte offsets of the thread's pieces of a weight stage (the surplus ones repeat the last)
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it) woff[it] = (unsigned)min(it * STAGERS + st, wlast) * 16u;
    auto request_weights = [&](u32x4 (&w)[W_ITERS], int stage) {
        const unsigned char* src = A.wpk + (size_t)stage * W_STAGE;   // uniform
        const unsigned long long base = reinterpret_cast<unsigned long long>(src);
        const unsigned base_lo = (unsigned)base, base_hi = (unsigned)(base >> 32);
        static_assert(W_ITERS == 3 || W_ITERS == 5, "one statement per weight stage");
        if constexpr (W_ITERS == 3)
            asm volatile("s_mov_b32 s60, %6\n\ts_mov_b32 s61, %7\n\tglobal_load_dwordx4 %0, %3, s[60:61]\n\t"
                         "global_load_dwordx4 %1, %4, s[60:61]\n\tglobal_load_dwordx4 %2, %5, s[60:61]"
                         : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2])
                         : "v"(woff[0]), "v"(woff[1]), "v"(woff[2]), "s"(base_lo), "s"(base_hi)
                         : "memory", "s60", "s61");
        else
            asm volatile("s_mov_b32 s60, %10\n\ts_mov_b32 s61, %11\n\tglobal_load_dwordx4 %0, %5, s[60:61]\n\t"
                         "global_load_dwordx4 %1, %6, s[60:61]\n\tglobal_load_dwordx4 %2, %7, s[60:61]\n\t"
                         "global_load_dwordx4 %3, %8, s[60:61]\n\tglobal_load_dwordx4 %4, %9, s[60:61]"
                         : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[W_ITERS > 3 ? 3 : 0]), "=&v"(w[W_ITERS > 4 ? 4 : 0])
                         : "v"(woff[0]), "v"(woff[1]), "v"(woff[2]), "v"(woff[W_ITERS > 3 ? 3 : 0]),
                           "v"(woff[W_ITERS > 4 ? 4 : 0]), "s"(base_lo), "s"(base_hi)
                         : "memory", "s60", "s61");
    };
    auto write_weights = [&](const u32x4 (&w)[W_ITERS], unsigned char* buf) {
        u32x4* dst = reinterpret_cast<u32x4*>(buf);
#pragma unroll
        for (int it = 0; it < W_ITERS; ++it) dst[min(it * STAGERS + st, wlast)] = w[it];
    };
    // folded InstanceNorm coefficients of the NEXT tile's (batch entry, plane) -> its table in LDS: part of every
    // stage's requests / writes, branch-free -- a load inside a branch merges with "no load" in a phi and the compiler
    // then drains ALL outstanding loads at the merge.  The writes of the first stages carry the placeholder tile's
    // values and are overwritten by the later ones well before the table is first read (stage 3 * nks - 3 >= 6).
    const unsigned coef_lane = (unsigned)coef_c * (A.a.per_plane ? (unsigned)A.D : 1u) * 4u;
    auto request_coef = [&](float& cs, float& ch, const Tile& tl) {
        if (NORM) {
            const size_t g0 = A.a.per_plane ? ((size_t)(tl.n * A.Cin) * A.D + tl.d) : (size_t)(tl.n * A.Cin);   // uniform
            const float *ps = A.a.scale + g0, *ph = A.a.shift + g0;
            const unsigned long long bs = reinterpret_cast<unsigned long long>(ps), bh = reinterpret_cast<unsigned long long>(ph);
            const unsigned s_lo = (unsigned)bs, s_hi = (unsigned)(bs >> 32), h_lo = (unsigned)bh, h_hi = (unsigned)(bh >> 32);
            asm volatile("s_mov_b32 s60, %2\n\ts_mov_b32 s61, %3\n\tglobal_load_dword %0, %6, s[60:61]\n\t"
                         "s_mov_b32 s60, %4\n\ts_mov_b32 s61, %5\n\tglobal_load_dword %1, %6, s[60:61]"
                         : "=&v"(cs), "=&v"(ch)
                         : "s"(s_lo), "s"(s_hi), "s"(h_lo), "s"(h_hi), "v"(coef_lane)
                         : "memory", "s60", "s61");
        }
    };
    auto write_coef = [&](float cs, float ch, int table) {
        if (NORM) {
            coef_tab[table * 2 * CMAX + coef_c] = cs * A.ascale;   // (exact power of two of the fp16 form; 1 for bf16)
            coef_tab[table * 2 * CMAX + CMAX + coef_c] = ch * A.ascale;
        }
    };
    // one fp64 (sum, sum of squares) record per tile and channel from the four MFMA waves' rows
    auto fold_statistics = [&](const Tile& tl) {
        if (A.partials && st < 128) {
            const float* red = reinterpret_cast<const float*>(lds + LDS_RED);
            const int oc = st >> 1, k = st & 1;
            double v = 0.0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) v += (double)red[(wv * 64 + oc) * 2 + k];
            A.partials[((((size_t)tl.n * A.Cout + oc) * A.D + tl.d) * A.tiles + tl.tile) * 2 + k] = v;
        }
    };

    // ---- prologue: the first tile's coefficient table, first K-step and weight stage; then the requests of what the
    // first two stages write (set 0: stage 0, set 1: stage 1)
    Tile cur = decode_tile(A, cur_id);
    Tile nxt = cur, done = cur;
    int nxt_id = -1;
    int tpar = 0;             // coefficient table of the current tile
    auto await_all = [&](X3In<CBI>& x, u32x4 (&w)[W_ITERS], float& cs, float& ch) {
        if constexpr (CBI == 2) {
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(x.q[0]), "+v"(x.q[1]), "+v"(x.q[2]), "+v"(x.q[3]), "+v"(x.h[0]), "+v"(x.h[1]), "+v"(x.h[2]), "+v"(x.h[3])
                         :
                         : "memory");
        } else if constexpr (CBI == 1) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(x.q[0]), "+v"(x.q[1]), "+v"(x.q[2]), "+v"(x.q[3]) : : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]), "+v"(x.v[5]), "+v"(x.v[6]),
                           "+v"(x.v[7]), "+v"(x.v[8]), "+v"(x.v[9]), "+v"(x.v[10]), "+v"(x.v[11]), "+v"(x.v[12]), "+v"(x.v[13]),
                           "+v"(x.v[14]), "+v"(x.v[15])
                         :
                         : "memory");
        }
#pragma unroll
        for (int it = 0; it < W_ITERS; ++it) asm volatile("" : "+v"(w[it]) : : "memory");
        asm volatile("" : "+v"(cs), "+v"(ch) : : "memory");
    };
#pragma unroll
    for (int set = 0; set < 2; ++set) {
        xin[set].clear();
#pragma unroll
        for (int it = 0; it < W_ITERS; ++it) win[set][it] = u32x4{0u, 0u, 0u, 0u};
    }
    request_coef(coef_s[0], coef_h[0], cur);
    request_weights(win[0], 0);
    await_all(xin[0], win[0], coef_s[0], coef_h[0]);
    write_coef(coef_s[0], coef_h[0], 0);
    write_weights(win[0], lds + LDS_W);
    x3_barrier();
    for (int third = 0; third < 3; ++third) {
        request_inputs(xin[0], cur, 0, third);
        await_all(xin[0], win[0], coef_s[0], coef_h[0]);
        write_inputs(xin[0], cur, 0, third, lds, coef_tab);
    }
#pragma unroll
    for (int set = 0; set < 2; ++set) {   // stage `set` writes third `set` of K-step 1 and weight stage `set` + 1
        request_coef(coef_s[set], coef_h[set], cur);
        request_weights(win[set], 1 + set);
        request_inputs(xin[set], cur, 1, set);
    }
    int upar = 0, wpar = 0;   // LDS buffer of the K-step / weight stage being consumed
    int rs = 0;               // stage within the current tile
    bool fold_pending = false;
    x3_barrier();

    // one stage on register set SET; returns true after the last stage of the last tile
    auto stage = [&](auto set_constant) -> bool {
        constexpr int SET = decltype(set_constant)::value;
        const int ks = rs / 3, dy = rs - 3 * ks;
        if (rs == 1) {   // the tile drawn during stage 0 (published by its barrier)
            nxt_id = __builtin_amdgcn_readfirstlane(next_slot[0]);
            if (nxt_id >= 0) nxt = decode_tile(A, nxt_id);
        }
        // a request whose answer this stage itself needs goes out first (its counted wait skips the younger ones)
        int drawn = -1;
        if (rs == 0 && st == 0) drawn = draw_tile(A.queue, home, A.planes, A.tiles, A.tiles_x, A.tiles_x_full, n_full);
#ifndef PDS_X3_NOSTAGE
        // -- write what was requested two stages ago: third (rs % 3) of K-step ks + 1 and weight stage rs + 1.  No
        // branch stands between a request and its use (see request_coef): past the last tile the sequence re-stages
        // the current tile into the idle buffer, which nobody reads.
        {
            await_set(xin[SET], win[SET], coef_s[SET], coef_h[SET]);
            const bool into_next = ks + 1 >= nks;
            write_inputs(xin[SET], pick_tile(into_next, nxt, cur), into_next ? 0 : ks + 1, dy,
                         lds + (upar ^ 1) * IN_BUF, coef_tab + ((into_next ? tpar ^ 1 : tpar) * 2 * CMAX));
            write_weights(win[SET], lds + LDS_W + (wpar ^ 1) * W_STAGE);
            write_coef(coef_s[SET], coef_h[SET], tpar ^ 1);
        }
        // -- requests of what stage rs + 2 writes: sequence position rs + 5, weight stage rs + 3
        {
            const int q = rs + 5;
            const int ksl = q / 3, third = q - 3 * ksl;
            const bool into_next = ksl >= nks;
            request_coef(coef_s[SET], coef_h[SET], nxt);
            request_inputs(xin[SET], pick_tile(into_next, nxt, cur), into_next ? ksl - nks : ksl, third);
            int ws = rs + 3;
            if (ws >= nstages) ws -= nstages;
            request_weights(win[SET], ws);
        }
#endif
        // -- housekeeping with the staging waves' spare time
        if (rs == 0 && st == 0) next_slot[0] = drawn;
        if (rs == 1 && fold_pending) fold_statistics(done);
        if (dy == 2) upar ^= 1;
        wpar ^= 1;
        x3_barrier();
        if (++rs < nstages) return false;
        // -- end of the tile
        rs = 0;
        done = cur;
        fold_pending = true;
        if (nxt_id < 0) return true;
        cur = nxt;   // (nxt stays a valid tile -- this one -- until the next draw is read at stage 1)
        nxt_id = -1;
        tpar ^= 1;
        return false;
    };
    for (;;) {
        if (stage(std::integral_constant<int, 0>{})) break;
        if (stage(std::integral_constant<int, 1>{})) break;
    }
    // the requests of the last stages are never consumed: they must not outlive the wave (their late writes would land
    // in registers that belong to another wave by then)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    x3_barrier();
    fold_statistics(done);
}

}  // namespace

template <int P, bool NORM, int CBI, bool CBO>
__global__ __launch_bounds__(THREADS, 2) void conv2d_x3_kernel(const X3Args A0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int* next_slot = reinterpret_cast<int*>(lds + X3Cfg<P>::LDS_NEXT);
    X3Args A = A0;
    A.ascale = 1.f;
    A.unscale = 1.f;
    if constexpr (P == 2) {
        // operand scales of the fp16 form: the source's range certificate -> as; 1 / ws from the packed weights' tail
        const float bound = block_bound(A.bound, A.bound_n, reinterpret_cast<float*>(lds + X3Cfg<P>::LDS_RED));
        // (wave-uniform: kept in scalar registers, the MFMA waves have no vector registers to spare)
        A.ascale = __builtin_bit_cast(
            float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pow2_scale(bound, kHalfTarget))));
        A.unscale = __builtin_bit_cast(
            float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(
                       int, reinterpret_cast<const float*>(A.queue)[13] * (1.f / A.ascale))));
    }
    if (tid == STAGERS)
        next_slot[0] = draw_tile(A.queue, blockIdx.x & 7, A.planes, A.tiles, A.tiles_x, A.tiles_x_full,
                                 A.tiles_y * A.tiles_x_full);
    __syncthreads();
    const int cur_id = __builtin_amdgcn_readfirstlane(next_slot[0]);
    if (cur_id >= 0) {
        if (wave < 4) x3_mfma_waves<P, NORM, CBO>(A, lds, wave, tid & 63, cur_id);
        else x3_staging_waves<P, NORM, CBI>(A, lds, tid - STAGERS, cur_id);
    }
    // A workgroup leaves only after it has found every queue empty, so the last one to leave may reset the counters
    // for the next launch on this workspace (no memset launch per layer; the packing launch zeroes them the first time).
    __syncthreads();
    if (tid == 0) {
        if (atomicAdd(A.queue + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
            for (int q = 0; q < 9; ++q) atomicExch(A.queue + q, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------
bool conv2d_x3_supported(const ConvLayer& L) {
    static const bool enabled = []() {  // PDS_X3=0 keeps the exact-fp32 MFMA kernels (A/B, debugging)
        const char* e = debug_switch("PDS_X3");
        return !(e && e[0] == '0');
    }();
    if (!enabled) return false;
    if (L.kd != 1 || L.stride != 1 || L.out_g.c != 64) return false;
    if (L.in.c % 16 != 0 || L.in.c < 48 || L.in.c > CMAX) return false;   // three K-steps at least (staging lead)
    if (L.b.p || L.l0A || L.side_out || L.plane_weight_sets > 0) return false;
    if ((size_t)L.in.d * L.in.h * L.in.w * 8 >= ((size_t)1 << 30)) return false;   // 32-bit channel offsets
    if ((size_t)L.in.n * L.in.d >= ((size_t)1 << 20)) return false;
    return true;
}

// statistics records per output plane
int conv2d_x3_tiles(const ConvLayer& L) {
    const Geom& o = L.out_g;
    return ((o.h + TH - 1) / TH) * ((o.w + TW - 1) / TW);
}

// dwords of packed weights (sized for the three-part form) + the eight queue counters behind them
static size_t x3_weight_dwords(int cin, int parts) { return (size_t)(cin / 16) * 3 * (3 * parts * 2 * W_FRAG / 4); }
size_t conv2d_x3_packed_floats(int cin) { return x3_weight_dwords(cin, 3) + 64; }

// fp16 form (three products) when the source carries a range certificate (Src::bound), else the range-safe bf16 form
static bool x3_use_fp16(const ConvLayer& L) {
    static const bool enabled = []() {  // PDS_X3_FP16=0: every launch on the range-safe bf16 form (A/B, debugging)
        const char* e = debug_switch("PDS_X3_FP16");
        return !(e && e[0] == '0');
    }();
    return enabled && L.a.bounded;
}

template <int P, int CBI, bool CBO>
static int x3_launch(const ConvLayer& L, X3Args& A, int workgroups, hipStream_t s) {
    using C = X3Cfg<P>;
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_x3_kernel<P, true, CBI, CBO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_x3_kernel<P, false, CBI, CBO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    }
    // (planes * tiles rides along as the "workgroups" of the probe record: it tells a 48-plane layer from a small one)
    const int probe = probe_before(P == 2 ? "conv2d_x3<fp16>" : "conv2d_x3<bf16>", s);
    if (L.a.scale)
        hipLaunchKernelGGL((conv2d_x3_kernel<P, true, CBI, CBO>), dim3(workgroups), dim3(THREADS), C::LDS_BYTES, s, A);
    else hipLaunchKernelGGL((conv2d_x3_kernel<P, false, CBI, CBO>), dim3(workgroups), dim3(THREADS), C::LDS_BYTES, s, A);
    probe_after(probe, A.planes * A.tiles, s);
    return check_launch("conv2d_x3");
}

// channel-blocked tensors ([N][D][C / 8][H][W][8], common.hpp Src::cb8): the fp16 form, whole tiles only, 64 channels out
bool conv2d_x3_cb8_ok(const ConvLayer& L, bool in_cb8, bool out_cb8) {
    if (!in_cb8 && !out_cb8) return true;
    if (!conv2d_x3_supported(L) || !x3_use_fp16(L)) return false;   // (both honour their debug switches)
    if (L.in.h % TH != 0 || L.in.w % 16 != 0) return false;
    if (out_cb8 && (L.out_batch_channels > 0 && L.out_batch_channels != 64)) return false;
    if (in_cb8 && L.in.c % 8 != 0) return false;
    return true;
}

int launch_conv2d_x3(const ConvLayer& L, hipStream_t s) {
    if (!L.packed) return set_error(-1, "conv2d_x3: packed weights missing");
    const bool fp16 = x3_use_fp16(L);
    const int total = (int)x3_weight_dwords(L.in.c, fp16 ? 2 : 3);
    const PackPhase phase = L.sink ? L.sink->phase : kPackInline;
    if (phase != kPackDone) {
        PackJob j;
        j.src = L.weight;
        j.dst = L.packed;
        j.cout = L.out_g.c;
        j.cin = L.in.c;
        j.mblocks = 2;
        j.kc = 4;
        j.taps = 9;
        // 6: three-way bf16 split, 7: two-way fp16 split of ws * w (ws: the power of two pack_wscale_kernel derives from max|w|); A-fragment order of v_mfma_f32_32x32x16_{bf16,f16}
        j.mode = fp16 ? 7 : 6;
        j.total = total + 16;   // + the queue counters (zeroed by the packing launch)
        if (phase == kPackCollect) return L.sink->push(j) ? 0 : set_error(-1, "pack job table full");
        if (int rc = launch_multi_pack(&j, 1, s)) return rc;
    }
    X3Args A;
    A.a = L.a;
    A.wpk = reinterpret_cast<const unsigned char*>(L.packed);
    A.bias = L.bias;
    A.out = L.out;
    A.partials = L.partials;
    A.queue = reinterpret_cast<int*>(L.packed + total);
    A.N = L.in.n;
    A.Cin = L.in.c;
    A.D = L.in.d;
    A.H = L.in.h;
    A.W = L.in.w;
    A.Cout = L.out_g.c;
    A.CoutStride = L.out_batch_channels > 0 ? L.out_batch_channels : L.out_g.c;
    A.lrelu = L.lrelu;
    A.tiles_x = (A.W + TW - 1) / TW;
    A.tiles_y = (A.H + TH - 1) / TH;
    A.tiles = A.tiles_x * A.tiles_y;
    const int rem = A.W % TW;
    A.tiles_x_full = A.tiles_x - ((rem != 0 && rem <= 16) ? 1 : 0);
    A.planes = A.N * A.D;
    A.nks = A.Cin / 16;
    A.epi_lds = 1;   // (round 3's PDS_X3_EPI_LDS=0 switch -- direct stores from the D fragment, 5 % slower -- is gone)
    A.bound = L.a.bound;
    A.bound_n = L.a.bound_n;
    if (fp16 && (!A.bound || A.bound_n <= 0)) return set_error(-1, "conv2d_x3: fp16 form without a range bound");
    static std::atomic<unsigned> cus_done{0};   // one bit per device
    static int cus[32] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (DeviceOnce once{cus_done}) {
        int n = 0;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        cus[dev & 31] = n > 0 ? n : 256;
    }
    const long long all = (long long)A.planes * A.tiles;
    const int workgroups = (int)(all < cus[dev & 31] ? all : cus[dev & 31]);
    A.in_cb8 = L.a.cb8;
    A.out_cb8 = L.out_cb8;
    A.l1B = L.l1B;
    A.l1H = L.l1H;
    A.l1_bstride = L.l1_bstride;
    A.l1_hstride = L.l1_hstride;
    A.l1_edge = L.l1_edge;
    A.l1_P = L.l1_P;
    A.l1_d0 = L.l1_d0;
    if (L.l1B) {   // layer-1 planes formed on the fly (always behind the deferred InstanceNorm of t1)
        if (!fp16 || !L.a.scale || A.in_cb8 || !conv2d_x3_cb8_ok(L, true, A.out_cb8 != 0))
            return set_error(-1, "conv2d_x3: the layer-1 source needs the fp16 form, a deferred InstanceNorm and whole tiles");
        return A.out_cb8 ? x3_launch<2, 2, true>(L, A, workgroups, s) : x3_launch<2, 2, false>(L, A, workgroups, s);
    }
    if (A.in_cb8 || A.out_cb8) {
        if (!fp16 || !conv2d_x3_cb8_ok(L, A.in_cb8, A.out_cb8))
            return set_error(-1, "conv2d_x3: channel-blocked tensors need the fp16 form and whole 16 x 16 tiles");
        if (A.in_cb8 && A.out_cb8) return x3_launch<2, 1, true>(L, A, workgroups, s);
        if (A.in_cb8) return x3_launch<2, 1, false>(L, A, workgroups, s);
        return x3_launch<2, 0, true>(L, A, workgroups, s);
    }
    return fp16 ? x3_launch<2, 0, false>(L, A, workgroups, s) : x3_launch<3, 0, false>(L, A, workgroups, s);
}

}  // namespace pds
