// 3x3 convolution 64 -> 8 channels over a stack of 2-D planes, written straight into the stacked signature layout:
// the last convolution of MatchingOperation for all disparity planes at once (reference
// practical_deep_stereo/matching.py:89-93 and the stack of :63; its input is the residual sum of
// network_blocks.py:143-144, formed here from the two sources while staging).
//
// Arithmetic (round 3).  The layer moves 850 MB for 15 GFLOP, but on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, 1/16 of
// the 16-bit rate) its matrix work alone took 160 of the kernel's 295 us (rocprofv3: 10.6 M MFMAs, pipe busy 55 %).  As in
// conv2d_x3.hip every fp32 operand is therefore split into two fp16 parts and a product is
// hi*lo + lo*hi + hi*hi on v_mfma_f32_16x16x16_f16 with fp32 accumulation -- measured more accurate than the fp32 fma
// chain (tools/ubench/fp16x2_probe.hip).  Both operands are pre-scaled by exact powers of two derived from the data
// (round 4): ws from max|w| (every workgroup reduces the 4 608 weights while it gathers them), as from the range
// certificates of the sources (common.hpp Src::bound; the sum of the two when the residual sum is formed here), so any
// finite input is in range; a source without a certificate keeps the exact-fp32 kernel of conv2d_mfma.hip.
//
// With 8 output channels an M = 16 tile would idle half of its rows; here (as in conv3d_t8.hip, along y instead of z)
// the M side is (output channel, parity of the output row): one MFMA makes 8 channels x 2 consecutive rows y, y+1 for 16
// pixels, and its K = 16 is FOUR input channels x the four input rows y-1 .. y+2 they touch:
//     A[(oc, py)][(yi, ic)] = ws W[oc][ic][dy = yi - py][dx]   (0 when dy is outside 0..2)
//     B[(yi, ic)][n]        = in[ic][y - 1 + yi][x + n + dx - 1]
// i.e. 9 MFMAs x 3 products per (4 channels x 8 channels x 2 rows x 16 pixels).
//
//   workgroup   4 waves, PERSISTENT (2 per CU), static lists of (plane, tile) in contiguous runs per XCD.  Tile:
//               16 rows x 32 columns of one (batch, plane); wave w owns rows 4w .. 4w+3: two row pairs x two column
//               blocks = four accumulators, so every A fragment read from LDS feeds four MFMAs.
//   weights     all 96 A fragments (16 chunks x 3 dx x 2 parts) live in LDS (48 KB), gathered and split once per workgroup
//               straight from the [8, 64, 3, 3] tensor: no packing launch.
//   LDS input   [buffer 2][part 2][18 rows][36 columns][4 channels] fp16: a lane's B fragment (row yi = lane >> 4, pixel
//               lane & 15, the chunk's four channels) is one aligned 8-byte slot.
//   pipeline    the 64 input channels stream through the two LDS buffers in chunks of 4; a thread owns up to three halo
//               positions and all four channels of the chunk; chunk g + 4 is being loaded (buffer loads, zero padding
//               from the range check), chunk g + 1 is normalised / summed / split / written in the shadow of the MFMAs
//               of chunk g (sched_group_barrier interleave); one barrier per chunk, the stream runs on across tiles.
//   epilogue    accumulators start at ws as bias; 1 / (ws as), optional LeakyReLU; 64-byte row segments.
#include <atomic>
#include <type_traits>

#include "common.hpp"

namespace pds {

namespace {

constexpr int C2_THREADS = 256;
constexpr int C2_CIN = 64, C2_COUT = 8, C2_KC = 4, C2_CHUNKS = C2_CIN / C2_KC;
constexpr int C2_TY = 16, C2_NB = 2, C2_TX = 16 * C2_NB, C2_RP = C2_TY / 8;   // row pairs per wave
constexpr int C2_YT = C2_TY + 2, C2_XT = C2_TX + 2, C2_RS = 36;               // halo tile, slots per LDS row
constexpr int C2_PART = C2_YT * C2_RS * 8;                 // bytes of one split part of a chunk: [row][column][4 x fp16]
constexpr int C2_BUF = 2 * C2_PART;                        // bytes per input buffer (hi | lo)
constexpr int C2_AFRAGS = C2_CHUNKS * 3 * 2;               // (chunk, dx, part): 64 lanes x 8 bytes each
constexpr int C2_ABYTES = C2_AFRAGS * 64 * 8;
constexpr int C2_NPOS = C2_YT * C2_XT;                     // halo positions of a tile: 612
constexpr int C2_POS = (C2_NPOS + C2_THREADS - 1) / C2_THREADS;   // positions per thread: 3
static_assert(C2_RS >= C2_XT, "LDS row");
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct C2Args {
    Src a, b;
    const float* __restrict__ w;     // [8][64][3][3]
    const float* __restrict__ bias;  // [8]
    float* __restrict__ out;         // [N][8][D][H][W]
    int N, D, H, W;
    int lrelu;
    int tiles_x, tiles_y, tiles;     // tiles = N * D * tiles_y * tiles_x
};

// two-way fp16 split of four fp32 values (round to nearest): hi carries 11 bits, lo the next 11
__device__ __forceinline__ void c2_split(const float (&v)[4], f16x4& hi, f16x4& lo) {
    pds_u32x2 h, l;   // (packed conversions, common.hpp: 3 instead of 5 instructions per value in the staging path)
    split_quad_f16(v, h, l);
    hi = __builtin_bit_cast(f16x4, h);
    lo = __builtin_bit_cast(f16x4, l);
}

// power-of-two operand scales of a launch (header comment), wave-uniform; called by ALL threads before anything else
// touches the LDS scratch `red`
struct C2Scales {
    float ws, as, unscale;
};
__device__ __forceinline__ float c2_uniform(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ C2Scales c2_scales(const C2Args& A, bool two, float* red) {
    // three maxima (weights, bound records of a, of b) in ONE pass: all loads go out together, one barrier pair
    float m[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < C2_COUT * C2_CIN * 9; i += blockDim.x) m[0] = fmaxf(m[0], fabsf(A.w[i]));
    for (int i = threadIdx.x; i < A.a.bound_n; i += blockDim.x) {
        const float v = fabsf(A.a.bound[i]);
        m[1] = fmaxf(m[1], v == v ? v : __builtin_inff());
    }
    if (two)
        for (int i = threadIdx.x; i < A.b.bound_n; i += blockDim.x) {
            const float v = fabsf(A.b.bound[i]);
            m[2] = fmaxf(m[2], v == v ? v : __builtin_inff());
        }
    const int wave = threadIdx.x >> 6, waves = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        m[k] = wave_max(m[k]);
        if ((threadIdx.x & 63) == 0) red[wave * 3 + k] = m[k];
    }
    __syncthreads();
    float r[3] = {0.f, 0.f, 0.f};
    for (int w = 0; w < waves; ++w)
#pragma unroll
        for (int k = 0; k < 3; ++k) r[k] = fmaxf(r[k], red[w * 3 + k]);
    __syncthreads();
    C2Scales S;
    S.ws = c2_uniform(pow2_scale(r[0], kHalfTarget));
    S.as = c2_uniform(pow2_scale(r[1] + r[2], kHalfTarget));
    S.unscale = c2_uniform((1.f / S.ws) * (1.f / S.as));
    return S;
}

}  // namespace

// TWO: second source present (the residual sum); NA / NB2: source a / b carries a deferred InstanceNorm
template <bool TWO, bool NA, bool NB2>
__global__ __launch_bounds__(C2_THREADS, 2) void conv2d_t8_kernel(const C2Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* abuf = lds;                           // [96 fragments][64 lanes][4 x fp16]
    unsigned char* ibuf = lds + C2_ABYTES;               // [2][hi | lo][18 rows][36][4 x fp16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, q = lane >> 4;
    const size_t plane = (size_t)A.H * A.W;
    const size_t cstride = (size_t)A.D * plane;
    const int cbytes = (int)(cstride * sizeof(float));
    // one resource per tensor for the whole launch: batch, channel and plane enter as the scalar offset
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A.a.p), 0, (int)((size_t)A.N * C2_CIN * cstride * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(TWO ? A.b.p : A.a.p), 0, (int)((size_t)A.N * C2_CIN * cstride * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        A.out, 0, (int)((size_t)A.N * C2_COUT * cstride * sizeof(float)), 0x00020000);

    const C2Scales SC = c2_scales(A, TWO, reinterpret_cast<float*>(ibuf));
    // ---- A fragments -> LDS: lane (m = lane & 15 -> oc = m >> 1, py = m & 1 ; yi = lane >> 4), four channels each ----
    for (int e = tid; e < C2_CHUNKS * 3 * 64; e += C2_THREADS) {
        const int l = e & 63, f = e >> 6;                // f = chunk * 3 + dx
        const int m = l & 15, oc = m >> 1, py = m & 1, yi = l >> 4, dy = yi - py;
        const int chunk = f / 3, dx = f % 3;
        float wv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            wv[i] = (dy >= 0 && dy <= 2)
                        ? A.w[((size_t)oc * C2_CIN + chunk * C2_KC + i) * 9 + dy * 3 + dx] * SC.ws
                        : 0.f;
        f16x4 hi, lo;
        c2_split(wv, hi, lo);
        *reinterpret_cast<f16x4*>(abuf + ((size_t)(f * 2 + 0) * 64 + l) * 8) = hi;
        *reinterpret_cast<f16x4*>(abuf + ((size_t)(f * 2 + 1) * 64 + l) * 8) = lo;
    }

    // ---- this workgroup's tiles: XCD x walks the x-th contiguous eighth of the (batch, plane, row, column) list ----
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int t_end = (int)(((long long)(xcd + 1) * A.tiles) >> 3);
    const int t_first = (int)(((long long)xcd * A.tiles) >> 3) + slot;
    const int my_tiles = t_first < t_end ? (t_end - t_first + per_xcd - 1) / per_xcd : 0;
    const int total_chunks = my_tiles * C2_CHUNKS;

    // staging role of this thread: halo positions tid + 256 k, all four channels of the chunk
    int lo_off[C2_POS];
#pragma unroll
    for (int k = 0; k < C2_POS; ++k) {
        const int p = min(tid + C2_THREADS * k, C2_NPOS - 1);    // (surplus threads re-stage the last position)
        lo_off[k] = ((p / C2_XT) * C2_RS + p % C2_XT) * 8;
    }

    struct TilePos {
        int nb, d, y0, x0;
    };
    auto tile_pos = [&](int local) {
        const int t = t_first + min(local, my_tiles - 1) * per_xcd;
        TilePos p;
        p.x0 = (t % A.tiles_x) * C2_TX;
        int r = t / A.tiles_x;
        p.y0 = (r % A.tiles_y) * C2_TY;
        r /= A.tiles_y;
        p.d = r % A.D;
        p.nb = r / A.D;
        return p;
    };

    // ---- staging state: offsets / padding mask / scalar position of the tile whose chunks are being FETCHED; they change
    //      every 16 chunks.  The new values are prepared in a (uniform) side block and selected into place inside the
    //      branch-free body, so that body -- stash, fetch, MFMAs -- stays one basic block.
    unsigned goff[C2_POS], goff_n[C2_POS];   // byte offsets inside (channel 0 of the chunk, plane 0); ~0: padding
    // padding as 0.0f / 1.0f factors of the InstanceNorm shift (a padded load already returns 0, so scale * 0 +
    // shift * 0 is the literal zero padding): one multiply instead of a bit test, a compare and a select per element
    float inside[C2_POS], inside_n[C2_POS];
    int sbase = 0, sbase_n = 0;              // scalar byte offset of (batch, plane) in the input
    // index of (batch, channel 0 [, plane]) in the scale / shift tables of the two sources and the channel stride there
    int gidx = 0, gidx_n = 0, gch = 0, gch_n = 0;
    const int stride_a = (NA && A.a.per_plane) ? A.D : 1, stride_b = (TWO && NB2 && A.b.per_plane) ? A.D : 1;
    auto prepare = [&](int local, unsigned* off, float* in, int& sb, int& gi, int& gc) {
        const TilePos P = tile_pos(local);
#pragma unroll
        for (int k = 0; k < C2_POS; ++k) {
            const int p = min(tid + C2_THREADS * k, C2_NPOS - 1);
            const int y = P.y0 + p / C2_XT - 1, x = P.x0 + p % C2_XT - 1;
            const bool ok = (unsigned)y < (unsigned)A.H && (unsigned)x < (unsigned)A.W;
            in[k] = ok ? 1.f : 0.f;
            off[k] = ok ? (unsigned)((size_t)y * A.W + x) * 4u : ~0u;
        }
        sb = (int)(((size_t)P.nb * C2_CIN * A.D + P.d) * plane * sizeof(float));
        gi = (NA && A.a.per_plane) ? P.nb * C2_CIN * A.D + P.d : P.nb * C2_CIN;               // source a
        gc = (TWO && NB2 && A.b.per_plane) ? P.nb * C2_CIN * A.D + P.d : P.nb * C2_CIN;       // source b
    };

    // four register sets: while chunk g is multiplied, chunk g + 1 is written to LDS and chunks g + 2 .. g + 4 are in
    // flight (some 15 MB of loads across the chip: what it takes to keep HBM busy at its latency; two sets in flight
    // measured 334 us, three 306)
    constexpr int DEPTH = 4;
    float va[DEPTH][C2_POS][C2_KC], vb[DEPTH][C2_POS][C2_KC];
    float vs[DEPTH][C2_KC], vh[DEPTH][C2_KC], vs2[DEPTH][C2_KC], vh2[DEPTH][C2_KC];   // (wave-uniform: scalar registers)
    float inside_regs[DEPTH][C2_POS];   // padding factors of the chunk held by each set
    auto fetch = [&](int g, auto set_c) {   // global -> registers of set SET; g already clamped
        constexpr int SET = decltype(set_c)::value;
        const int c0 = (g % C2_CHUNKS) * C2_KC;
        // (readfirstlane: the value IS wave-uniform, but the compiler cannot see it through the selects and would
        // wrap every buffer load in a waterfall loop)
        const int soff = __builtin_amdgcn_readfirstlane(sbase + c0 * cbytes);
#pragma unroll
        for (int k = 0; k < C2_POS; ++k) {
#pragma unroll
            for (int ch = 0; ch < C2_KC; ++ch) {
                va[SET][k][ch] =
                    __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, goff[k], soff + ch * cbytes, 0));
                if (TWO)
                    vb[SET][k][ch] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(rb, goff[k], soff + ch * cbytes, 0));
            }
            inside_regs[SET][k] = inside[k];
        }
        // deferred InstanceNorm of the producers: per (channel, plane) in Matching (per_plane), else per channel
#pragma unroll
        for (int ch = 0; ch < C2_KC; ++ch) {
            if (NA) {
                const int gi = __builtin_amdgcn_readfirstlane(gidx + (c0 + ch) * stride_a);
                vs[SET][ch] = A.a.scale[gi];
                vh[SET][ch] = A.a.shift[gi];
            }
            if (TWO && NB2) {
                const int gi = __builtin_amdgcn_readfirstlane(gch + (c0 + ch) * stride_b);
                vs2[SET][ch] = A.b.scale[gi];
                vh2[SET][ch] = A.b.shift[gi];
            }
        }
    };
    auto stash = [&](unsigned char* buf, auto set_c) {   // registers -> LDS: normalisation, the sum, the padding, the split
        constexpr int SET = decltype(set_c)::value;
        // The activation scale rides in the folded coefficients, multiplied HERE: the coefficients are scalar loads
        // issued by fetch() a chunk earlier; touching them there would wait for them on the spot (measured: +27 us per
        // launch of the full-width kernel).
        float cs[C2_KC], cs2[C2_KC], chs[C2_KC];
#pragma unroll
        for (int ch = 0; ch < C2_KC; ++ch) {
            cs[ch] = NA ? vs[SET][ch] * SC.as : SC.as;
            cs2[ch] = (TWO && NB2) ? vs2[SET][ch] * SC.as : SC.as;
            chs[ch] = ((NA ? vh[SET][ch] : 0.f) + ((TWO && NB2) ? vh2[SET][ch] : 0.f)) * SC.as;
        }
#pragma unroll
        for (int k = 0; k < C2_POS; ++k) {
            float v[4];
#pragma unroll
            for (int ch = 0; ch < C2_KC; ++ch) {
                float t = cs[ch] * va[SET][k][ch];
                if (TWO) t = fmaf(cs2[ch], vb[SET][k][ch], t);
                if (NA || (TWO && NB2)) t = fmaf(chs[ch], inside_regs[SET][k], t);   // both shifts vanish in the padding
                v[ch] = t;
            }
            f16x4 hi, lo;
            c2_split(v, hi, lo);
            *reinterpret_cast<f16x4*>(buf + lo_off[k]) = hi;
            *reinterpret_cast<f16x4*>(buf + C2_PART + lo_off[k]) = lo;
        }
    };

    const float bscale = SC.ws * SC.as;
    const float bias0 = (A.bias ? A.bias[2 * q] : 0.f) * bscale, bias1 = (A.bias ? A.bias[2 * q + 1] : 0.f) * bscale;
    const int b_base = ((2 * C2_RP * wave + q) * C2_RS + n16) * 8;     // halo row of the wave's first pair + yi, column n
    const unsigned out_lane = (unsigned)((size_t)(2 * q) * cstride + n16) * 4u;
    const unsigned out_c1 = (unsigned)cstride * 4u;

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    // offsets of the tile a chunk index falls into are brought up to date right before that chunk is fetched
    int fetched_tile = -1;
    auto fetch_chunk = [&](int g, auto set_c) {
        const int gf = min(g, total_chunks - 1);
        if (gf / C2_CHUNKS != fetched_tile) {      // uniform; only the prologue takes this path (the loop pre-selects)
            fetched_tile = gf / C2_CHUNKS;
            prepare(fetched_tile, goff, inside, sbase, gidx, gch);
        }
        fetch(gf, set_c);
    };
    if (total_chunks > 0) {
        fetch_chunk(0, S0());
        stash(ibuf, S0());
        fetch_chunk(1, S1());
        fetch_chunk(2, S2());
        fetch_chunk(3, S3());
    }
    __syncthreads();   // A fragments and the first chunk are in place

    f32x4 acc[C2_RP][C2_NB];
    TilePos Pcur = tile_pos(0);
    // one step of the stream: multiply chunk g, write chunk g + 1 (set (g + 1) % 4) to LDS, request chunk g + 4 (set g % 4)
    auto step = [&](int g, auto set_stash, auto set_fetch) {
        const int c = g % C2_CHUNKS;
        if (c == 0) {
            Pcur = tile_pos(g / C2_CHUNKS);
#pragma unroll
            for (int p = 0; p < C2_RP; ++p)
#pragma unroll
                for (int j = 0; j < C2_NB; ++j) acc[p][j] = f32x4{bias0, bias0, bias1, bias1};
        }
        // chunk g + 3 (fetched below) opens a new tile: its offsets are prepared here, on the side
        const int gf = min(g + DEPTH, total_chunks - 1);
        const bool new_tile = gf / C2_CHUNKS != fetched_tile;
        if (new_tile) prepare(gf / C2_CHUNKS, goff_n, inside_n, sbase_n, gidx_n, gch_n);
        fetched_tile = gf / C2_CHUNKS;

        // ---- branch-free body ----------------------------------------------------------------------------------------
        const unsigned char* buf = ibuf + (g & 1) * C2_BUF + b_base;
        stash(ibuf + ((g + 1) & 1) * C2_BUF, set_stash);   // chunk g + 1, masked with ITS tile's padding factors
#pragma unroll
        for (int k = 0; k < C2_POS; ++k) {
            inside[k] = new_tile ? inside_n[k] : inside[k];
            goff[k] = new_tile ? goff_n[k] : goff[k];
        }
        sbase = new_tile ? sbase_n : sbase;
        gidx = new_tile ? gidx_n : gidx;
        gch = new_tile ? gch_n : gch;
        fetch(gf, set_fetch);                       // lands three steps from now
        const unsigned char* af = abuf + ((size_t)c * 3 * 2 * 64 + lane) * 8;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const f16x4 a_hi = *reinterpret_cast<const f16x4*>(af + (dx * 2 + 0) * 512);
            const f16x4 a_lo = *reinterpret_cast<const f16x4*>(af + (dx * 2 + 1) * 512);
#pragma unroll
            for (int p = 0; p < C2_RP; ++p)
#pragma unroll
                for (int j = 0; j < C2_NB; ++j) {
                    const unsigned char* bp = buf + ((2 * p) * C2_RS + 16 * j + dx) * 8;
                    const f16x4 b_hi = *reinterpret_cast<const f16x4*>(bp);
                    const f16x4 b_lo = *reinterpret_cast<const f16x4*>(bp + C2_PART);
                    // small partial products first
                    acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_hi, b_lo, acc[p][j], 0, 0, 0);
                    acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_lo, b_hi, acc[p][j], 0, 0, 0);
                    acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_hi, b_hi, acc[p][j], 0, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < 3 * C2_RP * C2_NB * 3; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // VALU
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
        }
        if (c == C2_CHUNKS - 1) {
            // ---- epilogue of the tile: D row r = (channel 2q + (r >> 1), row parity r & 1) ------------------------------
            const int out_base = (int)((size_t)Pcur.nb * C2_COUT * cstride * sizeof(float));
#pragma unroll
            for (int p = 0; p < C2_RP; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int h = r >> 1, y = Pcur.y0 + 2 * (C2_RP * wave + p) + (r & 1);
                    const int row_bytes = __builtin_amdgcn_readfirstlane(
                        out_base + (int)(((size_t)Pcur.d * A.H + min(y, A.H - 1)) * A.W + Pcur.x0) * (int)sizeof(float));
#pragma unroll
                    for (int j = 0; j < C2_NB; ++j) {
                        float t = acc[p][j][r] * SC.unscale;
                        if (A.lrelu) t = fmaxf(t, t * kLeakySlope);
                        const bool ok = y < A.H && Pcur.x0 + 16 * j + n16 < A.W;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, t), ro,
                                                              ok ? out_lane + (h ? out_c1 : 0u) + 64u * j : ~0u, row_bytes, 0);
                    }
                }
        }
        __syncthreads();
    };
    for (int g = 0; g < total_chunks; g += 4) {   // the register sets rotate with period four (16 chunks per tile)
        step(g, S1(), S0());
        step(g + 1, S2(), S1());
        step(g + 2, S3(), S2());
        step(g + 3, S0(), S3());
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Full-width form (round 3).  The kernel above reads its 18 x 34 halo tiles as 136-byte row segments 960 bytes apart in 4
// channel planes x 2 tensors per step -- a pattern HBM serves at 2.7 TB/s (tools/ubench/tile_read_patterns.hip reproduces
// exactly its 0.31 ms with nothing but the loads), however much of the rest is removed (the fp16-split MFMAs cut the
// matrix time from 160 to 60 us and the launch stayed at 0.30 ms).  Tiles of 8 rows x the WHOLE row make the ten halo
// rows of a (channel, tile) one contiguous block read with 16-byte loads: 5.4 TB/s for the same bytes in the probe.
//   workgroup   8 waves, persistent, ONE per CU (the tile's two LDS buffers take 39 KB each at W = 240); wave w owns
//               the row pair w & 3 and the column half w >> 2 (NBH blocks of 16 columns).
//   LDS input   [buffer 2][part 2][10 rows][W + 4 slots][4 channels] fp16; slots 1 and W + 2 of every row are the zero
//               padding (written once; the image row starts at slot 2 so that a lane's four pixels are 16-byte aligned), rows outside the image are zero through the range check of the buffer loads.
//   pipeline    two register sets: chunk g + 1 is split / written while chunk g is multiplied, chunks g + 2 and g + 3
//               are in flight (a thread loads two (row, aligned quad) items x 4 channels x 2 sources per chunk).
// Round 6: tile height (8 or 6 rows) and staging items per thread are template parameters.  At W = 240 an 8-row tile has
// 10 x 60 = 600 staging items for 512 threads: six of the eight waves issued a second set of eight 16-byte loads (and
// converted its zeros) for nothing -- a build that staged only the first 512 items ran 279 -> 238 us (LAB_NOTES, round 5) --
// and three attempts to let them skip it spilled.  A 6-row tile has 8 x 60 = 480 items: ONE item per thread, half the
// staging registers; its fourth row pair does not exist, so waves 3 and 7 only stage.
constexpr int C2W_THREADS = 512;
constexpr int C2W_MAXW = 352;   // LDS: 48 KB of weights + 4 x 10 x (W + 4) x 8 bytes <= 160 KB
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool TWO, int NBH, int C2W_TY, int C2W_ITEMS>
__global__ __launch_bounds__(C2W_THREADS, 2) void conv2d_t8w_kernel(const C2Args A) {
    constexpr int C2W_YT = C2W_TY + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* abuf = lds;                           // [96 fragments][64 lanes][4 x fp16]
    unsigned char* ibuf = lds + C2_ABYTES;               // [2][hi | lo][10 rows][W + 2][4 x fp16]
    // LDS row: slot 1 = left zero column, slots 2 .. W + 1 = the image row (16-byte aligned quads), slot W + 2 = right zero
    const int RSW = A.W + 4, QW = A.W >> 2;
    const int part_bytes = C2W_YT * RSW * 8, buf_bytes = 2 * part_bytes;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave & 3, half = wave >> 2;
    const int n16 = lane & 15, q = lane >> 4;
    const size_t plane = (size_t)A.H * A.W;
    const size_t cstride = (size_t)A.D * plane;
    const int cbytes = (int)(cstride * sizeof(float));
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A.a.p), 0, (int)((size_t)A.N * C2_CIN * cstride * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(TWO ? A.b.p : A.a.p), 0, (int)((size_t)A.N * C2_CIN * cstride * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        A.out, 0, (int)((size_t)A.N * C2_COUT * cstride * sizeof(float)), 0x00020000);

    const C2Scales SC = c2_scales(A, TWO, reinterpret_cast<float*>(ibuf));
    // ---- A fragments -> LDS (as above); zero padding columns of the input buffers -------------------------------------
    for (int e = tid; e < C2_CHUNKS * 3 * 64; e += C2W_THREADS) {
        const int l = e & 63, f = e >> 6;
        const int m = l & 15, oc = m >> 1, py = m & 1, yi = l >> 4, dy = yi - py;
        const int chunk = f / 3, dx = f % 3;
        float wv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            wv[i] = (dy >= 0 && dy <= 2)
                        ? A.w[((size_t)oc * C2_CIN + chunk * C2_KC + i) * 9 + dy * 3 + dx] * SC.ws
                        : 0.f;
        f16x4 hi, lo;
        c2_split(wv, hi, lo);
        *reinterpret_cast<f16x4*>(abuf + ((size_t)(f * 2 + 0) * 64 + l) * 8) = hi;
        *reinterpret_cast<f16x4*>(abuf + ((size_t)(f * 2 + 1) * 64 + l) * 8) = lo;
    }
    for (int e = tid; e < 2 * 2 * C2W_YT * 2; e += C2W_THREADS) {   // (buffer, part, row) x (left | right)
        const int side = e & 1, row = e >> 1;
        *reinterpret_cast<unsigned long long*>(ibuf + ((size_t)row * RSW + (side ? A.W + 2 : 1)) * 8) = 0ull;
    }

    // ---- this workgroup's tiles: XCD x walks the x-th contiguous eighth of the (batch, plane, row tile) list -----------
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int t_end = (int)(((long long)(xcd + 1) * A.tiles) >> 3);
    const int t_first = (int)(((long long)xcd * A.tiles) >> 3) + slot;
    const int my_tiles = t_first < t_end ? (t_end - t_first + per_xcd - 1) / per_xcd : 0;
    const int total_chunks = my_tiles * C2_CHUNKS;

    // staging role of this thread: items tid + 512 k = (halo row, aligned quad of four columns), all four channels
    int item_row[C2W_ITEMS], item_col[C2W_ITEMS], lo_off[C2W_ITEMS], lo_step[C2W_ITEMS];
    bool item_ok[C2W_ITEMS];
#pragma unroll
    for (int k = 0; k < C2W_ITEMS; ++k) {
        // surplus threads (600 items for 1 024 slots at W = 240): their loads go out of range (zero, no memory access)
        // and the zeros land on slots 0 and 1 of halo row 0 (unused, left zero column; see the note at the stores) --
        // re-staging the last item kept 41 % of the loads in flight redundant
        item_ok[k] = tid + C2W_THREADS * k < C2W_YT * QW;
        const int it = min(tid + C2W_THREADS * k, C2W_YT * QW - 1);
        item_row[k] = it / QW;
        item_col[k] = 4 * (it % QW);
        lo_off[k] = item_ok[k] ? (item_row[k] * RSW + 2 + item_col[k]) * 8 : 0;
        lo_step[k] = item_ok[k] ? 16 : 0;
    }

    struct TilePos {
        int nb, d, y0;
    };
    auto tile_pos = [&](int local) {
        const int t = t_first + min(local, my_tiles - 1) * per_xcd;
        TilePos p;
        p.y0 = (t % A.tiles_y) * C2W_TY;
        const int r = t / A.tiles_y;
        p.d = r % A.D;
        p.nb = r / A.D;
        return p;
    };

    unsigned goff[C2W_ITEMS], goff_n[C2W_ITEMS];   // byte offsets inside (channel 0 of the chunk, plane 0); ~0: padding
    float inside[C2W_ITEMS], inside_n[C2W_ITEMS];
    int sbase = 0, sbase_n = 0;
    int gidx = 0, gidx_n = 0;
    const int stride_a = A.a.per_plane ? A.D : 1;
    auto prepare = [&](int local, unsigned* off, float* in, int& sb, int& gi) {
        const TilePos P = tile_pos(local);
#pragma unroll
        for (int k = 0; k < C2W_ITEMS; ++k) {
            const int y = P.y0 + item_row[k] - 1;
            const bool ok = (unsigned)y < (unsigned)A.H && item_ok[k];
            in[k] = ok ? 1.f : 0.f;
            off[k] = ok ? (unsigned)((size_t)y * A.W + item_col[k]) * 4u : ~0u;
        }
        sb = (int)(((size_t)P.nb * C2_CIN * A.D + P.d) * plane * sizeof(float));
        gi = A.a.per_plane ? P.nb * C2_CIN * A.D + P.d : P.nb * C2_CIN;
    };

    f32x4 va[2][C2W_ITEMS][C2_KC], vb[2][C2W_ITEMS][C2_KC];
    float vs[2][C2_KC], vh[2][C2_KC];
    float inside_regs[2][C2W_ITEMS];
    auto fetch = [&](int g, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        const int c0 = (g % C2_CHUNKS) * C2_KC;
        const int soff = __builtin_amdgcn_readfirstlane(sbase + c0 * cbytes);
#pragma unroll
        for (int k = 0; k < C2W_ITEMS; ++k) {
#pragma unroll
            for (int ch = 0; ch < C2_KC; ++ch) {
                va[SET][k][ch] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, goff[k], soff + ch * cbytes, 0));
                if (TWO)
                    vb[SET][k][ch] = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, goff[k], soff + ch * cbytes, 0));
            }
            inside_regs[SET][k] = inside[k];
        }
#pragma unroll
        for (int ch = 0; ch < C2_KC; ++ch) {
            const int gi = __builtin_amdgcn_readfirstlane(gidx + (c0 + ch) * stride_a);
            vs[SET][ch] = A.a.scale[gi];
            vh[SET][ch] = A.a.shift[gi];
        }
    };
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    auto stash = [&](unsigned char* buf, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        // the activation scale rides in the folded coefficients, multiplied here and not where they are loaded (a scalar
        // load touched in fetch() is waited for on the spot: +27 us per launch)
        float cs[C2_KC], chs[C2_KC];
#pragma unroll
        for (int ch = 0; ch < C2_KC; ++ch) {
            cs[ch] = vs[SET][ch] * SC.as;
            chs[ch] = vh[SET][ch] * SC.as;
        }
#pragma unroll
        for (int k = 0; k < C2W_ITEMS; ++k)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {   // two pixels = two slots = one 16-byte write per part (8-byte writes at the
                f16x8 hi2, lo2;                // lanes' 32-byte stride were 16-way bank conflicts)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int px = 2 * pp + e;
                    float v[4];
#pragma unroll
                    for (int ch = 0; ch < C2_KC; ++ch) {
                        float t = cs[ch] * va[SET][k][ch][px];
                        if (TWO) t = fmaf(SC.as, vb[SET][k][ch][px], t);
                        v[ch] = fmaf(chs[ch], inside_regs[SET][k], t);   // (the shift vanishes in the padding rows)
                    }
                    f16x4 hi, lo;
                    c2_split(v, hi, lo);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        hi2[4 * e + i] = hi[i];
                        lo2[4 * e + i] = lo[i];
                    }
                }
                // Surplus threads (out-of-range loads: zeros) store to slots 0 / 1 of halo row 0, i.e. onto the left zero
                // column: 0 * scale + 0 * shift = 0 for every finite coefficient.  A non-finite coefficient (NaN / inf
                // statistics of the producer) would put a NaN there -- but then every real pixel of that channel is NaN as
                // well and the convolution spreads it over all eight outputs of the plane in this form and in the planar
                // one alike, and pds_nonfinite_statistics() has counted it.  Predicating these stores off (tried in round
                // 6 after ADVICE r5) cost 56 us per launch: 279 -> 335 us (rocprofv3), the exec-mask juggling sits in the
                // kernel's critical issue path.
                *reinterpret_cast<f16x8*>(buf + lo_off[k] + pp * lo_step[k]) = hi2;
                *reinterpret_cast<f16x8*>(buf + part_bytes + lo_off[k] + pp * lo_step[k]) = lo2;
            }
    };

    const float bscale = SC.ws * SC.as;
    const float bias0 = (A.bias ? A.bias[2 * q] : 0.f) * bscale, bias1 = (A.bias ? A.bias[2 * q + 1] : 0.f) * bscale;
    // halo row of the wave's pair + yi, slot of column (16 * first block + n) - 1 (+ dx), i.e. + 2 - 1
    const int b_base = ((2 * pair + q) * RSW + 16 * half * NBH + n16 + 1) * 8;
    const unsigned out_lane = (unsigned)((size_t)(2 * q) * cstride + 16 * half * NBH + n16) * 4u;
    const unsigned out_c1 = (unsigned)cstride * 4u;

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    int fetched_tile = -1;
    auto fetch_chunk = [&](int g, auto set_c) {
        const int gf = min(g, total_chunks - 1);
        if (gf / C2_CHUNKS != fetched_tile) {
            fetched_tile = gf / C2_CHUNKS;
            prepare(fetched_tile, goff, inside, sbase, gidx);
        }
        fetch(gf, set_c);
    };
    __syncthreads();   // the zero columns are in place before the first stash
    if (total_chunks > 0) {
        fetch_chunk(0, S0());
        stash(ibuf, S0());
        fetch_chunk(1, S1());
        fetch_chunk(2, S0());
    }
    __syncthreads();

    f32x4 acc[NBH];
    TilePos Pcur = tile_pos(0);
#ifdef PDS_C2W_TIMING
    long long tmw[5] = {0, 0, 0, 0, 0};
#endif
    // one step: multiply chunk g, write chunk g + 1 (set (g + 1) & 1) to LDS and request chunk g + 3 into the same set
    auto step = [&](int g, auto set_c) {
        const int c = g % C2_CHUNKS;
        if (c == 0) {
            Pcur = tile_pos(g / C2_CHUNKS);
#pragma unroll
            for (int j = 0; j < NBH; ++j) acc[j] = f32x4{bias0, bias0, bias1, bias1};
        }
        const int gf = min(g + 3, total_chunks - 1);
        const bool new_tile = gf / C2_CHUNKS != fetched_tile;
        if (new_tile) prepare(gf / C2_CHUNKS, goff_n, inside_n, sbase_n, gidx_n);
        fetched_tile = gf / C2_CHUNKS;

        // ---- branch-free body ----------------------------------------------------------------------------------------
        const unsigned char* buf = ibuf + (g & 1) * buf_bytes + b_base;
#ifdef PDS_C2W_TIMING   // (debug builds only: cycle stamps around the phases of a step; the scheduling barriers below are
                        // disabled with it, so the phases run one after the other)
        const long long tt0 = __builtin_readcyclecounter();
#endif
        stash(ibuf + ((g + 1) & 1) * buf_bytes, set_c);
#ifdef PDS_C2W_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long tt1 = __builtin_readcyclecounter();
#endif
#pragma unroll
        for (int k = 0; k < C2W_ITEMS; ++k) {
            inside[k] = new_tile ? inside_n[k] : inside[k];
            goff[k] = new_tile ? goff_n[k] : goff[k];
        }
        sbase = new_tile ? sbase_n : sbase;
        gidx = new_tile ? gidx_n : gidx;
        fetch(gf, set_c);
#ifdef PDS_C2W_TIMING
        const long long tt2 = __builtin_readcyclecounter();
#endif
        const unsigned char* af = abuf + ((size_t)c * 3 * 2 * 64 + lane) * 8;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const f16x4 a_hi = *reinterpret_cast<const f16x4*>(af + (dx * 2 + 0) * 512);
            const f16x4 a_lo = *reinterpret_cast<const f16x4*>(af + (dx * 2 + 1) * 512);
#pragma unroll
            for (int j = 0; j < NBH; ++j) {
                // (blocks past the end of the row read the next row's slots -- or the slack behind the last buffer --
                // and are never stored)
                const unsigned char* bp = buf + (16 * j + dx) * 8;
                const f16x4 b_hi = *reinterpret_cast<const f16x4*>(bp);
                const f16x4 b_lo = *reinterpret_cast<const f16x4*>(bp + part_bytes);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_hi, b_lo, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_lo, b_hi, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_hi, b_hi, acc[j], 0, 0, 0);
            }
        }
#ifndef PDS_C2W_TIMING
#pragma unroll
        for (int i = 0; i < 3 * NBH * 3; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // VALU
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
        }
#else
        const long long tt3 = __builtin_readcyclecounter();
        tmw[0] += tt1 - tt0; tmw[1] += tt2 - tt1; tmw[2] += tt3 - tt2;
#endif
        if (c == C2_CHUNKS - 1) {
            // ---- epilogue of the tile: D row r = (channel 2q + (r >> 1), row parity r & 1) ------------------------------
            const int out_base = (int)((size_t)Pcur.nb * C2_COUT * cstride * sizeof(float));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int h = r >> 1, y = Pcur.y0 + 2 * pair + (r & 1);
                const int row_bytes = __builtin_amdgcn_readfirstlane(
                    out_base + (int)(((size_t)Pcur.d * A.H + min(y, A.H - 1)) * A.W) * (int)sizeof(float));
#pragma unroll
                for (int j = 0; j < NBH; ++j) {
                    float t = acc[j][r] * SC.unscale;
                    if (A.lrelu) t = fmaxf(t, t * kLeakySlope);
                    // (6-row tiles: the fourth row pair belongs to the next tile -- its waves multiplied whatever lies behind
                    // the eight halo rows and store nothing)
                    const bool ok = 2 * pair < C2W_TY && y < A.H && 16 * (half * NBH + j) + n16 < A.W;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, t), ro,
                                                          ok ? out_lane + (h ? out_c1 : 0u) + 64u * j : ~0u, row_bytes, 0);
                }
            }
        }
#ifdef PDS_C2W_TIMING
        const long long tt4 = __builtin_readcyclecounter();
#endif
        __syncthreads();
#ifdef PDS_C2W_TIMING
        tmw[3] += __builtin_readcyclecounter() - tt4;
        tmw[4] += 1;
#endif
    };
#ifdef PDS_C2W_TIMING
    const long long t_all = __builtin_readcyclecounter();
#endif
    for (int g = 0; g < total_chunks; g += 2) {   // the two register sets alternate (16 chunks per tile)
        step(g, S1());
        step(g + 1, S0());
    }
#ifdef PDS_C2W_TIMING
    if (lane == 0 && (blockIdx.x % 61) == 0 && (wave == 0 || wave == 5))
        printf("[t8w] wg %d wave %d: total %lld | stash(+wait) %lld fetch-issue %lld mfma %lld barrier %lld over %lld chunks (%d tiles)\n",
               (int)blockIdx.x, wave, (long long)__builtin_readcyclecounter() - t_all, tmw[0], tmw[1], tmw[2], tmw[3], tmw[4], my_tiles);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
namespace {

bool c2t8_enabled() {
    static const bool on = []() {  // PDS_CONV2D_T8=0: conv2d_mfma.hip serves the layer (A/B)
        const char* e = debug_switch("PDS_CONV2D_T8");
        return !(e && e[0] == '0');
    }();
    return on;
}

template <bool TWO, bool NA, bool NB2>
int launch_c2t8(const C2Args& A, hipStream_t s) {
    constexpr size_t lds_bytes = (size_t)C2_ABYTES + 2 * C2_BUF;
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_t8_kernel<TWO, NA, NB2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    }
    int wgs = 512;
    if (wgs > A.tiles) wgs = (A.tiles + 7) / 8 * 8;
    hipLaunchKernelGGL((conv2d_t8_kernel<TWO, NA, NB2>), dim3(wgs), dim3(C2_THREADS), lds_bytes, s, A);
    return check_launch("conv2d_t8");
}

}  // namespace

// full-width form: both-source-a-normalised launches on rows that fit the LDS (PDS_CONV2D_T8W=0: tiles of 16 x 32)
bool c2t8_wide(const ConvLayer& L) {
    static const bool on = []() {
        const char* e = debug_switch("PDS_CONV2D_T8W");
        return !(e && e[0] == '0');
    }();
    if (!on) return false;
    if (!L.a.scale || (L.b.p && L.b.scale)) return false;            // (source a normalised, source b plain or absent)
    return (L.in.w & 3) == 0 && L.in.w >= 64 && L.in.w <= C2W_MAXW && L.in.h >= 8;
}

template <bool TWO, int NBH, int TY, int ITEMS>
int launch_c2t8w(C2Args& A, hipStream_t s) {
    A.tiles_x = 1;
    A.tiles_y = (A.H + TY - 1) / TY;
    A.tiles = A.N * A.D * A.tiles_y;
    // (+ two halo rows and 256 bytes of slack: column blocks past the end of the last row -- and, with 6-row tiles, the
    // rows of the non-existent fourth row pair -- are read, never used)
    const size_t lds_bytes = (size_t)C2_ABYTES + 2 * 2 * (size_t)(TY + 2) * (A.W + 4) * 8 + 2 * (size_t)(A.W + 4) * 8 + 256;
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    static int cus[32] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_t8w_kernel<TWO, NBH, TY, ITEMS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        int n = 0;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        cus[dev & 31] = n > 0 ? n : 256;
    }
    int wgs = cus[dev & 31] / 8 * 8;
    if (wgs > A.tiles) wgs = (A.tiles + 7) / 8 * 8;
    const int probe = probe_before("conv2d_t8w", s);
    hipLaunchKernelGGL((conv2d_t8w_kernel<TWO, NBH, TY, ITEMS>), dim3(wgs), dim3(C2W_THREADS), lds_bytes, s, A);
    probe_after(probe, A.tiles, s);
    return check_launch("conv2d_t8w");
}

// tile height of the full-width form: 6 rows where that brings the staging items of a tile (halo rows x quads of columns)
// down to one per thread (W <= 256), else 8 (PDS_CONV2D_T8W_ROWS=8 | 6 forces one, A/B)
int c2t8w_rows(int w) {
    static const int forced = []() {
        const char* e = debug_switch("PDS_CONV2D_T8W_ROWS");
        return e ? atoi(e) : 0;
    }();
    if (forced == 8 || forced == 6) return forced;
    return 8 * (w >> 2) <= C2W_THREADS ? 6 : 8;
}

// the bare 64 -> 8 convolution (no statistics wanted), plain Matching layer without the layer-0 riders
bool conv2d_t8_supported(const ConvLayer& L) {
    if (!c2t8_enabled()) return false;
    if (L.kd != 1 || L.stride != 1 || L.in.c != C2_CIN || L.out_g.c != C2_COUT) return false;
    if (L.l0A || L.side_out || L.plane_weight_sets > 0) return false;   // (the caller checks that no statistics are wanted)
    // the fp16-split arithmetic scales its operands by the sources' range certificates (Src::bound)
    if (!L.a.bounded) return false;
    if (L.b.p && !L.b.bounded) return false;
    if (L.b.p && L.b.bcast_d) return false;
    if ((size_t)L.in.n * C2_CIN * L.in.d * L.in.h * L.in.w >= ((size_t)1 << 29)) return false;   // 31-bit byte offsets
    const long tiles = (long)L.in.n * L.in.d * ((L.in.h + C2_TY - 1) / C2_TY) * ((L.in.w + C2_TX - 1) / C2_TX);
    return tiles < (1L << 30);
}

int launch_conv2d_t8(const ConvLayer& L, hipStream_t s) {
    C2Args A;
    A.a = L.a;
    A.b = L.b;
    A.w = L.weight;
    A.bias = L.bias;
    A.out = L.out;
    A.N = L.in.n;
    A.D = L.in.d;
    A.H = L.in.h;
    A.W = L.in.w;
    A.lrelu = L.lrelu;
    A.tiles_x = (A.W + C2_TX - 1) / C2_TX;
    A.tiles_y = (A.H + C2_TY - 1) / C2_TY;
    A.tiles = A.N * A.D * A.tiles_y * A.tiles_x;
    if (!A.a.bound || A.a.bound_n <= 0 || (L.b.p && (!A.b.bound || A.b.bound_n <= 0)))
        return set_error(-1, "conv2d_t8: a source without a range bound");
    if (c2t8_wide(L)) {
        const int blocks = (A.W + 15) / 16, nbh = (blocks + 1) / 2;
        const int rows = c2t8w_rows(A.W), items = ((rows + 2) * (A.W >> 2) + C2W_THREADS - 1) / C2W_THREADS;
        if (nbh <= 8 && rows == 6 && items == 1)
            return L.b.p ? launch_c2t8w<true, 8, 6, 1>(A, s) : launch_c2t8w<false, 8, 6, 1>(A, s);
        if (nbh <= 8) return L.b.p ? launch_c2t8w<true, 8, 8, 2>(A, s) : launch_c2t8w<false, 8, 8, 2>(A, s);
        return L.b.p ? launch_c2t8w<true, 11, 8, 2>(A, s) : launch_c2t8w<false, 11, 8, 2>(A, s);
    }
    const bool na = L.a.scale != nullptr, nb2 = L.b.p && L.b.scale;
    if (L.b.p) {
        if (na) return nb2 ? launch_c2t8<true, true, true>(A, s) : launch_c2t8<true, true, false>(A, s);
        return nb2 ? launch_c2t8<true, false, true>(A, s) : launch_c2t8<true, false, false>(A, s);
    }
    return na ? launch_c2t8<false, true, false>(A, s) : launch_c2t8<false, false, false>(A, s);
}

}  // namespace pds
