// 3x3 convolution 64 -> 8 channels over a stack of 2-D planes, written straight into the stacked signature layout:
// the last convolution of MatchingOperation for all disparity planes at once (reference
// practical_deep_stereo/matching.py:89-93 and the stack of :63; its input is the residual sum of
// network_blocks.py:143-144, formed here from the two sources while staging).
//
// With 8 output channels the generic kernel (conv2d_mfma.hip) idles half of the 16 rows of v_mfma_f32_16x16x4_f32
// and synchronises every 45 MFMAs.  Here (as in conv3d_t8.hip, along y instead of z) the M side is
// (output channel, parity of the output row): one MFMA makes 8 channels x 2 consecutive rows y, y+1 for 16 pixels and
// its K = 4 is the four input rows y-1 .. y+2 they touch:
//     A[(oc, py)][yi] = W[oc][ic][dy = yi - py][dx]   (0 when dy is outside 0..2)
//     B[yi][n]        = in[ic][y - 1 + yi][x + n + dx - 1]
// 192 MFMAs per (8 channels x 2 rows x 16 pixels) instead of 288: 2/3 of the matrix work, exact fp32.
//
//   workgroup   4 waves, PERSISTENT (2 per CU), static lists of (plane, tile) in contiguous runs per XCD.  Tile:
//               16 rows x 32 columns of one (batch, plane); wave w owns rows 4w .. 4w+3: two row pairs x two column
//               blocks = four accumulators, so every A fragment read from LDS feeds four MFMAs.
//   weights     all 192 A fragments live in LDS (48 KB), gathered once per workgroup straight from the
//               [8, 64, 3, 3] tensor: no packing launch.
//   pipeline    the 64 input channels stream through two LDS buffers in chunks of 4; chunk g + 2 is being loaded
//               (buffer loads, zero padding from the range check), chunk g + 1 is normalised / summed / written
//               in the shadow of the MFMAs of chunk g (sched_group_barrier interleave); one barrier per chunk, the
//               stream runs on across tile boundaries.  Row stride 48 == 16 (mod 32): conflict-free B reads.
//   epilogue    accumulators start at the bias; optional LeakyReLU; 64-byte row segments.
#include <atomic>
#include <type_traits>

#include "common.hpp"

namespace pds {

namespace {

constexpr int C2_THREADS = 256;
constexpr int C2_CIN = 64, C2_COUT = 8, C2_KC = 4, C2_CHUNKS = C2_CIN / C2_KC;
constexpr int C2_TY = 16, C2_NB = 2, C2_TX = 16 * C2_NB, C2_RP = C2_TY / 8;   // row pairs per wave
constexpr int C2_YT = C2_TY + 2, C2_XT = C2_TX + 2, C2_RS = 48, C2_CS = C2_YT * C2_RS;
constexpr int C2_BUF = C2_KC * C2_CS;                      // floats per input buffer
constexpr int C2_AFRAGS = C2_CIN * 3;                      // (ic, dx)
constexpr int C2_NPOS = C2_YT * C2_XT;                     // halo positions per channel
constexpr int C2_TPC = C2_THREADS / C2_KC;                 // threads sharing one channel of the chunk
constexpr int C2_POS = (C2_NPOS + C2_TPC - 1) / C2_TPC;    // positions per thread
static_assert(C2_RS % 32 == 16 && C2_RS >= C2_XT, "bank layout");
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct C2Args {
    Src a, b;
    const float* __restrict__ w;     // [8][64][3][3]
    const float* __restrict__ bias;  // [8]
    float* __restrict__ out;         // [N][8][D][H][W]
    int N, D, H, W;
    int lrelu;
    int tiles_x, tiles_y, tiles;     // tiles = N * D * tiles_y * tiles_x
};

}  // namespace

// TWO: second source present (the residual sum); NA / NB2: source a / b carries a deferred InstanceNorm
template <bool TWO, bool NA, bool NB2>
__global__ __launch_bounds__(C2_THREADS, 2) void conv2d_t8_kernel(const C2Args A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* abuf = lds;                                   // [192][64] A fragments
    float* ibuf = lds + C2_AFRAGS * 64;                  // [2][4 ch][18 rows][48]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, q = lane >> 4;
    const size_t plane = (size_t)A.H * A.W;
    const size_t cstride = (size_t)A.D * plane;
    const int plane_bytes = (int)(plane * sizeof(float));
    const int cbytes = (int)(cstride * sizeof(float));
    // one resource per tensor for the whole launch: batch, channel chunk and plane enter as the scalar offset
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A.a.p), 0, (int)((size_t)A.N * C2_CIN * cstride * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(TWO ? A.b.p : A.a.p), 0, (int)((size_t)A.N * C2_CIN * cstride * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        A.out, 0, (int)((size_t)A.N * C2_COUT * cstride * sizeof(float)), 0x00020000);

    // ---- A fragments -> LDS: lane (m = lane & 15 -> oc = m >> 1, py = m & 1 ; k = lane >> 4 = input row yi) --------
    for (int e = tid; e < C2_AFRAGS * 64; e += C2_THREADS) {
        const int l = e & 63, f = e >> 6;                // fragment f = ic * 3 + dx
        const int m = l & 15, oc = m >> 1, py = m & 1, yi = l >> 4, dy = yi - py;
        const int ic = f / 3, dx = f % 3;
        abuf[e] = (dy >= 0 && dy <= 2) ? A.w[((size_t)oc * C2_CIN + ic) * 9 + dy * 3 + dx] : 0.f;
    }

    // ---- this workgroup's tiles: XCD x walks the x-th contiguous eighth of the (batch, plane, row, column) list ----
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int t_end = (int)(((long long)(xcd + 1) * A.tiles) >> 3);
    const int t_first = (int)(((long long)xcd * A.tiles) >> 3) + slot;
    const int my_tiles = t_first < t_end ? (t_end - t_first + per_xcd - 1) / per_xcd : 0;
    const int total_chunks = my_tiles * C2_CHUNKS;

    // staging role of this thread: channel tid / 64 of the chunk (a whole wave per channel), halo positions
    // (tid % 64) + 64 k
    const int sch = tid / C2_TPC, sp0 = tid % C2_TPC;
    int lo[C2_POS];
#pragma unroll
    for (int k = 0; k < C2_POS; ++k) {
        const int p = min(sp0 + C2_TPC * k, C2_NPOS - 1);    // (surplus threads re-stage the last position)
        lo[k] = sch * C2_CS + (p / C2_XT) * C2_RS + p % C2_XT;
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
            const int p = min(sp0 + C2_TPC * k, C2_NPOS - 1);
            const int y = P.y0 + p / C2_XT - 1, x = P.x0 + p % C2_XT - 1;
            const bool ok = (unsigned)y < (unsigned)A.H && (unsigned)x < (unsigned)A.W;
            in[k] = ok ? 1.f : 0.f;
            off[k] = ok ? (unsigned)((size_t)sch * cstride + (size_t)y * A.W + x) * 4u : ~0u;
        }
        sb = (int)(((size_t)P.nb * C2_CIN * A.D + P.d) * plane * sizeof(float));
        gi = (NA && A.a.per_plane) ? P.nb * C2_CIN * A.D + P.d : P.nb * C2_CIN;               // source a
        gc = (TWO && NB2 && A.b.per_plane) ? P.nb * C2_CIN * A.D + P.d : P.nb * C2_CIN;       // source b
    };

    // four register sets: while chunk g is multiplied, chunk g + 1 is written to LDS and chunks g + 2 .. g + 4 are in
    // flight (some 15 MB of loads across the chip: what it takes to keep HBM busy at its latency; two sets in flight
    // measured 334 us, three 306)
    constexpr int DEPTH = 4;
    float va[DEPTH][C2_POS], vb[DEPTH][C2_POS], vs[DEPTH], vh[DEPTH], vs2[DEPTH], vh2[DEPTH];
    float inside_regs[DEPTH][C2_POS];   // padding factors of the chunk held by each set
    auto fetch = [&](int g, auto set_c) {   // global -> registers of set SET; g already clamped
        constexpr int SET = decltype(set_c)::value;
        const int c0 = (g % C2_CHUNKS) * C2_KC;
        // (readfirstlane: the value IS wave-uniform, but the compiler cannot see it through the selects and would
        // wrap every buffer load in a waterfall loop)
        const int soff = __builtin_amdgcn_readfirstlane(sbase + c0 * cbytes);
#pragma unroll
        for (int k = 0; k < C2_POS; ++k) {
            va[SET][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, goff[k], soff, 0));
            if (TWO) vb[SET][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, goff[k], soff, 0));
            inside_regs[SET][k] = inside[k];
        }
        // deferred InstanceNorm of the producers: per (channel, plane) in Matching (per_plane), else per channel
        const int ch = c0 + sch;
        if (NA) {
            const int gi = gidx + ch * stride_a;
            vs[SET] = A.a.scale[gi];
            vh[SET] = A.a.shift[gi];
        }
        if (TWO && NB2) {
            const int gi = gch + ch * stride_b;
            vs2[SET] = A.b.scale[gi];
            vh2[SET] = A.b.shift[gi];
        }
    };
    auto stash = [&](float* buf, auto set_c) {   // registers -> LDS: producers' normalisation, the sum, the zero padding
        constexpr int SET = decltype(set_c)::value;
        const float hsum = (NA ? vh[SET] : 0.f) + ((TWO && NB2) ? vh2[SET] : 0.f);   // both shifts vanish in the padding
#pragma unroll
        for (int k = 0; k < C2_POS; ++k) {
            float v = NA ? vs[SET] * va[SET][k] : va[SET][k];
            if (TWO) v = NB2 ? fmaf(vs2[SET], vb[SET][k], v) : v + vb[SET][k];
            if (NA || (TWO && NB2)) v = fmaf(hsum, inside_regs[SET][k], v);
            buf[lo[k]] = v;
        }
    };

    const float bias0 = A.bias ? A.bias[2 * q] : 0.f, bias1 = A.bias ? A.bias[2 * q + 1] : 0.f;
    const int b_base = (2 * C2_RP * wave + q) * C2_RS + n16;     // halo row of the wave's first pair + yi, column n
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
        const float* buf = ibuf + (g & 1) * C2_BUF + b_base;
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
        const float* af = abuf + (size_t)c * C2_KC * 3 * 64 + lane;
#pragma unroll
        for (int ic = 0; ic < C2_KC; ++ic)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float a = af[(ic * 3 + dx) * 64];
#pragma unroll
                for (int p = 0; p < C2_RP; ++p)
#pragma unroll
                    for (int j = 0; j < C2_NB; ++j) {
                        const float b = buf[ic * C2_CS + 2 * p * C2_RS + 16 * j + dx];
                        acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[p][j], 0, 0, 0);
                    }
            }
#pragma unroll
        for (int i = 0; i < C2_KC * 3 * C2_RP * C2_NB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
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
                        float t = acc[p][j][r];
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
namespace {

bool c2t8_enabled() {
    static const bool on = []() {  // PDS_CONV2D_T8=0: conv2d_mfma.hip serves the layer (A/B)
        const char* e = getenv("PDS_CONV2D_T8");
        return !(e && e[0] == '0');
    }();
    return on;
}

template <bool TWO, bool NA, bool NB2>
int launch_c2t8(const C2Args& A, hipStream_t s) {
    constexpr size_t lds_bytes = (size_t)(C2_AFRAGS * 64 + 2 * C2_BUF) * sizeof(float);
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_t8_kernel<TWO, NA, NB2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    }
    int wgs = 512;
    if (wgs > A.tiles) wgs = (A.tiles + 7) / 8 * 8;
    hipLaunchKernelGGL((conv2d_t8_kernel<TWO, NA, NB2>), dim3(wgs), dim3(C2_THREADS), lds_bytes, s, A);
    return check_launch("conv2d_t8");
}

}  // namespace

// the bare 64 -> 8 convolution (no statistics wanted), plain Matching layer without the layer-0 riders
bool conv2d_t8_supported(const ConvLayer& L) {
    if (!c2t8_enabled()) return false;
    if (L.kd != 1 || L.stride != 1 || L.in.c != C2_CIN || L.out_g.c != C2_COUT) return false;
    if (L.l0A || L.side_out || L.plane_weight_sets > 0) return false;   // (the caller checks that no statistics are wanted)
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
    const bool na = L.a.scale != nullptr, nb2 = L.b.p && L.b.scale;
    if (L.b.p) {
        if (na) return nb2 ? launch_c2t8<true, true, true>(A, s) : launch_c2t8<true, true, false>(A, s);
        return nb2 ? launch_c2t8<true, false, true>(A, s) : launch_c2t8<true, false, false>(A, s);
    }
    return na ? launch_c2t8<false, true, false>(A, s) : launch_c2t8<false, false, false>(A, s);
}

}  // namespace pds
