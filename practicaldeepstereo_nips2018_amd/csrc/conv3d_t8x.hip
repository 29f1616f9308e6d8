// 3x3x3 convolution, stride 1, pad 1, 8 -> 8 channels on the 16-bit matrix pipe with split fp32 operands: the two
// full-resolution smoothing layers of the hourglass (reference practical_deep_stereo/regularization.py:77-78 and the last
// expansion block :51-52,56-57; network_blocks.py:61-72,106-112).  Round 4 successor of conv3d_t8.hip, whose launch time
// (78 / 89 us) was mostly matrix time on the fp32 MFMA (63 us of v_mfma_f32_16x16x4_f32 at 1/16 of the 16-bit rate).
//
// Same Toeplitz-along-z mapping as conv3d_t8.hip -- the M side is (output channel, parity of the output plane), one MFMA
// produces 8 channels x 2 consecutive output planes for 16 pixels -- but on v_mfma_f32_16x16x16_{f16,bf16}, whose K = 16 is
// FOUR input channels x the four input planes z-1 .. z+2 the two output planes touch:
//     A[(oc, pz)][(ic, zi)] = W[oc][ic][dz = zi - pz][dy][dx]   (0 when dz is outside 0..2)
//     B[(ic, zi)][n]        = in[ic][z - 1 + zi][y + dy - 1][x + n + dx - 1]
// i.e. 18 MFMAs x (3 or 6 partial products) per (8 channels x 2 planes x 16 pixels) instead of 72 fp32 MFMAs at a
// quarter of the rate: a fifth (P = 2) / two fifths (P = 3) of the matrix time.  Operand splitting as in conv2d_x3.hip:
//   P = 2  fp16 hi / lo (22 significand bits, three products), operands pre-scaled by powers of two derived from the
//          data: weights by max|w| (reduced by every workgroup while it gathers its fragments), activations by the range
//          certificate of the source(s) (common.hpp Src::bound); used when every source carries one;
//   P = 3  bf16 x 3 (exact truncation split, six products), range-safe: sources without a certificate (the matching
//          signatures handed over by the caller).
//
//   workgroup   4 waves, PERSISTENT: 2 workgroups per CU, static tile lists in contiguous runs per XCD (as conv3d_t8).
//   tile        2 output planes x 4 rows x 16*NB columns; wave w owns row w: NB accumulators.
//   LDS         [buffer 2][part P][8 ch][6 rows][16*NB + 2][4 planes] 16-bit: a lane's B fragment -- the four planes of
//               one (channel, row, column) -- is one aligned 8-byte slot; 16 consecutive columns cover all 32 banks.
//   weights     18 x P A fragments per lane, gathered, scaled and split ONCE per workgroup from the PyTorch-layout
//               tensor (no packing launch), kept in registers.
//   pipeline    next tile: channels 0-3 are requested before the first half of the MFMAs (channels 0-3 of this tile),
//               converted and written to the other buffer after it while channels 4-7 are requested, and so on; the
//               second workgroup of the CU covers what is not hidden.
//   epilogue    1 / (ws as), bias, LeakyReLU, 64-byte row segments, per-channel statistics in fp64 across the workgroup's
//               tiles: ONE deterministic record per (workgroup, channel) -- identical to conv3d_t8.hip.
#include <atomic>

#include "common.hpp"

namespace pds {

namespace {

constexpr int TX_THREADS = 256;
constexpr int TX_C = 8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct TXArgs {
    Src a, b;
    const float* __restrict__ w;     // [8][8][3][3][3]
    const float* __restrict__ bias;  // [8]
    float* __restrict__ out;
    double* __restrict__ partials;   // [(n, oc)][records][2]
    int D, H, W;
    int lrelu;
    int tiles_x, tiles_y, tiles;     // per batch element
    int records;                     // persistent workgroups per batch element (= gridDim.x)
};

template <int NB, int P>
struct TXCfg {
    static constexpr int XT = 16 * NB + 2, YT = 6;
    static constexpr int YX = YT * XT;                        // (row, column) slots of one channel
    static constexpr int PART = TX_C * YX * 8;                // bytes of one split part of a buffer
    static constexpr int BUF = P * PART;
    static constexpr int SLOTS_HALF = 4 * YX;                 // slots of four channels
    static constexpr int SPT = (SLOTS_HALF + TX_THREADS - 1) / TX_THREADS;   // slots per thread and half
    static constexpr int PRODUCTS = P == 3 ? 6 : 3;
};

// split of four fp32 values into P 16-bit parts, each part the 8 bytes of an MFMA operand
template <int P>
__device__ __forceinline__ void tx_split(const float (&v)[4], u32x2 (&part)[P]) {
    if constexpr (P == 2) {
        pds_u32x2 h, l;   // (packed conversions, common.hpp)
        split_quad_f16(v, h, l);
        part[0] = __builtin_bit_cast(u32x2, h);
        part[1] = __builtin_bit_cast(u32x2, l);
    } else {
        // truncation split: every part is the top 16 bits of what is left, the remainder is exact
        unsigned short h[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float r = v[i];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned u = __builtin_bit_cast(unsigned, r);
                h[q][i] = (unsigned short)(u >> 16);
                if (q < 2) r -= __builtin_bit_cast(float, u & 0xffff0000u);
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
            part[q] = u32x2{(unsigned)h[q][0] | ((unsigned)h[q][1] << 16), (unsigned)h[q][2] | ((unsigned)h[q][3] << 16)};
    }
}

template <int P>
__device__ __forceinline__ f32x4 tx_mma(const u32x2& a, const u32x2& b, const f32x4& c) {
    if constexpr (P == 3)
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
}

__device__ __forceinline__ float tx_uniform(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

}  // namespace

// SRC: 0 = one plain source, 1 = one source with a deferred InstanceNorm, 2 = two sources (each plain or deferred).
// EXACT: D, H, W are multiples of the tile (2, 4, 16 * NB): the epilogue needs no masks.
template <int NB, int P, int SRC, bool EXACT>
__global__ __launch_bounds__(TX_THREADS, 2) void conv3d_t8x_kernel(const TXArgs A) {
    using C = TXCfg<NB, P>;
    constexpr bool TWO = SRC == 2;
    constexpr bool NORM = SRC != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = blockIdx.y;
    const size_t plane = (size_t)A.H * A.W;
    const size_t cstride = (size_t)A.D * plane;
    const size_t cstride_b = (TWO && A.b.bcast_d) ? plane : cstride;
    // buffer resources: an offset of ~0 reads as 0.0f / drops the store (hardware range check)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A.a.p + (size_t)nb * TX_C * cstride), 0, (int)(TX_C * cstride * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(TWO ? A.b.p + (size_t)nb * TX_C * cstride_b : A.a.p), 0,
        (int)(TX_C * (TWO ? cstride_b : cstride) * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        A.out + (size_t)nb * TX_C * cstride, 0, (int)(TX_C * cstride * sizeof(float)), 0x00020000);
    const int cbytes = (int)(cstride * sizeof(float)), cbytes_b = (int)(cstride_b * sizeof(float));
    const int pbytes = (int)(plane * sizeof(float));

    // ---- operand scales (P = 2): powers of two from max|w| and from the sources' range certificates -----------------
    float ws = 1.f, as = 1.f;
    if constexpr (P == 2) {
        float* red = reinterpret_cast<float*>(lds);
        float wm = 0.f;
        for (int i = tid; i < TX_C * TX_C * 27; i += TX_THREADS) wm = fmaxf(wm, fabsf(A.w[i]));
        wm = block_max(wm, red);
        float bound = block_bound(A.a.bound, A.a.bound_n, red);
        if (TWO) bound += block_bound(A.b.bound, A.b.bound_n, red);
        ws = tx_uniform(pow2_scale(wm, kHalfTarget));
        as = tx_uniform(pow2_scale(bound, kHalfTarget));
    }
    const float unscale = tx_uniform((1.f / ws) * (1.f / as));

    // ---- this workgroup's tiles (as conv3d_t8.hip) -----------------------------------------------------------------
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int t_end = (int)(((long long)(xcd + 1) * A.tiles) >> 3);
    int tile = (int)(((long long)xcd * A.tiles) >> 3) + slot;
    int tx = tile % A.tiles_x, ty = (tile / A.tiles_x) % A.tiles_y, tz = tile / (A.tiles_x * A.tiles_y);
    const int step_x = per_xcd % A.tiles_x, step_y = (per_xcd / A.tiles_x) % A.tiles_y,
              step_z = per_xcd / (A.tiles_x * A.tiles_y);
    auto advance = [&](int& ax, int& ay, int& az) {
        ax += step_x;
        int carry = ax >= A.tiles_x ? 1 : 0;
        ax -= carry ? A.tiles_x : 0;
        ay += step_y + carry;
        carry = ay >= A.tiles_y ? 1 : 0;
        ay -= carry ? A.tiles_y : 0;
        az += step_z + carry;
    };

    // ---- deferred InstanceNorm coefficients of the sources, times the activation scale: a 32-float table in LDS (behind
    //      the two tile buffers) -- 32 live registers would not fit beside the A fragments ----------------------------------
    float* coef = reinterpret_cast<float*>(lds + 2 * C::BUF);   // [scale a | shift a | scale b | shift b][8 channels]
    if (tid < 4 * TX_C) {
        const int c = tid & 7, kind = tid >> 3;
        float v = (kind & 1) ? 0.f : 1.f;
        if (kind < 2 && NORM && A.a.scale) v = (kind ? A.a.shift : A.a.scale)[nb * TX_C + c];
        if (kind >= 2 && TWO && A.b.scale) v = ((kind & 1) ? A.b.shift : A.b.scale)[nb * TX_C + c];
        coef[tid] = v * as;
    }

    // ---- staging: per half (four channels) a thread owns up to SPT (channel, row, column) slots = 4 planes each -----
    int s_slot[C::SPT];
#pragma unroll
    for (int k = 0; k < C::SPT; ++k) s_slot[k] = min(tid + k * TX_THREADS, C::SLOTS_HALF - 1);   // (surplus threads repeat the last)
    unsigned goff[C::SPT];        // byte offset of (channel of the half, plane z0 - 1, row, column); ~0: row / column outside
    unsigned zmask[C::SPT];       // bit zi: plane z0 - 1 + zi is inside the volume (and the row / column are)
    int zbase = 0;
    auto prepare = [&](int ax, int ay, int az) {
        const int z0 = az * 2, y0 = ay * 4, x0 = ax * 16 * NB;
        zbase = z0 - 1;
#pragma unroll
        for (int k = 0; k < C::SPT; ++k) {
            const int c = s_slot[k] / C::YX, yx = s_slot[k] - c * C::YX;
            const int y = y0 + yx / C::XT - 1, x = x0 + yx % C::XT - 1;
            const bool in = (unsigned)y < (unsigned)A.H && (unsigned)x < (unsigned)A.W;
            unsigned m = 0;
#pragma unroll
            for (int zi = 0; zi < 4; ++zi) m |= (in && (unsigned)(zbase + zi) < (unsigned)A.D) ? (1u << zi) : 0u;
            zmask[k] = m;
            goff[k] = in ? (unsigned)(y * A.W + x) * 4u : ~0u;
        }
    };
    float va[C::SPT][4], vb[TWO ? C::SPT : 1][4];
    auto fetch_half = [&](int half) {
#pragma unroll
        for (int k = 0; k < C::SPT; ++k) {
            const int c = s_slot[k] / C::YX;
#pragma unroll
            for (int zi = 0; zi < 4; ++zi) {
                const bool ok = (zmask[k] >> zi) & 1u;
                const unsigned off = ok ? goff[k] + (unsigned)(c * cbytes + (zbase + zi) * pbytes) : ~0u;
                va[k][zi] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, off, (4 * half) * cbytes, 0));
                if (TWO) {
                    const unsigned offb = ok ? goff[k] + (unsigned)(c * cbytes_b + (A.b.bcast_d ? 0 : (zbase + zi) * pbytes))
                                             : ~0u;
                    vb[k][zi] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(rb, offb, (4 * half) * cbytes_b, 0));
                }
            }
        }
    };
    auto stash_half = [&](int half, unsigned char* buf) {
#pragma unroll
        for (int k = 0; k < C::SPT; ++k) {
            const int c = 4 * half + s_slot[k] / C::YX;
            const float cs = coef[c], ch = coef[TX_C + c];
            const float cs2 = TWO ? coef[2 * TX_C + c] : 0.f, ch2 = TWO ? coef[3 * TX_C + c] : 0.f;
            float v[4];
#pragma unroll
            for (int zi = 0; zi < 4; ++zi) {
                float t = NORM ? fmaf(cs, va[k][zi], ch) : va[k][zi] * cs;   // (cs carries the activation scale)
                if (TWO) t += fmaf(cs2, vb[k][zi], ch2);
                v[zi] = ((zmask[k] >> zi) & 1u) ? t : 0.f;                    // literal zero padding
            }
            u32x2 part[P];
            tx_split<P>(v, part);
#pragma unroll
            for (int p = 0; p < P; ++p)
                *reinterpret_cast<u32x2*>(buf + p * C::PART + half * (4 * C::YX * 8) + s_slot[k] * 8) = part[p];
        }
    };

    float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};
    const int n16 = lane & 15, q = lane >> 4;
    const float bias0 = A.bias ? A.bias[2 * q] : 0.f, bias1 = A.bias ? A.bias[2 * q + 1] : 0.f;
    // B fragment of (channel group g, dy, dx, column block j): channel 4 g + q, halo row wave + dy, halo column n + dx + 16 j
    const int b_base = ((q * C::YT + wave) * C::XT + n16) * 8;
    const unsigned out_lane = (unsigned)((size_t)(2 * q) * cstride + n16) * 4u;
    const unsigned out_c1 = (unsigned)cstride * 4u;

    // ---- A fragments: lane (m = lane & 15 -> oc = m >> 1, pz = m & 1 ; kg = lane >> 4 = channel within the group) --
    u32x2 af[2][9][P];
    {
        const int m = lane & 15, oc = m >> 1, pz = m & 1;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float wv[4];
#pragma unroll
                for (int zi = 0; zi < 4; ++zi) {
                    const int dz = zi - pz;
                    const bool valid = dz >= 0 && dz <= 2;
                    const float v = A.w[(((size_t)oc * TX_C + 4 * g + q) * 3 + (valid ? dz : 0)) * 9 + t];
                    wv[zi] = valid ? v * ws : 0.f;
                }
                tx_split<P>(wv, af[g][t]);
            }
    }

    int cur = 0;
    __syncthreads();   // the coefficient table is in place
    if (tile < t_end) {
        prepare(tx, ty, tz);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            fetch_half(half);
            stash_half(half, lds);
        }
    }
    __syncthreads();

    for (; tile < t_end; tile += per_xcd) {
        const int z0 = tz * 2, y0 = ty * 4, x0 = tx * 16 * NB;
        int nx = tx, ny = ty, nz = tz;
        advance(nx, ny, nz);
        const bool more = tile + per_xcd < t_end;
        // (the last tile stages itself once more into the idle buffer: no branch around the loads)
        prepare(more ? nx : tx, more ? ny : ty, more ? nz : tz);
        tx = nx;
        ty = ny;
        tz = nz;
        unsigned char* nxt = lds + (cur ^ 1) * C::BUF;
        const unsigned char* bp = lds + cur * C::BUF + b_base;

        f32x4 acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            fetch_half(g);   // next tile, channels 4 g .. 4 g + 3: lands during the MFMAs below
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                // one kernel column at a time (all three at once cost 24 more registers: spills in the two-source form)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    u32x2 bf[NB][P];
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int p = 0; p < P; ++p)
                            bf[j][p] = *reinterpret_cast<const u32x2*>(
                                bp + p * C::PART + ((g * 4 * C::YT + dy) * C::XT + dx + 16 * j) * 8);
                    // small partial products first; consecutive MFMAs hit different accumulators
#pragma unroll
                    for (int c = 0; c < C::PRODUCTS; ++c) {
                        // (weight part, pixel part): P = 3: (0,2) (2,0) (1,1) (0,1) (1,0) (0,0);  P = 2: (0,1) (1,0) (0,0)
                        const int pa = P == 3 ? (c == 0 ? 0 : c == 1 ? 2 : c == 2 ? 1 : c == 3 ? 0 : c == 4 ? 1 : 0) : (c == 1 ? 1 : 0);
                        const int pb = P == 3 ? (c == 0 ? 2 : c == 1 ? 0 : c == 2 ? 1 : c == 3 ? 1 : c == 4 ? 0 : 0) : (c == 0 ? 1 : 0);
#pragma unroll
                        for (int j = 0; j < NB; ++j) acc[j] = tx_mma<P>(af[g][dy * 3 + dx][pa], bf[j][pb], acc[j]);
                    }
                }
            }
            stash_half(g, nxt);
        }

        // ---- epilogue of this tile: scale back, bias, LeakyReLU, 64-byte row segments, statistics -------------------
        {
            const int y = y0 + wave;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int h = r >> 1, z = z0 + (r & 1);
                const bool rowok = EXACT || (z < A.D && y < A.H);
                const int row_bytes = EXACT ? ((z * A.H + y) * A.W + x0) * (int)sizeof(float)
                                            : ((min(z, A.D - 1) * A.H + min(y, A.H - 1)) * A.W + x0) * (int)sizeof(float);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    float t = fmaf(acc[j][r], unscale, h ? bias1 : bias0);
                    if (A.lrelu) t = fmaxf(t, t * kLeakySlope);
                    const bool ok = EXACT || (rowok && x0 + 16 * j + n16 < A.W);
                    const unsigned off = out_lane + (h ? out_c1 : 0u) + 64u * j;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, t), ro, ok ? off : ~0u, row_bytes, 0);
                    t = ok ? t : 0.f;
                    ssum[h] += t;
                    ssq[h] = fmaf(t, t, ssq[h]);
                }
            }
        }
        __syncthreads();   // the current buffer is free, the other one is complete
        cur ^= 1;
    }

    // ---- one record per (workgroup, channel) ----------------------------------------------------------------------
    if (A.partials) {
        double* red = reinterpret_cast<double*>(lds);   // [256 threads][2 channels][2]
        red[tid * 4 + 0] = (double)ssum[0];
        red[tid * 4 + 1] = (double)ssq[0];
        red[tid * 4 + 2] = (double)ssum[1];
        red[tid * 4 + 3] = (double)ssq[1];
        __syncthreads();
        if (tid < TX_C * 2) {
            const int oc = tid >> 1, k = tid & 1, qq = oc >> 1, hh = oc & 1;
            double sum = 0.0;
            for (int wv = 0; wv < 4; ++wv)
                for (int n = 0; n < 16; ++n) sum += red[(wv * 64 + qq * 16 + n) * 4 + hh * 2 + k];
            A.partials[(((size_t)nb * TX_C + oc) * A.records + blockIdx.x) * 2 + k] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
namespace {

template <int NB, int P, int SRC, bool EXACT>
int launch_t8x(const TXArgs& A, int batch, hipStream_t s) {
    using C = TXCfg<NB, P>;
    constexpr size_t lds_bytes = (size_t)2 * C::BUF + 4 * TX_C * sizeof(float);   // + the coefficient table
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_t8x_kernel<NB, P, SRC, EXACT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    }
    hipLaunchKernelGGL((conv3d_t8x_kernel<NB, P, SRC, EXACT>), dim3(A.records, batch), dim3(TX_THREADS), lds_bytes, s, A);
    return check_launch("conv3d_t8x");
}

}  // namespace

bool conv3d_t8x_enabled() {
    static const bool on = []() {  // PDS_CONV3D_T8X=0: the exact-fp32 kernel of conv3d_t8.hip serves these layers (A/B)
        const char* e = debug_switch("PDS_CONV3D_T8X");
        return !(e && e[0] == '0');
    }();
    return on;
}

// same tiling, records and arguments as launch_conv3d_t8 (conv3d_t8.hip), which calls this when the form is enabled
int launch_conv3d_t8x(const ConvLayer& L, int nb, int tiles_x, int tiles_y, int tiles, int records, hipStream_t s) {
    TXArgs A;
    A.a = L.a;
    A.b = L.b;
    A.w = L.weight;
    A.bias = L.bias;
    A.out = L.out;
    A.partials = L.partials;
    A.D = L.in.d;
    A.H = L.in.h;
    A.W = L.in.w;
    A.lrelu = L.lrelu;
    A.tiles_x = tiles_x;
    A.tiles_y = tiles_y;
    A.tiles = tiles;
    A.records = records;
    const int src = L.b.p != nullptr ? 2 : (L.a.scale != nullptr ? 1 : 0);
    const bool exact = A.D % 2 == 0 && A.H % 4 == 0 && A.W % (16 * nb) == 0;
    // fp16 form when every source carries a range certificate, else the range-safe bf16 form
    const bool fp16 = L.a.bound && L.a.bound_n > 0 && (!L.b.p || (L.b.bound && L.b.bound_n > 0));
    if (!fp16 && src == 2) return set_error(-1, "conv3d_t8x: two sources need range certificates");
#define PDS_T8X_CASE(NB_, SRC_)                                                                                  \
    if (nb == NB_ && src == SRC_) {                                                                              \
        if (fp16) return exact ? launch_t8x<NB_, 2, SRC_, true>(A, L.in.n, s) : launch_t8x<NB_, 2, SRC_, false>(A, L.in.n, s); \
        if constexpr (SRC_ != 2)                                                                                 \
            return exact ? launch_t8x<NB_, 3, SRC_, true>(A, L.in.n, s) : launch_t8x<NB_, 3, SRC_, false>(A, L.in.n, s);       \
    }
    PDS_T8X_CASE(2, 0)
    PDS_T8X_CASE(2, 1)
    PDS_T8X_CASE(2, 2)
    PDS_T8X_CASE(3, 0)
    PDS_T8X_CASE(3, 1)
    PDS_T8X_CASE(3, 2)
#undef PDS_T8X_CASE
    return set_error(-1, "conv3d_t8x: no configuration");
}

}  // namespace pds
