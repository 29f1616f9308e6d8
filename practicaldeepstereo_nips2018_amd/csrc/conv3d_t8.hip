// 3x3x3 convolution, stride 1, pad 1, 8 -> 8 channels on the fp32 MFMA units with NO idle half tile: the two
// full-resolution smoothing layers of the hourglass (reference practical_deep_stereo/regularization.py:77-78 and the
// last expansion block :51-52,56-57; network_blocks.py:61-72,106-112).
//
// With 8 output channels the generic kernel (conv3d_mfma.hip) fills only half of the 16 rows of
// v_mfma_f32_16x16x4_f32.  Here the M side is (output channel, parity of the output plane): one MFMA produces
// 8 channels x 2 consecutive output planes z, z+1 for 16 pixels, and its K = 4 is exactly the four input planes
// z-1 .. z+2 those two outputs touch ("Toeplitz along z"):
//     A[(oc, pz)][zi] = W[oc][ic][dz = zi - pz][dy][dx]   (0 when dz is outside 0..2)
//     B[zi][n]        = in[ic][z - 1 + zi][y + dy - 1][x + n + dx - 1]
// 12 of the 16 A entries of a column pair are non-zero: 72 MFMAs per (8 channels x 2 planes x 16 pixels) instead of
// 108, i.e. 2/3 of the generic kernel's matrix work, exact fp32 (an fmaf chain, like every other kernel here).
//
//   workgroup   4 waves, PERSISTENT: grid = 2 workgroups per CU, each walks a static list of tiles (contiguous runs
//               per XCD, so the halo planes neighbouring tiles share stay in one L2).
//   tile        2 output planes x 4 rows x 16*NB columns; wave w owns row w: NB accumulators.
//   weights     72 A fragments per lane, gathered ONCE per workgroup straight from the PyTorch-layout tensor
//               (no packing launch) and kept in registers.
//   LDS         one halo tile [8 ch][4 planes][6 rows][16*NB + 2], plane stride == 16 (mod 32) floats so the
//               two k-halves of a 32-lane ds_read_b32 group hit disjoint banks.
//   pipeline    the global loads of tile t+1 are issued before the MFMA loop of tile t and parked in registers;
//               after the loop: barrier, write them (deferred InstanceNorm of the producer(s), skip sum, literal
//               zero padding applied here) to LDS, barrier.  Global latency never shows; the LDS write phase of one
//               workgroup overlaps the MFMA phase of the other workgroup on the CU.
//   epilogue    bias, LeakyReLU(0.1), 64-byte row segments stored, per-channel sum / sum of squares accumulated in
//               fp64 across the workgroup's tiles: ONE deterministic record per (workgroup, channel).
#include <atomic>

#include "common.hpp"

namespace pds {

namespace {

constexpr int T8_THREADS = 256;
constexpr int T8_C = 8;  // input and output channels
typedef float f32x4 __attribute__((ext_vector_type(4)));

// timing-decomposition hooks (tools/build_variant_one.sh): any of them set makes the results wrong
#ifdef PDS_T8_NOFETCH
#define PDS_T8_LOAD(x) (float)(tid)
#else
#define PDS_T8_LOAD(x) (x)
#endif
#ifdef PDS_T8_NOBREAD
#define PDS_T8_BREAD(x) (float)(lane + g)
#else
#define PDS_T8_BREAD(x) (x)
#endif
#ifdef PDS_T8_NOMFMA
#define PDS_T8_MFMA(c, a, b) (c)[0] += (a) * (b)
#else
#define PDS_T8_MFMA(c, a, b) (c) = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif

struct T8Args {
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

template <int NB>
struct T8Cfg {
    static constexpr int XT = 16 * NB + 2, YT = 6, ZT = 4;
    static constexpr int PLANE_RAW = YT * XT;
    static constexpr int PS = (PLANE_RAW + 15) / 32 * 32 + 16;   // == 16 (mod 32), >= PLANE_RAW
    static constexpr int CS = ZT * PS;
    static constexpr int LDS_FLOATS = T8_C * CS;                  // one buffer; the kernel allocates two
    static constexpr int NPOS = ZT * YT * XT;
    static constexpr int POS = (NPOS + T8_THREADS - 1) / T8_THREADS;
    static_assert(PS >= PLANE_RAW && PS % 32 == 16, "bad plane padding");
};

__device__ __forceinline__ float t8_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

}  // namespace

// SRC: 0 = one plain source, 1 = one source with a deferred InstanceNorm, 2 = two sources (each plain or deferred).
// EXACT: D, H, W are multiples of the tile (2, 4, 16 * NB): the epilogue needs no masks.
template <int NB, int SRC, bool EXACT>
__global__ __launch_bounds__(T8_THREADS, 2) void conv3d_t8_kernel(const T8Args A) {
    using C = T8Cfg<NB>;
    constexpr bool TWO = SRC == 2;
    constexpr bool NORM = SRC != 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = blockIdx.y;
    const size_t plane = (size_t)A.H * A.W;
    const size_t cstride = (size_t)A.D * plane;
    const size_t cstride_b = (TWO && A.b.bcast_d) ? plane : cstride;
    // buffer resources (wave-uniform base + 32-bit lane offset + scalar channel offset): no 64-bit vector address
    // math, and an offset of ~0 reads as 0.0f / drops the store (hardware range check) -- that is the zero padding
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A.a.p + (size_t)nb * T8_C * cstride), 0, (int)(T8_C * cstride * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(TWO ? A.b.p + (size_t)nb * T8_C * cstride_b : A.a.p), 0,
        (int)(T8_C * (TWO ? cstride_b : cstride) * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        A.out + (size_t)nb * T8_C * cstride, 0, (int)(T8_C * cstride * sizeof(float)), 0x00020000);
    const int cbytes = (int)(cstride * sizeof(float)), cbytes_b = (int)(cstride_b * sizeof(float));

    // ---- this workgroup's tiles: XCD x gets the x-th contiguous eighth of the tile list; tile coordinates advance
    //      incrementally (scalar adds with carry instead of divisions) ------------------------------------------------
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

    // ---- deferred InstanceNorm coefficients of the sources (uniform per channel) ---------------------------------
    float sa[T8_C], ha[T8_C], sb[T8_C], hb[T8_C];
#pragma unroll
    for (int c = 0; c < T8_C; ++c) {
        sa[c] = (NORM && A.a.scale) ? A.a.scale[nb * T8_C + c] : 1.f;
        ha[c] = (NORM && A.a.scale) ? A.a.shift[nb * T8_C + c] : 0.f;
        sb[c] = (TWO && A.b.scale) ? A.b.scale[nb * T8_C + c] : 1.f;
        hb[c] = (TWO && A.b.scale) ? A.b.shift[nb * T8_C + c] : 0.f;
    }

    // ---- staging: thread t owns halo positions t, t + 256, ... of every channel ----------------------------------
    // (threads past the last position are clamped onto it: they rewrite the same value)
    int pzz[C::POS], pyy[C::POS], pxx[C::POS], lo[C::POS];
#pragma unroll
    for (int k = 0; k < C::POS; ++k) {
        const int p = min(tid + k * T8_THREADS, C::NPOS - 1);
        pxx[k] = p % C::XT - 1;
        pyy[k] = (p / C::XT) % C::YT - 1;
        pzz[k] = p / (C::XT * C::YT) - 1;
        lo[k] = (pzz[k] + 1) * C::PS + (pyy[k] + 1) * C::XT + pxx[k] + 1;
    }
    unsigned ga[C::POS], gb[C::POS];   // byte offsets inside one channel; ~0 = outside the volume
    unsigned inside_bits = 0;

    auto prepare = [&](int ax, int ay, int az) {   // offsets and padding mask of the tile at (ax, ay, az)
        const int z0 = az * 2, y0 = ay * 4, x0 = ax * 16 * NB;
        inside_bits = 0;
#pragma unroll
        for (int k = 0; k < C::POS; ++k) {
            const int z = z0 + pzz[k], y = y0 + pyy[k], x = x0 + pxx[k];
            const bool in = (unsigned)z < (unsigned)A.D && (unsigned)y < (unsigned)A.H && (unsigned)x < (unsigned)A.W;
            if (NORM) inside_bits |= in ? (1u << k) : 0u;
            ga[k] = in ? (unsigned)((z * A.H + y) * A.W + x) * 4u : ~0u;
            gb[k] = (TWO && A.b.bcast_d) ? (in ? (unsigned)(y * A.W + x) * 4u : ~0u) : ga[k];
        }
    };

    float va[T8_C][C::POS], vb[TWO ? T8_C : 1][C::POS];
    auto fetch_channel = [&](int c) {
#pragma unroll
        for (int k = 0; k < C::POS; ++k)
            va[c][k] = PDS_T8_LOAD(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, ga[k], c * cbytes, 0)));
        if (TWO) {
#pragma unroll
            for (int k = 0; k < C::POS; ++k)
                vb[c][k] = PDS_T8_LOAD(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, gb[k], c * cbytes_b, 0)));
        }
    };
    auto stash_channel = [&](int c, float* buf) {
#pragma unroll
        for (int k = 0; k < C::POS; ++k) {
            float v = va[c][k];   // a plain source: the range check already delivered the padding zeros
            if (NORM) {
                v = fmaf(sa[c], v, ha[c]);
                if (TWO) v += fmaf(sb[c], vb[c][k], hb[c]);
                v = ((inside_bits >> k) & 1u) ? v : 0.f;
            }
            buf[c * C::CS + lo[k]] = v;
        }
    };

    // per-lane statistics of channels 2q, 2q + 1 over all tiles of this workgroup: at most a few dozen values per lane,
    // summed in fp32 and reduced in fp64 at the end
    float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};
    const int n16 = lane & 15, q = lane >> 4;
    const float bias0 = A.bias ? A.bias[2 * q] : 0.f, bias1 = A.bias ? A.bias[2 * q + 1] : 0.f;
    const int b_base = q * C::PS + wave * C::XT + n16;
    // output: scalar byte offset of the row per (tile, plane parity) + this 32-bit lane offset (channel 2q, column n)
    const unsigned out_lane = (unsigned)((size_t)(2 * q) * cstride + n16) * 4u;
    const unsigned out_c1 = (unsigned)cstride * 4u;   // next channel

    int cur = 0;
    if (tile < t_end) {   // first tile: two half-tiles of four channels (register pressure)
        prepare(tx, ty, tz);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int c = 4 * half; c < 4 * half + 4; ++c) fetch_channel(c);
#pragma unroll
            for (int c = 4 * half; c < 4 * half + 4; ++c) stash_channel(c, lds);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- A fragments: lane (m = lane & 15 -> oc = m >> 1, pz = m & 1 ; k = lane >> 4 = input plane zi) ----------
    float af[T8_C * 9];
    {
        const int m = lane & 15, oc = m >> 1, pz = m & 1, zi = lane >> 4;
        const int dz = zi - pz;
        const bool valid = dz >= 0 && dz <= 2;
        const float* wl = A.w + ((size_t)oc * T8_C * 3 + (valid ? dz : 0)) * 9;
#pragma unroll
        for (int ic = 0; ic < T8_C; ++ic)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float v = wl[(size_t)ic * 27 + t];
                af[ic * 9 + t] = valid ? v : 0.f;
            }
    }

    __syncthreads();

    for (; tile < t_end; tile += per_xcd) {
        // the last tile stages itself once more (into the idle buffer) instead of branching around the riders: the
        // MFMA loop stays one basic block, which is what lets the scheduler interleave it
        const int z0 = tz * 2, y0 = ty * 4, x0 = tx * 16 * NB;
        int nx = tx, ny = ty, nz = tz;
        advance(nx, ny, nz);
        const bool more = tile + per_xcd < t_end;
#ifndef PDS_T8_NOPREP
        prepare(more ? nx : tx, more ? ny : ty, more ? nz : tz);
#endif
        tx = nx;
        ty = ny;
        tz = nz;
        float* nxt = lds + (cur ^ 1) * C::LDS_FLOATS;

        // accumulators start at the bias: row r of the D fragment is (channel 2q + (r >> 1), plane parity r & 1)
        f32x4 acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = f32x4{bias0, bias0, bias1, bias1};
        const float* bp = lds + cur * C::LDS_FLOATS + b_base;
        // B operands of one (input channel, dy) group = 3 dx x NB column blocks, read one group ahead of the MFMAs
        // that consume them.  The next tile rides along: channel c is fetched at group 2c and written to the OTHER LDS
        // buffer at group 2c + 8 (864 MFMA cycles later), so at most four channels are parked in registers.
        float bb[2][3 * NB];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int j = 0; j < NB; ++j) bb[0][dx * NB + j] = bp[dx + 16 * j];
#pragma unroll
        for (int g = 0; g < T8_C * 3; ++g) {
            if (g + 1 < T8_C * 3) {
                const float* p = bp + ((g + 1) / 3) * C::CS + ((g + 1) % 3) * C::XT;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int j = 0; j < NB; ++j) bb[(g + 1) & 1][dx * NB + j] = PDS_T8_BREAD(p[dx + 16 * j]);
            }
            if (g % 2 == 0 && g / 2 < T8_C) fetch_channel(g / 2);
#ifndef PDS_T8_NOSTASH
            if (g >= 8 && g % 2 == 0 && (g - 8) / 2 < T8_C) stash_channel((g - 8) / 2, nxt);
#endif
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    PDS_T8_MFMA(acc[j], af[g * 3 + dx], bb[g & 1][dx * NB + j]);
            // emitted order: after every MFMA one LDS read of the next group and a few of the riders (global loads,
            // InstanceNorm VALU, LDS writes of the next tile), so they issue in the shadow of the 32-cycle MFMAs
#pragma unroll
            for (int i = 0; i < 3 * NB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
            }
        }

        // ---- epilogue of this tile: LeakyReLU, 64-byte row segments, statistics ----------------------------------
#ifdef PDS_T8_NOEPI
        if (acc[0][0] == 12345.678f && acc[1][1] == 3.f && acc[NB - 1][2] == 4.f) A.out[tid] = acc[0][3];
#else
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
                    float t = acc[j][r];
                    if (A.lrelu) t = fmaxf(t, t * kLeakySlope);   // slope < 1: max(t, slope * t) is LeakyReLU
                    const bool ok = EXACT || (rowok && x0 + 16 * j + n16 < A.W);
                    const unsigned off = out_lane + (h ? out_c1 : 0u) + 64u * j;
#ifndef PDS_T8_NOSTORE
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, t), ro, ok ? off : ~0u, row_bytes, 0);
#endif
                    t = ok ? t : 0.f;
                    ssum[h] += t;
                    ssq[h] = fmaf(t, t, ssq[h]);
                }
            }
        }
#endif
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
        if (tid < T8_C * 2) {
            // channel oc = 2 q + h lives in the lanes with (lane >> 4) == q of every wave
            const int oc = tid >> 1, k = tid & 1, qq = oc >> 1, hh = oc & 1;
            double sum = 0.0;
            for (int wv = 0; wv < 4; ++wv)
                for (int n = 0; n < 16; ++n) sum += red[(wv * 64 + qq * 16 + n) * 4 + hh * 2 + k];
            A.partials[(((size_t)nb * T8_C + oc) * A.records + blockIdx.x) * 2 + k] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
namespace {

// columns per tile = 16 * NB: the width that pads the row least; ties go to the earlier candidate (48 columns keep
// the staging registers of a two-source layer at 80)
int t8_choose_nb(int w) {
    const int candidates[2] = {3, 2};   // wider tiles would not leave room for two double-buffered workgroups per CU
    int best = 3;
    long best_padded = -1;
    for (int i = 0; i < 2; ++i) {
        const int tx = 16 * candidates[i];
        const long padded = (long)((w + tx - 1) / tx) * tx;
        if (best_padded < 0 || padded < best_padded) {
            best_padded = padded;
            best = candidates[i];
        }
    }
    return best;
}

bool t8_enabled() {
    static const bool on = []() {  // PDS_CONV3D_T8=0: the generic MFMA kernel serves these layers (A/B)
        const char* e = debug_switch("PDS_CONV3D_T8");
        return !(e && e[0] == '0');
    }();
    return on;
}

template <int NB, int SRC, bool EXACT>
int launch_t8(const T8Args& A, int batch, hipStream_t s) {
    using C = T8Cfg<NB>;
    constexpr size_t lds_bytes = (size_t)2 * C::LDS_FLOATS * sizeof(float);
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_t8_kernel<NB, SRC, EXACT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    }
    hipLaunchKernelGGL((conv3d_t8_kernel<NB, SRC, EXACT>), dim3(A.records, batch), dim3(T8_THREADS), lds_bytes, s, A);
    return check_launch("conv3d_t8");
}

}  // namespace

bool conv3d_t8_supported(const ConvLayer& L) {
    if (!t8_enabled()) return false;
    if (L.kd != 3 || L.stride != 1) return false;
    if (L.in.c != T8_C || L.out_g.c != T8_C) return false;
    if (L.stat_per_plane) return false;
    if ((L.a.scale && L.a.per_plane) || (L.b.scale && L.b.per_plane)) return false;
    if ((size_t)L.in.c * L.in.d * L.in.h * L.in.w >= ((size_t)1 << 30)) return false;   // 32-bit byte offsets
    if (L.in.n > 65535) return false;
    return true;
}

static int t8_tiles(const Geom& o, int nb) {
    return ((o.w + 16 * nb - 1) / (16 * nb)) * ((o.h + 3) / 4) * ((o.d + 1) / 2);
}

// persistent workgroups (= partial records) per batch element: two per CU over the whole batch, a multiple of 8
int conv3d_t8_records(const Geom& o) {
    const int tiles = t8_tiles(o, t8_choose_nb(o.w));
    int per_n = 512 / (o.n > 0 ? o.n : 1);
    if (per_n > tiles) per_n = tiles;
    per_n = (per_n + 7) / 8 * 8;
    return per_n < 8 ? 8 : per_n;
}

bool conv3d_t8x_enabled();   // conv3d_t8x.hip: the same layer on the 16-bit matrix pipe (split operands)
int launch_conv3d_t8x(const ConvLayer& L, int nb, int tiles_x, int tiles_y, int tiles, int records, hipStream_t s);

int launch_conv3d_t8(const ConvLayer& L, hipStream_t s) {
    const int nb = t8_choose_nb(L.out_g.w);
    // Only layers whose sources all carry range certificates take the fp16-split kernel (measured at config 2: 79 us
    // against 95-104 for the two-source layer of the last expansion block).  Its range-safe bf16 form (six products, 12
    // conversion instructions per value) measured SLOWER than the exact-fp32 kernel below for the un-certified first
    // layer (the caller's matching signatures): 112 against 85 us -- PDS_CONV3D_T8X=2 forces it for tests.
    const bool certified = L.a.bound && L.a.bound_n > 0 && (!L.b.p || (L.b.bound && L.b.bound_n > 0));
    static const bool force_x = []() {
        const char* e = debug_switch("PDS_CONV3D_T8X");
        return e && e[0] == '2';
    }();
    if (conv3d_t8x_enabled() && (certified || (force_x && !L.b.p)))
        return launch_conv3d_t8x(L, nb, (L.in.w + 16 * nb - 1) / (16 * nb), (L.in.h + 3) / 4, t8_tiles(L.out_g, nb),
                                 conv3d_t8_records(L.out_g), s);
    T8Args A;
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
    A.tiles_x = (A.W + 16 * nb - 1) / (16 * nb);
    A.tiles_y = (A.H + 3) / 4;
    A.tiles = t8_tiles(L.out_g, nb);
    A.records = conv3d_t8_records(L.out_g);
    const int src = L.b.p != nullptr ? 2 : (L.a.scale != nullptr ? 1 : 0);
    const bool exact = A.D % 2 == 0 && A.H % 4 == 0 && A.W % (16 * nb) == 0;
#define PDS_T8_CASE(NB_, SRC_)                                                              \
    if (nb == NB_ && src == SRC_)                                                           \
        return exact ? launch_t8<NB_, SRC_, true>(A, L.in.n, s) : launch_t8<NB_, SRC_, false>(A, L.in.n, s);
    PDS_T8_CASE(2, 0)
    PDS_T8_CASE(2, 1)
    PDS_T8_CASE(2, 2)
    PDS_T8_CASE(3, 0)
    PDS_T8_CASE(3, 1)
    PDS_T8_CASE(3, 2)
#undef PDS_T8_CASE
    return set_error(-1, "conv3d_t8: no configuration");
}

}  // namespace pds
