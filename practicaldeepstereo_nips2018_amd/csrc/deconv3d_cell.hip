// ConvTranspose3d(kernel 4, stride 2, padding 1) as a DENSE GEMM per "cell" on the fp32 MFMA units: the two large
// up-sampling layers of the hourglass (reference practical_deep_stereo/regularization.py:48-50 at full size and
// :87-89, network_blocks.py:75-85,124-131).
//
// out[o] += in[i] * W[k] with o = 2 i - 1 + k.  Pair the outputs as (2c + 1, 2c + 2): both depend on exactly the
// inputs (c, c + 1), with taps k = 2 + p - 2 s for output parity p and input corner s in {0, 1} -- all four in range.
// In 3-D a cell c = (cz, cy, cx), c in [-1, n - 1] per axis, maps its 2 x 2 x 2 input corners (x Cin) to the
// 2 x 2 x 2 outputs (2c + 1 + p) (x Cout) with NO structurally zero weight: a dense [8 Cout] x [8 Cin] matrix.
// (conv3d_mfma.hip's form -- a 3x3x3 stencil over 8 parity classes with per-block tap masks -- multiplies by zeros for
// 56 % (Cout = 4) or 33 % (Cout = 8) of its MFMAs.)
//
//   GEMM view   M = 8 Cout "virtual channels" v = class * Cout + oc in blocks of 16, N = 16 consecutive cells along x,
//               K = 4 = the (yi, xi) corners of one (input channel, zi): v_mfma_f32_16x16x4_f32, exact fp32.
//               B[k][n] = in[ic][cz + zi][cy + yi][cx0 + n + xi]: the two k of a 32-lane LDS read group differ by one
//               float, so the reads are conflict-free for any row stride.
//   workgroup   4 waves, PERSISTENT (2 or 4 per CU), static tile lists in contiguous runs per XCD.  A tile is TZ x TY rows
//               of 16*NB cells; a wave owns RW rows and MBW of the M blocks: every B operand read from LDS feeds MBW
//               MFMAs; its A fragments (MBW x Cin x 2) are gathered once per workgroup straight from the
//               [Cin, Cout, 4, 4, 4] tensor (no packing launch) and stay in registers.
//   pipeline    as conv3d_t8.hip: double-buffered LDS halo tile, the next tile's global loads / deferred InstanceNorm
//               / LDS writes ride in the shadow of the MFMAs (one basic block, sched_group_barrier interleave), one
//               barrier per tile.  Out-of-volume inputs read as zero through the buffer range check.
//   epilogue    accumulators start at the bias; LeakyReLU; scatter to (2c + 1 + p); per-lane fp32 statistics of its
//               four channels, reduced in fp64 to ONE record per (workgroup, channel).
//
// Round 4, template flag X: the same contraction on v_mfma_f32_16x16x16_f16 with split fp32 operands (as conv2d_x3.hip):
// 56 of the 117 us of the (8, 4) layer were fp32 matrix time, and unlike the 8 -> 8 layers (conv3d_t8x.hip) the input of a
// transposed convolution is an eighth of its output, so the conversion of the staged values is cheap.  K = 16 = the four
// (yi, xi) corners (k group = lane >> 4) x FOUR input channels (a lane's four consecutive k): the LDS tile holds
// [part][channel group][z][y][x][4 ch] fp16, a B fragment is one aligned 8-byte slot, one MFMA x three partial products
// covers what four fp32 MFMAs did.  Operand scales are powers of two from max|w| (reduced per workgroup) and from the
// range certificate of the source (common.hpp Src::bound); a source without one keeps the exact-fp32 form.
#include <atomic>

#include "common.hpp"

namespace pds {

namespace {

constexpr int DC_THREADS = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void cell_split(const float (&v)[4], f16x4& hi, f16x4& lo) {
    pds_u32x2 h, l;   // (packed conversions, common.hpp)
    split_quad_f16(v, h, l);
    hi = __builtin_bit_cast(f16x4, h);
    lo = __builtin_bit_cast(f16x4, l);
}

struct CellArgs {
    Src a;
    const float* __restrict__ w;     // [Cin][Cout][4][4][4]
    const float* __restrict__ bias;  // [Cout]
    float* __restrict__ out;         // [N][Cout][2D][2H][2W]
    double* __restrict__ partials;   // [(n, oc)][records][2]
    int D, H, W;                     // input volume
    int lrelu;
    int tiles_x, tiles_y, tiles;     // per batch element (cells)
    int records;
};

// CIN, COUT: channels; MBW: M blocks per wave; TZ, TY: cell rows per tile; NB: 16-cell column blocks per tile
template <int CIN, int COUT, int MBW, int TZ, int TY, int NB>
struct CellCfg {
    static constexpr int MB = 8 * COUT / 16;           // M blocks in total
    static constexpr int WM = MB / MBW;                // wave groups along M
    static constexpr int WR = 4 / WM;                  // wave groups along the rows
    static constexpr int RW = TZ * TY / WR;            // rows per wave
    static constexpr int XT = 16 * NB + 1, YT = TY + 1, ZT = TZ + 1;
    static constexpr int RS = XT;
    static constexpr int CS = ZT * YT * RS;
    static constexpr int LDS_FLOATS = (CIN * CS + 3) / 4 * 4;
    static constexpr int NPOS = ZT * YT * XT;
    static constexpr int POS = (NPOS + DC_THREADS - 1) / DC_THREADS;
    static constexpr int GROUPS = CIN * 2;             // (ic, zi) k-steps
    static_assert(8 * COUT % 16 == 0 && MB % MBW == 0 && 4 % WM == 0 && (TZ * TY) % WR == 0, "bad tiling");
    static_assert(16 % COUT == 0 || COUT % 16 == 0, "channel blocks must align with parity classes");
    static_assert(TY % RW == 0, "the rows of a wave must share their z");
};

}  // namespace

// NORM: the source carries a deferred InstanceNorm; X: fp16-split operands on the 16-bit matrix pipe (header comment)
template <int CIN, int COUT, int MBW, int TZ, int TY, int NB, bool NORM, bool X>
__global__ __launch_bounds__(DC_THREADS, 2) void deconv3d_cell_kernel(const CellArgs A) {
    using C = CellCfg<CIN, COUT, MBW, TZ, TY, NB>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // X: one buffer = [part 2][CIN / 4 groups][CS slots of 8 bytes] = the same CIN * CS * 4 bytes as the fp32 tile
    constexpr int XPART = (CIN / 4) * C::CS * 8;   // bytes of one split part

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mgroup = wave % C::WM, rgroup = wave / C::WM;
    const int nb = blockIdx.y;
    const int Do = 2 * A.D, Ho = 2 * A.H, Wo = 2 * A.W;
    const size_t cstride = (size_t)A.D * A.H * A.W;
    const size_t cstride_o = (size_t)Do * Ho * Wo;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A.a.p + (size_t)nb * CIN * cstride), 0, (int)(CIN * cstride * sizeof(float)), 0x00020000);
    // cells of index -1 put the scalar part of a store address up to one plane + one row + one element BEFORE the
    // tensor (their valid lanes add it back through the parity offsets): the resource starts `guard` bytes early and
    // every scalar offset carries + guard, so it is never negative.  Nothing below the tensor is ever accessed.
    const int guard = (Ho * Wo + Wo + 1) * (int)sizeof(float);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(A.out + (size_t)nb * COUT * cstride_o) - guard, 0,
        (int)(COUT * cstride_o * sizeof(float)) + guard, 0x00020000);
    const int cbytes = (int)(cstride * sizeof(float));
    const int cobytes = (int)(cstride_o * sizeof(float));

    // ---- tiles of this workgroup (contiguous eighth of the list per XCD), coordinates advanced incrementally -----
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

    float ws = 1.f, as = 1.f;
    if constexpr (X) {   // power-of-two operand scales (every thread; before anything else touches the LDS scratch)
        float wm = 0.f;
        for (int i = tid; i < CIN * COUT * 64; i += DC_THREADS) wm = fmaxf(wm, fabsf(A.w[i]));
        wm = block_max(wm, lds);
        const float bound = block_bound(A.a.bound, A.a.bound_n, lds);
        ws = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pow2_scale(wm, kHalfTarget))));
        as = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pow2_scale(bound, kHalfTarget))));
    }
    const float unscale = (1.f / ws) * (1.f / as);
    float sa[CIN], ha[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
        sa[c] = ((NORM && A.a.scale) ? A.a.scale[nb * CIN + c] : 1.f) * as;   // (as = 1 in the exact form)
        ha[c] = ((NORM && A.a.scale) ? A.a.shift[nb * CIN + c] : 0.f) * as;
    }

    // ---- staging positions (halo tile of (TZ+1) x (TY+1) x (16 NB + 1) inputs per channel) -----------------------
    int pzz[C::POS], pyy[C::POS], pxx[C::POS], lo[C::POS];
#pragma unroll
    for (int k = 0; k < C::POS; ++k) {
        const int p = min(tid + k * DC_THREADS, C::NPOS - 1);
        pxx[k] = p % C::XT;
        pyy[k] = (p / C::XT) % C::YT;
        pzz[k] = p / (C::XT * C::YT);
        lo[k] = (pzz[k] * C::YT + pyy[k]) * C::RS + pxx[k];
    }
    unsigned ga[C::POS];
    unsigned inside_bits = 0;
    auto prepare = [&](int ax, int ay, int az) {
        // first cell of the tile is (az TZ - 1, ay TY - 1, ax 16 NB - 1): its corner zi = 0 is the input at that index
        const int z0 = az * TZ - 1, y0 = ay * TY - 1, x0 = ax * 16 * NB - 1;
        inside_bits = 0;
#pragma unroll
        for (int k = 0; k < C::POS; ++k) {
            const int z = z0 + pzz[k], y = y0 + pyy[k], x = x0 + pxx[k];
            const bool in = (unsigned)z < (unsigned)A.D && (unsigned)y < (unsigned)A.H && (unsigned)x < (unsigned)A.W;
            if (NORM) inside_bits |= in ? (1u << k) : 0u;
            ga[k] = in ? (unsigned)((z * A.H + y) * A.W + x) * 4u : ~0u;
        }
    };
    float va[CIN][C::POS];
    auto fetch_channel = [&](int c) {
#pragma unroll
        for (int k = 0; k < C::POS; ++k)
            va[c][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, ga[k], c * cbytes, 0));
    };
    auto stash_channel = [&](int c, float* buf) {
#pragma unroll
        for (int k = 0; k < C::POS; ++k) {
            float v = va[c][k];
            if (NORM) {
                v = fmaf(sa[c], v, ha[c]);
                v = ((inside_bits >> k) & 1u) ? v : 0.f;
            }
            buf[c * C::CS + lo[k]] = v;
        }
    };
    // X: the four channels of group cg at every position of the thread -> one 8-byte slot per part
    auto stash_group = [&](int cg, float* buf) {
        unsigned char* base = reinterpret_cast<unsigned char*>(buf) + cg * C::CS * 8;
#pragma unroll
        for (int k = 0; k < C::POS; ++k) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float t = NORM ? fmaf(sa[4 * cg + c], va[4 * cg + c][k], ha[4 * cg + c]) : va[4 * cg + c][k] * as;
                if (NORM) t = ((inside_bits >> k) & 1u) ? t : 0.f;   // (a plain source: the range check delivered the zeros)
                v[c] = t;
            }
            f16x4 hi, lo4;
            cell_split(v, hi, lo4);
            *reinterpret_cast<f16x4*>(base + lo[k] * 8) = hi;
            *reinterpret_cast<f16x4*>(base + XPART + lo[k] * 8) = lo4;
        }
    };

    // ---- lane roles -----------------------------------------------------------------------------------------------
    const int n16 = lane & 15, q = lane >> 4;
    // B: k = q -> (yi, xi) = (q >> 1, q & 1)
    // rows of this wave: cell rows (rz, ry0 + r), r < RW, inside the tile
    const int rz = (rgroup * C::RW) / TY, ry0 = (rgroup * C::RW) % TY;
    const int b_base = ((rz * C::YT + ry0 + (q >> 1)) * C::RS) + (q & 1) + n16;
    // D rows of block bl: v = (mgroup MBW + bl) 16 + 4 q + r -> class = v / COUT (independent of r), oc = ocb + r
    const int ocb = (4 * q) % COUT;
    int cls[MBW];
    unsigned out_lane[MBW];   // byte offset of (channel ocb, parity offsets of the class, column 2 n) in the output
#pragma unroll
    for (int bl = 0; bl < MBW; ++bl) {
        cls[bl] = ((mgroup * MBW + bl) * 16 + 4 * q) / COUT;
        const int pz = cls[bl] >> 2, py = (cls[bl] >> 1) & 1, px = cls[bl] & 1;
        out_lane[bl] = (unsigned)((size_t)ocb * cstride_o + ((size_t)pz * Ho + py) * Wo + px + 2 * n16) * 4u;
    }
    float bias4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = A.bias ? A.bias[ocb + r] : 0.f;
    __syncthreads();   // (X: the scale reductions are done with the LDS scratch)
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};

    int cur = 0;
    if (tile < t_end) {
        prepare(tx, ty, tz);
#pragma unroll
        for (int c0 = 0; c0 < CIN; c0 += 4) {
#pragma unroll
            for (int c = c0; c < c0 + 4; ++c) fetch_channel(c);
            if constexpr (X) {
                stash_group(c0 / 4, lds);
            } else {
#pragma unroll
                for (int c = c0; c < c0 + 4; ++c) stash_channel(c, lds);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- A fragments: lane (m = lane & 15, k = q): A[m][k] = W[ic][oc][2 + pz - 2 zi][2 + py - 2 yi][2 + px - 2 xi] ----
    constexpr int XG = (CIN / 4) * 2;            // X: k steps = (channel group, zi)
    float af[X ? 1 : MBW][X ? 1 : C::GROUPS];
    f16x4 afh[X ? MBW : 1][X ? XG : 1], afl[X ? MBW : 1][X ? XG : 1];
    {
        const int yi = q >> 1, xi = q & 1;
#pragma unroll
        for (int bl = 0; bl < MBW; ++bl) {
            const int v = (mgroup * MBW + bl) * 16 + n16;
            const int c8 = v / COUT, oc = v % COUT;
            const int pz = c8 >> 2, py = (c8 >> 1) & 1, px = c8 & 1;
            const int ky = 2 + py - 2 * yi, kx = 2 + px - 2 * xi;
            if constexpr (X) {
#pragma unroll
                for (int g = 0; g < XG; ++g) {
                    const int cg = g >> 1, zi = g & 1;
                    const int kz = 2 + pz - 2 * zi;
                    float wv[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        wv[c] = A.w[(((size_t)(4 * cg + c) * COUT + oc) * 4 + kz) * 16 + ky * 4 + kx] * ws;
                    cell_split(wv, afh[bl][g], afl[bl][g]);
                }
            } else {
#pragma unroll
                for (int g = 0; g < C::GROUPS; ++g) {
                    const int ic = g >> 1, zi = g & 1;
                    const int kz = 2 + pz - 2 * zi;
                    af[bl][g] = A.w[(((size_t)ic * COUT + oc) * 4 + kz) * 16 + ky * 4 + kx];
                }
            }
        }
    }
    __syncthreads();

    for (; tile < t_end; tile += per_xcd) {
        const int cz0 = tz * TZ - 1, cy0 = ty * TY - 1, cx0 = tx * 16 * NB - 1;   // first cell of the tile
        int nx = tx, ny = ty, nz = tz;
        advance(nx, ny, nz);
        const bool more = tile + per_xcd < t_end;
        prepare(more ? nx : tx, more ? ny : ty, more ? nz : tz);   // (the last tile re-stages itself: no branches below)
        tx = nx;
        ty = ny;
        tz = nz;
        float* nxt = lds + (cur ^ 1) * C::LDS_FLOATS;

        f32x4 acc[MBW][C::RW][NB];
#pragma unroll
        for (int bl = 0; bl < MBW; ++bl)
#pragma unroll
            for (int r = 0; r < C::RW; ++r)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[bl][r][j] = X ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{bias4[0], bias4[1], bias4[2], bias4[3]};

        if constexpr (X) {
            // the next tile's values are requested before the (short) matrix loop and converted after it; the other
            // workgroups of the CU cover what is not hidden
#pragma unroll
            for (int c = 0; c < CIN; ++c) fetch_channel(c);
            const unsigned char* bx = reinterpret_cast<const unsigned char*>(lds + cur * C::LDS_FLOATS) + b_base * 8;
#pragma unroll
            for (int g = 0; g < XG; ++g) {
                const int cg = g >> 1, zi = g & 1;
#pragma unroll
                for (int r = 0; r < C::RW; ++r) {
                    const unsigned char* p = bx + (cg * C::CS + (zi * C::YT + r) * C::RS) * 8;   // compile-time offsets
                    f16x4 bh[NB], bl2[NB];
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        bh[j] = *reinterpret_cast<const f16x4*>(p + 16 * j * 8);
                        bl2[j] = *reinterpret_cast<const f16x4*>(p + XPART + 16 * j * 8);
                    }
                    // small partial products first; consecutive MFMAs hit different accumulators
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int bl = 0; bl < MBW; ++bl)
                            acc[bl][r][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(afh[bl][g], bl2[j], acc[bl][r][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int bl = 0; bl < MBW; ++bl)
                            acc[bl][r][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(afl[bl][g], bh[j], acc[bl][r][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int bl = 0; bl < MBW; ++bl)
                            acc[bl][r][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(afh[bl][g], bh[j], acc[bl][r][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int cg = 0; cg < CIN / 4; ++cg) stash_group(cg, nxt);
        }
        const float* bp = lds + cur * C::LDS_FLOATS + b_base;
        float bb[2][C::RW * NB];
        auto read_group = [&](int g, float* dst) {
            const int ic = g >> 1, zi = g & 1;
#pragma unroll
            for (int r = 0; r < C::RW; ++r) {
                const float* p = bp + ic * C::CS + (zi * C::YT + r) * C::RS;   // compile-time offsets
#pragma unroll
                for (int j = 0; j < NB; ++j) dst[r * NB + j] = p[16 * j];
            }
        };
        if constexpr (!X) read_group(0, bb[0]);
#pragma unroll
        for (int g = 0; g < (X ? 0 : C::GROUPS); ++g) {
            if (g + 1 < C::GROUPS) read_group(g + 1, bb[(g + 1) & 1]);
            if (g < CIN) fetch_channel(g);
            if (g >= CIN) stash_channel(g - CIN, nxt);
#pragma unroll
            for (int r = 0; r < C::RW; ++r)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int bl = 0; bl < MBW; ++bl)
                        acc[bl][r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[bl][g], bb[g & 1][r * NB + j],
                                                                             acc[bl][r][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < C::RW * NB * MBW; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
            }
        }

        // ---- epilogue: LeakyReLU, scatter to (2 c + 1 + p), statistics --------------------------------------------
        // interior tiles (no cell on the border of the cell grid) need no masks
        const bool interior = cz0 >= 0 && cz0 + TZ <= A.D - 1 && cy0 >= 0 && cy0 + TY <= A.H - 1 && cx0 >= 0 &&
                              cx0 + 16 * NB <= A.W - 1;
#pragma unroll
        for (int r = 0; r < C::RW; ++r) {
            const int cz = cz0 + rz, cy = cy0 + ry0 + r;
            // scalar part: output (2 cz + 1, 2 cy + 1, 2 cx0 + 1) (+ guard); lane part: class parities, channel, 2 n
            const int row_bytes = (((2 * cz + 1) * Ho + (2 * cy + 1)) * Wo + 2 * cx0 + 1) * (int)sizeof(float) + guard;
#pragma unroll
            for (int bl = 0; bl < MBW; ++bl) {
                bool lane_ok = true;
                if (!interior) {
                    const int pz = cls[bl] >> 2, py = (cls[bl] >> 1) & 1;
                    const int oz = 2 * cz + 1 + pz, oy = 2 * cy + 1 + py;
                    lane_ok = (unsigned)oz < (unsigned)Do && (unsigned)oy < (unsigned)Ho;
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    bool ok = lane_ok;
                    if (!interior) {
                        const int ox = 2 * (cx0 + 16 * j + n16) + 1 + (cls[bl] & 1);
                        ok = ok && (unsigned)ox < (unsigned)Wo;
                    }
                    const unsigned off = ok ? out_lane[bl] + 128u * j : ~0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = X ? fmaf(acc[bl][r][j][e], unscale, bias4[e]) : acc[bl][r][j][e];
                        if (A.lrelu) t = fmaxf(t, t * kLeakySlope);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, t), ro, off, row_bytes + e * cobytes, 0);
                        t = ok ? t : 0.f;
                        ssum[e] += t;
                        ssq[e] = fmaf(t, t, ssq[e]);
                    }
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- one record per (workgroup, channel) ----------------------------------------------------------------------
    if (A.partials) {
        double* red = reinterpret_cast<double*>(lds);   // [256 threads][4 channels][2]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[tid * 8 + e * 2 + 0] = (double)ssum[e];
            red[tid * 8 + e * 2 + 1] = (double)ssq[e];
        }
        __syncthreads();
        if (tid < COUT * 2) {
            const int oc = tid >> 1, k = tid & 1;
            double sum = 0.0;
            for (int t = 0; t < DC_THREADS; ++t) {
                const int tq = (t & 63) >> 4;
                const int tocb = (4 * tq) % COUT;
                if (oc >= tocb && oc < tocb + 4) sum += red[t * 8 + (oc - tocb) * 2 + k];
            }
            A.partials[(((size_t)nb * COUT + oc) * A.records + blockIdx.x) * 2 + k] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
namespace {

bool cell_enabled() {
    static const bool on = []() {  // PDS_DECONV_CELL=0: the generic MFMA kernel serves these layers (A/B)
        const char* e = debug_switch("PDS_DECONV_CELL");
        return !(e && e[0] == '0');
    }();
    return on;
}

struct CellPlan {
    int id, tz, ty, nb;
};

// configurations: (Cin, Cout) = (8, 4): id 0; (16, 8): id 1
// (8, 4) runs 32-cell-wide tiles on FOUR workgroups per CU: with 64-cell tiles the kernel needed 251 VGPRs (two waves per
// SIMD, two workgroups per CU) -- 112 -> 102 us at 48 x 144 x 240; half-height tiles instead: 105-107; (16, 8) on 32-cell
// tiles: 66 -> 82, it keeps 64.
CellPlan cell_plan(int cin, int cout) {
    if (cin == 8 && cout == 4) return CellPlan{0, 2, 4, 2};
    if (cin == 16 && cout == 8) return CellPlan{1, 2, 2, 4};
    return CellPlan{-1, 0, 0, 0};
}

int cell_tiles(const Geom& in, const CellPlan& p) {
    const int cx = in.w + 1, cy = in.h + 1, cz = in.d + 1;
    return ((cx + 16 * p.nb - 1) / (16 * p.nb)) * ((cy + p.ty - 1) / p.ty) * ((cz + p.tz - 1) / p.tz);
}

template <int CIN, int COUT, int MBW, int TZ, int TY, int NB, bool NORM, bool X>
int launch_cell(const CellArgs& A, int batch, hipStream_t s) {
    using C = CellCfg<CIN, COUT, MBW, TZ, TY, NB>;
    constexpr size_t lds_bytes = (size_t)2 * C::LDS_FLOATS * sizeof(float);
    static_assert(lds_bytes >= (size_t)DC_THREADS * 8 * sizeof(double), "reduction scratch must fit");
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&deconv3d_cell_kernel<CIN, COUT, MBW, TZ, TY, NB, NORM, X>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    }
    hipLaunchKernelGGL((deconv3d_cell_kernel<CIN, COUT, MBW, TZ, TY, NB, NORM, X>), dim3(A.records, batch), dim3(DC_THREADS),
                       lds_bytes, s, A);
    return check_launch("deconv3d_cell");
}

}  // namespace

bool deconv3d_cell_supported(const DeconvLayer& L) {
    if (!cell_enabled()) return false;
    if (L.kd != 4 || L.b.p != nullptr) return false;
    if (cell_plan(L.in.c, L.out_g.c).id < 0) return false;
    if (L.a.scale && L.a.per_plane) return false;
    if ((size_t)L.out_g.c * L.out_g.d * L.out_g.h * L.out_g.w >= ((size_t)1 << 30)) return false;   // 32-bit byte offsets
    if ((size_t)L.in.c * L.in.d * L.in.h * L.in.w >= ((size_t)1 << 30)) return false;
    if (L.in.n > 65535) return false;
    return true;
}

int deconv3d_cell_records(const Geom& in, int cout) {
    const CellPlan p = cell_plan(in.c, cout);
    const int tiles = cell_tiles(in, p);
    int per_n = (p.id == 0 ? 1024 : 512) / (in.n > 0 ? in.n : 1);   // persistent workgroups: 4 resp. 2 per CU
    if (per_n > tiles) per_n = tiles;
    per_n = (per_n + 7) / 8 * 8;
    return per_n < 8 ? 8 : per_n;
}

int launch_deconv3d_cell(const DeconvLayer& L, hipStream_t s) {
    const CellPlan p = cell_plan(L.in.c, L.out_g.c);
    CellArgs A;
    A.a = L.a;
    A.w = L.weight;
    A.bias = L.bias;
    A.out = L.out;
    A.partials = L.partials;
    A.D = L.in.d;
    A.H = L.in.h;
    A.W = L.in.w;
    A.lrelu = L.lrelu;
    A.tiles_x = (A.W + 1 + 16 * p.nb - 1) / (16 * p.nb);
    A.tiles_y = (A.H + 1 + p.ty - 1) / p.ty;
    A.tiles = cell_tiles(L.in, p);
    A.records = deconv3d_cell_records(L.in, L.out_g.c);
    const bool norm = L.a.scale != nullptr;
    static const bool split_on = []() {  // PDS_DECONV_CELL_X=0: exact-fp32 MFMAs also for certified sources (A/B)
        const char* e = debug_switch("PDS_DECONV_CELL_X");
        return !(e && e[0] == '0');
    }();
    // fp16-split form when the source carries a range certificate (inside the hourglass: always)
    const bool x = split_on && norm && L.a.bound && L.a.bound_n > 0;
    if (p.id == 0) {
        if (x) return launch_cell<8, 4, 2, 2, 4, 2, true, true>(A, L.in.n, s);
        return norm ? launch_cell<8, 4, 2, 2, 4, 2, true, false>(A, L.in.n, s)
                    : launch_cell<8, 4, 2, 2, 4, 2, false, false>(A, L.in.n, s);
    }
    if (p.id == 1) {
        if (x) return launch_cell<16, 8, 2, 2, 2, 4, true, true>(A, L.in.n, s);
        return norm ? launch_cell<16, 8, 2, 2, 2, 4, true, false>(A, L.in.n, s)
                    : launch_cell<16, 8, 2, 2, 2, 4, false, false>(A, L.in.n, s);
    }
    return set_error(-1, "deconv3d_cell: no configuration");
}

}  // namespace pds
