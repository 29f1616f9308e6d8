// 3-D convolutions and k4/s2 transposed convolutions of the hourglass's INNER levels -- 16 to 128 channels over
// volumes of 405 to 207 360 voxels (reference practical_deep_stereo/regularization.py:22-26,48-52 at levels 1-3,
// network_blocks.py:61-85,106-131) -- on the exact-fp32 MFMA units.
//
// These layers are small GEMMs with a long K (Cin x 27 up to 3456) and few output positions: the tile-per-workgroup
// kernel (conv3d_mfma.hip) walks K in chunks behind a barrier each, one dependent accumulator per wave, a few dozen
// workgroups in flight -- 30 to 70 us per layer for 0.2 to 1.7 GFLOP.  Here K is SPLIT OVER THE WAVES instead:
//
//   workgroup   KSPLIT = 4, 8 or 16 waves share one output tile (one row of 16 NB columns, MBW channel blocks of 16);
//               wave k convolves input channels [k Cin/KSPLIT, (k+1) Cin/KSPLIT) only -- 4 or 8 of them -- so even the
//               405-voxel level runs > 1700 waves.  The waves never synchronise while they
//               accumulate: each stages ITS channels' halo tile into a private LDS region (deferred InstanceNorm of
//               the producer(s), skip sum and zero padding applied on the way; out-of-volume reads come back as zero
//               from the buffer range check), then streams its weights.
//   weights     ALL A fragments of a wave (<= 2 x 27 x MBW) go from global memory (L2-resident, fragment order
//               [ic/4][tap][block][64 lanes]: 256 contiguous bytes per load) straight into registers, requested before
//               the staging starts so both latencies overlap; every fragment feeds NB MFMAs.  No LDS copy, no barrier.
//               (A first version with 4 waves per workgroup and weights streamed tap group by tap group was no
//               faster than conv3d_mfma.hip: < 2 waves per CU, each serially waiting ~1.3 us per round trip.)
//   GEMM view   M = 16 output channels per block, N = 16 consecutive output x, K = 4 input channels per
//               v_mfma_f32_16x16x4_f32; the LDS channel stride is == 16 (mod 32) (stride 1) or odd (stride 2) so the
//               two k-halves of a 32-lane read group hit disjoint banks.
//   reduction   the four partial accumulators meet in LDS (one barrier), are summed in a fixed order, and the waves
//               share the epilogue: bias, LeakyReLU(0.1), store, per-channel statistics -> one record per workgroup.
//   MODE 2      ConvTranspose3d(k4, s2, p1) in the dense "cell" form of deconv3d_cell.hip: cell c maps its 2x2x2 input
//               corners to the 2x2x2 outputs 2c + 1 + p through 8 "taps"; virtual channel v = class * Cout + oc.
//
// Round 5, template flag X: the same kernel on v_mfma_f32_16x16x16_f16 with split fp32 operands (as conv2d_x3.hip /
// deconv3d_cell.hip).  PMC of the 32 -> 32 level-2 layer: its four waves per SIMD keep the fp32 matrix pipe busy for 27 600 of
// their 33 000 cycles (216 MFMAs x 32 cycles each) -- the larger of these layers are fp32-pipe-bound for ~40 % of their time.
// K = 16 = four input channels x the four x-taps of a kernel row (dx = 0..2 and a zero): one MFMA x three partial products
// per (dz, dy) covers what three fp32 MFMAs did; in the cell form K = four channels x the (yi, xi) corners, one MFMA per zi.
// The wave-private LDS region holds [group of 4 channels][part][position][4 x fp16] (the bytes of the fp32 tile), a B
// fragment is one 8-byte slot (k group = lane >> 4 selects the x-tap / corner).  Operand scales: powers of two from max|w|
// (pack.hip: pack_wscale_kernel) and from the sources' range certificates (Src::bound); a source without one keeps fp32.
#include <atomic>

#include "common.hpp"

namespace pds {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ks_f16x4 __attribute__((ext_vector_type(4)));

struct KsArgs {
    Src a, b;
    const float* __restrict__ wpk;   // [Cin / 4][taps][mblocks][64]
    const float* __restrict__ bias;
    float* __restrict__ out;
    double* __restrict__ partials;
    int Cin, Di, Hi, Wi;             // input volume
    int Cout, Do, Ho, Wo;            // REAL output channels and output volume
    int lrelu;
    int tiles_x, tiles_y, tiles;
    int mblocks;                     // blocks of 16 (virtual) channels
    int cs;                          // LDS floats per input channel (bank-padded)
    // X form: the 16-dword tail behind the packed weights (dwords 12, 13: ws, 1 / ws)
    const float* __restrict__ wtail;
};

// MODE 0: conv 3x3x3 stride 1; 1: conv 3x3x3 stride 2; 2: transposed conv k4 s2 p1 (cell form)
template <int MODE, int TZ, int TY, int NB>
struct KsGeom {
    static constexpr int S = MODE == 1 ? 2 : 1;
    static constexpr int TAPS = MODE == 2 ? 8 : 27;
    static constexpr int GROUP = MODE == 2 ? 4 : 9;       // taps whose weights are fetched together
    static constexpr int NGROUPS = TAPS / GROUP;
    static constexpr int EXT = MODE == 2 ? 2 : 3;         // stencil extent per axis
    static constexpr int ZT = (TZ - 1) * S + EXT, YT = (TY - 1) * S + EXT, XT = (16 * NB - 1) * S + EXT;
    static constexpr int NPOS = ZT * YT * XT;
    static constexpr int CS = S == 1 ? ((NPOS + 15) / 32 * 32 + 16) : (NPOS | 1);
    static constexpr int R = TZ * TY;
};

__device__ __forceinline__ float ks_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

}  // namespace

// ---- how an item of work is ordered against its producers ---------------------------------------------------------
// Stand-alone launch: the kernel boundary orders everything, scalar loads of the folded InstanceNorm are fine.
struct KsAlone {
    __device__ __forceinline__ void mark(unsigned) const {}
    __device__ __forceinline__ void before_staging() const {}
    __device__ __forceinline__ float coef(const float* p) const { return *p; }
};
// Inside the persistent chain kernel (below): the sources were written by other workgroups of the SAME launch.  One lane
// polls the producer phase's `ready` word (relaxed, agent scope), ONE agent-scope acquire drops this CU's stale L1 lines,
// the barrier publishes that to the other waves, then plain loads (MI355X_MICROARCH.md, inter-workgroup visibility).  The
// folded coefficients must not travel through the scalar cache, which no fence of this kernel invalidates.
#ifndef PDS_KS_POLL_TIMEOUT_TICKS
#define PDS_KS_POLL_TIMEOUT_TICKS 200000000ll   // 2 s of the 100 MHz clock
#endif
constexpr long long kKsPollTimeoutTicks = PDS_KS_POLL_TIMEOUT_TICKS;
struct KsInLaunch {
    const unsigned* ready;     // nullptr: the phase has no in-launch producer
    unsigned* nonfinite;       // host-mapped error counter (pds_nonfinite_statistics): a poll that times out counts here
    unsigned* marks;           // debug builds (-DPDS_KS_CHAIN_MARK): the workgroup's progress word, tools/chain_debug.py
    __device__ __forceinline__ void mark(unsigned stage) const {
#ifdef PDS_KS_CHAIN_MARK
        if (threadIdx.x == 0) __hip_atomic_store(marks, stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    __device__ __forceinline__ void before_staging() const {
        if (!ready) return;    // (kernel argument: uniform)
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __builtin_amdgcn_s_sleep(4);
                if (wall_clock64() - t0 > kKsPollTimeoutTicks) {   // never hang the GPU, report instead
                    if (nonfinite) atomicAdd_system(nonfinite, 1u << 20);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    __device__ __forceinline__ float coef(const float* p) const {
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// One item of work: output tile `tile` (one row of 16 NB columns), channel blocks [mb0, mb0 + MBW), batch entry nb, by
// the KSPLIT waves of a (sub-)group: `wave` / `tid` count inside the group, `lds` is the group's own region.  Every wave
// of the WORKGROUP must call it (the barriers are workgroup-wide); a group without an item passes active = false: it
// recomputes the item it was handed and stores nothing.
// NKS: groups of four input channels per wave (Cin = 4 NKS KSPLIT)
template <int MODE, int TZ, int TY, int NB, int MBW, int KSPLIT, int NKS, bool X, class Sync>
__device__ __forceinline__ void ks_item(const KsArgs& A, const int tile, const int mb0, const int nb, float* lds,
                                        const int wave, const int tid, const bool active, const Sync& sync) {
    using G = KsGeom<MODE, TZ, TY, NB>;
    constexpr int S = G::S, R = G::R, NACC = MBW * R * NB;
    constexpr int TG = MODE == 2 ? 2 : 9;          // X form: MFMA K-steps per group of four channels ((dz, dy) / zi)

    const int lane = tid & 63;
    const int n16 = lane & 15, q = lane >> 4;
    const int tx = tile % A.tiles_x, ty = (tile / A.tiles_x) % A.tiles_y, tz = tile / (A.tiles_x * A.tiles_y);
    // output (conv) / cell (deconv) origin of the tile and the input coordinate of halo position (0, 0, 0)
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * 16 * NB;
    const int iz0 = MODE == 2 ? z0 - 1 : z0 * S - 1, iy0 = MODE == 2 ? y0 - 1 : y0 * S - 1,
              ix0 = MODE == 2 ? x0 - 1 : x0 * S - 1;
    constexpr int cq = 4 * NKS;           // input channels of this wave
    const int c_first = wave * cq;
    const size_t plane_i = (size_t)A.Hi * A.Wi, cstride = (size_t)A.Di * plane_i;
    const bool two = A.b.p != nullptr;
    const size_t cstride_b = (two && A.b.bcast_d) ? plane_i : cstride;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A.a.p + ((size_t)nb * A.Cin + c_first) * cstride), 0, (int)(cq * cstride * sizeof(float)),
        0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(two ? A.b.p + ((size_t)nb * A.Cin + c_first) * cstride_b : A.a.p), 0,
        (int)(cq * (two ? cstride_b : cstride) * sizeof(float)), 0x00020000);

    // ---- every A fragment of this wave: requested now, consumed after the staging ------------------------------------
    const size_t tap_stride = (size_t)A.mblocks * 64;                       // floats between consecutive taps
    float af[X ? 1 : NKS][X ? 1 : G::TAPS][X ? 1 : MBW];
    pds_u32x2 afx[X ? NKS : 1][X ? TG : 1][X ? MBW : 1][2];                  // X: [group][K-step][block][part], 8 bytes per lane
    if constexpr (X) {
        // packed [ic / 4][K-step][block][part][64 lanes][2 dwords] (pack.hip modes 8 / 9)
        const pds_u32x2* wl = reinterpret_cast<const pds_u32x2*>(A.wpk) +
                              (((size_t)(c_first >> 2) * TG) * A.mblocks + mb0) * 2 * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int t = 0; t < TG; ++t)
#pragma unroll
                for (int m = 0; m < MBW; ++m)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        afx[ks][t][m][p] = wl[((((size_t)ks * TG + t) * A.mblocks + m) * 2 + p) * 64];
    } else {
        const float* wl = A.wpk + ((size_t)(c_first >> 2) * G::TAPS) * tap_stride + (size_t)mb0 * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int t = 0; t < G::TAPS; ++t)
#pragma unroll
                for (int m = 0; m < MBW; ++m) af[ks][t][m] = wl[((size_t)ks * G::TAPS + t) * tap_stride + m * 64];
    }
    // (in-launch producers: the weights above are already on their way while this waits)
    sync.mark(0x20);
    sync.before_staging();
    sync.mark(0x21);
    // X: power-of-two operand scales -- as from the sources' range certificates (few records: every wave reduces them
    // itself, no barrier), 1 / ws from the packed weights' tail
    float ascale = 1.f, unscale = 1.f;
    if constexpr (X) {
        float bm = 0.f, bm2 = 0.f;
        for (int i = lane; i < A.a.bound_n; i += 64) {
            const float v = fabsf(A.a.bound[i]);
            bm = fmaxf(bm, v == v ? v : __builtin_inff());
        }
        if (two)
            for (int i = lane; i < A.b.bound_n; i += 64) {
                const float v = fabsf(A.b.bound[i]);
                bm2 = fmaxf(bm2, v == v ? v : __builtin_inff());
            }
        ascale = pow2_scale(wave_max(bm) + wave_max(bm2), kHalfTarget);
        unscale = A.wtail[13] * (1.f / ascale);
    }

    // ---- stage this wave's channels: private LDS region [cq][CS] ---------------------------------------------------
    float* mine = lds + (size_t)wave * cq * A.cs;
    {
        const float* sa = A.a.scale ? A.a.scale + (size_t)nb * A.Cin + c_first : nullptr;
        const float* ha = A.a.scale ? A.a.shift + (size_t)nb * A.Cin + c_first : nullptr;
        const float* sb = (two && A.b.scale) ? A.b.scale + (size_t)nb * A.Cin + c_first : nullptr;
        const float* hb = (two && A.b.scale) ? A.b.shift + (size_t)nb * A.Cin + c_first : nullptr;
        // all loads of the wave are issued back to back (a few dozen per lane), then written: one round trip.  The
        // elements are walked channel by channel (lane + 64 i inside a channel), so the channel -- hence its InstanceNorm
        // coefficients -- is a compile-time index: the first version walked one flat index and picked the coefficients
        // with a select chain per element, 1 750 VALU instructions per wave for 216 MFMAs.
        constexpr int PER_CH = (G::NPOS + 63) / 64;
        float va[cq][PER_CH], vb[cq][PER_CH];
        int lo_[PER_CH];
        bool in_[PER_CH];
        unsigned off_[PER_CH], offb_[PER_CH];
#pragma unroll
        for (int i = 0; i < PER_CH; ++i) {
            const int p = min(lane + 64 * i, G::NPOS - 1);            // (surplus lanes re-stage the last element)
            const int xx = p % G::XT, yy = (p / G::XT) % G::YT, zz = p / (G::XT * G::YT);
            lo_[i] = p;
            const int z = iz0 + zz, y = iy0 + yy, x = ix0 + xx;
            in_[i] = (unsigned)z < (unsigned)A.Di && (unsigned)y < (unsigned)A.Hi && (unsigned)x < (unsigned)A.Wi;
            off_[i] = in_[i] ? (unsigned)((size_t)z * plane_i + (size_t)y * A.Wi + x) * 4u : ~0u;
            offb_[i] = !in_[i] ? ~0u : (two && A.b.bcast_d ? (unsigned)((size_t)y * A.Wi + x) * 4u : off_[i]);
        }
        const unsigned cbytes_a = (unsigned)(cstride * sizeof(float)), cbytes_b = (unsigned)(cstride_b * sizeof(float));
#pragma unroll
        for (int c = 0; c < cq; ++c)
#pragma unroll
            for (int i = 0; i < PER_CH; ++i) {
                // (the channel offset goes into the vector offset: an out-of-range marker plus a scalar offset could wrap)
                const unsigned oa = in_[i] ? off_[i] + c * cbytes_a : ~0u;
                va[c][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, oa, 0, 0));
                if (two) {
                    const unsigned ob = in_[i] ? offb_[i] + c * cbytes_b : ~0u;
                    vb[c][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, ob, 0, 0));
                }
            }
        if constexpr (X) {
            // [group of 4 channels][part][position][4 x fp16]: the same bytes as the fp32 tile of those channels
            unsigned char* minex = reinterpret_cast<unsigned char*>(mine);
            // (coefficients pre-multiplied by the operand scale, in vector registers: measured 36 -> 30 us on the
            // 32 -> 32 layer against scalar coefficients + one multiply per element)
            float s1[cq], h1[cq], s2[cq], h2[cq];
#pragma unroll
            for (int c = 0; c < cq; ++c) {
                s1[c] = (sa ? sync.coef(sa + c) : 1.f) * ascale;
                h1[c] = (sa ? sync.coef(ha + c) : 0.f) * ascale;
                s2[c] = (sb ? sync.coef(sb + c) : 1.f) * ascale;
                h2[c] = (sb ? sync.coef(hb + c) : 0.f) * ascale;
            }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int i = 0; i < PER_CH; ++i) {
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = 4 * ks + k;
                        float t = fmaf(s1[c], va[c][i], h1[c]);
                        if (two) t += fmaf(s2[c], vb[c][i], h2[c]);
                        v[k] = in_[i] ? t : 0.f;
                    }
                    pds_u32x2 hi, lo;
                    split_quad_f16(v, hi, lo);
                    unsigned char* slot = minex + ((size_t)ks * 2 * G::NPOS + lo_[i]) * 8;
                    *reinterpret_cast<pds_u32x2*>(slot) = hi;
                    *reinterpret_cast<pds_u32x2*>(slot + (size_t)G::NPOS * 8) = lo;
                }
        } else {
#pragma unroll
            for (int c = 0; c < cq; ++c) {
                const float s1 = sa ? sync.coef(sa + c) : 1.f, h1 = sa ? sync.coef(ha + c) : 0.f;   // (stand-alone: scalar loads)
                const float s2 = sb ? sync.coef(sb + c) : 1.f, h2 = sb ? sync.coef(hb + c) : 0.f;
#pragma unroll
                for (int i = 0; i < PER_CH; ++i) {
                    float v = fmaf(s1, va[c][i], h1);
                    if (two) v += fmaf(s2, vb[c][i], h2);
                    mine[c * A.cs + lo_[i]] = in_[i] ? v : 0.f;
                }
            }
        }
    }

    sync.mark(0x22);
    // ---- K loop: this wave's NKS groups of four channels, all taps ------------------------------------------------------
    f32x4 acc[MBW][R][NB];
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[m][r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (X) {
        // B fragment: the 8-byte slot of position (column n = lane & 15 [+ the x-tap / corner of k group lane >> 4])
        // cell: (yi, xi) = (q >> 1, q & 1); conv: dx = q, and the zero fourth group re-reads dx = 0 (finite data: a slot
        // past the row could hold the bits of an fp16 infinity, and inf x 0 is not 0)
        const int kg_off = MODE == 2 ? ((q >> 1) * G::XT + (q & 1)) : (q == 3 ? 0 : q);
        const unsigned char* bl = reinterpret_cast<const unsigned char*>(mine) + (size_t)(n16 * S + kg_off) * 8;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const int dz = MODE == 2 ? t : t / 3, dy = MODE == 2 ? 0 : t % 3;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int rz = r / TY, ry = r % TY;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const unsigned char* bp =
                            bl + ((size_t)ks * 2 * G::NPOS + ((rz * S + dz) * G::YT + ry * S + dy) * G::XT + j * 16 * S) * 8;
                        const ks_f16x4 b_hi = *reinterpret_cast<const ks_f16x4*>(bp);
                        const ks_f16x4 b_lo = *reinterpret_cast<const ks_f16x4*>(bp + (size_t)G::NPOS * 8);
#pragma unroll
                        for (int m = 0; m < MBW; ++m) {
                            const ks_f16x4 a_hi = __builtin_bit_cast(ks_f16x4, afx[ks][t][m][0]);
                            const ks_f16x4 a_lo = __builtin_bit_cast(ks_f16x4, afx[ks][t][m][1]);
                            acc[m][r][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_hi, b_lo, acc[m][r][j], 0, 0, 0);
                            acc[m][r][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_lo, b_hi, acc[m][r][j], 0, 0, 0);
                            acc[m][r][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_hi, b_hi, acc[m][r][j], 0, 0, 0);
                        }
                    }
                }
            }
    } else {
        const float* bl = mine + q * A.cs + n16 * S;   // B fragment: channel k = lane >> 4, column n = lane & 15
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int tap = 0; tap < G::TAPS; ++tap) {
                const int dz = MODE == 2 ? tap >> 2 : tap / 9, dy = MODE == 2 ? (tap >> 1) & 1 : (tap / 3) % 3,
                          dx = MODE == 2 ? tap & 1 : tap % 3;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int rz = r / TY, ry = r % TY;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const float b = bl[ks * 4 * A.cs + ((rz * S + dz) * G::YT + ry * S + dy) * G::XT + j * 16 * S + dx];
#pragma unroll
                        for (int m = 0; m < MBW; ++m)
                            acc[m][r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][tap][m], b, acc[m][r][j], 0, 0, 0);
                    }
                }
            }
    }

    // ---- the four partial sums meet in LDS; wave w finishes the accumulators with index == w (mod 4) -------------------
    sync.mark(0x23);
    __syncthreads();                                   // every wave is done with its staging region
    sync.mark(0x24);
    f32x4* red = reinterpret_cast<f32x4*>(lds);        // [KSPLIT waves][NACC][64 lanes]
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < NB; ++j) red[((size_t)wave * NACC + (m * R + r) * NB + j) * 64 + lane] = acc[m][r][j];
    __syncthreads();

    const size_t plane_o = (size_t)A.Ho * A.Wo, cstride_o = (size_t)A.Do * plane_o;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        A.out + (size_t)nb * A.Cout * cstride_o, 0, (int)(A.Cout * cstride_o * sizeof(float)), 0x00020000);
    float ssum[MBW][4], ssq[MBW][4];
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) ssum[m][e] = ssq[m][e] = 0.f;
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int idx = (m * R + r) * NB + j;
                if (idx % KSPLIT != wave) continue;    // wave-uniform
                f32x4 t = red[((size_t)0 * NACC + idx) * 64 + lane];
#pragma unroll
                for (int w = 1; w < KSPLIT; ++w) t += red[((size_t)w * NACC + idx) * 64 + lane];
                const int rz = r / TY, ry = r % TY;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int v = (mb0 + m) * 16 + 4 * q + e;      // (virtual) output channel
                    int oc = v, oz = z0 + rz, oy = y0 + ry, ox = x0 + 16 * j + n16;
                    bool ok = active && v < (MODE == 2 ? 8 * A.Cout : A.Cout);
                    if (MODE == 2) {
                        const int cls = v / A.Cout;
                        oc = v - cls * A.Cout;
                        oz = 2 * (oz - 1) + 1 + ((cls >> 2) & 1);
                        oy = 2 * (oy - 1) + 1 + ((cls >> 1) & 1);
                        ox = 2 * (ox - 1) + 1 + (cls & 1);
                    }
                    ok = ok && (unsigned)oz < (unsigned)A.Do && (unsigned)oy < (unsigned)A.Ho && (unsigned)ox < (unsigned)A.Wo;
                    float val = X ? fmaf(t[e], unscale, (ok && A.bias) ? A.bias[oc] : 0.f)
                                  : t[e] + ((ok && A.bias) ? A.bias[oc] : 0.f);
                    if (A.lrelu) val = fmaxf(val, val * kLeakySlope);
                    const unsigned off = ok ? (unsigned)(((size_t)oc * A.Do + oz) * plane_o + (size_t)oy * A.Wo + ox) * 4u : ~0u;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, val), ro, off, 0, 0);
                    val = ok ? val : 0.f;
                    ssum[m][e] += val;
                    ssq[m][e] = fmaf(val, val, ssq[m][e]);
                }
            }

    sync.mark(0x25);
    // ---- statistics: one record per (workgroup, channel [, parity class]) ------------------------------------------------
    if (A.partials) {
        __syncthreads();                               // the reduction scratch has been read
        float* sred = lds;                             // [KSPLIT waves][MBW * 16][2]
#pragma unroll
        for (int m = 0; m < MBW; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = ks_row16_sum(ssum[m][e]), sq = ks_row16_sum(ssq[m][e]);
                if (n16 == 15) {
                    sred[((wave * MBW + m) * 16 + 4 * q + e) * 2 + 0] = s;
                    sred[((wave * MBW + m) * 16 + 4 * q + e) * 2 + 1] = sq;
                }
            }
        __syncthreads();
        if (tid < MBW * 16 * 2) {
            const int c16 = tid >> 1, k = tid & 1;
            const int v = mb0 * 16 + c16;
            const int vmax = MODE == 2 ? 8 * A.Cout : A.Cout;
            if (v < vmax && active) {
                double sum = 0.0;
#pragma unroll
                for (int w = 0; w < KSPLIT; ++w) sum += (double)sred[((w * MBW * 16) + c16) * 2 + k];
                if (MODE == 2) {
                    const int cls = v / A.Cout, oc = v - cls * A.Cout;   // records of one real channel: [tile][class]
                    A.partials[((((size_t)nb * A.Cout + oc) * A.tiles + tile) * 8 + cls) * 2 + k] = sum;
                } else {
                    A.partials[(((size_t)nb * A.Cout + v) * A.tiles + tile) * 2 + k] = sum;
                }
            }
        }
    }
}

// ---- stand-alone launch: one item per workgroup -------------------------------------------------------------------------
template <int MODE, int TZ, int TY, int NB, int MBW, int KSPLIT, int NKS, bool X = false>
__global__ __launch_bounds__(64 * KSPLIT) void conv3d_ks_kernel(const KsArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    ks_item<MODE, TZ, TY, NB, MBW, KSPLIT, NKS, X>(A, blockIdx.x, blockIdx.y * MBW, blockIdx.z, lds, wave, threadIdx.x, true,
                                                   KsAlone{});
}

// ---------------------------------------------------------------------------------------------------------------
// The persistent CHAIN kernel (round 6; opt-in, see conv3d_ks_chain_enabled): a run of consecutive K-split layers of the
// hourglass -- at 960x540, D = 192 the eleven layers of levels 1-3 (regularization.py:22-26, 48-52: contraction 1-3,
// expansion 0-1 and the transposed convolution of expansion 2) -- as ONE launch instead of eleven launches + eleven
// in_finalize launches.
//
//   work list   every phase (layer) is a list of tickets; a ticket is 8 / KSPLIT items (4-wave configurations run two
//               items side by side in the 8-wave workgroup).  Workgroups draw tickets from ONE atomic counter, in order.
//   ordering    a ticket of phase p waits for `ready[p - 1]` (set when every ticket of phase p - 1 is done AND its
//               InstanceNorm has been folded).  Tickets are drawn in order and depend on earlier tickets only, so the
//               scheme cannot deadlock however few workgroups are resident (another stream's kernel may hold the CUs, a
//               second chain kernel may run beside this one): the lowest unfinished ticket is always held by a running
//               workgroup whose dependencies are complete.  No grid barrier, no co-residency requirement.
//   visibility  producers: plain stores -> every wave drains its stores -> barrier -> lane 0: agent-scope release fence,
//               drained (asm), relaxed agent-scope fetch_add on done[p].  The workgroup that draws the last count folds the
//               InstanceNorm (the same wave-per-group order as in_finalize_kernel: identical bits), releases again and
//               stores ready[p].  Consumers: one lane polls ready[p] relaxed, ONE agent acquire, barrier, plain loads;
//               folded coefficients by agent-scope loads (never through the scalar cache).
//               (MI355X_MICROARCH.md, "inter-workgroup visibility"; cdna_hip_programming.md Guideline 16.)
//   state       the phase table lives in device memory next to the counters; a one-workgroup launch ahead of every chain
//               launch writes it and zeroes the counters (never a left-over of an earlier, possibly failed, launch).
constexpr int kKsChainMax = 11;            // (the set-up kernel takes the table by value: 11 phases x 336 bytes + header < 4 KB)
constexpr int kKsChainThreads = 512;
constexpr int kKsSyncWords = kKsChainSyncWords;   // [0] head, [16 + p] done, [32 + p] ready, [48 + p] time stamps (debug)

struct KsPhase {
    KsArgs A;
    int cfg;                  // instantiation (ks_cfg_id)
    int ksplit;               // waves per item
    int mgroups;              // channel-block groups per tile (mblocks / MBW)
    int mbw;
    int items, tickets;       // items = tiles * mgroups * batch; tickets = ceil(items / (8 / ksplit))
    int lds_stride;           // floats between the LDS regions of the sub-groups of a workgroup
    // InstanceNorm fold of this phase's output (what launch_in_finalize does behind a stand-alone launch)
    const float* gamma;
    const float* beta;
    float* scale;
    float* shift;
    float* mean;
    float* rstd;
    float* bound;
    int groups, per_group, channels;
    double count;
};

struct KsChainArgs {
    int n;
    int first[kKsChainMax + 1];   // first ticket of every phase; first[n] = total
    unsigned* sync;
    unsigned* nonfinite;
    KsPhase ph[kKsChainMax];
};

// configurations of the chain kernel: id = ((mode * 3 + nbi) * 4 + cini), nbi = log2(NB), cini = log2(Cin / 16)
#define PDS_KS_CHAIN_CONFIGS(F)                                                                                     \
    F(0, 1, 1, 4, 1, true, 16) F(0, 2, 1, 4, 1, true, 16) F(0, 4, 1, 4, 1, true, 16)                                 \
    F(0, 1, 2, 8, 1, true, 32) F(0, 2, 2, 8, 1, true, 32) F(0, 4, 2, 8, 1, true, 32)                                 \
    F(0, 1, 2, 8, 2, false, 64) F(0, 2, 2, 8, 2, false, 64) F(0, 4, 2, 8, 2, false, 64)                              \
    F(0, 1, 1, 8, 4, false, 128)                                                                                      \
    F(1, 1, 2, 4, 1, false, 16) F(1, 2, 2, 4, 1, false, 16) F(1, 4, 2, 4, 1, false, 16)                              \
    F(1, 1, 2, 8, 1, false, 32) F(1, 2, 2, 8, 1, false, 32) F(1, 4, 2, 8, 1, false, 32)                              \
    F(1, 1, 2, 8, 2, false, 64) F(1, 2, 2, 8, 2, false, 64) F(1, 4, 2, 8, 2, false, 64)                              \
    F(2, 1, 2, 4, 1, true, 16) F(2, 2, 2, 4, 1, true, 16) F(2, 4, 2, 4, 1, true, 16)                                 \
    F(2, 1, 2, 8, 1, true, 32) F(2, 2, 2, 8, 1, true, 32) F(2, 4, 2, 8, 1, true, 32)                                 \
    F(2, 1, 2, 8, 2, true, 64) F(2, 2, 2, 8, 2, true, 64) F(2, 4, 2, 8, 2, true, 64)                                 \
    F(2, 1, 2, 8, 4, true, 128) F(2, 2, 2, 8, 4, true, 128) F(2, 4, 2, 8, 4, true, 128)

__host__ __device__ constexpr int ks_cfg_id(int mode, int nb, int cin) {
    return (mode * 3 + (nb == 1 ? 0 : nb == 2 ? 1 : 2)) * 4 + (cin == 16 ? 0 : cin == 32 ? 1 : cin == 64 ? 2 : 3);
}

// one out-of-line function per configuration: compiled once each (the thirty bodies inlined into one kernel took the
// compiler five minutes and the register allocator 5 KB of scratch per lane)
#ifdef PDS_KS_CHAIN_MARK
#define PDS_KS_MARK(stage) do { if (tid == 0) __hip_atomic_store(C.sync + 1024 + blockIdx.x, (unsigned)(stage) | ((unsigned)ticket << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#else
#define PDS_KS_MARK(stage) do {} while (0)
#endif
// (function arguments arrive in vector registers: the wave-uniform ones are moved back to scalar registers, so that the
// phase description -- a table in device memory, written by ks_chain_setup_kernel -- is read with scalar loads)
template <class T>
__device__ __forceinline__ T* ks_uniform_ptr(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
template <int MODE, int NB, int MBW, int KSPLIT, int NKS, bool X>
__device__ __attribute__((noinline)) void ks_chain_item(const KsArgs* __restrict__ A, int tile, int mb0, int nb, int region_floats,
                                                        int gwave, int gtid, int active, const unsigned* ready,
                                                        unsigned* nonfinite, unsigned* marks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // the phase description through the CONSTANT address space: scalar loads into scalar registers (written by the set-up
    // launch, never during this one), so buffer descriptors and weight pointers derived from it are wave-uniform -- read
    // through the generic pointer every field was a flat vector load and every buffer load sat in a waterfall loop
    typedef const unsigned __attribute__((address_space(4))) * ConstWords;
    ConstWords src = reinterpret_cast<ConstWords>(reinterpret_cast<unsigned long long>(ks_uniform_ptr(A)));
    alignas(8) unsigned words[sizeof(KsArgs) / sizeof(unsigned)];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(KsArgs) / sizeof(unsigned)); ++i) words[i] = src[i];
    const KsArgs& args = *reinterpret_cast<const KsArgs*>(words);
    tile = __builtin_amdgcn_readfirstlane(tile);
    mb0 = __builtin_amdgcn_readfirstlane(mb0);
    nb = __builtin_amdgcn_readfirstlane(nb);
    gwave = __builtin_amdgcn_readfirstlane(gwave);
    active = __builtin_amdgcn_readfirstlane(active);
    region_floats = __builtin_amdgcn_readfirstlane(region_floats);
    ks_item<MODE, 1, 1, NB, MBW, KSPLIT, NKS, X>(args, tile, mb0, nb, lds + region_floats, gwave, gtid, active != 0,
                                                 KsInLaunch{ks_uniform_ptr(ready), ks_uniform_ptr(nonfinite), ks_uniform_ptr(marks)});
}

// writes the phase table of a launch into device memory and zeroes its synchronisation words (one small launch ahead of
// the chain kernel: a by-value table in the chain kernel's own arguments cannot be handed to out-of-line functions
// without a private copy per lane)
__global__ __launch_bounds__(256) void ks_chain_setup_kernel(const KsChainArgs C, unsigned* __restrict__ table,
                                                             unsigned* __restrict__ sync) {
    const unsigned* src = reinterpret_cast<const unsigned*>(&C);
    for (int i = threadIdx.x; i < (int)(sizeof(KsChainArgs) / sizeof(unsigned)); i += 256) table[i] = src[i];
    for (int i = threadIdx.x; i < kKsSyncWords; i += 256) sync[i] = 0u;
}

__global__ __launch_bounds__(kKsChainThreads) void conv3d_ks_chain_kernel(const KsChainArgs* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const KsChainArgs& C = *table;
    // (all LDS in the one dynamic array: the control words live behind the item regions)
    int* ctl = reinterpret_cast<int*>(lds + (160 * 1024 - 64) / sizeof(float));
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned* head = C.sync;
    unsigned* done = C.sync + 16;
    unsigned* ready = C.sync + 32;
    const int total = C.first[C.n];
    for (;;) {
        if (tid == 0) ctl[0] = (int)__hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int ticket = __builtin_amdgcn_readfirstlane(ctl[0]);
        __syncthreads();   // (ctl[0] is rewritten by the next draw / the last-arriver flag below)
        if (ticket >= total) break;
        if (ticket == 0 && tid == 0) C.sync[47] = (unsigned)wall_clock64();   // (debug: start stamp)
        PDS_KS_MARK(1);
        int p = 0;
        while (ticket >= C.first[p + 1]) ++p;
        const KsPhase& P = C.ph[p];
        const int per_wg = 8 / P.ksplit;                       // items side by side in this workgroup
        const int sub = wave / P.ksplit;
        int item = (ticket - C.first[p]) * per_wg + sub;
        const bool active = item < P.items;
        item = min(item, P.items - 1);
        const int tile = item % P.A.tiles, mg = (item / P.A.tiles) % P.mgroups, nb = item / (P.A.tiles * P.mgroups);
        const int region = sub * P.lds_stride;
        const int gwave = wave - sub * P.ksplit, gtid = tid - sub * P.ksplit * 64;
        const unsigned* wait_for = p > 0 ? ready + (p - 1) : nullptr;
#ifdef PDS_KS_CHAIN_TRACE
        if (tid == 0) printf("wg %d ticket %d phase %d cfg %d item %d tile %d mg %d\n", (int)blockIdx.x, ticket, p, P.cfg, item, tile, mg);
#endif
        switch (P.cfg) {
#define PDS_KS_CASE(MODE, NB, MBW, KSPLIT, NKS, X, CIN)                                                              \
    case ks_cfg_id(MODE, NB, CIN):                                                                                   \
        ks_chain_item<MODE, NB, MBW, KSPLIT, NKS, X>(&P.A, tile, mg * MBW, nb, region, gwave, gtid, active, wait_for, \
                                                     C.nonfinite, C.sync + 1024 + blockIdx.x);                       \
        break;
            PDS_KS_CHAIN_CONFIGS(PDS_KS_CASE)
#undef PDS_KS_CASE
            default: break;
        }
#ifdef PDS_KS_CHAIN_TRACE
        if (tid == 0) printf("wg %d ticket %d item done\n", (int)blockIdx.x, ticket);
#endif
        PDS_KS_MARK(3);
        // ---- publish this ticket; the workgroup that completes the phase folds its InstanceNorm --------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains
        __syncthreads();
        if (tid == 0) {
#ifndef PDS_KS_ABL_NOFENCE   // (timing ablation builds only: results are wrong without the release)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (restated where the compiler cannot drop it: Guideline 16 pitfall 12)
            const unsigned old = __hip_atomic_fetch_add(done + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old + 1u == (unsigned)P.tickets;
            if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            ctl[0] = last;
        }
        __syncthreads();
        const int last = __builtin_amdgcn_readfirstlane(ctl[0]);
        __syncthreads();
        PDS_KS_MARK(4);
        if (last) {
            PDS_KS_MARK(5);
#ifndef PDS_KS_ABL_NOFOLD    // (timing ablation builds only)
            if (P.scale) {
#else
            if (false) {
#endif
                const int lane = tid & 63;
                // (four groups of a wave at a time: this one workgroup folds the whole layer while every other waits)
                constexpr int kWaves = kKsChainThreads / 64, kBatch = 4;
                for (int g = wave; g < P.groups; g += kWaves * kBatch) {
                    const int left = (P.groups - g + kWaves - 1) / kWaves;
                    in_finalize_groups<kBatch>(P.A.partials, g, kWaves, left < kBatch ? left : kBatch, P.per_group, P.count,
                                               P.gamma, P.beta, P.channels, 1, P.scale, P.shift, P.mean, P.rstd, C.nonfinite,
                                               lane);
                }
                if (wave == 0 && P.bound) in_finalize_bound(P.gamma, P.beta, P.channels, P.count, P.bound, lane);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                C.sync[48 + p] = (unsigned)wall_clock64();     // (debug: phase completion stamps, 100 MHz)
                __hip_atomic_store(ready + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef PDS_KS_CHAIN_TRACE
                printf("wg %d phase %d READY\n", (int)blockIdx.x, p);
#endif
            }
            // NOT redundant: without a barrier between this lane-0 block and the lane-0 block at the head of the loop (the
            // next draw) the compiler threads the two together across the back edge -- lane 0 leaves the loop body early,
            // the other lanes of its wave run ahead into the next iteration's barriers and the workgroup never meets again
            // (found on hardware, round 6: every phase's last arriver hung here)
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
int conv3d_ks_tiles(const Geom& o);
int deconv3d_ks_tiles(const Geom& in, int cout);

namespace {

bool ks_enabled() {
    static const bool on = []() {  // PDS_CONV3D_KS=0: conv3d_mfma.hip serves these layers (A/B)
        const char* e = debug_switch("PDS_CONV3D_KS");
        return !(e && e[0] == '0');
    }();
    return on;
}

struct KsPlan {
    int nb;       // 16-column blocks per tile (tiles are single rows)
    int ksplit;   // waves per workgroup
    int nks;      // groups of four input channels per wave
};

// tiles are single rows of 16 / 32 / 64 columns; the waves of a workgroup take 4 (Cin <= 32), 8 (Cin = 64) or 16
// (Cin = 128) input channels each -- never more than 8 waves, the size of the chain kernel's workgroups (round 6: the
// 128-channel layers ran 16 waves x 8 channels; both forms of a layer must use ONE plan to give the same bits)
// 16-column blocks per tile: the ONE rule behind the plan, the tile counts and the record counts (32-wide tiles for rows
// of more than 32 columns measured neutral: 30.1 / 42.0 / 24.6 us against 29.8 / 44.2 / 23.8)
int ks_nb(int columns) { return columns <= 16 ? 1 : (columns <= 32 ? 2 : 4); }

KsPlan ks_plan(int cin, int columns) {
    KsPlan p;
    p.nb = ks_nb(columns);
    p.nks = cin >= 128 ? 4 : (cin >= 64 ? 2 : 1);
    p.ksplit = cin / (4 * p.nks);
    return p;
}

// channel blocks per item: two (every B operand feeds two MFMAs), except where the weight registers of a wave would not
// fit (the 128-channel convolutions: 4 x 27 fragments per block) and for 16 output channels
int ks_mbw(int mode, const KsPlan& p, int mblocks) {
    if (mode != 2 && p.nks == 4) return 1;
    if (mode == 0 && mblocks == 1) return 1;
    return 2;
}

template <int MODE, int NB, int KSPLIT, int NKS, int MBW, bool X>
int launch_ks(KsArgs A, int batch, hipStream_t s) {
    using G = KsGeom<MODE, 1, 1, NB>;
    A.cs = G::CS;
    constexpr int NACC = MBW * NB;
    const size_t staging = (size_t)A.Cin * G::CS * sizeof(float);
    const size_t reduce = (size_t)KSPLIT * NACC * 64 * sizeof(f32x4);
    size_t lds_bytes = staging > reduce ? staging : reduce;
    if (lds_bytes < 4096) lds_bytes = 4096;
    if (lds_bytes > 160 * 1024) return set_error(-1, "conv3d_ks: tile does not fit in LDS (%zu bytes)", lds_bytes);
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_ks_kernel<MODE, 1, 1, NB, MBW, KSPLIT, NKS, X>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    }
    dim3 grid(A.tiles, A.mblocks / MBW, batch);
    hipLaunchKernelGGL((conv3d_ks_kernel<MODE, 1, 1, NB, MBW, KSPLIT, NKS, X>), grid, dim3(64 * KSPLIT), lds_bytes, s, A);
    return check_launch("conv3d_ks");
}

// which configurations take the X form: measured per configuration on the hot path's layers (us per launch, fp32 -> X):
//   3x3x3 s1, 4 channels per wave   34.9 -> 30.0 (32 -> 32 at 12x36x60)       taken
//   3x3x3 s1, 8 channels per wave   20.4 -> 22.1, 16.2 -> 24.1                 not taken (144+ weight registers: spills)
//   3x3x3 s2                        27.1 -> 33.1, 17.1 -> 17.8, 15.6 -> 15.2   not taken
//   k4 s2 transposed (cell form)    49.1 -> 41.6, 33.7 -> 29.3, 31.3 -> 30.8   taken
constexpr bool ks_split_config(int mode, int nks) { return mode == 2 || (mode == 0 && nks == 1); }

template <int MODE, int NB, bool X>
int dispatch_split(const KsPlan& p, const KsArgs& A, int batch, int mbw, hipStream_t s) {
    if constexpr (!X || ks_split_config(MODE, 1)) {
        if (p.ksplit == 4 && p.nks == 1 && mbw == 1 && MODE == 0) return launch_ks<0, NB, 4, 1, 1, X>(A, batch, s);
        if (p.ksplit == 4 && p.nks == 1 && mbw == 2) return launch_ks<MODE, NB, 4, 1, 2, X>(A, batch, s);
        if (p.ksplit == 8 && p.nks == 1 && mbw == 2) return launch_ks<MODE, NB, 8, 1, 2, X>(A, batch, s);
    }
    if constexpr (!X || ks_split_config(MODE, 2)) {
        if (p.ksplit == 8 && p.nks == 2 && mbw == 2) return launch_ks<MODE, NB, 8, 2, 2, X>(A, batch, s);
        if constexpr (MODE == 2) {
            if (p.ksplit == 8 && p.nks == 4 && mbw == 2) return launch_ks<2, NB, 8, 4, 2, X>(A, batch, s);
        } else if constexpr (MODE == 0 && NB == 1 && !X) {
            if (p.ksplit == 8 && p.nks == 4 && mbw == 1) return launch_ks<0, 1, 8, 4, 1, false>(A, batch, s);
        }
    }
    return set_error(-1, "conv3d_ks: no configuration for %d x %d channels per wave (mode %d)", p.ksplit, p.nks, MODE);
}

template <int MODE>
int dispatch_ks(const KsPlan& p, const KsArgs& A, int batch, int mbw, hipStream_t s, bool x) {
    if (x) {
        if (p.nb == 1) return dispatch_split<MODE, 1, true>(p, A, batch, mbw, s);
        if (p.nb == 2) return dispatch_split<MODE, 2, true>(p, A, batch, mbw, s);
        return dispatch_split<MODE, 4, true>(p, A, batch, mbw, s);
    }
    if (p.nb == 1) return dispatch_split<MODE, 1, false>(p, A, batch, mbw, s);
    if (p.nb == 2) return dispatch_split<MODE, 2, false>(p, A, batch, mbw, s);
    return dispatch_split<MODE, 4, false>(p, A, batch, mbw, s);
}

// X form (split fp16 operands on v_mfma_f32_16x16x16_f16) when every source carries a range certificate
bool ks_use_split(const Src& a, const Src& b, int mode, const KsPlan& p) {
    if (!ks_split_config(mode, p.nks)) return false;
    static const bool enabled = []() {  // PDS_CONV3D_KSX=0: every launch on the fp32 matrix pipe (A/B, debugging)
        const char* e = debug_switch("PDS_CONV3D_KSX");
        return !(e && e[0] == '0');
    }();
    return enabled && a.bounded && (b.p == nullptr || b.bounded);
}

size_t ks_split_dwords(int cin, int mblocks, int ksteps) { return (size_t)(cin / 4) * ksteps * mblocks * 2 * 64 * 2; }

int ks_cs(int mode, int nb) {
    const int sx = mode == 1 ? 2 : 1, ext = mode == 2 ? 2 : 3;
    const int npos = ext * ext * ((16 * nb - 1) * sx + ext);
    return sx == 1 ? ((npos + 15) / 32 * 32 + 16) : (npos | 1);
}

size_t ks_lds_bytes(int mode, const KsPlan& p, int cin) { return (size_t)cin * ks_cs(mode, p.nb) * sizeof(float); }

bool ks_shape_ok(int mode, int cin, int vchannels, int columns, size_t in_elems, size_t out_elems, int batch) {
    if (cin != 16 && cin != 32 && cin != 64 && cin != 128) return false;
    if (vchannels % 32 != 0 && !(vchannels == 16 && mode == 0)) return false;   // two channel blocks per workgroup
    if (in_elems >= ((size_t)1 << 30) || out_elems >= ((size_t)1 << 30)) return false;   // 32-bit byte offsets
    if (batch > 65535) return false;
    const KsPlan p = ks_plan(cin, columns);
    if (p.nks == 4 && mode == 1) return false;            // (128 -> 256 stride 2: no configuration; conv3d_mfma.hip)
    if (p.nks == 4 && mode == 0 && p.nb != 1) return false;
    return ks_lds_bytes(mode, p, cin) <= 160 * 1024;
}

// the launch description of a layer: arguments, plan, pack job
struct KsLaunch {
    KsArgs A;
    KsPlan plan;
    int mode = 0, mbw = 2, batch = 1;
    bool x = false;
    PackJob job;
};

int ks_describe_conv(const ConvLayer& L, KsLaunch& K) {
    if (!L.packed) return set_error(-1, "conv3d_ks: packed weights missing");
    K.mode = L.stride == 2 ? 1 : 0;
    KsArgs& A = K.A;
    A.a = L.a;
    A.b = L.b;
    A.wpk = L.packed;
    A.bias = L.bias;
    A.out = L.out;
    A.partials = L.partials;
    A.Cin = L.in.c;
    A.Di = L.in.d;
    A.Hi = L.in.h;
    A.Wi = L.in.w;
    A.Cout = L.out_g.c;
    A.Do = L.out_g.d;
    A.Ho = L.out_g.h;
    A.Wo = L.out_g.w;
    A.lrelu = L.lrelu;
    A.mblocks = A.Cout / 16;
    K.plan = ks_plan(A.Cin, A.Wo);
    K.mbw = ks_mbw(K.mode, K.plan, A.mblocks);
    K.batch = L.in.n;
    A.tiles_x = (A.Wo + 16 * K.plan.nb - 1) / (16 * K.plan.nb);
    A.tiles_y = A.Ho;
    A.tiles = conv3d_ks_tiles(L.out_g);
    A.cs = ks_cs(K.mode, K.plan.nb);
    K.x = ks_use_split(L.a, L.b, K.mode, K.plan);
    const size_t split_total = ks_split_dwords(A.Cin, A.mblocks, 9) + 16;
    A.wtail = L.packed + split_total - 16;
    PackJob& j = K.job;
    j.src = L.weight;
    j.dst = L.packed;
    j.cout = A.Cout;
    j.cin = A.Cin;
    j.mblocks = A.mblocks;
    j.kc = 4;
    j.taps = 27;
    j.mode = K.x ? 8 : 0;
    j.total = K.x ? (int)split_total : (int)((size_t)(A.Cin / 4) * 27 * A.Cout * 4);
    if (K.x && (!L.a.bounded || (L.b.p && !L.b.bounded)))
        return set_error(-1, "conv3d_ks: split form without a range bound");
    return 0;
}

int ks_describe_deconv(const DeconvLayer& L, KsLaunch& K) {
    if (!L.packed) return set_error(-1, "deconv3d_ks: packed weights missing");
    K.mode = 2;
    KsArgs& A = K.A;
    A.a = L.a;
    A.b = no_src();
    A.wpk = L.packed;
    A.bias = L.bias;
    A.out = L.out;
    A.partials = L.partials;
    A.Cin = L.in.c;
    A.Di = L.in.d;
    A.Hi = L.in.h;
    A.Wi = L.in.w;
    A.Cout = L.out_g.c;
    A.Do = L.out_g.d;
    A.Ho = L.out_g.h;
    A.Wo = L.out_g.w;
    A.lrelu = L.lrelu;
    A.mblocks = 8 * A.Cout / 16;
    K.plan = ks_plan(A.Cin, A.Wi + 1);
    K.mbw = ks_mbw(2, K.plan, A.mblocks);
    K.batch = L.in.n;
    A.tiles_x = (A.Wi + 1 + 16 * K.plan.nb - 1) / (16 * K.plan.nb);
    A.tiles_y = A.Hi + 1;
    A.tiles = deconv3d_ks_tiles(L.in, A.Cout);
    A.cs = ks_cs(2, K.plan.nb);
    K.x = ks_use_split(L.a, no_src(), 2, K.plan);
    const size_t split_total = ks_split_dwords(A.Cin, A.mblocks, 2) + 16;
    A.wtail = L.packed + split_total - 16;
    PackJob& j = K.job;
    j.src = L.weight;
    j.dst = L.packed;
    j.cout = A.Cout;
    j.cin = A.Cin;
    j.mblocks = A.mblocks;
    j.kc = 4;
    j.taps = 8;
    j.mode = K.x ? 9 : 5;
    j.total = K.x ? (int)split_total : (int)((size_t)(A.Cin / 4) * 8 * 8 * A.Cout * 4);
    if (K.x && !L.a.bounded) return set_error(-1, "deconv3d_ks: split form without a range bound");
    return 0;
}

// pack (per the layer's phase) and, unless only collecting, launch stand-alone
int ks_pack_and_launch(const KsLaunch& K, PackSink* sink, hipStream_t s) {
    const PackPhase phase = sink ? sink->phase : kPackInline;
    if (phase != kPackDone) {
        if (phase == kPackCollect) return sink->push(K.job) ? 0 : set_error(-1, "pack job table full");
        if (int rc = launch_multi_pack(&K.job, 1, s)) return rc;
    }
    const Src& a = K.A.a;
    const Src& b = K.A.b;
    if (K.x && (!a.bound || a.bound_n <= 0 || (b.p && (!b.bound || b.bound_n <= 0))))
        return set_error(-1, "conv3d_ks: split form without a range bound");
    return K.mode == 2 ? dispatch_ks<2>(K.plan, K.A, K.batch, K.mbw, s, K.x)
                       : K.mode == 1 ? dispatch_ks<1>(K.plan, K.A, K.batch, K.mbw, s, K.x)
                                     : dispatch_ks<0>(K.plan, K.A, K.batch, K.mbw, s, K.x);
}

// is (mode, nb, cin, mbw, ksplit, nks, x) one of the chain kernel's instantiations?
bool ks_chain_has(const KsLaunch& K) {
    const int cin = K.A.Cin;
#define PDS_KS_HAS(MODE, NB, MBW, KSPLIT, NKS, X, CIN)                                                               \
    if (K.mode == MODE && K.plan.nb == NB && cin == CIN)                                                             \
        return K.mbw == MBW && K.plan.ksplit == KSPLIT && K.plan.nks == NKS && K.x == X;
    PDS_KS_CHAIN_CONFIGS(PDS_KS_HAS)
#undef PDS_KS_HAS
    return false;
}

}  // namespace

bool conv3d_ks_supported(const ConvLayer& L) {
    if (!ks_enabled()) return false;
    if (L.kd != 3 || L.stat_per_plane) return false;
    if ((L.a.scale && L.a.per_plane) || (L.b.scale && L.b.per_plane)) return false;
    static const size_t volume_limit = []() {   // PDS_CONV3D_KS_LIMIT: largest output volume served (experiments)
        const char* e = debug_switch("PDS_CONV3D_KS_LIMIT");
        return e ? (size_t)atol(e) : (size_t)30000;
    }();
    // the large 16-channel levels keep the tile-per-workgroup kernel: plenty of tiles there, and K is short
    if ((size_t)L.out_g.d * L.out_g.h * L.out_g.w > volume_limit) return false;
    return ks_shape_ok(L.stride == 2 ? 1 : 0, L.in.c, L.out_g.c, L.out_g.w, (size_t)L.in.c * L.in.d * L.in.h * L.in.w,
                       L.out_g.numel() / (L.out_g.n ? L.out_g.n : 1), L.in.n);
}

int conv3d_ks_tiles(const Geom& o) {
    const int nb = ks_nb(o.w);
    return ((o.w + 16 * nb - 1) / (16 * nb)) * o.h * o.d;
}

// (the larger of the two forms + the 16-dword tail of the split form: the choice follows the sources' certificates)
size_t conv3d_ks_packed_floats(int cin, int vchannels, int taps) {
    const size_t plain = (size_t)(cin / 4) * taps * vchannels * 4;
    const size_t split = ks_split_dwords(cin, vchannels / 16, taps == 27 ? 9 : 2) + 16;
    return plain > split ? plain : split;
}

int launch_conv3d_ks(const ConvLayer& L, hipStream_t s) {
    KsLaunch K;
    if (int rc = ks_describe_conv(L, K)) return rc;
    return ks_pack_and_launch(K, L.sink, s);
}

bool deconv3d_ks_supported(const DeconvLayer& L) {
    if (!ks_enabled()) return false;
    if (L.kd != 4 || L.b.p != nullptr) return false;
    if (L.a.scale && L.a.per_plane) return false;
    if ((size_t)L.in.d * L.in.h * L.in.w > 30000) return false;
    return ks_shape_ok(2, L.in.c, 8 * L.out_g.c, L.in.w + 1, (size_t)L.in.c * L.in.d * L.in.h * L.in.w,
                       L.out_g.numel() / (L.out_g.n ? L.out_g.n : 1), L.in.n);
}

int deconv3d_ks_tiles(const Geom& in, int cout) {
    const int cells = in.w + 1, nb = ks_nb(cells);
    return ((cells + 16 * nb - 1) / (16 * nb)) * (in.h + 1) * (in.d + 1);
}

int launch_deconv3d_ks(const DeconvLayer& L, hipStream_t s) {
    KsLaunch K;
    if (int rc = ks_describe_deconv(L, K)) return rc;
    return ks_pack_and_launch(K, L.sink, s);
}

// ---- the chain (common.hpp: KsChain) ---------------------------------------------------------------------------------
namespace {
int ks_chain_append(KsChain& chain, const KsLaunch& K, const KsChainFold& fold) {
    if (chain.count >= kKsChainMax) return set_error(-1, "conv3d_ks chain: more than %d layers", kKsChainMax);
    static_assert(sizeof(KsPhase) <= sizeof(chain.storage[0]), "KsChain::storage too small for a phase");
    KsPhase& P = *reinterpret_cast<KsPhase*>(&chain.storage[chain.count]);
    P.A = K.A;
    P.cfg = ks_cfg_id(K.mode, K.plan.nb, K.A.Cin);
    P.ksplit = K.plan.ksplit;
    P.mbw = K.mbw;
    P.mgroups = K.A.mblocks / K.mbw;
    P.items = K.A.tiles * P.mgroups * K.batch;
    const int per_wg = 8 / P.ksplit;
    P.tickets = (P.items + per_wg - 1) / per_wg;
    const size_t staging = (size_t)K.A.Cin * K.A.cs;
    const size_t reduce = (size_t)P.ksplit * (K.mbw * K.plan.nb) * 64 * 4;
    size_t region = staging > reduce ? staging : reduce;
    if (region < 1024) region = 1024;
    region = (region + 63) & ~(size_t)63;
    if (region * per_wg * sizeof(float) > 160 * 1024 - 64)
        return set_error(-1, "conv3d_ks chain: %d items of %zu bytes do not fit in LDS", per_wg, region * sizeof(float));
    P.lds_stride = (int)region;
    P.gamma = fold.gamma;
    P.beta = fold.beta;
    P.scale = fold.scale;
    P.shift = fold.shift;
    P.mean = fold.mean;
    P.rstd = fold.rstd;
    P.bound = fold.bound;
    P.groups = fold.groups;
    P.per_group = fold.per_group;
    P.channels = fold.channels;
    P.count = fold.count;
    ++chain.count;
    return 0;
}
}  // namespace

// OFF by default: measured SLOWER than one launch + in_finalize per layer (round 6, 960x540 D = 192, the eleven layers of
// levels 1-3: 706 us against 280 us of kernels + 60 us of in_finalize launches + ~45 us of boundaries; without its release
// fences 620, without the in-launch InstanceNorm fold 493, without both 394 -- docs/LAB_NOTES.md, round 6).  A layer of
// this network ends in a reduction over its whole output, so every phase boundary is a grid-wide hand-off: publish, fold
// by ONE workgroup, notify, acquire cost more inside a launch than a kernel boundary + a 5 us launch that folds with one
// wave per channel.  PDS_CONV3D_KS_CHAIN=1 (with PDS_DEBUG_SWITCHES=1) selects it; bit-identical results
// (tools/chain_check.py, tests/test_gpu_chain.py).
bool conv3d_ks_chain_enabled() {
    static const bool on = []() {
        const char* e = debug_switch("PDS_CONV3D_KS_CHAIN");
        return e && e[0] == '1';
    }();
    return on;
}

int conv3d_ks_chain_add(KsChain& chain, const ConvLayer& L, const KsChainFold& fold, bool* taken) {
    *taken = false;
    KsLaunch K;
    if (int rc = ks_describe_conv(L, K)) return rc;
    if (!ks_chain_has(K) || chain.count >= kKsChainMax) return 0;
    if (K.x && (!L.a.bound || L.a.bound_n <= 0 || (L.b.p && (!L.b.bound || L.b.bound_n <= 0)))) return 0;
    if (int rc = ks_chain_append(chain, K, fold)) return rc;
    *taken = true;
    return 0;
}

int deconv3d_ks_chain_add(KsChain& chain, const DeconvLayer& L, const KsChainFold& fold, bool* taken) {
    *taken = false;
    KsLaunch K;
    if (int rc = ks_describe_deconv(L, K)) return rc;
    if (!ks_chain_has(K) || chain.count >= kKsChainMax) return 0;
    if (K.x && (!L.a.bound || L.a.bound_n <= 0)) return 0;
    if (int rc = ks_chain_append(chain, K, fold)) return rc;
    *taken = true;
    return 0;
}

// debugging / profiling aid (pds_debug_chain_stamps): the synchronisation words and phase count of the last chain launch
static unsigned* g_last_chain_sync = nullptr;
static int g_last_chain_phases = 0;
int ks_chain_debug_stamps(unsigned* out, int capacity) {
    if (!g_last_chain_sync) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return set_error(-1, "chain stamps: hipDeviceSynchronize failed");
    unsigned words[kKsSyncWords];
    if (hipMemcpy(words, g_last_chain_sync, sizeof(words), hipMemcpyDeviceToHost) != hipSuccess)
        return set_error(-1, "chain stamps: hipMemcpy failed");
    const int n = g_last_chain_phases < capacity ? g_last_chain_phases : capacity;
    for (int i = 0; i < n; ++i) out[i] = words[48 + i] - words[47];   // ticks since the first ticket was drawn
    // (behind them, when there is room: the raw words -- head, done[], ready[] -- for debugging a stuck chain)
    if (capacity >= n + kKsSyncWords)
        for (int i = 0; i < kKsSyncWords; ++i) out[n + i] = words[i];
    return n;
}

int conv3d_ks_chain_launch(KsChain& chain, unsigned* sync_words, hipStream_t s) {
    if (chain.count == 0) return 0;
    if (!sync_words) return set_error(-1, "conv3d_ks chain: no synchronisation words");
    KsChainArgs C;
    C.n = chain.count;
    int t = 0;
    for (int p = 0; p < chain.count; ++p) {
        C.ph[p] = *reinterpret_cast<const KsPhase*>(&chain.storage[p]);
        C.first[p] = t;
        t += C.ph[p].tickets;
    }
    for (int p = chain.count; p <= kKsChainMax; ++p) C.first[p] = t;
    C.sync = sync_words;
    C.nonfinite = nonfinite_counter(s);
    g_last_chain_sync = sync_words;
    g_last_chain_phases = chain.count;
    chain.count = 0;
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    static int cu_count = 256;
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_ks_chain_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cu_count = n;
    }
    static_assert(sizeof(KsChainArgs) % sizeof(unsigned) == 0 &&
                      sizeof(KsChainArgs) <= (kKsChainStateWords - kKsChainSyncWords) * sizeof(unsigned),
                  "the chain's state carve does not hold the phase table");
    unsigned* table = sync_words + kKsSyncWords;
    C.sync = sync_words;
    hipLaunchKernelGGL(ks_chain_setup_kernel, dim3(1), dim3(256), 0, s, C, table, sync_words);
    if (int rc = check_launch("conv3d_ks_chain_setup")) return rc;
    const int grid = t < cu_count ? t : cu_count;    // one 8-wave workgroup per CU (256 registers per lane, 160 KB of LDS)
    const int probe = probe_before("conv3d_ks_chain", s);
    hipLaunchKernelGGL(conv3d_ks_chain_kernel, dim3(grid), dim3(kKsChainThreads), 160 * 1024, s,
                       reinterpret_cast<const KsChainArgs*>(table));
    probe_after(probe, grid, s);
    return check_launch("conv3d_ks_chain");
}

}  // namespace pds
