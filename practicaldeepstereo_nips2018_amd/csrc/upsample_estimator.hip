// Eval-mode fusion of Regularization's last layer with SubpixelMap:
//   cost = ConvTranspose3d(k=(3,4,4), s=(1,2,2), p=1)(InstanceNorm(half))   reference regularization.py:90-92,125-126
//   disparity = SubpixelMap(cost)                                           reference estimator.py:45-91, network.py:50-51
// The [B, 2D, 4h, 4w] cost volume (212 MB at 960x540, D=192) is never written or re-read: every lane
// sweeps the planes once for its 2 x 4 output pixels, accumulating the transposed convolution on the VALU
// (48 MACs per output voxel, weights as scalar operands) and feeding each finished plane to the
// streaming arg-max / soft-arg-max state of estimator.hip.
//
//   workgroup   ONE wave = input tile of 4 rows x 32 columns of the half-resolution tensor (output 8 x 64);
//               lane (r, cp) owns input positions (i0 + r, j0 + 2cp .. 2cp+1) -> output rows 2i, 2i+1 and four
//               consecutive output columns (16-byte disparity stores, 256 B contiguous per 16 lanes).
//   planes      input plane p contributes to output planes p-1, p, p+1 (od = id - 1 + kd), so three running
//               accumulators rotate; plane p-1 is complete after plane p.
//   LDS         double-buffered halo tile [C][6][34] of the normalised plane (InstanceNorm of the previous
//               layer and zero padding applied while staging); plane p+1 is fetched while p is consumed.
//   taps        out = 2*i - 1 + k:  even output: (i, k=1), (i-1, k=3);  odd output: (i, k=2), (i+1, k=0).
#include <type_traits>

#include "common.hpp"

namespace pds {

namespace {

constexpr int TR = 4, TC = 32;            // input tile
constexpr int HR = TR + 2, HC = TC + 2;   // halo tile
constexpr int NPOS = HR * HC;             // 204
constexpr int HS = 48;                    // LDS floats per halo row: == 16 (mod 32), the four rows of a wave read disjoint banks
constexpr int TILE_CH = HR * HS;          // LDS floats per channel
constexpr int POS = (NPOS + 63) / 64;     // 4
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct FusedArgs {
    const float* __restrict__ in;     // raw upsample_half output [B, C, D, Hi, Wi]
    const float* __restrict__ scale;  // [B*C] folded InstanceNorm
    const float* __restrict__ shift;
    const float* __restrict__ w;      // [parity][C][3][2][kw order 1, 2, 3, 0]: the layer's weights, see upsample_weight_pairs_kernel
    const float* __restrict__ bias;   // [1]
    float* __restrict__ disp;         // [B, 2Hi, 2Wi]
    float* __restrict__ cost;         // WRITE_COST variant: [B, D, 2Hi, 2Wi] instead of the disparity
    int D, Hi, Wi;
    int lo, hi;                       // tap range of the estimator
    float step;
    int crop_top, crop_left;          // SizeAdapter.unpad folded into the store: disp is [B, 2Hi - top, 2Wi - left]
};

}  // namespace

// WRITE_COST = true: training-mode forward, every finished plane is stored to the cost volume and no
// estimator state is kept (T is ignored).
// A tile is swept by 2 x DS waves: wave (PY, part) owns the output rows of parity PY (4 pixels per lane instead of 2 x 4)
// and the candidate planes [part D / DS, (part + 1) D / DS).  Round 5: one wave per (tile, parity) put 2 160 waves on
// 1 024 SIMDs -- some SIMDs carried three sweeps, most two, and the launch lasted as long as the three (the kernel is bound by
// VALU issue and by the un-hidden scalar-load / LDS waits of so few waves, NOT by its halo re-reads: with the halo loads
// redirected to the tile's own interior, -DPDS_UPS_NOHALO, the fabric traffic drops from 721 to 212 MB and the launch takes
// the same 154 us; profiles/r05_upsample_fetch_calibration.txt).  With the plane range cut in DS = 2 parts the chip carries
// 4 320 half-length sweeps (4.2 per SIMD, at most 5): better balance, twice the waves to hide each other's waits.  A part
// sweeps T + 1 planes beyond either end of its range (window values only, never candidates), so the parts' states are exact
// and the merge is a comparison of their maxima (ties: the lower plane, as estimator.py's argmax).
constexpr int UHALF = 128;                               // threads that stage one part's planes (two waves)
constexpr int UPOS = (NPOS + UHALF - 1) / UHALF;         // 2 staged positions per thread and channel
constexpr int EO = HC / 2;                               // a halo row is stored as [even columns 17][odd columns 17]

// One plane of the streaming arg-max for one pixel, T = 2 (see the sweep): selects on SGPR-pair masks, hand-scheduled.
// CAND: plane k is a candidate of this part (uniform); otherwise it only feeds the neighbours of an earlier maximum.
template <bool CAND>
__device__ __forceinline__ void estimator_update2(float v, float h1, float h2, int k, int km1, int km2, float& best, int& bi,
                                                  float& bp0, float& bp1, float& bn0, float& bn1) {
    unsigned long long m0, m1, mu;
    if constexpr (CAND) {
        asm volatile(
            "v_cmp_eq_u32_e64 %[m0], %[bi], %[km1]\n\t"
            "v_cmp_eq_u32_e64 %[m1], %[bi], %[km2]\n\t"
            "v_cmp_gt_f32_e64 %[mu], %[v], %[best]\n\t"
            "v_cndmask_b32_e64 %[bn0], %[bn0], %[v], %[m0]\n\t"
            "v_cndmask_b32_e64 %[bn1], %[bn1], %[v], %[m1]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32_e64 %[best], %[best], %[v], %[mu]\n\t"
            "v_cndmask_b32_e64 %[bi], %[bi], %[kv], %[mu]\n\t"
            "v_cndmask_b32_e64 %[bp0], %[bp0], %[h1], %[mu]\n\t"
            "v_cndmask_b32_e64 %[bp1], %[bp1], %[h2], %[mu]"
            : [m0] "=&s"(m0), [m1] "=&s"(m1), [mu] "=&s"(mu), [best] "+v"(best), [bi] "+v"(bi), [bp0] "+v"(bp0),
              [bp1] "+v"(bp1), [bn0] "+v"(bn0), [bn1] "+v"(bn1)
            : [v] "v"(v), [h1] "v"(h1), [h2] "v"(h2), [kv] "v"(k), [km1] "s"(km1), [km2] "s"(km2));
    } else {
        asm volatile(
            "v_cmp_eq_u32_e64 %[m0], %[bi], %[km1]\n\t"
            "v_cmp_eq_u32_e64 %[m1], %[bi], %[km2]\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32_e64 %[bn0], %[bn0], %[v], %[m0]\n\t"
            "v_cndmask_b32_e64 %[bn1], %[bn1], %[v], %[m1]"
            : [m0] "=&s"(m0), [m1] "=&s"(m1), [bn0] "+v"(bn0), [bn1] "+v"(bn1)
            : [v] "v"(v), [bi] "v"(bi), [km1] "s"(km1), [km2] "s"(km2));
    }
}

// Round 5b: the sweep's bookkeeping is compile-time indexed.  Profile of the first form (rocprofv3, config 2): 205 VALU
// instructions per plane step and wave for 96 packed FMAs, every one a 4-cycle issue -- the launch is VALU-issue-bound, so
// the 109 others were the lever:
//   * the three running accumulators and the estimator's window are ONE ring of R = 3 + T register pairs per pixel pair,
//     indexed by the step modulo R (plane loop unrolled by lcm(2, R)): output plane k lives in slot (k - pb + 1) mod R from
//     its first contribution until T planes later, so nothing is ever moved (was: 4 window moves per pixel, 6 accumulator
//     moves and ~20 loop-carried copies per step);
//   * the candidate is the plane that has just been finished (no delayed centre): on a new maximum the T previous planes are
//     taken from the ring, the T following ones are captured when they arrive (`bi == k - 1 - t`); a stale capture of an older
//     maximum is always overwritten or, at the end of the range, masked by the plane-index test of the soft-arg-max;
//   * loads are buffer loads whose plane / channel offset is a scalar operand (no 64-bit vector address arithmetic), positions
//     outside the image read as zero through the range check and their InstanceNorm shift is zero in a per-thread copy:
//     the stash is one fma per element (was: fma + select + a move of the scalar shift);
//   * a halo row is stored as [even columns][odd columns], so the pairs (v0, v2) and (v1, v3) a packed FMA wants are adjacent
//     LDS words (was: ds_read2_b64 + 8 moves per channel).
template <int CIN, int T, bool WRITE_COST, int PY, int DS>
__device__ __forceinline__ void upsample_sweep(const FusedArgs& A, float (*tile)[CIN][TILE_CH], float* merge, const int part) {
    constexpr int R = 3 + T;                        // ring of output planes: 3 accumulating / finishing + T of history
    constexpr int U = (R % 2 == 0) ? R : 2 * R;     // steps per unrolled group: the staging registers alternate as well
    const int tid = threadIdx.x & (UHALF - 1);
    const int lane = tid & 63;
    const int r = lane >> 4, cp = lane & 15;
    const int i0 = blockIdx.y * TR, j0 = blockIdx.x * TC;
    const int b = blockIdx.z;
    const size_t plane = (size_t)A.Hi * A.Wi;
    const size_t cstride = (size_t)A.D * plane;
    const unsigned plane_bytes = (unsigned)(plane * sizeof(float)), cbytes = (unsigned)(cstride * sizeof(float));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A.in + (size_t)b * CIN * cstride), 0, (int)(CIN * cstride * sizeof(float)), 0x00020000);

    // candidate planes of this part, the input planes it sweeps and the common number of steps (barriers are shared)
    const int per = (A.D + DS - 1) / DS;
    const int lo = DS == 1 ? 0 : part * per, hi = DS == 1 ? A.D : min(A.D, lo + per);
    const int pb = DS == 1 ? 0 : max(lo - T - 1, 0);         // first input plane: output plane pb + 1 is the first complete one
    const int pe = WRITE_COST ? A.D : hi + T;                // the last plane that matters: k = hi - 1 + T is complete after it
    int nsteps = pe - pb + 1;
    if (DS > 1) {
#pragma unroll
        for (int q = 0; q < DS; ++q) {
            const int lq = q * per, hq = min(A.D, lq + per);
            nsteps = max(nsteps, hq + T - max(lq - T - 1, 0) + 1);
        }
    }

    unsigned goff[UPOS];
    int loff[UPOS];
    bool inside[UPOS];
#pragma unroll
    for (int k = 0; k < UPOS; ++k) {
        const int p = min(tid + k * UHALF, NPOS - 1);
        const int rr = p / HC, cc = p % HC;
        const int y = i0 - 1 + rr, x = j0 - 1 + cc;
        inside[k] = y >= 0 && y < A.Hi && x >= 0 && x < A.Wi;
#ifdef PDS_UPS_NOHALO   // FETCH_SIZE calibration (wrong results by design): every tile reads its own 4 x 32 interior only,
                        // with the same load instructions -- the launch then fetches exactly the input tensor's bytes
        goff[k] = (unsigned)(min(max(y, i0), min(i0 + TR, A.Hi) - 1) * A.Wi + min(max(x, j0), min(j0 + TC, A.Wi) - 1)) * 4u;
#else
        // (outside the image: beyond the resource's range for every scalar offset, the load returns zero)
        goff[k] = inside[k] ? (unsigned)(y * A.Wi + x) * 4u : 0x80000000u;
#endif
        loff[k] = rr * HS + (cc & 1) * EO + (cc >> 1);
    }
    float sc[CIN], shv[CIN][UPOS];
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
        sc[c] = A.scale ? A.scale[b * CIN + c] : 1.f;
        const float sh = A.scale ? A.shift[b * CIN + c] : 0.f;
#pragma unroll
        for (int k = 0; k < UPOS; ++k) shv[c][k] = inside[k] ? sh : 0.f;   // literal zero padding: 0 * scale + 0
    }
    const float bias = A.bias ? A.bias[0] : 0.f;

    // Two register stages: plane p + 2 is requested while plane p is consumed and is stashed one iteration later, so a
    // request has a whole plane step to land; the two stages swap roles from step to step.
    float st[2][CIN][UPOS];
#ifdef PDS_UPS_NOLDSWRITE
    float sink = 0.f;
#define PDS_UPS_STORE(dst_, v_) sink += (v_);
#else
#define PDS_UPS_STORE(dst_, v_) dst_ = (v_);
#endif
#define PDS_FETCHP(set_, p_)                                                       \
    _Pragma("unroll") for (int c = 0; c < CIN; ++c) _Pragma("unroll") for (int k = 0; k < UPOS; ++k) \
        st[set_][c][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(               \
            rsrc, goff[k], (unsigned)(p_) * plane_bytes + c * cbytes, 0));
#define PDS_STASHP(set_, buf_)                                                     \
    _Pragma("unroll") for (int c = 0; c < CIN; ++c) _Pragma("unroll") for (int k = 0; k < UPOS; ++k) \
        PDS_UPS_STORE(tile[buf_][c][loff[k]], fmaf(sc[c], st[set_][c][k], shv[c][k]))

#ifdef PDS_UPS_TIMING
    long long tm[4] = {0, 0, 0, 0};
#endif
    // estimator state of the lane's 4 pixels (output row 2 i + PY, four consecutive columns)
    float best[4], bprev[4][T], bnext[4][T];
    int bi[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        best[o] = -INFINITY;
        bi[o] = lo;
#pragma unroll
        for (int t = 0; t < T; ++t) bprev[o][t] = bnext[o][t] = -INFINITY;
    }
    // ring[slot][pixel pair]: register pairs, the transposed convolution runs on packed fp32 FMAs (two output columns each)
    f32x2 ring[R][2];
#pragma unroll
    for (int q = 0; q < R; ++q) ring[q][0] = ring[q][1] = f32x2{bias, bias};

    PDS_FETCHP(1, pb)
    if (pb + 1 < A.D) PDS_FETCHP(0, pb + 1)
    PDS_STASHP(1, 0)
    __syncthreads();

    // halo row r, halo columns 2cp .. 2cp + 3 (input columns j0 + 2cp - 1 ..): the even pair (v0, v2) = E[cp], E[cp + 1] and
    // the odd pair (v1, v3) = O[cp], O[cp + 1] are the second operands of the packed FMAs below as they stand
    const int lbase = r * HS + cp;
    // One plane step, J = step modulo U: input plane p = pb + step adds to the output planes p - 1, p, p + 1 in the slots
    // J, J + 1, J + 2 (mod R); st[J & 1] holds plane p + 1 (requested a step ago), st[(J & 1) ^ 1] receives plane p + 2.
    auto plane_step = [&](auto jc, const int step) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value;
        constexpr int SF = J % R, S0 = (J + 1) % R, S1 = (J + 2) % R;
        constexpr int SX = J & 1, SY = SX ^ 1;
        const int p = pb + step;
        ring[S1][0] = ring[S1][1] = f32x2{bias, bias};   // (its previous tenant is more than T planes old)
#ifdef PDS_UPS_TIMING
        const long long t0 = __builtin_readcyclecounter();
        long long t1 = t0, t2 = t0;
#endif
        if (p < A.D && p <= pe) {
#ifndef PDS_UPS_NOSTAGE   // (PDS_UPS_NO*: timing ablations, wrong results by design, never defined in the product build)
            if (p + 2 < A.D && p + 2 <= pe) PDS_FETCHP(SY, p + 2)
#endif
#pragma nounroll
            for (int c2 = 0; c2 < CIN; c2 += 2) {  // not unrolled: keeps only 3 x 16 weight SGPRs live
                // the two halo rows this output-row parity uses: vr = 1 (tap kh = 1 / 2) and vr = 0 / 2 (kh = 3 / 0)
                f32x2 r02[2][2], r13[2][2];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int vr = (a == 0) ? 1 : (PY == 0 ? 0 : 2);
                        const float* row = &tile[SX][c2 + cc][lbase + vr * HS];
#ifdef PDS_UPS_NOLDSREAD
                        r02[cc][a] = f32x2{sc[cc], shv[a][0]};
                        r13[cc][a] = f32x2{shv[cc][1], sc[a]};
                        continue;
#endif
                        r02[cc][a] = f32x2{row[0], row[1]};
                        r13[cc][a] = f32x2{row[EO], row[EO + 1]};
                    }
                // the 2 x 24 taps of this parity and channel pair as SGPR operands: explicit s_load_dwordx16 (hipcc turns
                // plain reads of the weight pointer into vector loads parked in VGPRs, which spills this kernel); the three
                // loads are issued back to back and share one wait -- two waits per plane step instead of the four of the
                // per-channel table with both parities (round 4).  Table: [parity][channel][kd][a][kw in the order 1, 2, 3, 0],
                // so that the pairs a packed FMA needs -- (w1, w2) and (w3, w0) -- are aligned SGPR pairs.
                f32x16 wk[3];
#ifdef PDS_UPS_NOSLOAD
                asm volatile("s_nop 0" : "=s"(wk[0]), "=s"(wk[1]), "=s"(wk[2]));
                if (0)
#endif
                asm volatile(
                    "s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx16 %1, %3, %5\n\ts_load_dwordx16 %2, %3, %6\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&s"(wk[0]), "=&s"(wk[1]), "=&s"(wk[2])
                    : "s"(A.w), "s"(((PY * CIN + c2) * 24 + 0) * 4), "s"(((PY * CIN + c2) * 24 + 16) * 4),
                      "s"(((PY * CIN + c2) * 24 + 32) * 4)
                    : "memory");
                // out column 2j + px reads in(j) with kw = 1 + px and in(j - 1 + 2 px) with kw = 3 - 3 px:
                //   columns (0, 1): (w1, w2) * v1 + (w3, w0) * (v0, v2)      columns (2, 3): (w1, w2) * v2 + (w3, w0) * (v1, v3)
#ifdef PDS_UPS_NOFMA
                if (r02[0][0][0] == 12345.f)
#endif
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd) {
                        constexpr int slot[3] = {SF, S0, S1};
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            const int f = (cc * 3 + kd) * 8 + a * 4;   // first of the four taps in the 48-float record
                            const f32x2 w12 = {wk[f >> 4][(f & 15) + 0], wk[f >> 4][(f & 15) + 1]},
                                        w30 = {wk[f >> 4][(f & 15) + 2], wk[f >> 4][(f & 15) + 3]};
                            ring[slot[kd]][0] = __builtin_elementwise_fma(w12, f32x2{r13[cc][a][0], r13[cc][a][0]}, ring[slot[kd]][0]);
                            ring[slot[kd]][0] = __builtin_elementwise_fma(w30, r02[cc][a], ring[slot[kd]][0]);
                            ring[slot[kd]][1] = __builtin_elementwise_fma(w12, f32x2{r02[cc][a][1], r02[cc][a][1]}, ring[slot[kd]][1]);
                            ring[slot[kd]][1] = __builtin_elementwise_fma(w30, r13[cc][a], ring[slot[kd]][1]);
                        }
                    }
            }
#ifdef PDS_UPS_TIMING
            t1 = __builtin_readcyclecounter();
#endif
#ifndef PDS_UPS_NOSTAGE
            if (p + 1 < A.D && p + 1 <= pe) PDS_STASHP(SX, SX ^ 1)
#endif
#ifdef PDS_UPS_TIMING
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t2 = __builtin_readcyclecounter();
#endif
        }
#ifndef PDS_UPS_NOBARRIER
        __syncthreads();
#endif
#ifdef PDS_UPS_TIMING
        const long long t3 = __builtin_readcyclecounter();
        tm[0] += t1 - t0; tm[1] += t2 - t1; tm[2] += t3 - t2;
#endif
        const int k = p - 1;   // the plane that is complete now (slot SF)
        if (WRITE_COST) {
            if (k >= 0) {
                const int i = i0 + r, j = j0 + 2 * cp;
                if (i < A.Hi && j < A.Wi) {
                    const int Wo = 2 * A.Wi;
                    float* dst = A.cost + (((size_t)b * A.D + k) * 2 * A.Hi + 2 * i + PY) * Wo + 2 * j;
                    if (j + 1 < A.Wi) {
                        *reinterpret_cast<float4*>(dst) = make_float4(ring[SF][0][0], ring[SF][0][1], ring[SF][1][0], ring[SF][1][1]);
                    } else {
                        dst[0] = ring[SF][0][0];
                        dst[1] = ring[SF][0][1];
                    }
                }
            }
        } else if (k >= 0 && k < A.D) {
#ifdef PDS_UPS_NOEST
            if (ring[SF][0][0] != 12345.f) return;
#endif
            const bool candidate = k >= lo && k < hi;   // uniform
            if constexpr (T == 2) {
                // Selects on SGPR-pair masks, hand-scheduled.  hipcc compares into VCC and chains v_cndmask_b32_e32 on it;
                // on gfx950 every v_cndmask after the first that reads the SAME VCC value costs ~22 cycles
                // (tools/ubench/valu_rate2.hip: v_cmp + 4 v_cndmask on VCC = 73 cycles per wave, on an SGPR pair 5 per
                // instruction) -- 12 of the 24 selects of a step were of that kind, as much issue time as the 96 FMAs.
                // A VALU-written SGPR must not be read by a VALU within two wait states (gfx940 hazard; the compiler does
                // not look into the statement): every mask has three instructions between its producer and first reader.
                const int km1 = k - 1, km2 = k - 2;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const float v = ring[SF][o >> 1][o & 1];
                    const float h1 = ring[(J - 1 + R) % R][o >> 1][o & 1], h2 = ring[(J - 2 + R) % R][o >> 1][o & 1];
                    if (candidate)
                        estimator_update2<true>(v, h1, h2, k, km1, km2, best[o], bi[o], bprev[o][0], bprev[o][1], bnext[o][0],
                                                bnext[o][1]);
                    else
                        estimator_update2<false>(v, h1, h2, k, km1, km2, best[o], bi[o], bprev[o][0], bprev[o][1],
                                                 bnext[o][0], bnext[o][1]);
                }
            } else {
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const float v = ring[SF][o >> 1][o & 1];
                    // the planes after an earlier maximum arrive: neighbour t + 1 of plane bi is plane k = bi + 1 + t
#pragma unroll
                    for (int t = 0; t < T; ++t) bnext[o][t] = (bi[o] == k - 1 - t) ? v : bnext[o][t];
                    const bool up = candidate && v > best[o];   // strict: first occurrence wins
                    best[o] = up ? v : best[o];
                    bi[o] = up ? k : bi[o];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const int SP = ((J - 1 - t) % R + R) % R;   // slot of plane k - 1 - t (garbage below plane 0: masked later)
                        bprev[o][t] = up ? ring[SP][o >> 1][o & 1] : bprev[o][t];
                    }
                }
            }
        }
#ifdef PDS_UPS_TIMING
        tm[3] += __builtin_readcyclecounter() - t3;
#endif
    };
#ifdef PDS_UPS_TIMING
    const long long t_begin = __builtin_readcyclecounter();
#endif
    for (int base = 0; base < nsteps; base += U) {
#define PDS_STEP(J_) if (base + J_ < nsteps) plane_step(std::integral_constant<int, J_>{}, base + J_);
        PDS_STEP(0) PDS_STEP(1) PDS_STEP(2) PDS_STEP(3)
        if constexpr (U > 4) { PDS_STEP(4) PDS_STEP(5) PDS_STEP(6) PDS_STEP(7) PDS_STEP(8) PDS_STEP(9) }
        if constexpr (U > 10) { PDS_STEP(10) PDS_STEP(11) PDS_STEP(12) PDS_STEP(13) }
#undef PDS_STEP
    }
    static_assert(U == 4 || U == 10 || U == 14, "unrolled groups of T = 1, 2, 4");
#ifdef PDS_UPS_TIMING
    {
        const int wgl = blockIdx.y * gridDim.x + blockIdx.x;
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (lane == 0 && (wgl % 97) == 0)
            printf("[ups] wg %d part %d py %d xcc %u hwid %08x: start %lld total %lld | channels %lld stash %lld barrier %lld estimator %lld (%d steps)\n",
                   wgl, part, PY, xcc & 15, hwid, t_begin, (long long)__builtin_readcyclecounter() - t_begin, tm[0], tm[1], tm[2], tm[3], nsteps);
    }
#endif
#undef PDS_FETCHP
#undef PDS_STASHP
#undef PDS_UPS_STORE
#ifdef PDS_UPS_NOLDSWRITE
    if (sink == 12345.f) best[0] = sink;
#endif

    if (WRITE_COST) return;
    // the parts' states meet in LDS (the tiles are idle behind the last barrier): part q > 0 publishes, part 0 folds them in
    // ascending order -- a later part wins only with a strictly larger maximum, i.e. the first occurrence wins
    if (DS > 1) {
        constexpr int ITEMS = 4 * (2 + 2 * T);
        float* mine = merge + ((size_t)(part > 0 ? part - 1 : 0) * 2 + PY) * ITEMS * 64 + lane;
        if (part > 0) {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                mine[(o * (2 + 2 * T) + 0) * 64] = best[o];
                mine[(o * (2 + 2 * T) + 1) * 64] = __int_as_float(bi[o]);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    mine[(o * (2 + 2 * T) + 2 + t) * 64] = bprev[o][t];
                    mine[(o * (2 + 2 * T) + 2 + T + t) * 64] = bnext[o][t];
                }
            }
        }
        __syncthreads();
        if (part > 0) return;
#pragma unroll
        for (int q = 1; q < DS; ++q) {
            const float* theirs = merge + ((size_t)(q - 1) * 2 + PY) * ITEMS * 64 + lane;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const float other = theirs[(o * (2 + 2 * T) + 0) * 64];
                const bool up = other > best[o];
                best[o] = up ? other : best[o];
                bi[o] = up ? __float_as_int(theirs[(o * (2 + 2 * T) + 1) * 64]) : bi[o];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float pv = theirs[(o * (2 + 2 * T) + 2 + t) * 64], nv = theirs[(o * (2 + 2 * T) + 2 + T + t) * 64];
                    bprev[o][t] = up ? pv : bprev[o][t];
                    bnext[o][t] = up ? nv : bnext[o][t];
                }
            }
        }
    }
    // soft-arg-max around the best plane (estimator.py:84-91)
    const int planes = A.D;
    const int i = i0 + r, j = j0 + 2 * cp;
    float res[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        float den = 1.f;
        float num = A.step * (float)bi[o];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int kb = bi[o] - (t + 1);
            const bool okb = -(t + 1) >= A.lo && kb >= 0;
            const float eb = okb ? expf(bprev[o][t] - best[o]) : 0.f;
            den += eb;
            num = fmaf(eb, A.step * (float)kb, num);
            const int ka = bi[o] + (t + 1);
            const bool oka = (t + 1) <= A.hi && ka < planes;
            const float ea = oka ? expf(bnext[o][t] - best[o]) : 0.f;
            den += ea;
            num = fmaf(ea, A.step * (float)ka, num);
        }
        res[o] = num / den;
    }
    if (i < A.Hi && j < A.Wi) {
        // the crop of SizeAdapter.unpad (size_adapter.py:45-52: rows from the top, columns from the left) is part of
        // the store: the output is the contiguous [B, 2Hi - top, 2Wi - left] image itself
        const int Wc = 2 * A.Wi - A.crop_left, Hc = 2 * A.Hi - A.crop_top;
        const int row = 2 * i + PY - A.crop_top, col = 2 * j - A.crop_left;
        if (row >= 0) {
            float* dst = A.disp + ((size_t)b * Hc + row) * Wc + col;
            const int valid = (j + 1 < A.Wi) ? 4 : 2;
            if (valid == 4 && col >= 0 && ((A.crop_left | Wc) & 3) == 0) {
                *reinterpret_cast<float4*>(dst) = make_float4(res[0], res[1], res[2], res[3]);
            } else {
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o < valid && col + o >= 0) dst[o] = res[o];
            }
        }
    }
}

template <int CIN, int T, bool WRITE_COST, int DS>
// (five workgroups of 2 DS waves per CU -- the busiest CUs of a 1 080-tile launch hold five -- i.e. at most 96 registers)
__global__ __launch_bounds__(UHALF * DS, (DS == 2 && T <= 2) ? 5 : 1) void upsample_full_subpixel_kernel(const FusedArgs A) {
    constexpr int TILE = 2 * CIN * TILE_CH, MERGE = (DS - 1) * 2 * 4 * (2 + 2 * T) * 64;
    constexpr int FLOATS = DS * TILE > MERGE ? DS * TILE : MERGE;
    __shared__ __attribute__((aligned(16))) float lds[FLOATS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int part = wave >> 1;
    float (*tile)[CIN][TILE_CH] = reinterpret_cast<float (*)[CIN][TILE_CH]>(lds + part * TILE);
    if ((wave & 1) == 0)
        upsample_sweep<CIN, T, WRITE_COST, 0, DS>(A, tile, lds, part);
    else
        upsample_sweep<CIN, T, WRITE_COST, 1, DS>(A, tile, lds, part);
}

// [C][3][4][4] weights of the (3, 4, 4) transposed convolution -> [parity][C][kd 3][a 2][4]: the two kernel rows an output-row
// parity uses (a = 0: kh = 1 + parity, a = 1: kh = 3 - 3 parity), their four kw taps in the order 1, 2, 3, 0 (see the sweep);
// w_pairs has C * 48 floats
__global__ void upsample_weight_pairs_kernel(const float* __restrict__ w, float* __restrict__ w_pairs, int cin) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cin * 48) return;
    const int j = e & 3, a = (e >> 2) & 1, kd = (e >> 3) % 3, c = (e / 24) % cin, py = e / (24 * cin);
    const int kh = a == 0 ? 1 + py : 3 - 3 * py;
    w_pairs[e] = w[((c * 3 + kd) * 4 + kh) * 4 + ((j + 1) & 3)];
}

int launch_upsample_weight_pairs(const float* w, float* w_pairs, int cin, hipStream_t s) {
    const int total = cin * 48;
    hipLaunchKernelGGL(upsample_weight_pairs_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, w_pairs, cin);
    return check_launch("upsample_weight_pairs");
}

bool upsample_estimator_supported(int cin, int lo, int hi) {
    const int t = (-lo > hi) ? -lo : hi;
    return cin == 4 && t >= 1 && t <= 4;
}

int launch_upsample_estimator(const float* in, const float* scale, const float* shift, const float* w,
                              const float* bias, float* disp, int batch, int cin, int d, int hi_, int wi, int lo,
                              int hi, int step, int crop_top, int crop_left, hipStream_t s) {
    FusedArgs A;
    A.crop_top = crop_top;
    A.crop_left = crop_left;
    A.in = in;
    A.scale = scale;
    A.shift = shift;
    A.w = w;
    A.bias = bias;
    A.disp = disp;
    A.D = d;
    A.Hi = hi_;
    A.Wi = wi;
    A.lo = lo;
    A.hi = hi;
    A.step = (float)step;
    dim3 grid((wi + TC - 1) / TC, (hi_ + TR - 1) / TR, batch);
    const int t = (-lo > hi) ? -lo : hi;
    if (cin != 4) return set_error(-1, "upsample_estimator: unsupported channel count %d", cin);
    A.cost = nullptr;
    if (t <= 1)
        hipLaunchKernelGGL((upsample_full_subpixel_kernel<4, 1, false, 2>), grid, dim3(2 * UHALF), 0, s, A);
    else if (t <= 2)
        hipLaunchKernelGGL((upsample_full_subpixel_kernel<4, 2, false, 2>), grid, dim3(2 * UHALF), 0, s, A);
    else
        hipLaunchKernelGGL((upsample_full_subpixel_kernel<4, 4, false, 2>), grid, dim3(2 * UHALF), 0, s, A);
    return check_launch("upsample_full_subpixel");
}

bool upsample_full_valu_supported(int cin) { return cin == 4; }

// Training-mode forward of _upsample_to_fullsize (regularization.py:90-92): same sweep, cost volume stored.
int launch_upsample_full(const float* in, const float* scale, const float* shift, const float* w, const float* bias,
                         float* cost, int batch, int cin, int d, int hi_, int wi, hipStream_t s) {
    if (cin != 4) return set_error(-1, "upsample_full: unsupported channel count %d", cin);
    FusedArgs A;
    A.in = in;
    A.scale = scale;
    A.shift = shift;
    A.w = w;
    A.bias = bias;
    A.disp = nullptr;
    A.cost = cost;
    A.D = d;
    A.Hi = hi_;
    A.Wi = wi;
    A.lo = 0;
    A.hi = 0;
    A.step = 0.f;
    A.crop_top = A.crop_left = 0;
    dim3 grid((wi + TC - 1) / TC, (hi_ + TR - 1) / TR, batch);
    hipLaunchKernelGGL((upsample_full_subpixel_kernel<4, 1, true, 1>), grid, dim3(UHALF), 0, s, A);
    return check_launch("upsample_full");
}

}  // namespace pds
