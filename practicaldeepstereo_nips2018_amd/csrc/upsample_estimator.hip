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
#include "common.hpp"

namespace pds {

namespace {

constexpr int TR = 4, TC = 32;            // input tile
constexpr int HR = TR + 2, HC = TC + 2;   // halo tile
constexpr int NPOS = HR * HC;             // 204
constexpr int POS = (NPOS + 63) / 64;     // 4
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct FusedArgs {
    const float* __restrict__ in;     // raw upsample_half output [B, C, D, Hi, Wi]
    const float* __restrict__ scale;  // [B*C] folded InstanceNorm
    const float* __restrict__ shift;
    const float* __restrict__ w;      // [C][3][4 kh][kw order 1, 2, 3, 0]: the layer's weights, see upsample_weight_pairs_kernel
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

template <int CIN, int T, bool WRITE_COST, int PY, int DS>
__device__ __forceinline__ void upsample_sweep(const FusedArgs& A, float (*tile)[CIN][NPOS + 4], float* merge, const int part) {
    const int tid = threadIdx.x & (UHALF - 1);
    const int lane = tid & 63;
    const int r = lane >> 4, cp = lane & 15;
    const int i0 = blockIdx.y * TR, j0 = blockIdx.x * TC;
    const int b = blockIdx.z;
    const size_t plane = (size_t)A.Hi * A.Wi;
    const size_t cstride = (size_t)A.D * plane;
    const float* src = A.in + (size_t)b * CIN * cstride;

    // candidate planes of this part, the input planes it sweeps and the common number of steps (barriers are shared)
    const int per = (A.D + DS - 1) / DS;
    const int lo = DS == 1 ? 0 : part * per, hi = DS == 1 ? A.D : min(A.D, lo + per);
    const int pb = DS == 1 ? 0 : max(lo - T - 1, 0);         // first input plane: output plane pb + 1 is the first complete one
    const int pe = WRITE_COST ? A.D : hi + T;                // last step: plane pe - 1 enters the window, centre = hi - 1
    int nsteps = pe - pb + 1;
    if (DS > 1) {
#pragma unroll
        for (int q = 0; q < DS; ++q) {
            const int lq = q * per, hq = min(A.D, lq + per);
            nsteps = max(nsteps, hq + T - max(lq - T - 1, 0) + 1);
        }
    }

    int goff[UPOS], loff[UPOS];
    bool inside[UPOS];
#pragma unroll
    for (int k = 0; k < UPOS; ++k) {
        const int p = min(tid + k * UHALF, NPOS - 1);
        const int rr = p / HC, cc = p % HC;
        const int y = i0 - 1 + rr, x = j0 - 1 + cc;
        inside[k] = y >= 0 && y < A.Hi && x >= 0 && x < A.Wi;
#ifdef PDS_UPS_NOHALO   // FETCH_SIZE calibration (wrong results by design): every tile reads its own 4 x 32 interior only,
                        // with the same load instructions -- the launch then fetches exactly the input tensor's bytes
        goff[k] = min(max(y, i0), min(i0 + TR, A.Hi) - 1) * A.Wi + min(max(x, j0), min(j0 + TC, A.Wi) - 1);
#else
        goff[k] = min(max(y, 0), A.Hi - 1) * A.Wi + min(max(x, 0), A.Wi - 1);
#endif
        loff[k] = rr * HC + (cc & 1) * EO + (cc >> 1);
    }
    float sc[CIN], sh[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
        sc[c] = A.scale ? A.scale[b * CIN + c] : 1.f;
        sh[c] = A.scale ? A.shift[b * CIN + c] : 0.f;
    }
    const float bias = A.bias ? A.bias[0] : 0.f;

    // Two register stages: plane p + 2 is requested while plane p is consumed and is stashed one iteration later, so a
    // request has a whole plane step (~1 500 cycles) to land; with one stage (request at the top of a step, stash at its
    // end) every step waited for HBM.  The two stages swap roles from step to step (plane loop unrolled by two).
    float stA[CIN][UPOS], stB[CIN][UPOS];
#define PDS_FETCHP(st_, p_)                                                        \
    _Pragma("unroll") for (int c = 0; c < CIN; ++c) _Pragma("unroll") for (int k = 0; k < UPOS; ++k) \
        st_[c][k] = src[c * cstride + (size_t)(p_) * plane + goff[k]];
#define PDS_STASHP(st_, buf_)                                                      \
    _Pragma("unroll") for (int c = 0; c < CIN; ++c) _Pragma("unroll") for (int k = 0; k < UPOS; ++k) \
        tile[buf_][c][loff[k]] = inside[k] ? fmaf(sc[c], st_[c][k], sh[c]) : 0.f;

    // estimator state for the 4 pixels of this lane (output row 2 i + PY): a delayed window win[0..2T] of the last
    // finished planes (win[2T] newest).  When the centre win[T] (plane k - T) beats the running maximum, its T
    // neighbours on either side are captured from the window -- static register indices only.
    float best[4], win[4][2 * T + 1], bprev[4][T], bnext[4][T];
    int bi[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        best[o] = -INFINITY;
        bi[o] = lo;
#pragma unroll
        for (int t = 0; t < T; ++t) bprev[o][t] = bnext[o][t] = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2 * T + 1; ++t) win[o][t] = -INFINITY;
    }
    // accumulators: [0] -> output plane p-1, [1] -> p, [2] -> p+1 ; pixel index q (four consecutive output columns)
    // (register pairs: the transposed convolution runs on packed fp32 FMAs, two output columns per instruction)
    f32x2 acc[3][2];
#pragma unroll
    for (int o = 0; o < 2; ++o) acc[0][o] = acc[1][o] = acc[2][o] = f32x2{bias, bias};

    PDS_FETCHP(stB, pb)
    if (pb + 1 < A.D) PDS_FETCHP(stA, pb + 1)
    PDS_STASHP(stB, 0)
    __syncthreads();

    // halo row r, halo columns 2cp .. 2cp + 3 (input columns j0 + 2cp - 1 ..): the even pair (v0, v2) = E[cp], E[cp + 1] and
    // the odd pair (v1, v3) = O[cp], O[cp + 1] are the second operands of the packed FMAs below as they stand
    const int lbase = r * HC + cp;
    // one plane step: sx holds plane p + 1 (requested a step ago), sy receives plane p + 2
    auto plane_step = [&](const int step, float (&sx)[CIN][UPOS], float (&sy)[CIN][UPOS]) __attribute__((always_inline)) {
        const int p = pb + step;
        if (p < A.D && p <= pe) {
            const int cur = step & 1;
            if (p + 2 < A.D && p + 2 <= pe) PDS_FETCHP(sy, p + 2)
#pragma nounroll
            for (int c = 0; c < CIN; ++c) {  // not unrolled: keeps only 3 x 16 weight SGPRs live
                // the two halo rows this output-row parity uses: vr = 1 (tap kh = 1 / 2) and vr = 0 / 2 (kh = 3 / 0)
                f32x2 r02[2], r13[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int vr = (a == 0) ? 1 : (PY == 0 ? 0 : 2);
                    const float* row = &tile[cur][c][lbase + vr * HC];
                    r02[a] = f32x2{row[0], row[1]};
                    r13[a] = f32x2{row[EO], row[EO + 1]};
                }
                // the 3 x 16 taps of channel c as SGPR operands: explicit s_load_dwordx16 (hipcc turns plain reads of
                // the weight pointer into vector loads parked in VGPRs, which spills this kernel); the three loads
                // are issued back to back and share one wait.  The table is stored with the kw taps of a kernel row in
                // the order 1, 2, 3, 0, so that the pairs a packed FMA needs -- (w1, w2) and (w3, w0) -- are aligned
                // SGPR pairs (taken from the natural order they cost 24 s_mov per channel and plane: the loop was bound
                // by instruction issue, 470 instructions per plane for 192 FMAs; rocprofv3: 2 waves per SIMD, both busy).
                f32x16 wk[3];
                asm volatile(
                    "s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx16 %1, %3, %5\n\ts_load_dwordx16 %2, %3, %6\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&s"(wk[0]), "=&s"(wk[1]), "=&s"(wk[2])
                    : "s"(A.w), "s"((c * 3 + 0) * 64), "s"((c * 3 + 1) * 64), "s"((c * 3 + 2) * 64)
                    : "memory");
                // out column 2j + px reads in(j) with kw = 1 + px and in(j - 1 + 2 px) with kw = 3 - 3 px:
                //   columns (0, 1): (w1, w2) * v1 + (w3, w0) * (v0, v2)      columns (2, 3): (w1, w2) * v2 + (w3, w0) * (v1, v3)
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const f32x16 wp = wk[kd];
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int kh = (PY == 0) ? (a == 0 ? 1 : 3) : (a == 0 ? 2 : 0);
                        const f32x2 w12 = {wp[kh * 4 + 0], wp[kh * 4 + 1]}, w30 = {wp[kh * 4 + 2], wp[kh * 4 + 3]};
                        acc[kd][0] = __builtin_elementwise_fma(w12, f32x2{r13[a][0], r13[a][0]}, acc[kd][0]);
                        acc[kd][0] = __builtin_elementwise_fma(w30, r02[a], acc[kd][0]);
                        acc[kd][1] = __builtin_elementwise_fma(w12, f32x2{r02[a][1], r02[a][1]}, acc[kd][1]);
                        acc[kd][1] = __builtin_elementwise_fma(w30, r13[a], acc[kd][1]);
                    }
                }
            }
            if (p + 1 < A.D && p + 1 <= pe) PDS_STASHP(sx, cur ^ 1)
        }
        __syncthreads();
        if (WRITE_COST) {
            if (p >= 1) {  // store the finished plane p - 1
                const int i = i0 + r, j = j0 + 2 * cp;
                if (i < A.Hi && j < A.Wi) {
                    const int Wo = 2 * A.Wi;
                    float* dst = A.cost + (((size_t)b * A.D + (p - 1)) * 2 * A.Hi + 2 * i + PY) * Wo + 2 * j;
                    if (j + 1 < A.Wi) {
                        *reinterpret_cast<float4*>(dst) = make_float4(acc[0][0][0], acc[0][0][1], acc[0][1][0], acc[0][1][1]);
                    } else {
                        dst[0] = acc[0][0][0];
                        dst[1] = acc[0][0][1];
                    }
                }
            }
        } else if (p >= 1) {
            const int k = p - 1;          // plane entering the window (a real plane while k < D)
            const int centre = k - T;     // plane now at the centre of the window
#pragma unroll
            for (int o = 0; o < 4; ++o) {
#pragma unroll
                for (int t = 0; t < 2 * T; ++t) win[o][t] = win[o][t + 1];
                win[o][2 * T] = k < A.D ? acc[0][o >> 1][o & 1] : -INFINITY;
                const bool up = centre >= lo && centre < hi && win[o][T] > best[o];  // strict: first occurrence wins
                best[o] = up ? win[o][T] : best[o];
                bi[o] = up ? centre : bi[o];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    bprev[o][t] = up ? win[o][T - 1 - t] : bprev[o][t];
                    bnext[o][t] = up ? win[o][T + 1 + t] : bnext[o][t];
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            acc[0][o] = acc[1][o];
            acc[1][o] = acc[2][o];
            acc[2][o] = f32x2{bias, bias};
        }
    };
    for (int step = 0; step < nsteps; step += 2) {
        plane_step(step, stA, stB);
        if (step + 1 < nsteps) plane_step(step + 1, stB, stA);
    }
#undef PDS_FETCHP
#undef PDS_STASHP

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
__global__ __launch_bounds__(UHALF * DS) void upsample_full_subpixel_kernel(const FusedArgs A) {
    constexpr int TILE = 2 * CIN * (NPOS + 4), MERGE = (DS - 1) * 2 * 4 * (2 + 2 * T) * 64;
    constexpr int FLOATS = DS * TILE > MERGE ? DS * TILE : MERGE;
    __shared__ __attribute__((aligned(16))) float lds[FLOATS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int part = wave >> 1;
    float (*tile)[CIN][NPOS + 4] = reinterpret_cast<float (*)[CIN][NPOS + 4]>(lds + part * TILE);
    if ((wave & 1) == 0)
        upsample_sweep<CIN, T, WRITE_COST, 0, DS>(A, tile, lds, part);
    else
        upsample_sweep<CIN, T, WRITE_COST, 1, DS>(A, tile, lds, part);
}

// [C][3][4][4] weights of the (3, 4, 4) transposed convolution -> the same table with the four kw taps of every kernel
// row stored in the order 1, 2, 3, 0 (see the sweep); w_pairs has C * 48 floats
__global__ void upsample_weight_pairs_kernel(const float* __restrict__ w, float* __restrict__ w_pairs, int total) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) w_pairs[e] = w[(e & ~3) + ((e + 1) & 3)];
}

int launch_upsample_weight_pairs(const float* w, float* w_pairs, int cin, hipStream_t s) {
    const int total = cin * 48;
    hipLaunchKernelGGL(upsample_weight_pairs_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, w_pairs, total);
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
