// Weight gradient of the 64k -> 64 and 64k -> (at most 16) 3x3 convolutions of MatchingOperation / the embedding on the 16-bit
// matrix pipe (round 4; reference: autograd through network_blocks.py:47-58, driven by pds_trainer.py:40-46):
//   dW[oc][c][tap] = sum over (n, d, y, x) of dz[oc][p] * xhat[c][p + tap]
// wgrad2d_mfma.hip runs this contraction on v_mfma_f32_16x16x4_f32 (1/16 of the 16-bit MFMA rate): 11 launches x 1.2 ms =
// a third of the config-5 training step.  Here both operands are split into two fp16 parts exactly as in conv2d_x3.hip
// (hi = fp16(v), lo = fp16(v - hi): 22 significand bits) and a product is hi*lo + lo*hi + hi*hi on
// v_mfma_f32_16x16x32_f16 with fp32 accumulation.  Both operands are pre-scaled by powers of two derived from the data:
// xhat by its range certificate (common.hpp Src::bound), dz by max|dz|, which the InstanceNorm backward that writes dz
// collects (backward.hip in_bwd_apply_kernel); the partial sums are multiplied back when they are written.  A layer
// without both certificates keeps the exact-fp32 kernel.
//
// GEMM view: M = 64 output channels, N = (input channel, tap), K = positions, 32 per MFMA.
//   work unit   a strip 32 columns wide of one (n, d) plane, walked downwards two rows (four K-steps) at a time over a
//               chunk of rows.  The four xhat rows of a step live in an LDS ring: a step stages only its two NEW rows
//               (the one-step-per-item form of the first version staged all four every time: 1.6 GB through the
//               memory path for a 64 -> 64 layer of MatchingOperation, which is what bounded it), and their global
//               loads are issued before the MFMAs of the step before.  Persistent workgroups stride over the units,
//               keep their partial dW in registers and write ONE partial per workgroup (fp32; a second kernel sums
//               the partials in fp64 in a fixed order: deterministic).
//   workgroup   4 waves, two workgroups per CU (67.6 KB of LDS each): one stages while the other multiplies.  Wave w owns
//               the 16 input channels of block w, all nine taps and all 64 output channels: 4 x 9 accumulator tiles =
//               144 registers.  The B fragments of a wave are its own; the A fragments (dz) are shared by the four waves.
//   LDS         xhat [part][64 ch][4 rows][40] fp16, S[j] holds column x0 - 4 + j, so that an aligned 16-byte global
//               load lands as one aligned 8-byte LDS write per part; dz [part][64 oc][2 rows x 32].  Channel strides
//               == 2 dwords (mod 32): the 16 lanes x 8 bytes of a fragment read cover all 32 banks once.
//   taps        a lane's B fragment is eight CONSECUTIVE positions of one channel (v_mfma_f32_16x16x32_f16: K = 32 = one
//               row of the strip; the 16-position v_mfma_f32_16x16x16_f16 of the first version issues at HALF the rate
//               on gfx950 -- tools/ubench/mfma_rate.hip: 1.10 against 2.12 PFLOP/s chip-wide -- and bounded the kernel:
//               606 -> 428 us per 64 -> 64 layer); the three kernel columns are the windows S[b + 3 + dx .. b + 10 + dx]:
//               four aligned 8-byte reads (b, b + 4, b + 8, b + 12) per part and v_alignbit give all of them -- no
//               unaligned LDS access, no re-staging per tap.  (A v_mfma_f32_32x32x16_f16 form -- wave = 32 input channels x
//               32 output channels, the rate of the pipe from a single wave where the 16 x 16 forms need two -- was built
//               and measured: 428 -> 564 us; its K-step is 16 positions, so the tap windows are rebuilt twice per row.
//               So was ONE workgroup of eight waves per CU -- both waves of a SIMD in the matrix phase at the same time, a
//               six-row ring and double-buffered dz behind one barrier per step, the next step's operands in registers:
//               470-555 us; its waves stage together and multiply together, so neither the memory path nor the matrix
//               pipe ever works while the other does, which the two independent workgroups of this form provide.)
#include "common.hpp"

namespace pds {

namespace {

constexpr int THREADS = 256;
constexpr int TR = 2, TWG = 32;            // rows x columns of positions per work item
constexpr int XR = TR + 2;                 // staged xhat rows
constexpr int RSX = 40;                    // xhat row stride (fp16 elements): columns x0 - 4 .. x0 + 35
constexpr int CSX = XR * RSX + 36;         // xhat channel stride: 196 == 4 (mod 64) elements = 2 dwords (mod 32)
constexpr int CSD = TR * TWG + 4;          // dz channel stride: 68 == 4 (mod 64)
constexpr int CG = 64;                     // input channels per workgroup (grid.y walks further groups)
constexpr int XPART = CG * CSX;            // elements of one split part of xhat
constexpr int dpart(int mb) { return 16 * mb * CSD; }   // ... of dz, for mb blocks of 16 output channels
constexpr int lds_bytes(int mb) { return (2 * XPART + 2 * dpart(mb)) * 2; }
static_assert(CSX % 64 == 4 && CSD % 64 == 4, "conflict-free fragment reads");
static_assert(lds_bytes(4) <= 80 * 1024, "two workgroups per CU");
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WX3Args {
    Src a, b;
    const float* __restrict__ dz;
    const float* __restrict__ dz_bound;
    int dz_bound_n;
    float* __restrict__ partial;  // [workgroup][64][Cin][9]
    int N, Cin, D, H, W, Cout;
    int units, segs, rowpairs, chunks, ch;   // a unit = one 32-column strip of one plane over `ch` row pairs
};

__device__ __forceinline__ void split4(const float (&v)[4], f16x4& hi, f16x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = (_Float16)v[i];
        lo[i] = (_Float16)(v[i] - (float)hi[i]);
    }
}

template <int V>
using IC = std::integral_constant<int, V>;

__device__ __forceinline__ f16x8 join8(u32x2 lo, u32x2 hi) {
    return __builtin_bit_cast(f16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
}

}  // namespace

// MB = blocks of 16 output channels: 4 for the 64 -> 64 layers, 1 for the 64 -> 8 layer that closes MatchingOperation
// (rows 8..15 of its block are zeros: the layer is bound by staging xhat, not by the matrix pipe).
template <bool HAS_B, int MB>
__global__ __launch_bounds__(THREADS, 2) void wgrad2d_x3_kernel(const WX3Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int DPART = dpart(MB);
    _Float16* xs = reinterpret_cast<_Float16*>(lds_raw);     // [2][CG][CSX]
    _Float16* dl = xs + 2 * XPART;                           // [2][16 MB][CSD]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = the wave's input-channel block
    const int cg0 = blockIdx.y * CG;
    const size_t plane = (size_t)A.H * A.W;

    // operand scales (powers of two): xhat by the sources' range certificates, dz by its recorded maximum
    float bound = block_bound(A.a.bound, A.a.bound_n, reinterpret_cast<float*>(lds_raw));
    if (HAS_B) bound += block_bound(A.b.bound, A.b.bound_n, reinterpret_cast<float*>(lds_raw));
    const float as = pow2_scale(bound, kHalfTarget);
    const float ds = pow2_scale(block_bound(A.dz_bound, A.dz_bound_n, reinterpret_cast<float*>(lds_raw)), kHalfTarget);
    const float unscale = (1.f / as) * (1.f / ds);

    f32x4 acc[MB][9];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses (elements): A: row m = lane & 15 of output block mb, positions 4 kg ..; B: channel n = lane & 15
    const int frag_k = 8 * (lane >> 4);
    const _Float16* arow = dl + (lane & 15) * CSD + frag_k;
    const _Float16* brow = xs + (wave * 16 + (lane & 15)) * CSX + frag_k;

    // what a thread stages per step: 5 quads of two xhat rows (64 channels x 2 rows x 10 quads = 1280 slots), MB quads of dz
    // the 16-output-channel form has registers for the next step's operands in flight during its MFMAs; the 64 -> 64 forms
    // (144 accumulator registers + 32 of A fragments) load after the MFMAs, a quad at a time -- measured with three of the
    // five xhat quads in flight: 428 -> 471 us (the allocator spills inside the K loop); the other workgroup of the CU
    // covers the latency
    constexpr bool PREFETCH = MB == 1;
    f32x4 qa[5], qb[HAS_B ? 5 : 1], qz[MB];

    for (int unit = blockIdx.x; unit < A.units; unit += gridDim.x) {
        int r = unit;
        const int chunk = r % A.chunks;
        r /= A.chunks;
        const int seg = r % A.segs;
        r /= A.segs;
        const int d = r % A.D;
        const int n = r / A.D;
        const int x0 = seg * TWG;
        const int ry0 = chunk * A.ch, ry1 = min(ry0 + A.ch, A.rowpairs);
        const unsigned cstride = (unsigned)(A.D * (int)plane);
        const float* abase = A.a.p + ((size_t)(n * A.Cin + cg0) * A.D + d) * plane;
        const float* bbase = HAS_B ? A.b.p + ((size_t)(n * A.Cin + cg0) * A.D + d) * plane : nullptr;
        const float* zbase = A.dz + ((size_t)n * A.Cout * A.D + d) * plane;

        // ---- xhat rows ybase, ybase + 1: columns x0 - 4 .. x0 + 35 as 10 aligned quads; a run of 10 consecutive threads
        // covers one (channel, row): 160 contiguous bytes.  Row y lives in ring slot (y + 1) & 3 of the 4-row tile: a step
        // (rows y0 - 1 .. y0 + 2) keeps the two lower rows of the step before it and stages only two new ones.
        auto load_rows = [&](int ybase, auto J0, auto J1) __attribute__((always_inline)) {   // quads J0 .. J1 - 1 of the thread's five
#pragma unroll
            for (int j = decltype(J0)::value; j < decltype(J1)::value; ++j) {
                const int slot = tid + j * THREADS;     // < 1280
                const int q = slot % 10, cr = slot / 10, c = cr >> 1, rr = cr & 1;
                const int x = x0 - 4 + 4 * q, y = ybase + rr;
                const int xc = min(max(x, 0), A.W - 4), yc = min(max(y, 0), A.H - 1);
                // a wave-uniform base and a 32-bit lane offset (64 channel volumes stay below 2^31 elements: launcher)
                const unsigned off = (unsigned)c * cstride + (unsigned)(yc * A.W + xc);
                qa[j] = *reinterpret_cast<const f32x4*>(abase + off);
                if (HAS_B) qb[j] = *reinterpret_cast<const f32x4*>(bbase + off);
            }
        };
        // the deferred InstanceNorm, the skip sum, the literal zero padding and the fp16 split, on the way to LDS
        auto store_rows = [&](int ybase, auto J0, auto J1) __attribute__((always_inline)) {
#pragma unroll
            for (int j = decltype(J0)::value; j < decltype(J1)::value; ++j) {
                const int slot = tid + j * THREADS;
                const int q = slot % 10, cr = slot / 10, c = cr >> 1, rr = cr & 1;
                const int x = x0 - 4 + 4 * q, y = ybase + rr;
                const bool ok = x >= 0 && x < A.W && y >= 0 && y < A.H;
                const int ch = cg0 + c;
                float sa = 1.f, ha = 0.f, sb2 = 1.f, hb2 = 0.f;
                if (A.a.scale) {
                    const int g = A.a.per_plane ? ((n * A.Cin + ch) * A.D + d) : (n * A.Cin + ch);
                    sa = A.a.scale[g];
                    ha = A.a.shift[g];
                }
                if (HAS_B && A.b.scale) {
                    const int g = A.b.per_plane ? ((n * A.Cin + ch) * A.D + d) : (n * A.Cin + ch);
                    sb2 = A.b.scale[g];
                    hb2 = A.b.shift[g];
                }
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = fmaf(sa, qa[j][e], ha);
                    if (HAS_B) t += fmaf(sb2, qb[j][e], hb2);
                    v[e] = ok ? t * as : 0.f;
                }
                f16x4 hi, lo;
                split4(v, hi, lo);
                const int dst = c * CSX + ((y + 1) & 3) * RSX + 4 * q;
                *reinterpret_cast<f16x4*>(xs + dst) = hi;
                *reinterpret_cast<f16x4*>(xs + XPART + dst) = lo;
            }
        };
        // ---- dz rows y0, y0 + 1: 16 MB output channels x 2 rows x 8 quads, MB quads per thread --------------------------
        auto load_dz = [&](int y0) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const int slot = tid + j * THREADS;     // < 256 MB
                const int q = slot & 7, rr = (slot >> 3) & 1, oc = slot >> 4;
                const int xc = min(x0 + 4 * q, A.W - 4), yc = min(y0 + rr, A.H - 1), occ = min(oc, A.Cout - 1);
                qz[j] = *reinterpret_cast<const f32x4*>(zbase + ((unsigned)occ * cstride + (unsigned)(yc * A.W + xc)));
            }
        };
        auto store_dz = [&](int y0) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const int slot = tid + j * THREADS;
                const int q = slot & 7, rr = (slot >> 3) & 1, oc = slot >> 4;
                const bool ok = x0 + 4 * q < A.W && y0 + rr < A.H && oc < A.Cout;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok ? qz[j][e] * ds : 0.f;
                f16x4 hi, lo;
                split4(v, hi, lo);
                const int dst = oc * CSD + rr * TWG + 4 * q;
                *reinterpret_cast<f16x4*>(dl + dst) = hi;
                *reinterpret_cast<f16x4*>(dl + DPART + dst) = lo;
            }
        };

        // ---- prologue: all four rows of the first step (the barrier closing the previous unit released the tiles) ------
        auto stage_rows = [&](int ybase) __attribute__((always_inline)) {   // load and store back to back, in batches the registers have room for
            if (PREFETCH) {
                load_rows(ybase, IC<0>{}, IC<5>{});
                store_rows(ybase, IC<0>{}, IC<5>{});
            } else {
                load_rows(ybase, IC<0>{}, IC<1>{});
                store_rows(ybase, IC<0>{}, IC<1>{});
                load_rows(ybase, IC<1>{}, IC<2>{});
                store_rows(ybase, IC<1>{}, IC<2>{});
                load_rows(ybase, IC<2>{}, IC<3>{});
                store_rows(ybase, IC<2>{}, IC<3>{});
                load_rows(ybase, IC<3>{}, IC<4>{});
                store_rows(ybase, IC<3>{}, IC<4>{});
                load_rows(ybase, IC<4>{}, IC<5>{});
                store_rows(ybase, IC<4>{}, IC<5>{});
            }
        };
        stage_rows(2 * ry0 - 1);
        stage_rows(2 * ry0 + 1);
        load_dz(2 * ry0);
        store_dz(2 * ry0);
        __syncthreads();

        for (int ry = ry0; ry < ry1; ++ry) {
            const int y0 = ry * TR;
            const bool more = ry + 1 < ry1;
            if (PREFETCH && more) {   // the next step's two new rows and its dz: in flight during the MFMAs
                load_rows(y0 + 3, IC<0>{}, IC<5>{});
                load_dz(y0 + 2);
            }
            const int ring = y0 & 3;   // slot of row y0 - 1
            // ---- two K-steps of 32 positions (one row each) on v_mfma_f32_16x16x32_f16: a lane holds EIGHT consecutive
            // positions of its row of A (dz) and of its channel of B (xhat); the sum over K does not care which eight, as
            // long as A and B agree.  The kernel columns are the windows S[b + 3 + dx .. b + 10 + dx], b = 8 (lane >> 4):
            // four aligned 8-byte reads (b, b + 4, b + 8, b + 12) per part and v_alignbit give all three.
#pragma unroll 1
            for (int r2 = 0; r2 < TR; ++r2) {
                f16x8 ah[MB], al[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const _Float16* ap = arow + m * 16 * CSD + r2 * TWG;
                    ah[m] = join8(*reinterpret_cast<const u32x2*>(ap), *reinterpret_cast<const u32x2*>(ap + 4));
                    al[m] = join8(*reinterpret_cast<const u32x2*>(ap + DPART), *reinterpret_cast<const u32x2*>(ap + DPART + 4));
                }
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const _Float16* bp = brow + ((ring + r2 + dy) & 3) * RSX;
                    u32x2 g[2][4];
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int k = 0; k < 4; ++k) g[p][k] = *reinterpret_cast<const u32x2*>(bp + p * XPART + 4 * k);
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        f16x8 bw[2];   // [part]: window S[b + 3 + dx .. b + 10 + dx]
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const unsigned mid1 = __builtin_amdgcn_alignbit(g[p][1][1], g[p][1][0], 16);   // {G1.e1, G1.e2}
                            const unsigned mid2 = __builtin_amdgcn_alignbit(g[p][2][1], g[p][2][0], 16);   // {G2.e1, G2.e2}
                            const unsigned c12 = __builtin_amdgcn_alignbit(g[p][2][0], g[p][1][1], 16);    // {G1.e3, G2.e0}
                            if (dx == 0)
                                bw[p] = join8(u32x2{__builtin_amdgcn_alignbit(g[p][1][0], g[p][0][1], 16), mid1}, u32x2{c12, mid2});
                            else if (dx == 1)
                                bw[p] = join8(g[p][1], g[p][2]);
                            else
                                bw[p] = join8(u32x2{mid1, c12}, u32x2{mid2, __builtin_amdgcn_alignbit(g[p][3][0], g[p][2][1], 16)});
                        }
                        // small partial products first; consecutive MFMAs hit different accumulators
#pragma unroll
                        for (int prod = 0; prod < 3; ++prod)
#pragma unroll
                            for (int m = 0; m < MB; ++m) {
                                const f16x8 av = prod == 1 ? al[m] : ah[m];
                                const f16x8 bv = prod == 0 ? bw[1] : bw[0];
                                acc[m][dy * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[m][dy * 3 + dx], 0, 0, 0);
                            }
                    }
                }
            }
            __syncthreads();
            if (more) {
                if (!PREFETCH) load_dz(y0 + 2);
                if (PREFETCH) store_rows(y0 + 3, IC<0>{}, IC<5>{});   // into the slots of rows y0 - 1 and y0
                else stage_rows(y0 + 3);
                store_dz(y0 + 2);
                __syncthreads();
            }
        }
    }

    // ---- one partial per workgroup: [Cout][Cin][9] -----------------------------------------------------------------
    float* dst = A.partial + (size_t)blockIdx.x * A.Cout * A.Cin * 9;
    const int c = cg0 + wave * 16 + (lane & 15);
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int oc = m * 16 + 4 * (lane >> 4) + rr;
                if (MB == 4 || oc < A.Cout) dst[((size_t)oc * A.Cin + c) * 9 + t] = acc[m][t][rr] * unscale;
            }
}

bool wgrad2d_x3_supported(const Src& a, const Src& b, const Src& dz, const Geom& in, const Geom& out) {
    static const bool enabled = []() {  // PDS_WGRAD2D_X3=0 keeps the exact-fp32 kernel (A/B, debugging)
        const char* e = debug_switch("PDS_WGRAD2D_X3");
        return !(e && e[0] == '0');
    }();
    if (!enabled || !(out.c == 64 || out.c <= 16) || in.c % CG != 0 || (in.w & 3) != 0) return false;
    if ((size_t)64 * in.d * in.h * in.w >= ((size_t)1 << 31)) return false;   // 32-bit offsets inside a 64-channel group
    if (!dz.bound || dz.bound_n <= 0 || !a.bound || a.bound_n <= 0 || (b.p && (!b.bound || b.bound_n <= 0 || b.bcast_d)))
        return false;
    if ((reinterpret_cast<uintptr_t>(a.p) | reinterpret_cast<uintptr_t>(b.p) | reinterpret_cast<uintptr_t>(dz.p)) & 15)
        return false;
    return true;
}

int launch_wgrad2d_x3(const Src& a, const Src& b, const Src& dz, float* partial, int workgroups, const Geom& in,
                      const Geom& out, hipStream_t s) {
    WX3Args A;
    A.Cout = out.c;
    A.a = a;
    A.b = b;
    A.dz = dz.p;
    A.dz_bound = dz.bound;
    A.dz_bound_n = dz.bound_n;
    A.partial = partial;
    A.N = in.n;
    A.Cin = in.c;
    A.D = in.d;
    A.H = in.h;
    A.W = in.w;
    A.segs = (in.w + TWG - 1) / TWG;
    A.rowpairs = (in.h + TR - 1) / TR;
    // strips are cut into chunks of row pairs: a chunk stages two halo rows once, so long chunks stage less; at least
    // ~4 units per workgroup keep the persistent workgroups balanced
    const int strips = in.n * in.d * A.segs;
    int chunks = (4 * workgroups + strips - 1) / strips;
    chunks = chunks < 1 ? 1 : chunks > A.rowpairs ? A.rowpairs : chunks;
    A.ch = (A.rowpairs + chunks - 1) / chunks;
    A.chunks = (A.rowpairs + A.ch - 1) / A.ch;
    A.units = strips * A.chunks;
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad2d_x3_kernel<true, 4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(4));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad2d_x3_kernel<false, 4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(4));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad2d_x3_kernel<true, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(1));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad2d_x3_kernel<false, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(1));
    }
    const dim3 grid(workgroups, in.c / CG);
    if (out.c == 64) {
        if (b.p) hipLaunchKernelGGL((wgrad2d_x3_kernel<true, 4>), grid, dim3(THREADS), lds_bytes(4), s, A);
        else hipLaunchKernelGGL((wgrad2d_x3_kernel<false, 4>), grid, dim3(THREADS), lds_bytes(4), s, A);
    } else {
        if (b.p) hipLaunchKernelGGL((wgrad2d_x3_kernel<true, 1>), grid, dim3(THREADS), lds_bytes(1), s, A);
        else hipLaunchKernelGGL((wgrad2d_x3_kernel<false, 1>), grid, dim3(THREADS), lds_bytes(1), s, A);
    }
    return check_launch("wgrad2d_x3");
}

}  // namespace pds
