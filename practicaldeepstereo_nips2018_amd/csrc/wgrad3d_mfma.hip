// Weight gradient of the stride-1 3x3x3 convolutions of the hourglass (reference regularization.py:77-82, 85-86 under
// loss.backward(), pds_trainer.py:40-46) on the fp32 MFMA units:
//   dW[oc][c][tap] = sum over (n, z, y, x) of dz[oc][p] * xhat[c][p + tap]
// GEMM view: M = output channels (one block of 16), N = 16 input channels per tap (27 column blocks), K = positions,
// walked 4 at a time with v_mfma_f32_16x16x4_f32 (exact fp32: needs no range certificate).
//   work unit   a column of the volume: 4 rows x 32 columns of positions, walked along z over a chunk of planes.  The
//               three xhat planes a step needs live in an LDS ring: every step stages ONE new plane (6 rows x 34
//               columns x 16 channels), i.e. 1.6 staged elements per position and channel -- the round-3 form staged
//               nine rows for every single row of positions (9.6) and spent 7/8 of its time doing so.
//   staging     the (channel, row, column) a thread stages is the same in every step: global offsets, LDS addresses,
//               bounds and the deferred-InstanceNorm coefficients are computed once per unit; the loads of step z + 1
//               are issued before the MFMAs of step z and land in LDS after them.
//   workgroup   one (output-channel block, input-channel block) pair (grid.y); 4 waves split the 27 taps; persistent
//               over units (grid.x strides) with the partial dW in registers, ONE partial per workgroup; a second
//               kernel sums the partials in fp64 in a fixed order (wgrad2d_mfma.hip: wgrad_reduce_f32_kernel).
//   LDS         xhat [16 ch][3 planes][6 rows][36] and dz [16][4 x 32]; channel strides == 2 (mod 32), so the 32 lanes
//               of a half-wave (16 channels x 2 positions) read 32 different banks.
#include "common.hpp"

namespace pds {

namespace {

constexpr int THREADS = 256;
constexpr int TY = 4, TWG = 32;                 // rows x columns of positions per step
constexpr int ROWS = TY + 2, COLS = TWG + 2;    // staged rows y0 - 1 .. y0 + TY, columns x0 - 1 .. x0 + TWG
constexpr int RSX = 36;                         // xhat row stride
constexpr int PSX = ROWS * RSX;                 // plane (ring slot) stride
constexpr int XS = 3 * PSX + 26;                // xhat channel stride: 674 == 2 (mod 32)
constexpr int DS = TY * TWG + 2;                // dz channel stride: 130 == 2 (mod 32)
constexpr int xslots(bool pair) { return (pair ? 8 : 16) * ROWS * COLS; }   // staged xhat elements per plane
constexpr int NDZ = 16 * TY * TWG / THREADS;    // 8 per thread
constexpr int TPW = 7;                          // taps per wave (4 x 7 >= 27)
typedef float f32x4 __attribute__((ext_vector_type(4)));
static_assert(XS % 32 == 2 && DS % 32 == 2, "bank layout");
static_assert((16 * XS + 16 * DS) * 4 <= 53 * 1024, "three workgroups per CU");

struct W3Args {
    Src a, b;
    const float* __restrict__ dz;
    float* __restrict__ partial;  // [workgroup][Cout][Cin][27]
    int N, Cin, D, H, W, Cout;
    int units, segs, yblocks, zchunks, zc, ocbs;
};

}  // namespace

// PAIR (layers with at most 8 input channels: the two full-resolution 8 -> 8 layers, the biggest of the hourglass): the 16
// columns of an MFMA are 8 channels x TWO consecutive taps instead of 16 channels of which 8 are padding -- 14 tap pairs
// instead of 27 taps per K-step.
template <bool HAS_B, bool PAIR>
__global__ __launch_bounds__(THREADS, 2) void wgrad3d_mfma_kernel(const W3Args A) {
    constexpr int NACC = PAIR ? 4 : TPW;   // 14 pairs or 27 taps over 4 waves
    constexpr int XSLOTS = xslots(PAIR), NSTG = (XSLOTS + THREADS - 1) / THREADS;   // 13 (7) staged elements per thread
    // channel stride: 2 (mod 32) for 16 channels x 2 positions per half-wave; the tap-pair form reads (8 channels, 2 taps,
    // 2 positions) at c XSK + {0, 1 (mostly)} + {0, 1}: 4 (mod 32) keeps the channels apart (with 2, a quarter of the lanes
    // of every read collided with the next channel)
    constexpr int XSK = PAIR ? XS + 2 : XS;
    __shared__ __attribute__((aligned(16))) float xl[(PAIR ? 8 : 16) * XSK];
    __shared__ __attribute__((aligned(16))) float dzl[16 * DS];
    __shared__ f32x4 coef[16];   // per input channel: scale and shift of the two sources (deferred InstanceNorm)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ocb = blockIdx.y % A.ocbs, cb = blockIdx.y / A.ocbs;
    const int oc0 = ocb * 16, c0 = cb * 16;
    const int plane = A.H * A.W;
    const size_t vol = (size_t)A.D * plane;
    const size_t bstride = HAS_B && A.b.bcast_d ? (size_t)plane : vol;  // channel stride of the second source
    const int bz = HAS_B && A.b.bcast_d ? 0 : plane;                    // its plane stride

    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* arow = dzl + (lane & 15) * DS + (lane >> 4);
    const float* brow = xl + (lane & (PAIR ? 7 : 15)) * XSK + (lane >> 4);
    const bool second = PAIR && (lane & 8);   // this lane's column belongs to the second tap of the pair

    for (int unit = blockIdx.x; unit < A.units; unit += gridDim.x) {
        int r = unit;
        const int seg = r % A.segs;
        r /= A.segs;
        const int yb = r % A.yblocks;
        r /= A.yblocks;
        const int zci = r % A.zchunks;
        const int n = r / A.zchunks;
        const int x0 = seg * TWG, y0 = yb * TY;
        const int z0 = zci * A.zc, z1 = min(z0 + A.zc, A.D);
        const int rows = min(TY, A.H - y0);   // rows of positions that exist

        // ---- what this thread stages in every step -----------------------------------------------------------------
        int xoff[NSTG], xdst[NSTG];      // offset inside the (n, c0) block of one plane index 0, or -1; LDS address or -1
        if (tid < 16) {
            const int g = n * A.Cin + min(c0 + tid, A.Cin - 1);
            f32x4 cf{1.f, 0.f, 1.f, 0.f};
            if (A.a.scale) {
                cf[0] = A.a.scale[g];
                cf[1] = A.a.shift[g];
            }
            if (HAS_B && A.b.scale) {
                cf[2] = A.b.scale[g];
                cf[3] = A.b.shift[g];
            }
            coef[tid] = cf;
        }
#pragma unroll
        for (int j = 0; j < NSTG; ++j) {
            const int e = tid + j * THREADS;
            const int c = e / (ROWS * COLS), rem = e - c * (ROWS * COLS);
            const int row = rem / COLS, col = rem - row * COLS;
            const int y = y0 - 1 + row, x = x0 - 1 + col, ch = c0 + c;
            const bool slot = e < XSLOTS;
            const bool ok = slot && ch < A.Cin && y >= 0 && y < A.H && x >= 0 && x < A.W;
            xdst[j] = slot ? c * XSK + row * RSX + col : -1;
            xoff[j] = ok ? c * (int)vol + y * A.W + x : -1;   // 16 channel volumes stay below 2^31 elements (launcher)
        }
        const float* abase = A.a.p + (size_t)(n * A.Cin + c0) * vol;
        const float* bbase = HAS_B ? A.b.p + (size_t)(n * A.Cin + c0) * bstride : nullptr;
        const unsigned bshrink = (unsigned)(vol - bstride);   // a source broadcast along D has one plane per channel
        int doff[NDZ];
#pragma unroll
        for (int j = 0; j < NDZ; ++j) {
            const int e = tid + j * THREADS;
            const int o = e / (TY * TWG), rem = e % (TY * TWG);
            const int y = y0 + rem / TWG, x = x0 + rem % TWG;
            doff[j] = (oc0 + o < A.Cout && y < A.H && x < A.W) ? o * (int)vol + y * A.W + x : -1;
        }
        const float* dbase = A.dz + (size_t)(n * A.Cout + oc0) * vol;

        float xa[NSTG], xb[HAS_B ? NSTG : 1], dv[NDZ];
        // loads of xhat plane zz (zero outside the volume) / of dz plane z, into registers
        auto load_x = [&](int zz) {
            const bool zin = zz >= 0 && zz < A.D;
            const unsigned zo = (unsigned)(min(max(zz, 0), A.D - 1) * plane);
            const unsigned zob = (unsigned)(min(max(zz, 0), A.D - 1) * bz);
#pragma unroll
            for (int j = 0; j < NSTG; ++j) {
                const bool ok = zin && xoff[j] >= 0;
                const unsigned off = (unsigned)max(xoff[j], 0);
                xa[j] = ok ? abase[off + zo] : 0.f;
                if (HAS_B) {
                    const unsigned c = (unsigned)(tid + j * THREADS) / (ROWS * COLS);
                    xb[j] = ok ? bbase[off - c * bshrink + zob] : 0.f;
                }
            }
        };
        auto store_x = [&](int zz) {
            const bool zin = zz >= 0 && zz < A.D;
            const int so = ((zz + 3) % 3) * PSX;   // zz >= -1
#pragma unroll
            for (int j = 0; j < NSTG; ++j) {
                const f32x4 cf = coef[min((tid + j * THREADS) / (ROWS * COLS), 15)];
                float v = fmaf(cf[0], xa[j], cf[1]);
                if (HAS_B) v += fmaf(cf[2], xb[j], cf[3]);
                if (!(zin && xoff[j] >= 0)) v = 0.f;   // the literal zero padding, not the normalised zero
                if (xdst[j] >= 0) xl[xdst[j] + so] = v;
            }
        };
        auto load_dz = [&](int z) {
#pragma unroll
            for (int j = 0; j < NDZ; ++j) {
                dv[j] = doff[j] >= 0 ? dbase[(unsigned)doff[j] + (unsigned)(z * plane)] : 0.f;
            }
        };
        auto store_dz = [&]() {
#pragma unroll
            for (int j = 0; j < NDZ; ++j) {
                const int e = tid + j * THREADS;
                dzl[(e / (TY * TWG)) * DS + e % (TY * TWG)] = dv[j];
            }
        };

        // ---- prologue: planes z0 - 1 and z0 in the ring, plane z0 + 1 and dz(z0) in flight ---------------------------
        load_x(z0 - 1);
        __syncthreads();   // coef
        store_x(z0 - 1);
        load_x(z0);
        store_x(z0);
        load_x(z0 + 1);
        load_dz(z0);
        for (int z = z0; z < z1; ++z) {
            store_x(z + 1);      // the slot held plane z - 2: the barrier closing step z - 1 released it
            store_dz();
            __syncthreads();
            if (z + 1 < z1) {    // next step's operands: in flight during the MFMAs
                load_x(z + 2);
                load_dz(z + 1);
            }
            // plane z - 1 + dzt sits in ring slot (z + dzt + 2) % 3
            auto tap_offset = [&](int tap) {
                const int dzt = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
                return ((z + dzt + 2) % 3) * PSX + dy * RSX + dx;
            };
            int boff[NACC];
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (PAIR) {   // pairs w, w + 4, ...: taps 2 p and 2 p + 1 (the 14th pair's second tap does not exist)
                    const int p = min(wave + 4 * i, 13);
                    const int o0 = tap_offset(2 * p), o1 = tap_offset(min(2 * p + 1, 26));
                    boff[i] = second ? o1 : o0;
                } else {
                    boff[i] = tap_offset(min(wave + 4 * i, 26));   // taps w, w + 4, ...: wave-uniform
                }
            }
            for (int rr = 0; rr < rows; ++rr) {
#pragma unroll 2
                for (int ks = 0; ks < TWG / 4; ++ks) {
                    const float af = arow[rr * TWG + ks * 4];
#pragma unroll
                    for (int i = 0; i < NACC; ++i) {
                        if (PAIR ? (i < 3 || wave < 2) : (i < TPW - 1 || wave < 3)) {   // pair 14 / tap 27 does not exist
                            const float bf = brow[boff[i] + rr * RSX + ks * 4];
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- one partial per workgroup: [Cout][Cin][27] ----------------------------------------------------------
    float* dst = A.partial + (size_t)blockIdx.x * A.Cout * A.Cin * 27;
    const int c = c0 + (lane & (PAIR ? 7 : 15));
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        const int tap = PAIR ? 2 * (wave + 4 * i) + (second ? 1 : 0) : wave + 4 * i;
        if (tap >= 27) continue;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int oc = oc0 + 4 * (lane >> 4) + rr;
            if (oc < A.Cout && c < A.Cin) dst[((size_t)oc * A.Cin + c) * 27 + tap] = acc[i][rr];
        }
    }
}

int launch_wgrad_reduce_f32(const float* partial, size_t wcount, int parts, float* dw, int accumulate, hipStream_t s);

bool wgrad3d_mfma_supported(int transposed, int kd, int stride, const Geom& in, const Geom& out) {
    static const bool enabled = []() {  // PDS_WGRAD3D_MFMA=0 selects the direct kernel (A/B, debugging)
        const char* e = debug_switch("PDS_WGRAD3D_MFMA");
        return !(e && e[0] == '0');
    }();
    if (!enabled || transposed || kd != 3 || stride != 1) return false;
    if (in.d != out.d || in.h != out.h || in.w != out.w) return false;
    if ((size_t)in.d * in.h * in.w * 16 >= ((size_t)1 << 31)) return false;   // 32-bit offsets inside a channel block
    return true;
}

namespace {

// How the volume is cut into units and how many workgroups walk them.  A unit is one column (4 rows x 32 columns) over a
// chunk of `zc` planes and stages zc + 2 planes for zc steps: long chunks stage less, short ones give more units.
// The chip holds 3 workgroups per CU: with plenty of work the workgroups are persistent over >= 3 units each (chunks of
// 8, 6 or 4 planes); a small level gets the shortest chunks that still fit all units on the chip at once.
struct W3Plan {
    int segs, yblocks, zc, zchunks, units, wgs;
};

W3Plan wgrad3d_plan(const Geom& in, int pairs) {
    W3Plan P;
    P.segs = (in.w + TWG - 1) / TWG;
    P.yblocks = (in.h + TY - 1) / TY;
    const int columns = in.n * P.yblocks * P.segs;
    const int slots = 768 / pairs > 0 ? 768 / pairs : 1;
    P.zc = 0;
    for (int zc : {8, 6, 4})
        if (columns * ((in.d + zc - 1) / zc) >= 3 * slots) {
            P.zc = zc;
            break;
        }
    if (!P.zc) {
        P.zc = in.d;
        for (int zc = 1; zc <= in.d; ++zc)
            if (columns * ((in.d + zc - 1) / zc) <= slots) {
                P.zc = zc;
                break;
            }
    }
    P.zchunks = (in.d + P.zc - 1) / P.zc;
    P.units = columns * P.zchunks;
    P.wgs = P.units < slots ? P.units : slots;
    return P;
}

}  // namespace

size_t wgrad3d_mfma_scratch_floats(const Geom& in, const Geom& out) {
    const int pairs = ((out.c + 15) / 16) * ((in.c + 15) / 16);
    return (size_t)wgrad3d_plan(in, pairs).wgs * out.c * in.c * 27;
}

int launch_wgrad3d_mfma(const Src& a, const Src& b, const float* dz, float* dw, const Geom& in, const Geom& out,
                        int accumulate, float* scratch, hipStream_t s) {
    W3Args A;
    A.a = a;
    A.b = b;
    A.dz = dz;
    A.partial = scratch;
    A.N = in.n;
    A.Cin = in.c;
    A.D = in.d;
    A.H = in.h;
    A.W = in.w;
    A.Cout = out.c;
    A.ocbs = (out.c + 15) / 16;
    const int pairs = A.ocbs * ((in.c + 15) / 16);
    const W3Plan P = wgrad3d_plan(in, pairs);
    A.segs = P.segs;
    A.yblocks = P.yblocks;
    A.zc = P.zc;
    A.zchunks = P.zchunks;
    A.units = P.units;
    if (in.c <= 8) {
        if (b.p) hipLaunchKernelGGL((wgrad3d_mfma_kernel<true, true>), dim3(P.wgs, pairs), dim3(THREADS), 0, s, A);
        else hipLaunchKernelGGL((wgrad3d_mfma_kernel<false, true>), dim3(P.wgs, pairs), dim3(THREADS), 0, s, A);
    } else {
        if (b.p) hipLaunchKernelGGL((wgrad3d_mfma_kernel<true, false>), dim3(P.wgs, pairs), dim3(THREADS), 0, s, A);
        else hipLaunchKernelGGL((wgrad3d_mfma_kernel<false, false>), dim3(P.wgs, pairs), dim3(THREADS), 0, s, A);
    }
    if (int rc = check_launch("wgrad3d_mfma")) return rc;
    return launch_wgrad_reduce_f32(scratch, (size_t)out.c * in.c * 27, P.wgs, dw, accumulate, s);
}

}  // namespace pds
