// Weight gradient of the stride-1 3x3x3 convolutions of the hourglass (reference regularization.py:77-82, 85-86 under
// loss.backward(), pds_trainer.py:40-46) on the fp32 MFMA units:
//   dW[oc][c][tap] = sum over (n, z, y, x) of dz[oc][p] * xhat[c][p + tap]
// GEMM view: M = output channels (one block of 16), N = 16 input channels per tap (27 column blocks), K = positions,
// walked 4 at a time with v_mfma_f32_16x16x4_f32.
//   workgroup   one (output-channel block, input-channel block) pair (grid.y); 4 waves split the 27 taps; persistent
//               over work items (grid.x strides) with the partial dW kept in registers, ONE partial per workgroup;
//               a second kernel sums the partials in fp64 (wgrad2d_mfma.hip: wgrad_reduce_f32_kernel).
//   work item   a 32-position x-segment of one (n, z, y) row.
//   LDS         xhat tile [16 ch][9 rows (3 z x 3 y)][34 columns] with the deferred InstanceNorm, the skip sum (also
//               one broadcast along D, regularization.py:115) and zero padding applied while staging, and dz [16][32];
//               row strides chosen so that both fragment reads are bank-conflict free (channel stride == 2 mod 32).
#include "common.hpp"

namespace pds {

namespace {

constexpr int THREADS = 256;
constexpr int TWG = 32;          // positions per work item
constexpr int RSX = 50;          // xhat row stride: >= TWG + 2, and 9 * RSX == 2 (mod 32)
constexpr int XS = 9 * RSX;      // xhat channel stride
constexpr int DS = 34;           // dz row stride, == 2 (mod 32)
constexpr int TPW = 7;           // taps per wave (4 x 7 >= 27)
typedef float f32x4 __attribute__((ext_vector_type(4)));
static_assert(XS % 32 == 2 && DS % 32 == 2, "bank layout");

struct W3Args {
    Src a, b;
    const float* __restrict__ dz;
    float* __restrict__ partial;  // [workgroup][Cout][Cin][27]
    int N, Cin, D, H, W, Cout;
    int items, segs, ocbs;
};

}  // namespace

__global__ __launch_bounds__(THREADS) void wgrad3d_mfma_kernel(const W3Args A) {
    __shared__ __attribute__((aligned(16))) float xl[16 * XS];
    __shared__ __attribute__((aligned(16))) float dzl[16 * DS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ocb = blockIdx.y % A.ocbs, cb = blockIdx.y / A.ocbs;
    const int oc0 = ocb * 16, c0 = cb * 16;
    const size_t plane = (size_t)A.H * A.W;
    const size_t vol = (size_t)A.D * plane;
    const size_t bstride = A.b.bcast_d ? plane : vol;  // channel stride of the second source

    f32x4 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int item = blockIdx.x; item < A.items; item += gridDim.x) {
        int r = item;
        const int seg = r % A.segs;
        r /= A.segs;
        const int y = r % A.H;
        r /= A.H;
        const int z = r % A.D;
        const int n = r / A.D;
        const int x0 = seg * TWG;

        // ---- stage xhat: 16 channels x 9 rows x 34 columns (x0 - 1 .. x0 + 32) ----------------------------
        for (int e = tid; e < 16 * 9 * (TWG + 2); e += THREADS) {
            const int c = e / (9 * (TWG + 2));
            const int rem = e - c * 9 * (TWG + 2);
            const int rr = rem / (TWG + 2), xx = rem - rr * (TWG + 2);
            const int zz = z - 1 + rr / 3, yy = y - 1 + rr % 3, x = x0 - 1 + xx;
            const int ch = c0 + c;
            float v = 0.f;
            if (ch < A.Cin && zz >= 0 && zz < A.D && yy >= 0 && yy < A.H && x >= 0 && x < A.W) {
                const size_t inplane = (size_t)yy * A.W + x;
                const int g = n * A.Cin + ch;
                float sa = 1.f, ha = 0.f;
                if (A.a.scale) {
                    sa = A.a.scale[g];
                    ha = A.a.shift[g];
                }
                v = fmaf(sa, A.a.p[(size_t)g * vol + (size_t)zz * plane + inplane], ha);
                if (A.b.p) {
                    float sb = 1.f, hb = 0.f;
                    if (A.b.scale) {
                        sb = A.b.scale[g];
                        hb = A.b.shift[g];
                    }
                    v += fmaf(sb, A.b.p[(size_t)g * bstride + (A.b.bcast_d ? 0 : (size_t)zz * plane) + inplane], hb);
                }
            }
            xl[c * XS + rr * RSX + xx] = v;
        }
        // ---- stage dz: 16 output channels x 32 positions ----------------------------------------------------
        for (int e = tid; e < 16 * TWG; e += THREADS) {
            const int o = e / TWG, px = e - o * TWG;
            const int x = x0 + px, oc = oc0 + o;
            float v = 0.f;
            if (oc < A.Cout && x < A.W)
                v = A.dz[((size_t)(n * A.Cout + oc)) * vol + (size_t)z * plane + (size_t)y * A.W + x];
            dzl[o * DS + px] = v;
        }
        __syncthreads();

        const float* arow = dzl + (lane & 15) * DS + (lane >> 4);
        const float* brow = xl + (lane & 15) * XS + (lane >> 4);
#pragma unroll 2
        for (int ks = 0; ks < TWG / 4; ++ks) {
            const float af = arow[ks * 4];
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int tap = wave + 4 * i;            // taps w, w + 4, ...: wave-uniform
                if (tap < 27) {
                    const int rr = tap / 3, dx = tap % 3;  // rr = dz * 3 + dy
                    const float bf = brow[rr * RSX + ks * 4 + dx];
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- one partial per workgroup: [Cout][Cin][27] ----------------------------------------------------------
    float* dst = A.partial + (size_t)blockIdx.x * A.Cout * A.Cin * 27;
    const int c = c0 + (lane & 15);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tap = wave + 4 * i;
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
        const char* e = getenv("PDS_WGRAD3D_MFMA");
        return !(e && e[0] == '0');
    }();
    if (!enabled || transposed || kd != 3 || stride != 1) return false;
    if (in.d != out.d || in.h != out.h || in.w != out.w) return false;
    return true;
}

static int wgrad3d_workgroups(const Geom& in, int pairs) {
    const size_t items = (size_t)in.n * in.d * in.h * ((in.w + TWG - 1) / TWG);
    size_t wgs = items / 8;                       // >= ~8 items per workgroup so the partial write amortises
    const size_t cap = (size_t)(2048 / pairs) > 0 ? (size_t)(2048 / pairs) : 1;
    if (wgs > cap) wgs = cap;
    return wgs < 1 ? 1 : (int)wgs;
}

size_t wgrad3d_mfma_scratch_floats(const Geom& in, const Geom& out) {
    const int pairs = ((out.c + 15) / 16) * ((in.c + 15) / 16);
    return (size_t)wgrad3d_workgroups(in, pairs) * out.c * in.c * 27;
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
    A.segs = (in.w + TWG - 1) / TWG;
    A.items = in.n * in.d * in.h * A.segs;
    A.ocbs = (out.c + 15) / 16;
    const int pairs = A.ocbs * ((in.c + 15) / 16);
    const int wgs = wgrad3d_workgroups(in, pairs);
    hipLaunchKernelGGL(wgrad3d_mfma_kernel, dim3(wgs, pairs), dim3(THREADS), 0, s, A);
    if (int rc = check_launch("wgrad3d_mfma")) return rc;
    return launch_wgrad_reduce_f32(scratch, (size_t)out.c * in.c * 27, wgs, dw, accumulate, s);
}

}  // namespace pds
