// 3x3x3 convolutions (stride 1, pad 1) of the hourglass's QUARTER-resolution level -- 16 channels over 24 x 72 x 120 =
// 207 360 voxels at 960x540, D = 192 (reference practical_deep_stereo/regularization.py:25-26 [contraction 0, smoothing]
// and :51-52 [expansion 2, smoothing]; network_blocks.py:61-72, 106-112) -- on the 16-bit matrix pipe with split fp32
// operands (round 6).
//
// conv3d_mfma.hip served these two layers on v_mfma_f32_16x16x4_f32 at 51 us each: 2.87 GFLOP at the fp32 matrix rate
// (1/16 of the 16-bit one on gfx950) are 18 us of pipe time, and the tile-per-workgroup kernel reached 36 % of it.  The
// K-split kernel (conv3d_ks.hip) is the wrong shape here (its tiles are single rows: nine halo rows staged per output
// row).  This kernel keeps conv3d_mfma's shape -- a workgroup of four waves owns a tile of TZ x TY rows x 16 NB columns,
// K is walked in chunks of FOUR input channels through a double-buffered LDS halo tile -- and takes the arithmetic of
// conv3d_ks.hip's X form:
//   operands    every fp32 value v = hi + lo with hi = fp16(v), lo = fp16(v - hi) (22 of 24 significand bits), three
//               partial products a_hi b_lo + a_lo b_hi + a_hi b_hi per multiply on v_mfma_f32_16x16x16_f16, fp32
//               accumulation; activations scaled by the power of two derived from the sources' range certificates
//               (common.hpp Src::bound), weights by the one pack.hip derives from max |w| (mode 8), undone in the epilogue.
//   GEMM view   M = 16 output channels per block, N = 16 consecutive output x, K = 16 = four input channels x the four
//               x-taps of a kernel row (dx = 0..2 and a zero): one MFMA x three products per (dz, dy) covers what three
//               fp32 MFMAs did.  LDS tile of a chunk: [part][position][4 x fp16] (the bytes of four fp32 channels); a B
//               fragment is the 8-byte slot of position (column n + x-tap of the lane's k group).
//   weights     the chunk's A fragments (9 K-steps x MB blocks x 2 parts x 8 bytes per lane) go from global memory (L2;
//               pack.hip mode 8, the same packing as conv3d_ks's X form) straight into registers, one chunk ahead.
//   prologue / epilogue   as conv3d_mfma.hip: deferred InstanceNorm of the producer(s) + skip sum + zero padding while
//               staging; bias, LeakyReLU(0.1), store, one fp64 statistics record per (tile, channel).
// Sources without a range certificate keep conv3d_mfma.hip (exact fp32).
#include <atomic>

#include "common.hpp"

namespace pds {

namespace {

constexpr int NX_THREADS = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 nx_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 nx_f16x8 __attribute__((ext_vector_type(8)));

struct ArgsNX {
    Src a, b;
    const float* __restrict__ wpk;    // pack.hip mode 8: [Cin / 4][9][mblocks][2 parts][64 lanes][2 dwords] + 16-dword tail
    const float* __restrict__ wtail;  // dwords 12, 13 of the tail: ws, 1 / ws
    const float* __restrict__ bias;
    float* __restrict__ out;
    double* __restrict__ partials;
    int N, Cin, Di, Hi, Wi;
    int Cout, Do, Ho, Wo;
    int lrelu;
    int tiles_x, tiles_y, tiles;
    int mblocks;
    int xcd_run;                      // tiles per XCD (ceil(tiles / 8)); 0: identity mapping
};

template <int MB, int TZ, int TY, int NB>
struct CfgNX {
    static constexpr int RW = TZ * TY / 4;
    static constexpr int ZT = TZ + 2, YT = TY + 2, XT = 16 * NB + 2;
    static constexpr int NPOS = ZT * YT * XT;
    static constexpr int POS = (NPOS + NX_THREADS - 1) / NX_THREADS;
    static constexpr int PART = NPOS * 8;                  // bytes of one split part of a chunk
    static constexpr int BUF = 2 * PART + 64;              // hi | lo (+ slack: the zero k group of the last column reads its own slot)
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF;
    static_assert(TZ * TY % 4 == 0, "rows must split over 4 waves");
};

__device__ __forceinline__ float nx_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

}  // namespace

template <int MB, int TZ, int TY, int NB>
__global__ __launch_bounds__(NX_THREADS) void conv3d_nx_kernel(const ArgsNX A) {
    using C = CfgNX<MB, TZ, TY, NB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware placement as in conv3d_mfma.hip: every XCD works on a contiguous run of tiles (a slab of z)
    int tile = blockIdx.x;
    if (A.xcd_run > 0) {
        tile = (int)(blockIdx.x & 7) * A.xcd_run + (int)(blockIdx.x >> 3);
        if (tile >= A.tiles) return;  // grid.x is padded to a multiple of 8
    }
    const int mb0 = blockIdx.y * MB;
    const int n = blockIdx.z;
    const int tx = tile % A.tiles_x;
    const int ty = (tile / A.tiles_x) % A.tiles_y;
    const int tz = tile / (A.tiles_x * A.tiles_y);
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * 16 * NB;
    const size_t plane_i = (size_t)A.Hi * A.Wi;
    const size_t cstride_a = (size_t)A.Di * plane_i;
    const bool hasb = A.b.p != nullptr;
    const size_t cstride_b = A.b.bcast_d ? plane_i : cstride_a;
    const int nchunks = A.Cin / 4;

    // power-of-two operand scales: activations from the sources' range certificates, weights from the packing's tail
    // (few records -- one behind an InstanceNorm: every wave reduces them itself, no barrier)
    float bm = 0.f, bm2 = 0.f;
    for (int i = lane; i < A.a.bound_n; i += 64) {
        const float v = fabsf(A.a.bound[i]);
        bm = fmaxf(bm, v == v ? v : __builtin_inff());
    }
    if (hasb)
        for (int i = lane; i < A.b.bound_n; i += 64) {
            const float v = fabsf(A.b.bound[i]);
            bm2 = fmaxf(bm2, v == v ? v : __builtin_inff());
        }
    const float ascale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(
                                                       int, pow2_scale(wave_max(bm) + wave_max(bm2), kHalfTarget))));
    const float unscale = A.wtail[13] * (1.f / ascale);

    unsigned ga[C::POS], gb[C::POS];
    int lo[C::POS];
    bool inside[C::POS];
#pragma unroll
    for (int k = 0; k < C::POS; ++k) {
        const int p = min(tid + k * NX_THREADS, C::NPOS - 1);
        const int xx = p % C::XT, yy = (p / C::XT) % C::YT, zz = p / (C::XT * C::YT);
        const int z = z0 - 1 + zz, y = y0 - 1 + yy, x = x0 - 1 + xx;
        inside[k] = z >= 0 && z < A.Di && y >= 0 && y < A.Hi && x >= 0 && x < A.Wi;
        const int zc = min(max(z, 0), A.Di - 1), yc = min(max(y, 0), A.Hi - 1), xc = min(max(x, 0), A.Wi - 1);
        ga[k] = (unsigned)((zc * A.Hi + yc) * A.Wi + xc);
        gb[k] = A.b.bcast_d ? (unsigned)(yc * A.Wi + xc) : ga[k];
        lo[k] = p * 8;                                     // byte offset of the position's slot inside a part
    }
    const float* pa = A.a.p + (size_t)n * A.Cin * cstride_a;
    const float* pb = hasb ? A.b.p + (size_t)n * A.Cin * cstride_b : nullptr;
    const pds_u32x2* wl = reinterpret_cast<const pds_u32x2*>(A.wpk) + lane;

    float va[4][C::POS], vb[4][C::POS];
    pds_u32x2 aw[9][MB][2], awn[9][MB][2];                 // A fragments of the chunk being multiplied / of the next one

    auto fetch = [&](int chunk) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* ca = pa + (size_t)(chunk * 4 + c) * cstride_a;
#pragma unroll
            for (int k = 0; k < C::POS; ++k) va[c][k] = ca[ga[k]];
        }
        if (hasb) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float* cb = pb + (size_t)(chunk * 4 + c) * cstride_b;
#pragma unroll
                for (int k = 0; k < C::POS; ++k) vb[c][k] = cb[gb[k]];
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    awn[t][m][p] = wl[((((size_t)chunk * 9 + t) * A.mblocks + mb0 + m) * 2 + p) * 64];
    };
    auto stash = [&](int chunk, unsigned char* buf) {
        float s1[4], h1[4], s2[4], h2[4];                  // folded InstanceNorm x operand scale (wave-uniform: scalar loads)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ch = n * A.Cin + chunk * 4 + c;
            s1[c] = (A.a.scale ? A.a.scale[ch] : 1.f) * ascale;
            h1[c] = (A.a.scale ? A.a.shift[ch] : 0.f) * ascale;
            s2[c] = (hasb && A.b.scale ? A.b.scale[ch] : 1.f) * ascale;
            h2[c] = (hasb && A.b.scale ? A.b.shift[ch] : 0.f) * ascale;
        }
#pragma unroll
        for (int k = 0; k < C::POS; ++k) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float t = fmaf(s1[c], va[c][k], h1[c]);
                if (hasb) t += fmaf(s2[c], vb[c][k], h2[c]);
                v[c] = inside[k] ? t : 0.f;
            }
            pds_u32x2 hi, lw;
            split_quad_f16(v, hi, lw);
            *reinterpret_cast<pds_u32x2*>(buf + lo[k]) = hi;
            *reinterpret_cast<pds_u32x2*>(buf + C::PART + lo[k]) = lw;
        }
    };

    f32x4 acc[MB][C::RW][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < C::RW; ++r)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[m][r][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    stash(0, lds);
    __syncthreads();

    // B fragment of lane (k group q, column n16): slot of position (row, n16 + x-tap); the zero fourth group re-reads
    // x-tap 0 (finite data: a slot past the row could hold the bits of an fp16 infinity, and inf x 0 is not 0)
    const int n16 = lane & 15, q = lane >> 4;
    int b_row[C::RW];
#pragma unroll
    for (int r = 0; r < C::RW; ++r) {
        const int rho = wave * C::RW + r;
        const int zr = rho / TY, yr = rho % TY;
        b_row[r] = ((zr * C::YT + yr) * C::XT + n16 + (q == 3 ? 0 : q)) * 8;
    }

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const unsigned char* buf = lds + (chunk & 1) * C::BUF;
        unsigned char* nxt = lds + ((chunk + 1) & 1) * C::BUF;
        const bool more = chunk + 1 < nchunks;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p) aw[t][m][p] = awn[t][m][p];
        if (more) fetch(chunk + 1);
        // K-steps (dz, dy) in PAIRS on v_mfma_f32_16x16x32_f16 (K = 32 = two K-steps x four channels x four x-taps: twice
        // the work of the K = 16 instruction in the same 16 issue cycles; the ninth K-step runs alone on the K = 16 form).
        // The three partial products of an accumulator are dependent matrix instructions, so each product runs over all
        // accumulators before the next one starts.
#pragma unroll
        for (int t2 = 0; t2 < 5; ++t2) {
            const int ta = 2 * t2, tb = ta + 1 < 9 ? ta + 1 : ta;
            const int offa = ((ta / 3) * C::YT + ta % 3) * C::XT * 8, offb = ((tb / 3) * C::YT + tb % 3) * C::XT * 8;
#pragma unroll
            for (int r = 0; r < C::RW; ++r) {   // (one row of accumulators at a time: the B fragments of a row are 16 registers)
                nx_f16x4 ha[NB], la[NB], hb[NB], lb[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const unsigned char* bp = buf + b_row[r] + j * 16 * 8;
                    ha[j] = *reinterpret_cast<const nx_f16x4*>(bp + offa);
                    la[j] = *reinterpret_cast<const nx_f16x4*>(bp + offa + C::PART);
                    if (t2 < 4) {
                        hb[j] = *reinterpret_cast<const nx_f16x4*>(bp + offb);
                        lb[j] = *reinterpret_cast<const nx_f16x4*>(bp + offb + C::PART);
                    }
                }
#pragma unroll
                for (int prod = 0; prod < 3; ++prod)
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const nx_f16x4 a0 = __builtin_bit_cast(nx_f16x4, aw[ta][m][prod == 1 ? 1 : 0]);
                        const nx_f16x4 a1 = __builtin_bit_cast(nx_f16x4, aw[tb][m][prod == 1 ? 1 : 0]);
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            const nx_f16x4 b0 = prod == 0 ? la[j] : ha[j];
                            if (t2 < 4) {
                                const nx_f16x4 b1 = prod == 0 ? lb[j] : hb[j];
                                const nx_f16x8 a8 = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                                const nx_f16x8 b8 = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                                acc[m][r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[m][r][j], 0, 0, 0);
                            } else {
                                acc[m][r][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, acc[m][r][j], 0, 0, 0);
                            }
                        }
                    }
            }
        }
        if (more) stash(chunk + 1, nxt);
        __syncthreads();
    }

    // ---- epilogue (as conv3d_mfma.hip) ------------------------------------------------------------------------------
    const size_t plane_o = (size_t)A.Ho * A.Wo;
    float* sred = reinterpret_cast<float*>(lds);  // [4 waves][MB*16][2]
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int oc = (mb0 + m) * 16 + q * 4 + rr;
            const bool chok = oc < A.Cout;
            const float bv = (chok && A.bias) ? A.bias[oc] : 0.f;
            float s = 0.f, sq = 0.f;
#pragma unroll
            for (int r = 0; r < C::RW; ++r) {
                const int rho = wave * C::RW + r;
                const int z = z0 + rho / TY, y = y0 + rho % TY;
                const bool rowok = chok && z < A.Do && y < A.Ho;
                float* po = A.out + (((size_t)n * A.Cout + (chok ? oc : 0)) * A.Do + min(z, A.Do - 1)) * plane_o +
                            (size_t)min(y, A.Ho - 1) * A.Wo;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int x = x0 + j * 16 + n16;
                    float t = fmaf(acc[m][r][j][rr], unscale, bv);
                    if (A.lrelu) t = t > 0.f ? t : t * kLeakySlope;
                    if (rowok && x < A.Wo) {
                        po[x] = t;
                        s += t;
                        sq = fmaf(t, t, sq);
                    }
                }
            }
            if (A.partials) {
                s = nx_row16_sum(s);
                sq = nx_row16_sum(sq);
                if (n16 == 15) {
                    sred[((wave * MB + m) * 16 + q * 4 + rr) * 2 + 0] = s;
                    sred[((wave * MB + m) * 16 + q * 4 + rr) * 2 + 1] = sq;
                }
            }
        }
    }
    if (A.partials) {
        __syncthreads();
        if (tid < MB * 16 * 2) {
            const int ocl = tid >> 1, k = tid & 1;
            const int oc = mb0 * 16 + ocl;
            if (oc < A.Cout) {
                double sum = 0.0;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) sum += (double)sred[((wv * MB * 16) + ocl) * 2 + k];
                A.partials[(((size_t)n * A.Cout + oc) * A.tiles + tile) * 2 + k] = sum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
namespace {

// tile shapes (TZ x TY rows x 16 NB columns); PDS_CONV3D_NX_CFG selects one (A/B), the default is the measured best
struct NxShape {
    int tz, ty, nb;
};
constexpr NxShape kNxShapes[5] = {{2, 4, 4}, {2, 2, 4}, {2, 4, 2}, {4, 4, 4}, {2, 2, 2}};
int nx_shape_index() {
    static const int index = []() {
        const char* e = debug_switch("PDS_CONV3D_NX_CFG");
        const int v = e ? atoi(e) : 0;
        return v >= 0 && v < 5 ? v : 0;
    }();
    return index;
}

size_t nx_split_dwords(int cin, int mblocks) { return (size_t)(cin / 4) * 9 * mblocks * 2 * 64 * 2; }

template <int MB, int TZ, int TY, int NB>
int launch_nx(ArgsNX& A, hipStream_t s) {
    using C = CfgNX<MB, TZ, TY, NB>;
    static std::atomic<unsigned> attr_done{0};   // one bit per device
    if (DeviceOnce once{attr_done}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_nx_kernel<MB, TZ, TY, NB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    }
    A.xcd_run = A.tiles >= 64 ? (A.tiles + 7) / 8 : 0;
    dim3 grid(A.xcd_run > 0 ? 8 * A.xcd_run : A.tiles, A.mblocks / MB, A.N);
    hipLaunchKernelGGL((conv3d_nx_kernel<MB, TZ, TY, NB>), grid, dim3(NX_THREADS), C::LDS_BYTES, s, A);
    return check_launch("conv3d_nx");
}

}  // namespace

// 16- and 32-channel stride-1 layers over volumes too large for the K-split kernel, both sources with a range certificate
bool conv3d_nx_supported(const ConvLayer& L) {
    static const bool on = []() {   // PDS_CONV3D_NX=0: conv3d_mfma.hip (exact fp32) serves these layers (A/B, tests)
        const char* e = debug_switch("PDS_CONV3D_NX");
        return !(e && e[0] == '0');
    }();
    if (!on) return false;
    if (L.kd != 3 || L.stride != 1 || L.stat_per_plane) return false;
    if ((L.a.scale && L.a.per_plane) || (L.b.scale && L.b.per_plane)) return false;
    if (L.in.c != 16 || L.out_g.c != 16) return false;
    if (!L.a.bounded || (L.b.p && !L.b.bounded)) return false;
    if ((size_t)L.out_g.d * L.out_g.h * L.out_g.w < 100000) return false;   // (smaller volumes: conv3d_ks / conv3d_mfma)
    if ((size_t)L.in.d * L.in.h * L.in.w >= ((size_t)1 << 31) || L.in.n > 65535) return false;
    return true;
}

int conv3d_nx_tiles(const Geom& o) {
    const NxShape t = kNxShapes[nx_shape_index()];
    return ((o.w + 16 * t.nb - 1) / (16 * t.nb)) * ((o.h + t.ty - 1) / t.ty) * ((o.d + t.tz - 1) / t.tz);
}

size_t conv3d_nx_packed_floats(int cin, int cout) { return nx_split_dwords(cin, cout / 16) + 16; }

int launch_conv3d_nx(const ConvLayer& L, hipStream_t s) {
    if (!L.packed) return set_error(-1, "conv3d_nx: packed weights missing");
    ArgsNX A;
    A.a = L.a;
    A.b = L.b;
    A.wpk = L.packed;
    A.bias = L.bias;
    A.out = L.out;
    A.partials = L.partials;
    A.N = L.in.n;
    A.Cin = L.in.c;
    A.Di = L.in.d;
    A.Hi = L.in.h;
    A.Wi = L.in.w;
    A.Cout = L.out_g.c;
    A.Do = L.out_g.d;
    A.Ho = L.out_g.h;
    A.Wo = L.out_g.w;
    A.lrelu = L.lrelu;
    const NxShape shape = kNxShapes[nx_shape_index()];
    A.tiles_x = (A.Wo + 16 * shape.nb - 1) / (16 * shape.nb);
    A.tiles_y = (A.Ho + shape.ty - 1) / shape.ty;
    A.tiles = conv3d_nx_tiles(L.out_g);
    A.mblocks = A.Cout / 16;
    const size_t total = nx_split_dwords(A.Cin, A.mblocks) + 16;
    A.wtail = L.packed + total - 16;
    {
        const PackPhase phase = L.sink ? L.sink->phase : kPackInline;
        if (phase != kPackDone) {
            PackJob j;
            j.src = L.weight;
            j.dst = L.packed;
            j.cout = A.Cout;
            j.cin = A.Cin;
            j.mblocks = A.mblocks;
            j.kc = 4;
            j.taps = 27;
            j.mode = 8;
            j.total = (int)total;
            if (phase == kPackCollect) return L.sink->push(j) ? 0 : set_error(-1, "pack job table full");
            if (int rc = launch_multi_pack(&j, 1, s)) return rc;
        }
    }
    if (!L.a.bound || L.a.bound_n <= 0 || (L.b.p && (!L.b.bound || L.b.bound_n <= 0)))
        return set_error(-1, "conv3d_nx: a source without a range bound");
    switch (nx_shape_index()) {
        case 1: return launch_nx<1, 2, 2, 4>(A, s);
        case 2: return launch_nx<1, 2, 4, 2>(A, s);
        case 3: return launch_nx<1, 4, 4, 4>(A, s);
        case 4: return launch_nx<1, 2, 2, 2>(A, s);
        default: return launch_nx<1, 2, 4, 4>(A, s);
    }
}

}  // namespace pds
