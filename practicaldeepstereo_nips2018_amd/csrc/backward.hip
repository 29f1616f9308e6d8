// Backward kernels of the hot path (training: reference practical_deep_stereo/pds_trainer.py:40-46 calls
// loss.backward() through Matching and Regularization).  Correctness-first generic VALU kernels, any channel
// count; every reduction is two-stage and deterministic (no atomics).
//
// Layer forward (network_blocks.py:47-85):  z = conv(x) + b;  t = LeakyReLU(z);  y = gamma * (t - mu) * r + beta
// with mu, r = 1/sqrt(var + eps) per InstanceNorm group.  Given g = dL/dy:
//   S1 = sum g, S2 = sum g*t over the group;  Q = r * (S2 - mu * S1) = sum g * n,  n = (t - mu) * r
//   dgamma_c = sum_groups Q,  dbeta_c = sum_groups S1
//   dt = gamma * r * (g - S1/N - n * Q/N);   dz = dt * (t > 0 ? 1 : 0.1)
//   db = sum dz;  dW = correlate(x, dz);  dx = transposed-correlate(dz, W)
#include "common.hpp"

namespace pds {

// ---------------------------------------------------------------------------------------------------
// InstanceNorm + LeakyReLU backward
// ---------------------------------------------------------------------------------------------------
// partial sums of g and g*t: records [(n*C + c)][d][tile] x {S1, S2}
__global__ __launch_bounds__(256) void in_bwd_partial_kernel(const float* __restrict__ g, const float* __restrict__ t,
                                                             const Geom geom, double* __restrict__ partials) {
    const int nc = blockIdx.z, d = blockIdx.y, tile = blockIdx.x, tiles = gridDim.x;
    const size_t px = geom.plane();
    const size_t base = ((size_t)nc * geom.d + d) * px;
    double s1 = 0.0, s2 = 0.0;
    for (size_t i = (size_t)tile * 256 + threadIdx.x; i < px; i += (size_t)tiles * 256) {
        const float gv = g[base + i], tv = t[base + i];
        s1 += gv;
        s2 += (double)gv * tv;
    }
    __shared__ double red[4][2];
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave][0] = s1;
        red[wave][1] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int k = threadIdx.x;
        partials[((((size_t)nc * geom.d + d) * tiles) + tile) * 2 + k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    }
}

// per group: m1 = S1/N, m2 = Q/N, q = Q, s1 = S1 (kept for the parameter gradients)
__global__ __launch_bounds__(256) void in_bwd_finalize_kernel(const double* __restrict__ partials, int per_group,
                                                              double count, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* __restrict__ m1,
                                                              float* __restrict__ m2, double* __restrict__ qs) {
    const int grp = blockIdx.x;
    const double* p = partials + (size_t)grp * per_group * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < per_group; i += 256) {
        s1 += p[2 * i];
        s2 += p[2 * i + 1];
    }
    __shared__ double red[4][2];
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave][0] = s1;
        red[wave][1] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s1 = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        s2 = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        const double q = (double)rstd[grp] * (s2 - (double)mean[grp] * s1);
        m1[grp] = (float)(s1 / count);
        m2[grp] = (float)(q / count);
        qs[2 * grp] = q;
        qs[2 * grp + 1] = s1;
    }
}

// dgamma[c] (+)= sum over the groups of channel c of Q, dbeta[c] (+)= sum of S1.  groups = N*C*inner, group
// index = (n*C + c)*inner + d.
__global__ __launch_bounds__(64) void in_bwd_params_kernel(const double* __restrict__ qs, int n_batch, int channels,
                                                           int inner, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int accumulate,
                                                           float* __restrict__ dz_amax,
                                                           const double* __restrict__ bias_group = nullptr,
                                                           float* __restrict__ dbias = nullptr) {
    const int c = blockIdx.x;
    // the slots the apply kernel (next launch) collects max |dz| in -- the range certificate of dz (Src::bound) for the
    // fp16-split weight- and data-gradient kernels -- start from zero
    if (dz_amax && c == 0)
        for (int i = threadIdx.x; i < kDzAmaxSlots; i += 64) dz_amax[i] = 0.f;
    // (the one-pass kernel leaves ONE bias partial per group: its sum over the channel's groups rides along)
    double q = 0.0, s = 0.0, b = 0.0;
    const int per_c = n_batch * inner;
    for (int i = threadIdx.x; i < per_c; i += 64) {
        const int n = i / inner, d = i % inner;
        const size_t grp = ((size_t)n * channels + c) * inner + d;
        q += qs[2 * grp];
        s += qs[2 * grp + 1];
        if (bias_group) b += bias_group[grp];
    }
    q = wave_sum(q);
    s = wave_sum(s);
    if (bias_group) b = wave_sum(b);
    if (threadIdx.x == 0) {
        dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)q;
        dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
        if (bias_group) dbias[c] = (accumulate ? dbias[c] : 0.f) + (float)b;
    }
}

// dz = lrelu'(t) * gamma * r * (g - m1 - n * m2); every block also leaves the sum of the dz it wrote in bias_partial
// [(n*C + c)*D + d][block] (the bias gradient is that sum over everything but c: no extra pass over dz)
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ t,
                                                           const Geom geom, int per_plane,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ m1, const float* __restrict__ m2,
                                                           float* __restrict__ dz, double* __restrict__ bias_partial,
                                                           float* __restrict__ dz_amax) {
    const int nc = blockIdx.z, d = blockIdx.y;
    float seen = 0.f, poison = 0.f;
    const int c = nc % geom.c;
    const int grp = per_plane ? nc * geom.d + d : nc;
    const float mu = mean[grp], r = rstd[grp], a = gamma[c] * r, b1 = m1[grp], b2 = m2[grp];
    const size_t px = geom.plane();
    const size_t base = ((size_t)nc * geom.d + d) * px;
    float sum = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < px; i += (size_t)gridDim.x * 256) {
        const float tv = t[base + i];
        const float n = (tv - mu) * r;
        const float dt = a * (g[base + i] - b1 - n * b2);
        const float v = tv > 0.f ? dt : dt * kLeakySlope;
        dz[base + i] = v;
        sum += v;
        seen = fmaxf(seen, fabsf(v));
        poison = fmaf(v, 0.f, poison);   // NaN / inf stick (fmaxf alone drops a NaN): ADVICE r4
    }
    __shared__ double red[4];
    __shared__ float redmax[4];
    const double ws = wave_sum((double)sum);
    if (dz_amax) seen = wave_max(poison == poison ? seen : __builtin_inff());
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = ws;
        redmax[threadIdx.x >> 6] = seen;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bias_partial[((size_t)nc * geom.d + d) * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        if (dz_amax) {
            // One atomic per WORKGROUP that raises the maximum of its slot (non-negative floats order like their bit
            // patterns); a coherent read first keeps the workgroups that do not raise it off the atomic unit, and the
            // workgroups are spread over kDzAmaxSlots slots (memory channels).  A coherent read per WAVE of a kernel
            // with one element per thread tripled its duration (65 -> 180 us): hence the four elements per thread.
            const float m = fmaxf(fmaxf(redmax[0], redmax[1]), fmaxf(redmax[2], redmax[3]));
            float* slot = dz_amax + ((blockIdx.z * 7u + blockIdx.y * 13u + blockIdx.x) & (kDzAmaxSlots - 1));
            if (m > *reinterpret_cast<volatile float*>(slot))
                atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, m));
        }
    }
}

// db[c] (+)= sum over n and the [D * blocks] partials of (n, c)
__global__ __launch_bounds__(256) void bias_partial_reduce_kernel(const double* __restrict__ partial, int N, int C,
                                                                  int per_nc, float* __restrict__ db, int accumulate) {
    const int c = blockIdx.x;
    double s = 0.0;
    for (int n = 0; n < N; ++n) {
        const double* p = partial + (size_t)(n * C + c) * per_nc;
        for (int i = threadIdx.x; i < per_nc; i += 256) s += p[i];
    }
    __shared__ double red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) db[c] = (accumulate ? db[c] : 0.f) + (float)(red[0] + red[1] + red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------------
// One pass for per-plane groups (MatchingOperation: one InstanceNorm group per (channel, disparity plane), 34 560 elements
// at full size).  The two kernels above read g and t twice (statistics, then dz): 2.1 GB per layer of MatchingOperation.
// A plane fits the REGISTERS of one 1024-thread workgroup (nine 16-byte quads of g and of t per thread, all eighteen
// loads in flight): statistics and dz from one read -- 1.27 GB.  Sums in fp64 as above; the group's (Q, S1) go to the same
// `qs` records, the plane's sum of dz to the bias partials, max |dz| to the same slots.
// ---------------------------------------------------------------------------------------------------
constexpr int kPlaneThreads = 1024, kPlaneQuads = 9;

__global__ __launch_bounds__(kPlaneThreads) void in_bwd_plane_kernel(const float* __restrict__ g, const float* __restrict__ t,
                                                                   const Geom geom, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd,
                                                                   const float* __restrict__ gamma, float* __restrict__ dz,
                                                                   double* __restrict__ qs, double* __restrict__ bias_partial,
                                                                   float* __restrict__ dz_amax) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int grp = blockIdx.x;                 // (n * C + c) * D + d
    const int c = (grp / geom.d) % geom.c;
    const int quads = (int)(geom.plane() >> 2);
    const size_t base = (size_t)grp * geom.plane();
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g + base);
    const f32x4* t4 = reinterpret_cast<const f32x4*>(t + base);
    f32x4 gv[kPlaneQuads], tv[kPlaneQuads];
#pragma unroll
    for (int k = 0; k < kPlaneQuads; ++k) {
        const int q = threadIdx.x + k * kPlaneThreads;
        const bool ok = q < quads;
        gv[k] = g4[ok ? q : 0];
        tv[k] = t4[ok ? q : 0];
        if (!ok) gv[k] = f32x4{0.f, 0.f, 0.f, 0.f};   // contributes nothing to the sums; never stored
    }
    // s1 = sum g, s2 = sum g t; and, for the bias gradient (round 5), the same sums weighted by the LeakyReLU slope
    // w = 1 | 0.1 of every element: sw = sum w, swg = sum w g, swt = sum w t.  The bias gradient sum(w dt) is a difference
    // of large terms (sum dt == 0 in exact arithmetic): summed from the fp32 dz values it carried the rounding of
    // b1 = mean(g) in every one of its N terms -- 2.5e-2 of the tensor's largest entry at the 1/8-resolution level of the
    // full-size training step against 2.6e-3 for the reference's own fp32 run (fixture G13) -- so it is formed in fp64
    // from these sums instead (three more fp64 fmas per element of a bandwidth-bound pass).
    double s1 = 0.0, s2 = 0.0, sw = 0.0, swg = 0.0, swt = 0.0;
#pragma unroll
    for (int k = 0; k < kPlaneQuads; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double gd = gv[k][e], td = tv[k][e];
            const double w = tv[k][e] > 0.f ? 1.0 : (double)kLeakySlope;
            s1 += gd;
            s2 += gd * td;
            const bool real = threadIdx.x + k * kPlaneThreads < quads;
            sw += real ? w : 0.0;
            swg += w * gd;                     // (padding quads carry g == 0)
            swt += real ? w * td : 0.0;
        }
    __shared__ double red[kPlaneThreads / 64][5];
    __shared__ double tot[3];
    __shared__ float redmax[kPlaneThreads / 64];
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    sw = wave_sum(sw);
    swg = wave_sum(swg);
    swt = wave_sum(swt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave][0] = s1;
        red[wave][1] = s2;
        red[wave][2] = sw;
        red[wave][3] = swg;
        red[wave][4] = swt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a1 = 0.0, a2 = 0.0, aw = 0.0, awg = 0.0, awt = 0.0;
        for (int w = 0; w < kPlaneThreads / 64; ++w) {
            a1 += red[w][0];
            a2 += red[w][1];
            aw += red[w][2];
            awg += red[w][3];
            awt += red[w][4];
        }
        const double mu_d = (double)mean[grp], r_d = (double)rstd[grp];
        const double q = r_d * (a2 - mu_d * a1);
        tot[0] = a1;
        tot[1] = q;
        qs[2 * grp] = q;
        qs[2 * grp + 1] = a1;
        // sum_i w_i dt_i with dt_i = a (g_i - b1 - n_i b2), n_i = (t_i - mu) r
        const double cnt = (double)geom.plane();
        const double b1_d = a1 / cnt, b2_d = q / cnt;
        bias_partial[grp] = (double)gamma[c] * r_d * (awg - b1_d * aw - b2_d * r_d * (awt - mu_d * aw));
    }
    __syncthreads();
    const double count = (double)geom.plane();
    const float mu = mean[grp], r = rstd[grp], a = gamma[c] * r;
    const float b1 = (float)(tot[0] / count), b2 = (float)(tot[1] / count);
    float seen = 0.f, poison = 0.f;
    f32x4* o4 = reinterpret_cast<f32x4*>(dz + base);
#pragma unroll
    for (int k = 0; k < kPlaneQuads; ++k) {
        const int q = threadIdx.x + k * kPlaneThreads;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float n = (tv[k][e] - mu) * r;
            const float dt = a * (gv[k][e] - b1 - n * b2);
            v[e] = tv[k][e] > 0.f ? dt : dt * kLeakySlope;
        }
        if (q < quads) {
            o4[q] = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                seen = fmaxf(seen, fabsf(v[e]));
                poison = fmaf(v[e], 0.f, poison);
            }
        }
    }
    if (dz_amax) seen = wave_max(poison == poison ? seen : __builtin_inff());
    if (lane == 0) redmax[wave] = seen;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.f;
        for (int w = 0; w < kPlaneThreads / 64; ++w) m = fmaxf(m, redmax[w]);
        if (dz_amax) {   // as in_bwd_apply_kernel: one guarded atomic per workgroup
            float* slot = dz_amax + ((blockIdx.x * 7u) & (kDzAmaxSlots - 1));
            if (m > *reinterpret_cast<volatile float*>(slot))
                atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, m));
        }
    }
}

static bool in_bwd_plane_supported(const float* g, const float* t, const float* dz, const Geom& geom, int per_plane) {
    static const bool enabled = []() {  // PDS_IN_BWD_PLANE=0 keeps the two-pass kernels (A/B, debugging)
        const char* e = debug_switch("PDS_IN_BWD_PLANE");
        return !(e && e[0] == '0');
    }();
    const size_t px = geom.plane();
    return enabled && per_plane && (px & 3) == 0 && px <= (size_t)kPlaneThreads * kPlaneQuads * 4 &&
           ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(dz)) & 15) == 0;
}

static unsigned plane_tiles(const Geom& g) {
    unsigned t = (unsigned)((g.plane() + 1023) / 1024);
    return t < 1 ? 1 : (t > 64 ? 64 : t);
}

// scratch (doubles): partial records N*C*D*tiles*2 + group records 2*groups ; floats: m1, m2 [groups]
size_t in_bwd_scratch_doubles(const Geom& g) {
    return (size_t)g.n * g.c * g.d * plane_tiles(g) * 2 + (size_t)g.n * g.c * g.d * 2 +
           (size_t)g.n * g.c * g.d * plane_tiles(g) * 4;   // + the bias partials of the apply kernel
}

int launch_in_bwd(const float* g, const float* t, const Geom& geom, int per_plane, const float* mean,
                  const float* rstd, const float* gamma, double* scratch, float* m1, float* m2, float* dz,
                  float* dgamma, float* dbeta, float* dbias, int accumulate_params, hipStream_t s, float* dz_amax) {
    const unsigned tiles = plane_tiles(geom);
    double* partials = scratch;
    double* qs = scratch + (size_t)geom.n * geom.c * geom.d * tiles * 2;
    double* bias_partial = qs + (size_t)geom.n * geom.c * geom.d * 2;
    // a whole-volume group that is small enough (the deep levels of the hourglass: 12 x 36 x 60 and below) is one "plane"
    Geom view = geom;
    if (!per_plane && (size_t)geom.d * geom.plane() <= (size_t)kPlaneThreads * kPlaneQuads * 4) {
        view.w = geom.d * geom.h * geom.w;
        view.h = 1;
        view.d = 1;
    }
    if (in_bwd_plane_supported(g, t, dz, view, per_plane || view.d != geom.d || geom.d == 1)) {
        // the maxima are collected by the kernel that also produces the records in_bwd_params_kernel sums, so the slots
        // are cleared ahead of it instead of by that kernel
        if (dz_amax && hipMemsetAsync(dz_amax, 0, kDzAmaxSlots * sizeof(float), s) != hipSuccess)
            return set_error(-1, "in_bwd: clearing the range slots failed");
        hipLaunchKernelGGL(in_bwd_plane_kernel, dim3(view.n * view.c * view.d), dim3(kPlaneThreads), 0, s, g, t, view, mean,
                           rstd, gamma, dz, qs, bias_partial, dz_amax);
        hipLaunchKernelGGL(in_bwd_params_kernel, dim3(view.c), dim3(64), 0, s, qs, view.n, view.c, view.d, dgamma, dbeta,
                           accumulate_params, static_cast<float*>(nullptr), static_cast<const double*>(bias_partial), dbias);
        return check_launch("in_bwd_plane");
    }
    hipLaunchKernelGGL(in_bwd_partial_kernel, dim3(tiles, geom.d, geom.n * geom.c), dim3(256), 0, s, g, t, geom,
                       partials);
    const int inner = per_plane ? geom.d : 1;
    const int groups = geom.n * geom.c * inner;
    const int per_group = (int)tiles * (per_plane ? 1 : geom.d);
    const double count = (double)geom.plane() * (per_plane ? 1 : geom.d);
    hipLaunchKernelGGL(in_bwd_finalize_kernel, dim3(groups), dim3(256), 0, s, partials, per_group, count, mean, rstd,
                       m1, m2, qs);
    hipLaunchKernelGGL(in_bwd_params_kernel, dim3(geom.c), dim3(64), 0, s, qs, geom.n, geom.c, inner, dgamma, dbeta,
                       accumulate_params, dz_amax);
    // (one workgroup per 1024 positions of a plane: four elements per thread)
    hipLaunchKernelGGL(in_bwd_apply_kernel, dim3(tiles, geom.d, geom.n * geom.c), dim3(256), 0, s, g, t, geom,
                       per_plane, mean, rstd, gamma, m1, m2, dz, bias_partial, dz_amax);
    hipLaunchKernelGGL(bias_partial_reduce_kernel, dim3(geom.c), dim3(256), 0, s, bias_partial, geom.n, geom.c,
                       (int)(geom.d * tiles), dbias, accumulate_params);
    return check_launch("in_bwd");
}

// ---------------------------------------------------------------------------------------------------
// bias gradient: db[c] (+)= sum over n, d, y, x of dz
// ---------------------------------------------------------------------------------------------------
// two-stage: grid (C, splits) partial sums over a strided share of the positions, then one reduce.  The pass reads the
// whole of dz anyway: it also records max |dz| per workgroup (amax[split * C + c], may be null), the range certificate
// (common.hpp Src::bound) that lets the gradients of a bare layer run their fp16-split kernels.
__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const float* __restrict__ dz, const Geom geom,
                                                                  double* __restrict__ partial,
                                                                  float* __restrict__ amax) {
    const int c = blockIdx.x, split = blockIdx.y, splits = gridDim.y;
    const size_t vol = geom.volume();
    double s = 0.0;
    float m = 0.f;
    for (int n = 0; n < geom.n; ++n) {
        const float* p = dz + ((size_t)n * geom.c + c) * vol;
        for (size_t i = (size_t)split * 256 + threadIdx.x; i < vol; i += (size_t)splits * 256) {
            const float v = p[i];
            s += v;
            m = (fabsf(v) > m || v != v) ? fabsf(v) : m;   // a NaN sticks
        }
    }
    __shared__ float redm[4];
    if (amax) block_amax_record(m, amax + (size_t)split * geom.c + c, redm);
    __shared__ double red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)split * geom.c + c] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(64) void channel_sum_reduce_kernel(const double* __restrict__ partial, int channels,
                                                                int splits, float* __restrict__ db, int accumulate) {
    const int c = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < splits; i += 64) s += partial[(size_t)i * channels + c];
    s = wave_sum(s);
    if (threadIdx.x == 0) db[c] = (accumulate ? db[c] : 0.f) + (float)s;
}

int channel_sum_splits(const Geom& g) {
    const size_t work = g.volume() / 4096;
    int splits = (int)(work < 1 ? 1 : work);
    const int cap = 2048 / (g.c < 1 ? 1 : g.c);
    if (splits > cap) splits = cap;
    return splits < 1 ? 1 : splits;
}

// scratch: channel_sum_splits(g) * C doubles
// amax (optional): channel_sum_splits(g) * C floats
int launch_channel_sum(const float* dz, const Geom& g, float* db, int accumulate, double* scratch, hipStream_t s,
                       float* amax) {
    const int splits = channel_sum_splits(g);
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(g.c, splits), dim3(256), 0, s, dz, g, scratch, amax);
    hipLaunchKernelGGL(channel_sum_reduce_kernel, dim3(g.c), dim3(64), 0, s, scratch, g.c, splits, db, accumulate);
    return check_launch("channel_sum");
}

// ---------------------------------------------------------------------------------------------------
// backward data
// ---------------------------------------------------------------------------------------------------
struct BwdGeom {
    int N, Cin, Di, Hi, Wi;   // layer INPUT (the gradient being produced)
    int Cout, Do, Ho, Wo;     // layer OUTPUT (dz)
};

// conv (KD x 3 x 3, stride S, pad k/2):  dx[c][i] = sum_oc sum_k dz[oc][(i + p - k)/S] * W[oc][c][k]  (when divisible)
template <int KD, int S, int CB>
__global__ __launch_bounds__(256) void conv_bwd_data_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                            float* __restrict__ dx, const BwdGeom G) {
    const int cbs = (G.Cin + CB - 1) / CB;
    const int n = blockIdx.z / cbs, c0 = (blockIdx.z % cbs) * CB;
    const int iz = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int iy = idx / G.Wi, ix = idx % G.Wi;
    if (iy >= G.Hi) return;
    float acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = 0.f;
    const size_t plane_o = (size_t)G.Ho * G.Wo;
    for (int oc = 0; oc < G.Cout; ++oc) {
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) {
            int oz;
            if (KD == 1) {
                oz = iz;
            } else {
                const int tz = iz + 1 - kd;
                if (tz < 0 || tz % S != 0) continue;
                oz = tz / S;
                if (oz >= G.Do) continue;
            }
            const float* pz = dz + (((size_t)n * G.Cout + oc) * G.Do + oz) * plane_o;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int ty = iy + 1 - kh;
                if (ty < 0 || ty % S != 0) continue;
                const int oy = ty / S;
                if (oy >= G.Ho) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int tx = ix + 1 - kw;
                    if (tx < 0 || tx % S != 0) continue;
                    const int ox = tx / S;
                    if (ox >= G.Wo) continue;
                    const float v = pz[(size_t)oy * G.Wo + ox];
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        const int ci = min(c0 + c, G.Cin - 1);
                        acc[c] = fmaf(v, w[(((size_t)oc * G.Cin + ci) * KD + kd) * 9 + kh * 3 + kw], acc[c]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CB; ++c)
        if (c0 + c < G.Cin)
            dx[(((size_t)n * G.Cin + c0 + c) * G.Di + iz) * G.Hi * G.Wi + (size_t)iy * G.Wi + ix] = acc[c];
}

// transposed conv (kernel (KD,4,4), stride (KD == 4 ? 2 : 1, 2, 2), pad 1; weight [Cin, Cout, KD, 4, 4]):
//   forward out[oc][o] += x[c][i] * W[c][oc][k] for o = s*i - 1 + k   =>   dx[c][i] = sum_oc sum_k dz[oc][s*i - 1 + k] * W[c][oc][k]
template <int KD, int CB>
__global__ __launch_bounds__(256) void deconv_bwd_data_kernel(const float* __restrict__ dz,
                                                              const float* __restrict__ w, float* __restrict__ dx,
                                                              const BwdGeom G) {
    constexpr int SD = KD == 4 ? 2 : 1;
    const int cbs = (G.Cin + CB - 1) / CB;
    const int n = blockIdx.z / cbs, c0 = (blockIdx.z % cbs) * CB;
    const int iz = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int iy = idx / G.Wi, ix = idx % G.Wi;
    if (iy >= G.Hi) return;
    float acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = 0.f;
    const size_t plane_o = (size_t)G.Ho * G.Wo;
    for (int oc = 0; oc < G.Cout; ++oc) {
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) {
            const int oz = SD * iz - 1 + kd;
            if (oz < 0 || oz >= G.Do) continue;
            const float* pz = dz + (((size_t)n * G.Cout + oc) * G.Do + oz) * plane_o;
#pragma unroll
            for (int kh = 0; kh < 4; ++kh) {
                const int oy = 2 * iy - 1 + kh;
                if (oy < 0 || oy >= G.Ho) continue;
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    const int ox = 2 * ix - 1 + kw;
                    if (ox < 0 || ox >= G.Wo) continue;
                    const float v = pz[(size_t)oy * G.Wo + ox];
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        const int ci = min(c0 + c, G.Cin - 1);
                        acc[c] = fmaf(v, w[(((size_t)ci * G.Cout + oc) * KD + kd) * 16 + kh * 4 + kw], acc[c]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CB; ++c)
        if (c0 + c < G.Cin)
            dx[(((size_t)n * G.Cin + c0 + c) * G.Di + iz) * G.Hi * G.Wi + (size_t)iy * G.Wi + ix] = acc[c];
}

// Wf[c][oc][taps-1-t] = W[oc][c][t]: the data gradient of a stride-1 "same" convolution is the convolution of dz
// with these flipped, channel-swapped weights, so the forward (MFMA) kernels can compute it.
__global__ __launch_bounds__(256) void flip_weights_kernel(const float* __restrict__ w, float* __restrict__ wf,
                                                           int cout, int cin, int taps) {
    const int total = cout * cin * taps;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int t = i % taps;
        const int c = (i / taps) % cin;
        const int oc = i / (taps * cin);
        wf[((size_t)c * cout + oc) * taps + (taps - 1 - t)] = w[i];
    }
}

int launch_flip_weights(const float* w, float* wf, int cout, int cin, int taps, hipStream_t s) {
    const int total = cout * cin * taps;
    hipLaunchKernelGGL(flip_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, wf, cout, cin, taps);
    return check_launch("flip_weights");
}

// ---------------------------------------------------------------------------------------------------
// Data gradients of the stride-2 layers, second form (round 4).  The kernels above give a thread ONE input position and
// walk every output channel and every tap in sequence: the transposed layers issue one strided 4-byte load per 4 FMAs,
// the strided convolutions test all 27 taps for divisibility per lane (1 to 8 of them contribute), and the small
// levels of the hourglass (a few thousand positions, 64-128 output channels) run a chain of thousands of dependent
// iterations on a handful of workgroups.  Here:
//   unit    one WAVE: one plane iz, RY rows, PW = 64 / RY pairs of adjacent columns (2 j, 2 j + 1) and CB input
//           channels; a lane owns a column pair, so the loads of a row are shared by both positions (transposed: 6
//           loads for 8 taps x CB FMAs; strided convolution: 2 loads for 3 taps x CB FMAs) and the stores are dense.
//   parity  strided convolution: the wave's plane and rows all have ONE parity each (rows 2 r + py), so the taps
//           that contribute are known per wave -- no divisibility tests, no masked-off work.
//   OCG     the output channels are dealt to OCG (1, 2 or 4) waves of a workgroup, summed through LDS in wave order
//           (deterministic): the small levels get 4 x the waves and a quarter of the chain.
//   weights wave-uniform addresses: scalar loads.
// ---------------------------------------------------------------------------------------------------
struct Bwd2Launch {
    int ry_shift;   // log2(RY)
    int segs;       // column-pair segments per row
    int yblocks;    // row blocks (per parity for the strided convolution)
    int cbs;
    int units;
};

template <int CB, int OCG>
__device__ __forceinline__ void bwd2_reduce_store(float (&acc0)[CB], float (&acc1)[CB], float* red, int wave, int og,
                                                  int lane, bool live, bool p0, bool p1, float* __restrict__ dst,
                                                  size_t cstride, int nch) {
    if (OCG > 1) {
        // [wave][2 CB][64 lanes]
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            red[(wave * 2 * CB + 2 * c) * 64 + lane] = acc0[c];
            red[(wave * 2 * CB + 2 * c + 1) * 64 + lane] = acc1[c];
        }
        __syncthreads();
        if (og != 0) return;
#pragma unroll
        for (int k = 1; k < OCG; ++k)
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                acc0[c] += red[((wave + k) * 2 * CB + 2 * c) * 64 + lane];
                acc1[c] += red[((wave + k) * 2 * CB + 2 * c + 1) * 64 + lane];
            }
    }
    if (!live) return;
#pragma unroll
    for (int c = 0; c < CB; ++c)
        if (c < nch) {
            if (p0) dst[(size_t)c * cstride] = acc0[c];
            if (p1) dst[(size_t)c * cstride + 1] = acc1[c];
        }
}

// transposed conv (kernel (KD,4,4), stride (KD == 4 ? 2 : 1, 2, 2), pad 1; weight [Cin, Cout, KD, 4, 4]):
//   dx[c][i] = sum_oc sum_k dz[oc][s*i - 1 + k] * W[c][oc][k]
// VEC (dz rows 16-byte aligned): the four inner columns of a lane's six are ONE aligned 16-byte load
// (the lanes' quads are contiguous: 1 KB per instruction) instead of four 4-byte loads at a 16-byte stride, which cost the
// texture path as much each.
template <int KD, int CB, int OCG, bool VEC>
__global__ __launch_bounds__(256) void deconv_bwd_data2_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                               float* __restrict__ dx, const BwdGeom G,
                                                               const Bwd2Launch Q) {
    constexpr int SD = KD == 4 ? 2 : 1;
    __shared__ float red[OCG > 1 ? 4 * 2 * CB * 64 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * 4 + wave;
    const int og = gw % OCG;
    int u = gw / OCG;
    const bool live_unit = u < Q.units;
    u = min(u, Q.units - 1);
    const int seg = u % Q.segs;
    u /= Q.segs;
    const int yb = u % Q.yblocks;
    u /= Q.yblocks;
    const int iz = u % G.Di;
    u /= G.Di;
    const int c0 = (u % Q.cbs) * CB, n = u / Q.cbs;
    const int pw = 64 >> Q.ry_shift;
    const int iy = (yb << Q.ry_shift) + (lane / pw);
    const int j = seg * pw + (lane & (pw - 1));
    const int ix0 = 2 * j;
    const bool rowok = iy < G.Hi;
    const bool p0 = rowok && ix0 < G.Wi, p1 = rowok && ix0 + 1 < G.Wi;

    int col[6];
    bool cv[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int ox = 2 * ix0 - 1 + t;
        cv[t] = ox >= 0 && ox < G.Wo;
        col[t] = min(max(ox, 0), G.Wo - 1);
    }
    if (VEC) col[1] = min(col[1], G.Wo - 4);   // lanes past the end of the row: any aligned quad inside it (masked by cv)
    float acc0[CB], acc1[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc0[c] = acc1[c] = 0.f;
    const size_t plane_o = (size_t)G.Ho * G.Wo;
    if (live_unit)
        for (int oc = og; oc < G.Cout; oc += OCG) {
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) {
                const int oz = SD * iz - 1 + kd;
                if (oz < 0 || oz >= G.Do) continue;   // wave-uniform
                const float* pz = dz + (((size_t)n * G.Cout + oc) * G.Do + oz) * plane_o;
#pragma unroll
                for (int kh = 0; kh < 4; ++kh) {
                    const int oy = 2 * iy - 1 + kh;
                    const bool yv = oy >= 0 && oy < G.Ho;
                    const float* pr = pz + (size_t)min(max(oy, 0), G.Ho - 1) * G.Wo;
                    float v[6];
                    if (VEC) {
                        typedef float f32x4 __attribute__((ext_vector_type(4)));
                        const f32x4 m = *reinterpret_cast<const f32x4*>(pr + col[1]);   // columns 4 j .. 4 j + 3
                        v[0] = pr[col[0]];
                        v[5] = pr[col[5]];
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[1 + t] = m[t];
                    } else {
#pragma unroll
                        for (int t = 0; t < 6; ++t) v[t] = pr[col[t]];
                    }
#pragma unroll
                    for (int t = 0; t < 6; ++t) v[t] = (yv && cv[t]) ? v[t] : 0.f;
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        const int ci = min(c0 + c, G.Cin - 1);
                        const float* wp = w + (((size_t)ci * G.Cout + oc) * KD + kd) * 16 + kh * 4;
                        const float w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
                        acc0[c] = fmaf(v[0], w0, fmaf(v[1], w1, fmaf(v[2], w2, fmaf(v[3], w3, acc0[c]))));
                        acc1[c] = fmaf(v[2], w0, fmaf(v[3], w1, fmaf(v[4], w2, fmaf(v[5], w3, acc1[c]))));
                    }
                }
            }
        }
    const size_t cstride = (size_t)G.Di * G.Hi * G.Wi;
    float* dst = dx + ((size_t)n * G.Cin + c0) * cstride + ((size_t)iz * G.Hi + min(iy, G.Hi - 1)) * G.Wi + min(ix0, G.Wi - 1);
    bwd2_reduce_store<CB, OCG>(acc0, acc1, red, wave, og, lane, live_unit, p0, p1, dst, cstride, G.Cin - c0);
}

// conv (3 x 3 x 3, stride 2, pad 1; weight [Cout, Cin, 3, 3, 3]):  dx[c][i] = sum over (o, k) with 2 o - 1 + k = i.
// Per axis: i even -> k = 1, o = i / 2;  i odd -> k = 0, o = (i + 1) / 2 (if it exists) and k = 2, o = (i - 1) / 2.
template <int CB, int OCG>
__global__ __launch_bounds__(256) void conv_s2_bwd_data2_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                                float* __restrict__ dx, const BwdGeom G,
                                                                const Bwd2Launch Q) {
    __shared__ float red[OCG > 1 ? 4 * 2 * CB * 64 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * 4 + wave;
    const int og = gw % OCG;
    int u = gw / OCG;
    const bool live_unit = u < Q.units;
    u = min(u, Q.units - 1);
    const int seg = u % Q.segs;
    u /= Q.segs;
    const int yb = u % Q.yblocks;
    u /= Q.yblocks;
    const int py = u & 1;
    u >>= 1;
    const int iz = u % G.Di;
    u /= G.Di;
    const int c0 = (u % Q.cbs) * CB, n = u / Q.cbs;
    const int pw = 64 >> Q.ry_shift;
    const int iy = 2 * ((yb << Q.ry_shift) + (lane / pw)) + py;
    const int j = seg * pw + (lane & (pw - 1));
    const int ix0 = 2 * j;
    const bool rowok = iy < G.Hi;
    const bool p0 = rowok && ix0 < G.Wi, p1 = rowok && ix0 + 1 < G.Wi;

    // the contributing taps of the plane (wave-uniform) and of the row (same count for every lane, positions differ)
    int nz, kdl[2], ozl[2];
    if (iz & 1) {
        nz = 0;
        if ((iz + 1) / 2 < G.Do) {
            kdl[nz] = 0;
            ozl[nz++] = (iz + 1) / 2;
        }
        kdl[nz] = 2;
        ozl[nz++] = (iz - 1) / 2;
    } else {
        nz = 1;
        kdl[0] = 1;
        ozl[0] = iz / 2;
    }
    const int ny = py ? 2 : 1;
    const bool xv0 = j < G.Wo, xv1 = j + 1 < G.Wo;
    const int xc0 = min(j, G.Wo - 1), xc1 = min(j + 1, G.Wo - 1);

    float acc0[CB], acc1[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc0[c] = acc1[c] = 0.f;
    const size_t plane_o = (size_t)G.Ho * G.Wo;
    if (live_unit)
        for (int oc = og; oc < G.Cout; oc += OCG) {
            for (int a = 0; a < nz; ++a) {
                const float* pz = dz + (((size_t)n * G.Cout + oc) * G.Do + ozl[a]) * plane_o;
                for (int b = 0; b < ny; ++b) {
                    // row taps: even row -> kh = 1, oy = iy / 2; odd row -> (kh = 0, oy = (iy + 1) / 2), (kh = 2, oy = (iy - 1) / 2)
                    const int kh = py ? 2 * b : 1;
                    const int oy = py ? (b == 0 ? (iy + 1) / 2 : (iy - 1) / 2) : iy / 2;
                    const bool yv = oy < G.Ho;
                    const float* pr = pz + (size_t)min(oy, G.Ho - 1) * G.Wo;
                    float v0 = pr[xc0], v1 = pr[xc1];
                    v0 = (yv && xv0) ? v0 : 0.f;
                    v1 = (yv && xv1) ? v1 : 0.f;
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        const int ci = min(c0 + c, G.Cin - 1);
                        const float* wp = w + ((((size_t)oc * G.Cin + ci) * 3 + kdl[a]) * 3 + kh) * 3;
                        acc0[c] = fmaf(v0, wp[1], acc0[c]);                      // even column: kw = 1, ox = j
                        acc1[c] = fmaf(v1, wp[0], fmaf(v0, wp[2], acc1[c]));     // odd column: kw = 0 at j + 1, kw = 2 at j
                    }
                }
            }
        }
    const size_t cstride = (size_t)G.Di * G.Hi * G.Wi;
    float* dst = dx + ((size_t)n * G.Cin + c0) * cstride + ((size_t)iz * G.Hi + min(iy, G.Hi - 1)) * G.Wi + min(ix0, G.Wi - 1);
    bwd2_reduce_store<CB, OCG>(acc0, acc1, red, wave, og, lane, live_unit, p0, p1, dst, cstride, G.Cin - c0);
}

namespace {

template <int CB, int OCG>
void launch_bwd2_variant(int transposed, int kd, const float* dz, const float* w, float* dx, const BwdGeom& G,
                         const Bwd2Launch& Q, hipStream_t s) {
    const unsigned wgs = (unsigned)(((size_t)Q.units * OCG + 3) / 4);
    if (!transposed)
        hipLaunchKernelGGL((conv_s2_bwd_data2_kernel<CB, OCG>), dim3(wgs), dim3(256), 0, s, dz, w, dx, G, Q);
    else {
        const bool vec = (G.Wo & 3) == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0;
        if (kd == 4 && vec)
            hipLaunchKernelGGL((deconv_bwd_data2_kernel<4, CB, OCG, true>), dim3(wgs), dim3(256), 0, s, dz, w, dx, G, Q);
        else if (kd == 4)
            hipLaunchKernelGGL((deconv_bwd_data2_kernel<4, CB, OCG, false>), dim3(wgs), dim3(256), 0, s, dz, w, dx, G, Q);
        else if (vec)
            hipLaunchKernelGGL((deconv_bwd_data2_kernel<3, CB, OCG, true>), dim3(wgs), dim3(256), 0, s, dz, w, dx, G, Q);
        else
            hipLaunchKernelGGL((deconv_bwd_data2_kernel<3, CB, OCG, false>), dim3(wgs), dim3(256), 0, s, dz, w, dx, G, Q);
    }
}

// transposed (kd 3 or 4) or strided convolution (kd 3, stride 2)
int launch_bwd_data2(int transposed, int kd, const float* dz, const float* w, float* dx, const BwdGeom& G, hipStream_t s) {
    Bwd2Launch Q;
    const int pairs = (G.Wi + 1) / 2;
    int pw = 64;
    while (pw > 8 && pw / 2 >= pairs) pw /= 2;        // narrow levels: several rows per wave
    Q.ry_shift = pw == 64 ? 0 : pw == 32 ? 1 : pw == 16 ? 2 : 3;
    const int ry = 64 / pw;
    Q.segs = (pairs + pw - 1) / pw;
    const int rows = transposed ? G.Hi : (G.Hi + 1) / 2;   // rows per parity for the strided convolution
    Q.yblocks = (rows + ry - 1) / ry;
    // 8 input channels per wave share its loads best; the deep levels (a few hundred units) take 4 for twice the waves
    int cb = (G.Cin % 8 == 0) ? 8 : 4;
    const int per_cb = G.N * G.Di * (transposed ? 1 : 2) * Q.yblocks * Q.segs;
    if (cb == 8 && per_cb * (G.Cin / 8) < 512) cb = 4;
    Q.cbs = (G.Cin + cb - 1) / cb;
    Q.units = per_cb * Q.cbs;
    // enough waves to fill the chip (256 CUs x 16 wave slots) before the output channels stay with one wave
    int ocg = 1;
    while (ocg < 4 && (size_t)Q.units * ocg < 4096 && G.Cout >= 8 * ocg) ocg *= 2;
    if (cb == 8) {
        if (ocg == 1) launch_bwd2_variant<8, 1>(transposed, kd, dz, w, dx, G, Q, s);
        else if (ocg == 2) launch_bwd2_variant<8, 2>(transposed, kd, dz, w, dx, G, Q, s);
        else launch_bwd2_variant<8, 4>(transposed, kd, dz, w, dx, G, Q, s);
    } else {
        if (ocg == 1) launch_bwd2_variant<4, 1>(transposed, kd, dz, w, dx, G, Q, s);
        else if (ocg == 2) launch_bwd2_variant<4, 2>(transposed, kd, dz, w, dx, G, Q, s);
        else launch_bwd2_variant<4, 4>(transposed, kd, dz, w, dx, G, Q, s);
    }
    return check_launch("bwd_data2");
}

}  // namespace

int launch_bwd_data(int transposed, int kd, int stride, const float* dz, const float* w, float* dx, const Geom& in,
                    const Geom& out, hipStream_t s) {
    BwdGeom G{in.n, in.c, in.d, in.h, in.w, out.c, out.d, out.h, out.w};
    static const bool v2 = []() {   // PDS_BWD_DATA_V2=0 keeps the one-position-per-thread kernels (A/B, debugging)
        const char* e = debug_switch("PDS_BWD_DATA_V2");
        return !(e && e[0] == '0');
    }();
    if (v2 && ((transposed && (kd == 3 || kd == 4)) || (!transposed && kd == 3 && stride == 2)))
        return launch_bwd_data2(transposed, kd, dz, w, dx, G, s);
    constexpr int CB = 4;
    dim3 grid((unsigned)((in.h * in.w + 255) / 256), in.d, in.n * ((in.c + CB - 1) / CB));
    if (!transposed) {
        if (kd == 1 && stride == 1)
            hipLaunchKernelGGL((conv_bwd_data_kernel<1, 1, CB>), grid, dim3(256), 0, s, dz, w, dx, G);
        else if (kd == 3 && stride == 1)
            hipLaunchKernelGGL((conv_bwd_data_kernel<3, 1, CB>), grid, dim3(256), 0, s, dz, w, dx, G);
        else if (kd == 3 && stride == 2)
            hipLaunchKernelGGL((conv_bwd_data_kernel<3, 2, CB>), grid, dim3(256), 0, s, dz, w, dx, G);
        else
            return set_error(-1, "bwd_data: unsupported conv kd=%d stride=%d", kd, stride);
    } else {
        if (kd == 4)
            hipLaunchKernelGGL((deconv_bwd_data_kernel<4, CB>), grid, dim3(256), 0, s, dz, w, dx, G);
        else if (kd == 3)
            hipLaunchKernelGGL((deconv_bwd_data_kernel<3, CB>), grid, dim3(256), 0, s, dz, w, dx, G);
        else
            return set_error(-1, "bwd_data: unsupported deconv kd=%d", kd);
    }
    return check_launch("bwd_data");
}

// ---------------------------------------------------------------------------------------------------
// backward weight: one block per (oc, c) pair, TAPS running sums per thread over all positions
// ---------------------------------------------------------------------------------------------------
struct BwdWArgs {
    Src a, b;      // the layer's input (deferred-normalised sources, as in the forward)
    const float* __restrict__ dz;
    double* __restrict__ partial;  // [split][weight element]
    BwdGeom G;
};

__device__ __forceinline__ float src_value(const Src& a, const Src& b, int n, int C, int c, int D, int d, int H,
                                           int W, int y, int x) {
    // normalised input value at an in-range position
    const size_t plane = (size_t)H * W;
    const size_t off = (size_t)y * W + x;
    float v;
    {
        float sc = 1.f, sh = 0.f;
        if (a.scale) {
            const int grp = a.per_plane ? ((n * C + c) * D + d) : (n * C + c);
            sc = a.scale[grp];
            sh = a.shift[grp];
        }
        v = fmaf(sc, a.p[((size_t)(n * C + c) * D + d) * plane + off], sh);
    }
    if (b.p) {
        float sc = 1.f, sh = 0.f;
        if (b.scale) {
            const int grp = b.per_plane ? ((n * C + c) * D + d) : (n * C + c);
            sc = b.scale[grp];
            sh = b.shift[grp];
        }
        const size_t base = b.bcast_d ? (size_t)(n * C + c) * plane : ((size_t)(n * C + c) * D + d) * plane;
        v += fmaf(sc, b.p[base + off], sh);
    }
    return v;
}

// conv: dW[oc][c][k] = sum_{n,o} dz[oc][o] * x[c][o*S - p + k]
template <int KD, int S>
__global__ __launch_bounds__(256) void conv_bwd_weight_kernel(const BwdWArgs A) {
    constexpr int TAPS = KD * 9;
    const BwdGeom& G = A.G;
    const int oc = blockIdx.x, c = blockIdx.y;
    double acc[TAPS];  // fp64: each thread sums thousands of signed terms
#pragma unroll
    for (int k = 0; k < TAPS; ++k) acc[k] = 0.0;
    const size_t vol_o = (size_t)G.Do * G.Ho * G.Wo;
    for (int n = 0; n < G.N; ++n) {
        const float* pz = A.dz + ((size_t)n * G.Cout + oc) * vol_o;
        for (size_t o = (size_t)blockIdx.z * 256 + threadIdx.x; o < vol_o; o += (size_t)gridDim.z * 256) {
            const float dzv = pz[o];
            const int ox = (int)(o % G.Wo);
            const int oy = (int)((o / G.Wo) % G.Ho);
            const int oz = (int)(o / ((size_t)G.Wo * G.Ho));
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) {
                const int iz = KD == 1 ? oz : oz * S - 1 + kd;
                if (iz < 0 || iz >= G.Di) continue;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int iy = oy * S - 1 + kh;
                    if (iy < 0 || iy >= G.Hi) continue;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int ix = ox * S - 1 + kw;
                        if (ix < 0 || ix >= G.Wi) continue;
                        const float xv = src_value(A.a, A.b, n, G.Cin, c, G.Di, iz, G.Hi, G.Wi, iy, ix);
                        acc[(kd * 3 + kh) * 3 + kw] += (double)dzv * (double)xv;
                    }
                }
            }
        }
    }
    __shared__ double red[4][TAPS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < TAPS; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < TAPS) {
        const int k = threadIdx.x;
        const size_t wcount = (size_t)G.Cout * G.Cin * TAPS;
        A.partial[(size_t)blockIdx.z * wcount + ((size_t)oc * G.Cin + c) * TAPS + k] =
            red[0][k] + red[1][k] + red[2][k] + red[3][k];
    }
}

// transposed conv: dW[c][oc][k] = sum_{n,i} x[c][i] * dz[oc][s*i - 1 + k]
template <int KD>
__global__ __launch_bounds__(256) void deconv_bwd_weight_kernel(const BwdWArgs A) {
    constexpr int TAPS = KD * 16;
    constexpr int SD = KD == 4 ? 2 : 1;
    const BwdGeom& G = A.G;
    const int oc = blockIdx.x, c = blockIdx.y;
    double acc[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; ++k) acc[k] = 0.0;
    const size_t vol_i = (size_t)G.Di * G.Hi * G.Wi;
    const size_t plane_o = (size_t)G.Ho * G.Wo;
    for (int n = 0; n < G.N; ++n) {
        for (size_t i = (size_t)blockIdx.z * 256 + threadIdx.x; i < vol_i; i += (size_t)gridDim.z * 256) {
            const int ix = (int)(i % G.Wi);
            const int iy = (int)((i / G.Wi) % G.Hi);
            const int iz = (int)(i / ((size_t)G.Wi * G.Hi));
            const float xv = src_value(A.a, A.b, n, G.Cin, c, G.Di, iz, G.Hi, G.Wi, iy, ix);
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) {
                const int oz = SD * iz - 1 + kd;
                if (oz < 0 || oz >= G.Do) continue;
                const float* pz = A.dz + (((size_t)n * G.Cout + oc) * G.Do + oz) * plane_o;
#pragma unroll
                for (int kh = 0; kh < 4; ++kh) {
                    const int oy = 2 * iy - 1 + kh;
                    if (oy < 0 || oy >= G.Ho) continue;
#pragma unroll
                    for (int kw = 0; kw < 4; ++kw) {
                        const int ox = 2 * ix - 1 + kw;
                        if (ox < 0 || ox >= G.Wo) continue;
                        acc[(kd * 4 + kh) * 4 + kw] += (double)xv * (double)pz[(size_t)oy * G.Wo + ox];
                    }
                }
            }
        }
    }
    __shared__ double red[4][TAPS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < TAPS; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < TAPS) {
        const int k = threadIdx.x;
        const size_t wcount = (size_t)G.Cout * G.Cin * TAPS;
        A.partial[(size_t)blockIdx.z * wcount + ((size_t)c * G.Cout + oc) * TAPS + k] =
            red[0][k] + red[1][k] + red[2][k] + red[3][k];
    }
}

__global__ __launch_bounds__(256) void weight_reduce_kernel(const double* __restrict__ partial, size_t wcount,
                                                            int splits, float* __restrict__ dw, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < wcount; i += (size_t)gridDim.x * 256) {
        double s = 0.0;
        for (int k = 0; k < splits; ++k) s += partial[(size_t)k * wcount + i];
        dw[i] = (accumulate ? dw[i] : 0.f) + (float)s;
    }
}

static int weight_taps(int transposed, int kd) { return transposed ? kd * 16 : kd * 9; }

// position splits: enough workgroups to fill the chip when Cout * Cin is small
int bwd_weight_splits(int transposed, const Geom& in, const Geom& out) {
    const size_t positions = transposed ? in.volume() : out.volume();
    size_t splits = positions / 8192;
    const size_t cap = 4096 / ((size_t)in.c * out.c < 1 ? 1 : (size_t)in.c * out.c);
    if (splits > cap) splits = cap;
    if (splits > 512) splits = 512;
    return splits < 1 ? 1 : (int)splits;
}

size_t bwd_weight_scratch_doubles(int transposed, int kd, const Geom& in, const Geom& out) {
    return (size_t)bwd_weight_splits(transposed, in, out) * in.c * out.c * weight_taps(transposed, kd);
}

int launch_bwd_weight(int transposed, int kd, int stride, const Src& a, const Src& b, const float* dz, float* dw,
                      const Geom& in, const Geom& out, int accumulate, double* scratch, hipStream_t s) {
    BwdWArgs A;
    A.a = a;
    A.b = b;
    A.dz = dz;
    A.partial = scratch;
    A.G = BwdGeom{in.n, in.c, in.d, in.h, in.w, out.c, out.d, out.h, out.w};
    const int splits = bwd_weight_splits(transposed, in, out);
    dim3 grid(out.c, in.c, splits);
    if (!transposed) {
        if (kd == 1 && stride == 1)
            hipLaunchKernelGGL((conv_bwd_weight_kernel<1, 1>), grid, dim3(256), 0, s, A);
        else if (kd == 3 && stride == 1)
            hipLaunchKernelGGL((conv_bwd_weight_kernel<3, 1>), grid, dim3(256), 0, s, A);
        else if (kd == 3 && stride == 2)
            hipLaunchKernelGGL((conv_bwd_weight_kernel<3, 2>), grid, dim3(256), 0, s, A);
        else
            return set_error(-1, "bwd_weight: unsupported conv kd=%d stride=%d", kd, stride);
    } else {
        if (kd == 4)
            hipLaunchKernelGGL((deconv_bwd_weight_kernel<4>), grid, dim3(256), 0, s, A);
        else if (kd == 3)
            hipLaunchKernelGGL((deconv_bwd_weight_kernel<3>), grid, dim3(256), 0, s, A);
        else
            return set_error(-1, "bwd_weight: unsupported deconv kd=%d", kd);
    }
    const size_t wcount = (size_t)in.c * out.c * weight_taps(transposed, kd);
    unsigned bx = (unsigned)((wcount + 255) / 256);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(weight_reduce_kernel, dim3(bx), dim3(256), 0, s, scratch, wcount, splits, dw, accumulate);
    return check_launch("bwd_weight");
}

// ---------------------------------------------------------------------------------------------------
// gradient routing: dst (+)= src ; dst[n,c,y,x] (+)= sum_d src[n,c,d,y,x] (broadcast source)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_add_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                       size_t count, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
        dst[i] = accumulate ? dst[i] + src[i] : src[i];
}

__global__ __launch_bounds__(256) void grad_reduce_d_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                            const Geom g, int accumulate) {
    const size_t px = g.plane();
    const int nc = blockIdx.y;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < px; i += (size_t)gridDim.x * 256) {
        float s = 0.f;
        for (int d = 0; d < g.d; ++d) s += src[((size_t)nc * g.d + d) * px + i];
        dst[(size_t)nc * px + i] = accumulate ? dst[(size_t)nc * px + i] + s : s;
    }
}

int launch_grad_add(float* dst, const float* src, size_t count, int accumulate, hipStream_t s) {
    unsigned bx = (unsigned)((count + 255) / 256);
    if (bx > 16384) bx = 16384;
    hipLaunchKernelGGL(grad_add_kernel, dim3(bx), dim3(256), 0, s, dst, src, count, accumulate);
    return check_launch("grad_add");
}

int launch_grad_reduce_d(float* dst, const float* src, const Geom& g, int accumulate, hipStream_t s) {
    unsigned bx = (unsigned)((g.plane() + 255) / 256);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(grad_reduce_d_kernel, dim3(bx, g.n * g.c), dim3(256), 0, s, dst, src, g, accumulate);
    return check_launch("grad_reduce_d");
}

// shift_concat backward (matching.py:50-61): g [d_count, B, 2C, h, w] ->
//   dleft[b,c,y,x] = sum_d g[d,b,c,y,x];  dright[b,c,y,x'] = sum_d g[d,b,C+c,y,x'+d] (x'+d < w)
__global__ __launch_bounds__(256) void shift_concat_bwd_kernel(const float* __restrict__ g, float* __restrict__ dleft,
                                                               float* __restrict__ dright, int batch, int C, int h,
                                                               int w, int d_begin, int d_count) {
    const size_t rows = (size_t)batch * C * h;
    for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const int y = (int)(row % h);
        const int c = (int)((row / h) % C);
        const int b = (int)(row / ((size_t)h * C));
        for (int x = threadIdx.x; x < w; x += 256) {
            float sl = 0.f, sr = 0.f;
            for (int di = 0; di < d_count; ++di) {
                const int d = d_begin + di;
                const float* gl = g + ((((size_t)di * batch + b) * 2 * C + c) * h + y) * w;
                const float* gr = g + ((((size_t)di * batch + b) * 2 * C + C + c) * h + y) * w;
                sl += gl[x];
                if (x + d < w) sr += gr[x + d];
            }
            dleft[row * w + x] = sl;
            dright[row * w + x] = sr;
        }
    }
}

int launch_shift_concat_bwd(const float* g, float* dleft, float* dright, int batch, int channels, int h, int w,
                            int d_begin, int d_count, hipStream_t s) {
    const size_t rows = (size_t)batch * channels * h;
    const unsigned grid = (unsigned)(rows < 65536 ? rows : 65536);
    hipLaunchKernelGGL(shift_concat_bwd_kernel, dim3(grid), dim3(256), 0, s, g, dleft, dright, batch, channels, h, w,
                       d_begin, d_count);
    return check_launch("shift_concat_bwd");
}

}  // namespace pds
