// Shared declarations of libpds_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/pds_hip.h"

namespace pds {

// Kernel-selection switches (DESIGN.md "Switches": PDS_X3, PDS_WINOGRAD, PDS_CONV3D_KS, ...) exist for A/B measurements and
// for tests/test_gpu_switches.py.  They are honoured only when PDS_DEBUG_SWITCHES=1 is set as well, so that a stray
// variable in a production environment cannot move the library off its measured-best, parity-tested default paths.
inline const char* debug_switch(const char* name) {
    static const bool armed = []() {
        const char* e = std::getenv("PDS_DEBUG_SWITCHES");
        return e && e[0] == '1';
    }();
    return armed ? std::getenv(name) : nullptr;
}

constexpr float kLeakySlope = 0.1f;  // reference network_blocks.py:57,71,84
constexpr double kInEps = 1e-5;      // torch InstanceNorm default eps

// ---- error reporting across the C ABI ------------------------------------------------
int set_error(int code, const char* fmt, ...);
int check_launch(const char* what);
// Launch probe (pds_probe_begin / pds_probe_end, api.hip): a launcher brackets its launch with
//     const int probe = probe_before("conv2d_x3", stream);  <launch>  probe_after(probe, workgroups, stream);
// (one relaxed atomic load when the probe is not armed)
int probe_before(const char* name, hipStream_t s);
void probe_after(int slot, int workgroups, hipStream_t s);
long long nonfinite_statistics(int reset);   // conv_direct.hip: the host-mapped counter behind pds_nonfinite_statistics
unsigned* nonfinite_counter(hipStream_t s);  // its device-visible address (nullptr: unavailable, e.g. while capturing)

// Per-function attributes (hipFuncSetAttribute) and per-device launch data are set up once per DEVICE of the process:
//     static std::atomic<unsigned> done{0};          // one bit per device
//     if (DeviceOnce once{done}) { ... set up ... }
// The first caller on a device runs the body holding a mutex and publishes the bit (release) when the guard leaves
// the if statement; a concurrent first caller (nn.DataParallel threads, a threaded server) waits on the mutex and
// then sees the bit, so nobody launches before the set-up is complete.
struct DeviceOnce {
    std::atomic<unsigned>& done;
    unsigned bit = 0;
    bool first = false;
    explicit DeviceOnce(std::atomic<unsigned>& d) : done(d) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        bit = 1u << (dev & 31);
        if (done.load(std::memory_order_acquire) & bit) return;
        mutex().lock();
        if (done.load(std::memory_order_relaxed) & bit) {
            mutex().unlock();
            return;
        }
        first = true;
    }
    ~DeviceOnce() {
        if (first) {
            done.fetch_or(bit, std::memory_order_release);
            mutex().unlock();
        }
    }
    DeviceOnce(const DeviceOnce&) = delete;
    DeviceOnce& operator=(const DeviceOnce&) = delete;
    explicit operator bool() const { return first; }
    static std::mutex& mutex() {
        static std::mutex m;
        return m;
    }
};

#define PDS_REQUIRE(cond, ...)                       \
    do {                                             \
        if (!(cond)) return pds::set_error(-1, __VA_ARGS__); \
    } while (0)

// ---- a "deferred-normalisation" input source -------------------------------------------
// Convolution kernels read their input as  scale * raw + shift  where (scale, shift) fold the
// InstanceNorm of the PRODUCING layer (conv -> LeakyReLU -> IN, network_blocks.py:50-58): the
// producer stores LeakyReLU(conv) plus per-group partial sums; in_finalize turns them into
// scale = gamma * rstd, shift = beta - mean * scale.  A plain tensor has scale == nullptr.
struct Src {
    const float* p;      // NCDHW (or NCHW when bcast_d)
    const float* scale;  // [N*C] or [N*C*D] (per_plane)
    const float* shift;
    int per_plane;       // statistics per (n, c, d) -- Matching's per-disparity InstanceNorm2d
    int bcast_d;         // tensor has no D axis and is broadcast along it (regularization.py:115)
    int id = -1;         // host-side only: index of the tensor on the backward tape
    int normed = 0;      // host-side only: scale / shift are (or, in a planning walk with null pointers, will be) set
    // Range certificate for the fp16-split kernels (conv2d_x3 P = 2, conv2d_t8): device floats whose maximum bounds
    // |value as the consumer reads it| -- the NORMALISED value of a deferred InstanceNorm (written by in_finalize from
    // gamma, beta and the group size: |gamma| sqrt(count) + |beta| is rigorous), the raw value of a plain tensor
    // (per-workgroup maxima written by the kernel that produced it).  nullptr: nothing is known about the range and
    // the consumer takes a range-safe form (three-way bf16 split / exact fp32).
    const float* bound = nullptr;
    int bound_n = 0;
    int bounded = 0;     // host-side only: a bound goes with the tensor (true in planning walks too, where bound is null)
    // Channel-blocked layout [N][D][C / 8][H][W][8] instead of NCDHW (round 5): private to the fused inference chain of
    // Matching, whose 64-channel tensors never leave the library; only the kernels of that chain accept it.
    int cb8 = 0;
};

inline Src plain_src(const float* p) { return Src{p, nullptr, nullptr, 0, 0}; }
inline Src no_src() { return Src{nullptr, nullptr, nullptr, 0, 0}; }

// Activation tensor geometry (contiguous NCDHW).
struct Geom {
    int n, c, d, h, w;
    __host__ __device__ size_t plane() const { return (size_t)h * w; }
    __host__ __device__ size_t volume() const { return (size_t)d * h * w; }
    __host__ __device__ size_t numel() const { return (size_t)n * c * d * h * w; }
};

// ---- weight packing jobs (pack.hip) -------------------------------------------------------------
struct PackJob {
    const float* src = nullptr;  // PyTorch-layout weights
    float* dst = nullptr;        // fragment-ordered weights
    unsigned* mask = nullptr;    // transposed convolutions: per-block tap masks
    int cout = 0, cin = 0, mblocks = 0, kc = 4, taps = 9;
    int mode = 0;                // 0 conv, 1 transposed k4 s2, 2 transposed k(3,4,4) s(1,2,2), 3 conv 3x3 -> F(2,3) along x, 4 conv 3x3 -> F(2x2,3x3) per-lane order,
                                 // 5 transposed k4 s2 in the dense cell form (8 taps)
                                 // 6 / 7 conv2d_x3 (bf16 x 3 / fp16 x 2 split), 8 / 9 conv3d_ks X form (fp16 x 2 split of modes 0 / 5)
    int total = 0;               // floats in dst
};
int launch_multi_pack(const PackJob* jobs, int count, hipStream_t s);

// How an MFMA launcher treats its weight packing:
//   kPackInline  pack (own launch) then run            -- stand-alone layer entry points
//   kPackCollect append the job(s) to `jobs`, launch NOTHING -- first walk of a module pipeline
//   kPackDone    weights already packed, just run      -- second walk
enum PackPhase { kPackInline = 0, kPackCollect = 1, kPackDone = 2 };
struct PackSink {
    PackPhase phase = kPackInline;
    PackJob* jobs = nullptr;
    int count = 0, capacity = 0;
    bool push(const PackJob& j) {
        if (count >= capacity) return false;
        jobs[count++] = j;
        return true;
    }
};

// ---- launchers (defined in the .hip files) -------------------------------------------------
struct ConvLayer {
    Src a, b;            // input = a (+ b)
    Geom in;             // geometry of a
    const float* weight; // PyTorch layout
    const float* bias;
    float* out;          // raw output: conv + bias (+ LeakyReLU)
    Geom out_g;
    int kd;              // 1 or 3 (kH = kW = 3)
    int stride;          // 1 or 2 (all convolved dims)
    int lrelu;
    int stat_per_plane;  // statistics grouping of THIS layer's InstanceNorm
    double* partials;    // nullptr: no statistics wanted
    float* packed;       // scratch for MFMA-ordered weights (MFMA kernels only)
    // conv2d_mfma only: layer-0 terms added on the fly (x0 = l0A + shift_d(l0G)) and an
    // optional copy of the staged input (the residual sum the NEXT block needs)
    // all three have row stride l0_rs and channel stride l0_cstride; l0A points at the column of x = 0 of its rows,
    // l0G / l0G2 hold G[u] at index u + 1
    const float* l0A = nullptr;
    const float* l0G = nullptr;
    const float* l0G2 = nullptr;
    size_t l0_cstride = 0;
    int l0_rs = 0;
    // conv2d Winograd kernels only: the output tensor has this many channels per batch entry (0: out_g.c) -- the launch
    // writes a 64-channel slice of a wider tensor (data gradients of layers with more than 64 input channels)
    int out_batch_channels = 0;
    int d_begin = 0;
    float* side_out = nullptr;
    // > 0: every d-plane has its own weight / bias set (weight + d * Cout*Cin*9, bias + d * Cout)
    int plane_weight_sets = 0;
    PackSink* sink = nullptr;  // nullptr: pack inline
    int out_cb8 = 0;           // conv2d_x3 only: write the output channel-blocked (Src::cb8)
    // conv2d_x3 only: the input is LeakyReLU(B + shift_d(H)) of the layer-1 factorisation, formed while it is staged from the
    // channel-blocked planes of misc.hip: launch_l1_blocked (X3Args::l1B); `a` then carries only the deferred InstanceNorm
    const float* l1B = nullptr;
    const float* l1H = nullptr;
    unsigned l1_bstride = 0, l1_hstride = 0, l1_edge = 0;
    int l1_P = 0, l1_d0 = 0;
};

// direct VALU convolution, any channel count
int launch_conv_direct(const ConvLayer& L, hipStream_t s);
int conv_direct_tiles_for(const Geom& out_g, int stride);  // tiles per output plane (partials sizing)

struct DeconvLayer {
    Src a, b;
    Geom in;
    const float* weight;  // [Cin, Cout, kD, 4, 4]
    const float* bias;
    float* out;
    Geom out_g;
    int kd;       // 4 (stride 2 in D) or 3 (stride 1 in D)
    int lrelu;
    double* partials;
    float* packed;  // scratch for the MFMA path (virtual weights + tap masks)
    PackSink* sink = nullptr;
};
int launch_deconv_direct(const DeconvLayer& L, hipStream_t s);
int deconv_direct_tiles(const Geom& out_g);

// ---- the persistent chain of K-split hourglass layers (conv3d_ks.hip) ------------------------------------------------
// A module walk hands consecutive conv3d_ks / deconv3d_ks layers to a chain instead of launching them; the chain runs
// them -- and the InstanceNorm fold behind each -- as ONE launch when the next other launch is due.
struct KsChainFold {          // what launch_in_finalize would be given behind a stand-alone launch of the layer
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
struct KsChain {
    int count = 0;
    struct alignas(8) Slot {
        unsigned char bytes[384];
    } storage[12];            // opaque phases (conv3d_ks.hip: KsPhase)
};
constexpr int kKsChainSyncWords = 64;     // unsigned words of device memory the chain kernel synchronises through ...
constexpr int kKsChainStateWords = 2048;  // ... at the head of this many words of state (counters + the phase table)
bool conv3d_ks_chain_enabled();
// `taken` = the layer went into the chain (otherwise: launch it the usual way, after flushing the chain)
int conv3d_ks_chain_add(KsChain& chain, const ConvLayer& L, const KsChainFold& fold, bool* taken);
int deconv3d_ks_chain_add(KsChain& chain, const DeconvLayer& L, const KsChainFold& fold, bool* taken);
int conv3d_ks_chain_launch(KsChain& chain, unsigned* sync_words, hipStream_t s);

// partial sums -> (scale, shift).  groups = N*C*(per_plane ? D : 1); each group reduces
// `per_group` consecutive partial records of (sum, sumsq); count = elements per group.
// bound (may be null): one float, max over channels of |gamma| sqrt(count) + |beta| -- a rigorous bound of the normalised values
int launch_in_finalize(const double* partials, int groups, int per_group, double count,
                       const float* gamma, const float* beta, int channels, int groups_per_channel_block,
                       float* scale, float* shift, float* mean, float* rstd, hipStream_t s, float* bound = nullptr);

// ---- backward (backward.hip) ------------------------------------------------------------------------
size_t in_bwd_scratch_doubles(const Geom& g);
int launch_in_bwd(const float* g, const float* t, const Geom& geom, int per_plane, const float* mean,
                  const float* rstd, const float* gamma, double* scratch, float* m1, float* m2, float* dz,
                  float* dgamma, float* dbeta, float* dbias, int accumulate_params, hipStream_t s,
                  float* dz_amax = nullptr);   // (kDzAmaxSlots floats whose maximum is max |dz|: the range certificate of dz)
constexpr int kDzAmaxSlots = 1024;
int channel_sum_splits(const Geom& g);
int launch_channel_sum(const float* dz, const Geom& g, float* db, int accumulate, double* scratch, hipStream_t s,
                       float* amax = nullptr);
int launch_flip_weights(const float* w, float* wf, int cout, int cin, int taps, hipStream_t s);
int launch_bwd_data(int transposed, int kd, int stride, const float* dz, const float* w, float* dx, const Geom& in,
                    const Geom& out, hipStream_t s);
size_t bwd_weight_scratch_doubles(int transposed, int kd, const Geom& in, const Geom& out);
int launch_bwd_weight(int transposed, int kd, int stride, const Src& a, const Src& b, const float* dz, float* dw,
                      const Geom& in, const Geom& out, int accumulate, double* scratch, hipStream_t s);
// MFMA weight gradient of the 2-D 3x3 convolutions (wgrad2d_mfma.hip)
// wgrad3d_s2_mfma.hip: weight gradient of the full-resolution k(3,4,4) s(1,2,2) transposed convolution (Cout = 1)
bool wgrad_up_full_mfma_supported(int transposed, int kd, const Src& b, const Geom& in, const Geom& out);
size_t wgrad_up_full_mfma_scratch_floats(const Geom& in);
int launch_wgrad_up_full_mfma(const Src& a, const float* dz, float* dw, const Geom& in, int accumulate, float* scratch,
                              hipStream_t s);
// wgrad3d_s2_mfma.hip: weight gradients of the stride-2 convolutions (kd 3) and the k4 s2 transposed convolutions
bool wgrad3d_s2_mfma_supported(int transposed, int kd, int stride, const Geom& in, const Geom& out);
size_t wgrad3d_s2_mfma_scratch_floats(int transposed, const Geom& in, const Geom& out);
int launch_wgrad3d_s2_mfma(int transposed, const Src& a, const Src& b, const float* dz, float* dw, const Geom& in,
                           const Geom& out, int accumulate, float* scratch, hipStream_t s);
bool wgrad3d_mfma_supported(int transposed, int kd, int stride, const Geom& in, const Geom& out);  // wgrad3d_mfma.hip
size_t wgrad3d_mfma_scratch_floats(const Geom& in, const Geom& out);
int launch_wgrad3d_mfma(const Src& a, const Src& b, const float* dz, float* dw, const Geom& in, const Geom& out,
                        int accumulate, float* scratch, hipStream_t s);
bool wgrad2d_mfma_supported(int transposed, int kd, int stride, const Src& b, const Geom& in, const Geom& out);
size_t wgrad2d_mfma_scratch_floats(const Geom& in, const Geom& out);
// dz: the gradient with its range certificate (Src::bound; none: plain_src).  With it and certificates on a (and b) the
// 64 -> 64 layers run the fp16-split kernel of wgrad2d_x3.hip
int launch_wgrad2d_mfma(const Src& a, const Src& b, const Src& dz, float* dw, const Geom& in, const Geom& out,
                        int accumulate, float* scratch, hipStream_t s);
bool wgrad2d_x3_supported(const Src& a, const Src& b, const Src& dz, const Geom& in, const Geom& out);   // wgrad2d_x3.hip
int launch_wgrad2d_x3(const Src& a, const Src& b, const Src& dz, float* partial, int workgroups, const Geom& in,
                      const Geom& out, hipStream_t s);
int launch_wgrad_reduce_f32(const float* partial, size_t wcount, int parts, float* dw, int accumulate, hipStream_t s);
int launch_grad_add(float* dst, const float* src, size_t count, int accumulate, hipStream_t s);
int launch_grad_reduce_d(float* dst, const float* src, const Geom& g, int accumulate, hipStream_t s);
int launch_shift_concat_bwd(const float* g, float* dleft, float* dright, int batch, int channels, int h, int w,
                            int d_begin, int d_count, hipStream_t s);

// out = a (+ b), both deferred-normalised; amax (may be null): materialize_records(g) floats, the maximum |out| of every workgroup
int launch_materialize(const Src& a, const Src& b, const Geom& g, float* out, hipStream_t s, float* amax = nullptr);
int materialize_records(const Geom& g);
// out = norm(a) + (A + shift_d(G)): the first residual sum of the fused Matching path; amax as above (materialize_l0_records)
int launch_materialize_l0(const Src& a, const Geom& g, const float* A, const float* G, const float* G2,
                          size_t l0_cstride, int l0_rs, int d_begin, float* out, hipStream_t s, float* amax = nullptr);
int materialize_l0_records(const Geom& g);

int launch_subpixel_map(const float* sim, float* disp, int batch, int planes, int height, int width,
                        int taps_lo, int taps_hi, int step, hipStream_t s);

int launch_shift_concat(const float* left, const float* right, float* out, int batch, int channels,
                        int h, int w, int d_begin, int d_count, hipStream_t s);

// Matching layer 0, factorised: x0[b,c,d,y,x] = A[b,c,y,x] + G[b,c,y,x-d] with the
// right-edge fix.  A = conv_L(L)+bias [B,C,h,w]; G, G2 [B,C,h,w+1] indexed by u+1, u = x-d.
// A, G, G2: row stride w + 1, channel stride `cstride`; A already points at column 1.
// amax (may be null): l0_combine_records floats, the largest |x0| of every workgroup (the range certificate of x0)
int launch_l0_combine(const float* A, const float* G, const float* G2, size_t cstride, float* x0, int batch,
                      int channels, int h, int w, int d_begin, int d_count, hipStream_t s, float* amax = nullptr);
int l0_combine_records(int batch, int channels, int d_count);
// adjoint of l0_combine and the glue of the layer-0 backward (misc.hip; pds_matching_bwd)
int launch_l0_combine_bwd(const float* g, float* gy_a, float* gy_gs, float* gy_g, float* gy_g2, int batch,
                          int channels, int h, int w, int d_begin, int d_count, hipStream_t s);
int launch_first_weight_grads(const float* dwl, const float* dws, const float* dwg, float* dw0, int cout, int cin_half,
                              hipStream_t s);
int launch_crop_left1_add(const float* a, const float* b, float* out, size_t rows, int w, hipStream_t s);
int launch_l0_stack_inputs(const float* left, const float* right, float* out, size_t bc_count, int h, int w,
                           int planes, int pad, hipStream_t s);

// layer-1 factorisation (misc.hip): planes B, H, Ha, Hb, H0
constexpr int kL1Planes = 5;
int launch_l1_stack_inputs(const float* y3, float* x4, int batch, int channels, int h, int w, hipStream_t s);
int launch_l1_weights(const float* w1, const float* b1, float* w4, float* bias4, int cout, int channels, hipStream_t s);
int l1_combine_tiles(int h, int w);
int launch_l1_combine(const float* y4, const float* corr, const float* corr0, float* t1, double* partials,
                      int batch, int channels, int h, int w, int d_begin, int d_count, hipStream_t s);
// column form (misc.hip): only A, G / B, H are full planes; G2, Ha, Hb, H0 are corrections at the columns that are read
int launch_column_weights(const float* wt, int cin_total, int cin_off, int channels, int cout, float* out,
                          hipStream_t s);
// G / G2: channel stride cstride, row stride rs, G[u] at column u + co; A (same strides): first zero_cols columns zeroed
int launch_l0_column_fix(const float* G, const float* R, const float* wcol, float* G2, float* A, int zero_cols,
                         size_t cstride, int rs, int co, int batch, int channels, int cout, int h, int w, int d_begin,
                         int d_count, hipStream_t s);
int launch_l1_column_terms(const float* G, const float* G2, const float* wcol, float* corr, float* corr0,
                           size_t cstride, int rs, int co, int batch, int channels, int cout, int h, int w, int d_begin,
                           int d_count, hipStream_t s);
int launch_l1_weights2(const float* w1, const float* b1, float* w2, float* bias2, int cout, int channels,
                       hipStream_t s);
// layer-1 planes B / H (y4 [n][C][2][h][w + 2]) + column corrections -> the channel-blocked form conv2d_x3 stages its first
// launch from (ConvLayer::l1B): sizes in floats per (batch entry, channel group); pad = zero columns left of H
__host__ __device__ size_t l1_blocked_b_floats(int h, int w);
__host__ __device__ size_t l1_blocked_h_floats(int h, int w, int pad, int d_count);
__host__ __device__ size_t l1_blocked_edge_offset_floats(int h, int w, int pad);
int launch_l1_blocked(const float* y4, const float* corr, const float* corr0, float* Bc, float* Hx, int batch, int channels,
                      int h, int w, int pad, int d_begin, int d_count, hipStream_t s);

// small utility: zero-pad one column on the left ([.., w] -> [.., w+1]); split conv0 weights
int launch_pad_left1(const float* in, float* out, size_t rows, int w, hipStream_t s);
int launch_split_first_weights(const float* w0, const float* b0, float* wl, float* wr, float* wr2, float* bias3,
                               int cout, int cin_half, hipStream_t s);

// ---- descriptor network helpers (embedding.hip) --------------------------------------------------------
int image_stats_chunks(int h, int w);
int launch_image_stats(const float* img, int nc, int h, int w, double* partials, hipStream_t s);
// [N, C, H, W] (virtually zero-padded by top rows / left columns) -> [N, 4C, ceil((H+top)/2), ceil((W+left)/2)]
// bound_out (may be null): one float, the bound of `a` carried over (a re-layout changes no value; pad zeros are inside any bound)
int launch_space_to_depth(const Src& a, int n, int c, int h, int w, int top, int left, float* out, hipStream_t s,
                          float* bound_out = nullptr);
int launch_depth_to_space(const float* g, int n, int c, int h, int w, float* out, hipStream_t s);
// d image from the gradient of the space-to-depth tensor: depth-to-space + InstanceNorm2d (no affine) backward over the padded plane
int launch_image_grad(const float* g, const float* img, const float* scale, const float* shift, int n, int c, int h,
                      int w, int top, int left, float* out, hipStream_t s);
int launch_s2d_weights(const float* w5, float* w3, int cout, int cin, hipStream_t s);
int launch_s2d_weights_bwd(const float* g3, float* g5, int cout, int cin, int accumulate, hipStream_t s);

// ---- evaluation metrics (errors.hip) ----------------------------------------------------------------
size_t disparity_errors_partial_doubles(size_t total);
int launch_disparity_errors(const float* est, const float* gt, size_t total, float n, float* abs_out, float* bad_out,
                            double* stats, double* partials, hipStream_t s);

// ---- loss (loss.hip) --------------------------------------------------------------------------------
size_t sce_partial_doubles(size_t total_px);
int launch_sce_fwd(const float* sim, const float* gt, const float* weights, float* loss, float* lse, float* stats,
                   double* partials, int n, int planes, int h, int w, float diversity, int step, hipStream_t s);
int launch_sce_bwd(const float* sim, const float* gt, const float* weights, const float* lse, const float* stats,
                   const float* grad_loss, float* gsim, int n, int planes, int h, int w, float diversity, int step,
                   hipStream_t s);
int launch_sce_weights_bwd(const float* sim, const float* gt, const float* lse, const float* stats,
                           const float* grad_loss, float* gweights, int n, int planes, int h, int w, float diversity,
                           int step, hipStream_t s);

// ---- power-of-two operand scales of the fp16-split kernels -----------------------------------------
// Largest power of two s with s * bound <= target (exact to apply and to undo); 1 when the bound is zero, negative
// or not finite (the result is then as meaningless as the reference's own for such an input, but never a trap).
__host__ __device__ inline float pow2_scale(float bound, float target) {
    if (!(bound > 0.f) || !(bound < 3.0e38f)) return 1.f;
    float f = target / bound;
    f = f < 0x1p-60f ? 0x1p-60f : (f > 0x1p60f ? 0x1p60f : f);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, f) & 0x7f800000u);
#else
    union { float f; unsigned u; } v;
    v.f = f;
    v.u &= 0x7f800000u;
    return v.f;
#endif
}
constexpr float kHalfTarget = 16384.f;   // scaled operands stay below 2^14 (fp16 overflows at 65 504)

// ---- two-way fp16 split of fp32 values (the operands of the fp16-split MFMA kernels) ---------------------------------
// hi = fp16(r), lo = fp16(r - hi), both round-to-nearest-even; a PAIR at a time with the packed conversion of gfx950
// (conv2d_x3.hip, round 4): 3 instructions per pair instead of the 4 conversions + 2 subtractions + 2 packs of the scalar
// form, bit-identical results.  The packed words hold element 0 in their low half.
__device__ __forceinline__ void split_pair_f16(float r0, float r1, unsigned& hi, unsigned& lo) {
    // (the low parts straight from the mixed-precision fma: lo = f16(r - hi), hi read as the fp16 half it is; r - hi is exact in
    // fp32, so the only rounding is the conversion -- 3 instructions per pair)
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(r0), "v"(r1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(r0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(r1));
}
typedef unsigned pds_u32x2 __attribute__((ext_vector_type(2)));
// four values -> the 8 bytes of hi parts and the 8 bytes of lo parts
__device__ __forceinline__ void split_quad_f16(const float (&v)[4], pds_u32x2& hi, pds_u32x2& lo) {
    unsigned h0, l0, h1, l1;
    split_pair_f16(v[0], v[1], h0, l0);
    split_pair_f16(v[2], v[3], h1, l1);
    hi = pds_u32x2{h0, h1};
    lo = pds_u32x2{l0, l1};
}

// ---- wave helpers ------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {   // every lane receives the maximum
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// Maximum of one value per thread, for every thread of the workgroup (call it from ALL threads, before any divergence;
// red = 16 floats of LDS; a NaN counts as +inf).
__device__ __forceinline__ float block_max(float v, float* red) {
    float mm = wave_max(v == v ? v : __builtin_inff());
    const int wave = threadIdx.x >> 6, waves = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = mm;
    __syncthreads();
    float r = 0.f;
    for (int k = 0; k < waves; ++k) r = fmaxf(r, red[k]);
    __syncthreads();
    return r;
}
// Maximum of the n bound records of a source (Src::bound), same calling convention.
__device__ __forceinline__ float block_bound(const float* __restrict__ b, int n, float* red) {
    float mm = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = fabsf(b[i]);
        mm = fmaxf(mm, v == v ? v : __builtin_inff());   // (fmaxf alone would drop a NaN record)
    }
    return block_max(mm, red);
}
// one record per workgroup of a producer: the maximum |v| its threads have seen (NaN / inf -> +inf)
__device__ __forceinline__ void block_amax_record(float m, float* __restrict__ rec, float* red) {
    float mm = (m == m) ? m : __builtin_inff();
    mm = wave_max(mm);
    const int wave = threadIdx.x >> 6, waves = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = mm;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = 0.f;
        for (int k = 0; k < waves; ++k) r = fmaxf(r, red[k]);
        *rec = r;
    }
}

// ---- InstanceNorm fold: partial records -> (scale, shift) of ONE group, by ONE wave (every lane calls it) --------------
// The single definition behind in_finalize_kernel (conv_direct.hip) and the in-launch fold of the persistent chain kernel
// (conv3d_ks.hip): lane l sums the records l, l + 64, ... in order, one shuffle reduction, lane 0 finishes -- the same
// order wherever it runs, so both paths give the same bits.  Biased variance, eps 1e-5 (torch.nn.InstanceNorm defaults,
// network_blocks.py:58,72,85).  g: group index; c = (g / inner) % channels its channel.
// B groups at a time (g, g + gstride, ...; the first `valid` of them exist): the B record streams are independent, so a
// lane has B (x the unroll factor) loads in flight instead of one group's -- what matters where ONE workgroup folds a
// whole layer (the chain kernel's last arriver).  Per group the order of the additions is the same for every B.
template <int B>
__device__ __forceinline__ void in_finalize_groups(const double* __restrict__ partials, int g, int gstride, int valid,
                                                   int per_group, double count, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, int channels, int inner,
                                                   float* __restrict__ scale, float* __restrict__ shift,
                                                   float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                   unsigned* __restrict__ nonfinite, int lane) {
    const double2* p[B];
    double s[B], q[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        // (a missing group re-reads the first one: no branch in the load loop; its result is dropped)
        p[b] = reinterpret_cast<const double2*>(partials) + (size_t)(b < valid ? g + b * gstride : g) * per_group;
        s[b] = q[b] = 0.0;
    }
#pragma unroll 4
    for (int i = lane; i < per_group; i += 64) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const double2 r = p[b][i];
            s[b] += r.x;
            q[b] += r.y;
        }
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const double sum = wave_sum(s[b]), sq = wave_sum(q[b]);
        if (lane == 0 && b < valid) {
            const int gb = g + b * gstride;
            const double mean = sum / count;
            double var = sq / count - mean * mean;
            // a NaN / inf reached this layer (or it overflowed): counted in host-mapped memory, pds_nonfinite_statistics()
            if (nonfinite && !(fabs(mean) < 1.7e308 && fabs(var) < 1.7e308)) atomicAdd_system(nonfinite, 1u);
            if (var < 0.0) var = 0.0;
            const double rstd = 1.0 / sqrt(var + kInEps);
            const int c = (gb / inner) % channels;
            const double sc = (gamma ? (double)gamma[c] : 1.0) * rstd;  // no affine: embedding.py:32
            scale[gb] = (float)sc;
            shift[gb] = (float)((beta ? (double)beta[c] : 0.0) - mean * sc);
            if (mean_out) {  // kept for the backward pass
                mean_out[gb] = (float)mean;
                rstd_out[gb] = (float)rstd;
            }
        }
    }
}
__device__ __forceinline__ void in_finalize_group(const double* __restrict__ partials, int g, int per_group, double count,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  int channels, int inner, float* __restrict__ scale,
                                                  float* __restrict__ shift, float* __restrict__ mean_out,
                                                  float* __restrict__ rstd_out, unsigned* __restrict__ nonfinite, int lane) {
    in_finalize_groups<1>(partials, g, 0, 1, per_group, count, gamma, beta, channels, inner, scale, shift, mean_out, rstd_out,
                          nonfinite, lane);
}
// Range certificate of the normalised tensor (Src::bound): a group of `count` values with unit (biased) variance has no
// z-score beyond sqrt(count - 1), so |gamma| sqrt(count) + |beta| bounds every value.  One wave, every lane calls it.
__device__ __forceinline__ void in_finalize_bound(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  int channels, double count, float* __restrict__ bound_out, int lane) {
    float m = 0.f;
    const float root = sqrtf((float)count);
    for (int c = lane; c < channels; c += 64) {
        const float v = fabsf(gamma ? gamma[c] : 1.f) * root + fabsf(beta ? beta[c] : 0.f);
        m = fmaxf(m, v == v ? v : __builtin_inff());
    }
    m = wave_max(m);
    if (lane == 0) *bound_out = m;
}

}  // namespace pds
