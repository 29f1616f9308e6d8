// Internals shared by the api*.hip translation units (the C ABI of libpds_hip.so, include/pds_hip.h): the workspace
// arena, the backward tape, deferred-normalisation tensors and the two layer builders every module walk is made of.
//   api.hip                 error reporting, launch probe, conv_block / deconv_block, the small entry points
//   api_matching.hip        Matching / MatchingOperation walks and entry points (matching.py)
//   api_regularization.hip  Regularization / ContractionBlock3d / ExpansionBlock3d (regularization.py)
//   api_embedding.hip       Embedding (embedding.py)
//   api_training.hip        the reverse walk over a tape (every pds_*_bwd entry point ends in it)
#pragma once
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace pds {

int launch_conv2d_mfma(const ConvLayer& L, hipStream_t s);        // conv2d_mfma.hip
bool conv2d_mfma_supported(const ConvLayer& L);
int conv2d_mfma_tiles(const Geom& out_g);
size_t conv2d_mfma_packed_floats(int cin, int cout);
int launch_conv2d_wino(const ConvLayer& L, hipStream_t s);        // conv2d_wino.hip
bool conv2d_wino_eligible(const ConvLayer& L);
int conv2d_wino_tiles(const Geom& out_g);
size_t conv2d_wino_packed_floats(int cin, int cout);
int launch_conv2d_x3(const ConvLayer& L, hipStream_t s);          // conv2d_x3.hip (Cin -> 64, fp32 on the bf16 pipe)
bool conv2d_x3_supported(const ConvLayer& L);
bool conv2d_x3_cb8_ok(const ConvLayer& L, bool in_cb8, bool out_cb8);   // channel-blocked input / output (Src::cb8)
int conv2d_x3_tiles(const ConvLayer& L);   // statistics records per plane (depends on the form chosen)
size_t conv2d_x3_packed_floats(int cin);
int launch_conv2d_t8(const ConvLayer& L, hipStream_t s);          // conv2d_t8.hip (64 -> 8 channels, bare)
bool conv2d_t8_supported(const ConvLayer& L);
int launch_conv3d_mfma(const ConvLayer& L, hipStream_t s);        // conv3d_mfma.hip
bool conv3d_mfma_supported(const ConvLayer& L);
int conv3d_mfma_tiles(const Geom& out_g, int cin, int stride);
size_t conv3d_mfma_packed_floats(const Geom& out_g, int cin, int stride);
int launch_conv3d_nx(const ConvLayer& L, hipStream_t s);          // conv3d_nx.hip (16 -> 16 channels, quarter resolution, fp16 split)
bool conv3d_nx_supported(const ConvLayer& L);
int conv3d_nx_tiles(const Geom& out_g);
size_t conv3d_nx_packed_floats(int cin, int cout);
int launch_conv3d_t8(const ConvLayer& L, hipStream_t s);          // conv3d_t8.hip (8 -> 8 channels, stride 1)
bool conv3d_t8_supported(const ConvLayer& L);
int conv3d_t8_records(const Geom& out_g);
int launch_deconv3d_mfma(const DeconvLayer& L, hipStream_t s);
bool deconv3d_mfma_supported(const DeconvLayer& L);
int deconv3d_mfma_tiles(const Geom& in_g);
size_t deconv3d_mfma_packed_floats(const Geom& in_g, int cout, int kd);
int launch_conv3d_ks(const ConvLayer& L, hipStream_t s);          // conv3d_ks.hip (inner hourglass levels, K split over waves)
bool conv3d_ks_supported(const ConvLayer& L);
int conv3d_ks_tiles(const Geom& out_g);
size_t conv3d_ks_packed_floats(int cin, int vchannels, int taps);
int launch_deconv3d_ks(const DeconvLayer& L, hipStream_t s);
int ks_chain_debug_stamps(unsigned* out, int capacity);           // (measurement aid: pds_debug_chain_stamps)
bool deconv3d_ks_supported(const DeconvLayer& L);
int deconv3d_ks_tiles(const Geom& in_g, int cout);
int launch_deconv3d_cell(const DeconvLayer& L, hipStream_t s);    // deconv3d_cell.hip (dense cell form, k4 s2)
bool deconv3d_cell_supported(const DeconvLayer& L);
int deconv3d_cell_records(const Geom& in_g, int cout);
bool upsample_estimator_supported(int cin, int lo, int hi);       // upsample_estimator.hip
int launch_upsample_estimator(const float* in, const float* scale, const float* shift, const float* w_pairs,
                              const float* bias, float* disp, int batch, int cin, int d, int hi_, int wi, int lo,
                              int hi, int step, int crop_top, int crop_left, hipStream_t s);
int launch_upsample_weight_pairs(const float* w, float* w_pairs, int cin, hipStream_t s);   // kw order 1, 2, 3, 0


// ---- backward tape ---------------------------------------------------------------------------------
// Recorded while a pipeline is (re-)walked over the forward workspace; the arena is deterministic, so the
// backward entry points rebuild the tape from the preserved workspace instead of keeping library state.
struct TapeTensor {
    const float* raw = nullptr;   // stored values (raw layer output, or a plain tensor)
    const float* scale = nullptr; // folded InstanceNorm (nullptr: plain)
    const float* shift = nullptr;
    const float* mean = nullptr;
    const float* rstd = nullptr;
    Geom g{0, 0, 0, 0, 0};
    int per_plane = 0;
    int bcast_d = 0;              // [N, C, H, W] tensor broadcast along D (g.d is the broadcast extent)
    bool needs_grad = true;       // false: nothing upstream wants a gradient (the image)
    const float* bound = nullptr; // range certificate of the forward pass (Src::bound), still in the forward workspace
    int bound_n = 0;
    bool bounded = false;
    Src src() const {
        Src s{raw, scale, shift, per_plane, bcast_d};
        s.bound = bound;
        s.bound_n = bound_n;
        s.bounded = bounded ? 1 : 0;
        return s;
    }
};
struct TapeLayer {
    int type = 0;                 // 0 conv, 1 transposed conv, 2 sum (out = a^ + b^, plain),
                                  // 3 space-to-depth (out = s2d(a^), plain; embedding.hip)
    int kd = 3, stride = 1;
    int a = -1, b = -1, out = -1; // tensor ids
    Geom in_g{0, 0, 0, 0, 0}, out_g{0, 0, 0, 0, 0};
    const PdsConvBlockParams* P = nullptr;  // address inside the caller's parameter struct
    bool norm = false;
    // k5 s2 convolution run as k3 s1 over space-to-depth input: the 3x3 weights actually used, and the
    // channel count of the 5x5 kernel they were derived from (0: ordinary layer)
    const float* weight_used = nullptr;
    int s2d_cin = 0;
};
struct Tape {
    std::vector<TapeTensor> tensors;
    std::vector<TapeLayer> layers;
    int add(const TapeTensor& t) {
        tensors.push_back(t);
        return (int)tensors.size() - 1;
    }
};

// ---- workspace arena: plan mode only measures ---------------------------------------------------
struct Ctx {
    char* base;
    size_t off = 0;
    bool plan;          // true: measure only (null pointers) or collect pack jobs (real pointers): NO launches
    hipStream_t s;
    int err = 0;
    PackSink* sink = nullptr;
    Tape* tape = nullptr;   // non-null: record layers for the backward pass (and keep every layer tape-friendly)
    size_t limit = ~(size_t)0;  // bytes behind `base`: carving past it is an error, never a wild write
    // Regularization only (round 6): consecutive K-split layers are collected here and run as ONE persistent launch
    // (conv3d_ks.hip: conv3d_ks_chain_kernel) when the next other launch is due -- flush_chain()
    KsChain* chain = nullptr;
    unsigned* chain_sync = nullptr;
    void flush_chain() {
        if (chain && chain->count > 0) run(conv3d_ks_chain_launch(*chain, chain_sync, s));
    }

    template <class T>
    T* get(size_t count) {
        const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += bytes;
        static const bool debug_arena = getenv("PDS_DEBUG_ARENA") != nullptr && atoi(getenv("PDS_DEBUG_ARENA")) > 1;
        if (debug_arena) fprintf(stderr, "[pds]   get %zu\n", bytes);
        if (base && off > limit) {
            if (!err) err = set_error(-1, "workspace arena overflow (%zu > %zu bytes)", off, limit);
            plan = true;  // nothing more is launched
        }
        return p;
    }
    void run(int rc) {
        if (!err && rc) err = rc;
    }
    // Kernels whose output feeds the batched weight packing must be enqueued in the collect walk (before
    // the pack launch), or right away when packing is inline.
    bool before_packing() const {
        if (sink) return base != nullptr && sink->phase == kPackCollect;
        return !plan;
    }
};

// Runs a module pipeline in two walks over the same (deterministic) arena: the first only collects the
// weight-packing jobs of every MFMA layer, which are then executed by ONE launch; the second enqueues the
// layers with their weights already packed.
// weights_resident: the caller vouches that this workspace still holds the packed weights (and the weight-derived
// tensors) a previous call of the same entry point with the same shapes and parameter values left there: the first walk
// and the packing launch are skipped.
template <class Pipeline>
static int run_with_batched_packing(void* workspace, hipStream_t stream, Pipeline&& pipeline,
                                    bool weights_resident = false) {
    PackJob table[64];
    PackSink sink;
    sink.jobs = table;
    sink.capacity = 64;
    if (!weights_resident) {
        sink.phase = kPackCollect;
        Ctx collect{(char*)workspace, 0, true, stream};
        collect.sink = &sink;
        pipeline(collect);
        if (collect.err) return collect.err;
        if (sink.count > 0)
            if (int rc = launch_multi_pack(table, sink.count, stream)) return rc;
    }
    sink.phase = kPackDone;
    Ctx run{(char*)workspace, 0, false, stream};
    run.sink = &sink;
    pipeline(run);
    return run.err;
}

// A tensor whose InstanceNorm is deferred to its consumers.
struct DT {
    float* raw = nullptr;
    float* scale = nullptr;
    float* shift = nullptr;
    float* mean = nullptr;
    float* rstd = nullptr;
    Geom g{0, 0, 0, 0, 0};
    int per_plane = 0;
    int id = -1;
    bool normed = false;   // a deferred InstanceNorm goes with the tensor (true in planning walks too, where scale is null)
    // range certificate (common.hpp Src::bound): written by in_finalize for a normalised tensor, by the producing
    // kernel (per-workgroup maxima) for a plain one
    float* bound = nullptr;
    int bound_n = 0;
    bool bounded = false;
    bool cb8 = false;      // stored channel-blocked ([N][D][C / 8][H][W][8], common.hpp Src::cb8)
    Src src() const {
        Src s{raw, scale, shift, per_plane, 0};
        s.cb8 = cb8 ? 1 : 0;
        s.id = id;
        s.normed = normed ? 1 : 0;
        s.bound = bound;
        s.bound_n = bound_n;
        s.bounded = bounded ? 1 : 0;
        return s;
    }
};


// (api.hip)
void carve_amax(Ctx& c, DT& t, int records);
Src external_src(Ctx& c, const float* p, const Geom& g, int bcast_d = 0, bool needs_grad = true);
void tape_layer(Ctx& c, int type, int kd, int stride, const Src& a, const Src& b, const Geom& in_g, DT& o,
                const PdsConvBlockParams* P, bool norm);
Geom conv_out_geom(const Geom& in, int cout, int kd, int stride);

// Extras of the fused Matching path (conv2d_mfma only): layer-0 terms formed in the loader, side output.
struct ConvExtra {
    const float* l0A = nullptr;
    const float* l0G = nullptr;
    const float* l0G2 = nullptr;
    size_t l0_cstride = 0;
    int l0_rs = 0;
    int out_batch_channels = 0;   // Winograd kernels only: write a channel slice of a wider tensor
    int d_begin = 0;
    float* side_out = nullptr;
    int plane_weight_sets = 0;
    // a k5 s2 layer evaluated as k3 s1 over space-to-depth input (any kernel): weights to use instead of P.weight
    const float* weight_used = nullptr;
    int s2d_cin = 0;
    bool out_cb8 = false;   // conv2d_x3 only: write the output channel-blocked (the consumer must accept Src::cb8)
    // conv2d_x3 only: input formed on the fly from the blocked layer-1 planes (ConvLayer::l1B)
    const float* l1B = nullptr;
    const float* l1H = nullptr;
    unsigned l1_bstride = 0, l1_hstride = 0, l1_edge = 0;
    int l1_P = 0, l1_d0 = 0;
    bool matching_extras() const { return l0A || side_out || plane_weight_sets > 0; }
};


// conv (+ LeakyReLU + deferred InstanceNorm when P.gamma) ; out_raw may be caller-provided   (api.hip)
DT conv_block(Ctx& c, const Src& a, const Src& b, const Geom& in, const PdsConvBlockParams& P, int cout, int kd, int stride,
              int per_plane, float* out_raw = nullptr, bool allow_mfma = true, float* scale_out = nullptr,
              float* shift_out = nullptr, const ConvExtra* extra = nullptr);
DT deconv_block(Ctx& c, const Src& a, const Src& b, const Geom& in, const PdsConvBlockParams& P, int cout, int kd,
                float* out_raw = nullptr);
int check_block(const PdsConvBlockParams& b, bool norm, const char* name);

// maps the address of a layer's parameters inside the caller's struct to the same slot of the gradient struct
struct GradMap {
    const char* params_base;
    const char* grads_base;
    size_t struct_bytes;
    const PdsConvBlockParams* blocks_params = nullptr;  // out-of-struct array (PdsMatchingParams::blocks)
    const PdsConvBlockParams* blocks_grads = nullptr;
    int blocks_count = 0;
    const PdsConvBlockParams* find(const PdsConvBlockParams* p) const {
        if (blocks_params && p >= blocks_params && p < blocks_params + blocks_count) return blocks_grads + (p - blocks_params);
        const char* q = reinterpret_cast<const char*>(p);
        if (q >= params_base && q < params_base + struct_bytes)
            return reinterpret_cast<const PdsConvBlockParams*>(grads_base + (q - params_base));
        return nullptr;
    }
};

// bytes behind the backward arena of the entry point being served (set by the pds_*_bwd functions): a planning

// bytes behind the backward arena of the entry point being served (set by the pds_*_bwd functions): a planning
// walk that under-estimates must surface as an error, not as a write past the caller's buffer
extern thread_local size_t g_backward_arena_bytes;   // (api_training.hip)
struct ArenaLimit {
    explicit ArenaLimit(size_t bytes) { g_backward_arena_bytes = bytes; }
    ~ArenaLimit() { g_backward_arena_bytes = ~(size_t)0; }
};

// dhat[i]: gradient with respect to the NORMALISED value of tensor i.  Entries preset by the caller (the
// gradient of the output, the gradient buffers of the external inputs) are used as they are; the others
// are carved from the backward arena on first use.   (api_training.hip)
void backward_walk(Ctx& c, const Tape& T, const GradMap& M, std::vector<float*>& dhat, std::vector<char>& written);

}  // namespace pds
