// C ABI of libpds_hip.so (include/pds_hip.h): argument checks, workspace carving and the
// per-module launch sequences.  No allocation, no synchronisation: everything is enqueued on the
// caller's stream into caller-owned memory.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace pds {

static thread_local char g_error[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error((int)e, "%s: %s", what, hipGetErrorString(e));
    // PDS_DEBUG_SYNC=1 (debugging only): wait for every launch and name it, so that a faulting kernel is the
    // last line printed.  Off by default: the library never synchronises.
    static const bool debug_sync = getenv("PDS_DEBUG_SYNC") != nullptr;
    if (debug_sync) {
        fprintf(stderr, "[pds] %s ...", what);
        fflush(stderr);
        const hipError_t es = hipDeviceSynchronize();
        fprintf(stderr, " %s\n", es == hipSuccess ? "ok" : hipGetErrorString(es));
        if (es != hipSuccess) return set_error((int)es, "%s: %s", what, hipGetErrorString(es));
    }
    return 0;
}

// ---- launch probe (include/pds_hip.h, ABI v5): in-situ kernel timing for bench.py --------------------------------
namespace {
constexpr int kProbeMax = 256;
std::atomic<int> g_probe_armed{0};
char g_probe_name[64] = "";
int g_probe_capacity = 0, g_probe_count = 0, g_probe_events = 0;
hipEvent_t g_probe_start[kProbeMax], g_probe_stop[kProbeMax];
int g_probe_wgs[kProbeMax];
}  // namespace

int probe_before(const char* name, hipStream_t s) {
    if (!g_probe_armed.load(std::memory_order_relaxed)) return -1;
    if (!strstr(name, g_probe_name) || g_probe_count >= g_probe_capacity) return -1;
    const int slot = g_probe_count++;
    (void)hipEventRecord(g_probe_start[slot], s);
    return slot;
}
void probe_after(int slot, int workgroups, hipStream_t s) {
    if (slot < 0) return;
    g_probe_wgs[slot] = workgroups;
    (void)hipEventRecord(g_probe_stop[slot], s);
}

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

// a plain tensor whose producer writes `records` per-workgroup maxima of |value|
static void carve_amax(Ctx& c, DT& t, int records) {
    t.bound = c.get<float>((size_t)records);
    t.bound_n = records;
    t.bounded = true;
}

// a caller-provided plain tensor as a source, registered on the tape when one is being recorded
static Src external_src(Ctx& c, const float* p, const Geom& g, int bcast_d = 0, bool needs_grad = true) {
    Src s{p, nullptr, nullptr, 0, bcast_d};
    if (c.tape) {
        TapeTensor t;
        t.raw = p;
        t.g = g;
        t.bcast_d = bcast_d;
        t.needs_grad = needs_grad;
        s.id = c.tape->add(t);
    }
    return s;
}

static void tape_layer(Ctx& c, int type, int kd, int stride, const Src& a, const Src& b, const Geom& in_g, DT& o,
                       const PdsConvBlockParams* P, bool norm) {
    if (!c.tape) return;
    TapeTensor t;
    t.raw = o.raw;
    t.scale = o.scale;
    t.shift = o.shift;
    t.mean = o.mean;
    t.rstd = o.rstd;
    t.g = o.g;
    t.per_plane = o.per_plane;
    t.bound = o.bound;
    t.bound_n = o.bound_n;
    t.bounded = o.bounded;
    o.id = c.tape->add(t);
    TapeLayer L;
    L.type = type;
    L.kd = kd;
    L.stride = stride;
    L.a = a.id;
    L.b = b.p ? b.id : -1;
    L.out = o.id;
    L.in_g = in_g;
    L.out_g = o.g;
    L.P = P;
    L.norm = norm;
    c.tape->layers.push_back(L);
}

static Geom conv_out_geom(const Geom& in, int cout, int kd, int stride) {
    Geom o = in;
    o.c = cout;
    if (stride == 2) {
        if (kd == 3) o.d = (in.d + 1) / 2;
        o.h = (in.h + 1) / 2;
        o.w = (in.w + 1) / 2;
    }
    return o;
}

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

// conv (+ LeakyReLU + deferred InstanceNorm when P.gamma) ; out_raw may be caller-provided
static DT conv_block(Ctx& c, const Src& a, const Src& b, const Geom& in, const PdsConvBlockParams& P, int cout,
                     int kd, int stride, int per_plane, float* out_raw = nullptr, bool allow_mfma = true,
                     float* scale_out = nullptr, float* shift_out = nullptr, const ConvExtra* extra = nullptr) {
    DT o;
    o.g = conv_out_geom(in, cout, kd, stride);
    o.per_plane = per_plane;
    o.raw = out_raw ? out_raw : c.get<float>(o.g.numel());
    const bool norm = P.gamma != nullptr;
    ConvLayer L;
    L.a = a;
    L.b = b;
    L.in = in;
    L.weight = (extra && extra->s2d_cin) ? extra->weight_used : P.weight;
    L.bias = P.bias;
    L.out = o.raw;
    L.out_g = o.g;
    L.kd = kd;
    L.stride = stride;
    L.lrelu = norm ? 1 : 0;
    L.stat_per_plane = per_plane;
    L.partials = nullptr;
    L.packed = nullptr;
    L.sink = c.sink;
    if (extra) {
        L.l0A = extra->l0A;
        L.l0G = extra->l0G;
        L.l0G2 = extra->l0G2;
        L.l0_cstride = extra->l0_cstride;
        L.l0_rs = extra->l0_rs;
        L.out_batch_channels = extra->out_batch_channels;
        L.d_begin = extra->d_begin;
        L.side_out = extra->side_out;
        L.plane_weight_sets = extra->plane_weight_sets;
        L.out_cb8 = extra->out_cb8 ? 1 : 0;
        L.l1B = extra->l1B;
        L.l1H = extra->l1H;
        L.l1_bstride = extra->l1_bstride;
        L.l1_hstride = extra->l1_hstride;
        L.l1_edge = extra->l1_edge;
        L.l1_P = extra->l1_P;
        L.l1_d0 = extra->l1_d0;
    }
    o.cb8 = L.out_cb8 != 0;
    // kernel choice: 0 = direct VALU, 2 = conv2d MFMA (kd 1), 3 = conv3d MFMA (kd 3), 4 = conv2d MFMA in the
    // Winograd domain (plain single-source Cin -> 64 layers), 9 = conv2d on the bf16 pipe with three-way split operands
    int kind = 0;
    if (allow_mfma && !norm && !c.tape && conv2d_t8_supported(L)) kind = 8;   // persistent y-Toeplitz kernel, no packing
    else if (allow_mfma && !(extra && extra->matching_extras()) && conv2d_x3_supported(L)) kind = 9;
    else if (allow_mfma && conv2d_mfma_supported(L)) kind = conv2d_wino_eligible(L) ? 4 : 2;
    else if (allow_mfma && conv3d_t8_supported(L)) kind = 6;   // persistent z-Toeplitz kernel, no weight packing
    else if (allow_mfma && conv3d_ks_supported(L)) kind = 7;   // inner levels: K split over the waves
    else if (allow_mfma && conv3d_mfma_supported(L)) kind = 3;
    if (kind == 2)
        L.packed = c.get<float>(conv2d_mfma_packed_floats(in.c, cout) * (L.plane_weight_sets > 0 ? L.plane_weight_sets : 1));
    if (kind == 4)
        L.packed = c.get<float>(conv2d_wino_packed_floats(in.c, cout) * (L.plane_weight_sets > 0 ? L.plane_weight_sets : 1));
    if (kind == 3) L.packed = c.get<float>(conv3d_mfma_packed_floats(o.g, in.c, stride));
    if (kind == 7) L.packed = c.get<float>(conv3d_ks_packed_floats(in.c, cout, 27));
    if (kind == 9) L.packed = c.get<float>(conv2d_x3_packed_floats(in.c));
    // (conv2d_t8 / conv2d_t8w read planar tensors only: nothing but conv2d_x3 takes or writes the blocked layout)
    if ((a.cb8 || b.cb8 || L.out_cb8) && !(kind == 9 && !b.p && conv2d_x3_cb8_ok(L, a.cb8 != 0, L.out_cb8 != 0))) {
        c.run(set_error(-1, "conv_block: a channel-blocked tensor reached a kernel that does not take it"));
        return o;
    }
    if (L.l1B && kind != 9) {
        c.run(set_error(-1, "conv_block: the on-the-fly layer-1 source needs conv2d_x3"));
        return o;
    }
    if (L.out_batch_channels && kind != 4 && kind != 9) {
        c.run(set_error(-1, "conv_block: a channel-slice output needs the Winograd kernel"));
        return o;
    }
    if (extra && extra->matching_extras() && kind != 2 && kind != 4 && kind != 8) {
        c.run(set_error(-1, "conv_block: fused Matching extras need the conv2d MFMA kernel"));
        return o;
    }
    auto launch = [&]() {
        return kind == 2 ? launch_conv2d_mfma(L, c.s)
                         : kind == 4 ? launch_conv2d_wino(L, c.s)
                                     : kind == 3 ? launch_conv3d_mfma(L, c.s)
                                                 : kind == 6 ? launch_conv3d_t8(L, c.s)
                                                             : kind == 7 ? launch_conv3d_ks(L, c.s)
                                                                         : kind == 8 ? launch_conv2d_t8(L, c.s)
                                                                                     : kind == 9 ? launch_conv2d_x3(L, c.s) : launch_conv_direct(L, c.s);
    };
    const bool collecting = c.sink && c.sink->phase == kPackCollect && c.base != nullptr;
    if (collecting && kind != 0 && kind != 6 && kind != 8) c.run(launch());  // registers the pack job(s) only
    if (norm) {
        // partial records: direct / 2-D kernels write [(n, c, d)][tile]; the 3-D kernel writes [(n, c)][tile]
        const int tiles = kind == 2 ? conv2d_mfma_tiles(o.g)
                                    : kind == 4 ? conv2d_wino_tiles(o.g)
                                    : kind == 9 ? conv2d_x3_tiles(L)
                                    : kind == 3 ? conv3d_mfma_tiles(o.g, in.c, stride)
                                    : kind == 6 ? conv3d_t8_records(o.g)
                                    : kind == 7 ? conv3d_ks_tiles(o.g) : conv_direct_tiles_for(o.g, stride);
        const bool volume_records = kind == 3 || kind == 6 || kind == 7;   // [(n, c)][record] instead of [(n, c, d)][tile]
        const size_t records = (size_t)o.g.n * o.g.c * (volume_records ? 1 : o.g.d) * tiles;
        L.partials = c.get<double>(records * 2);
        const int groups = o.g.n * o.g.c * (per_plane ? o.g.d : 1);
        o.normed = true;
        o.scale = scale_out ? scale_out : c.get<float>(groups);
        o.shift = shift_out ? shift_out : c.get<float>(groups);
        o.mean = c.get<float>(groups);
        o.rstd = c.get<float>(groups);
        o.bound = c.get<float>(1);
        o.bound_n = 1;
        o.bounded = true;
        if (!c.plan) {
            const int per_group = volume_records ? tiles : tiles * (per_plane ? 1 : o.g.d);
            const double count = (double)o.g.h * o.g.w * (per_plane ? 1 : o.g.d);
            bool chained = false;
            if (c.chain && kind == 7 && !per_plane) {
                const KsChainFold fold{P.gamma, P.beta, o.scale, o.shift, o.mean, o.rstd, o.bound, groups, per_group, o.g.c, count};
                c.run(conv3d_ks_chain_add(*c.chain, L, fold, &chained));
            }
            if (!chained) {
                c.flush_chain();
                c.run(launch());
                c.run(launch_in_finalize(L.partials, groups, per_group, count, P.gamma, P.beta, o.g.c,
                                         per_plane ? o.g.d : 1, o.scale, o.shift, o.mean, o.rstd, c.s, o.bound));
            }
        }
    } else if (!c.plan) {
        c.flush_chain();
        c.run(launch());
    }
    tape_layer(c, 0, kd, stride, a, b, in, o, &P, norm);
    if (c.tape && extra && extra->s2d_cin) {  // (the pointer is null in a planning walk: test the count)
        c.tape->layers.back().weight_used = extra->weight_used;
        c.tape->layers.back().s2d_cin = extra->s2d_cin;
    }
    return o;
}

static DT deconv_block(Ctx& c, const Src& a, const Src& b, const Geom& in, const PdsConvBlockParams& P, int cout,
                       int kd, float* out_raw = nullptr) {
    DT o;
    o.g = in;
    o.g.c = cout;
    o.g.d = (kd == 4) ? in.d * 2 : in.d;
    o.g.h = in.h * 2;
    o.g.w = in.w * 2;
    o.per_plane = 0;
    o.raw = out_raw ? out_raw : c.get<float>(o.g.numel());
    const bool norm = P.gamma != nullptr;
    DeconvLayer L;
    L.a = a;
    L.b = b;
    L.in = in;
    L.weight = P.weight;
    L.bias = P.bias;
    L.out = o.raw;
    L.out_g = o.g;
    L.kd = kd;
    L.lrelu = norm ? 1 : 0;
    L.partials = nullptr;
    L.packed = nullptr;
    L.sink = c.sink;
    const bool cell = deconv3d_cell_supported(L);   // persistent dense-cell kernel: no weight packing
    const bool ks = !cell && deconv3d_ks_supported(L);   // inner levels: cell form, K split over the waves
    const bool mfma = !cell && !ks && deconv3d_mfma_supported(L);
    if (mfma) L.packed = c.get<float>(deconv3d_mfma_packed_floats(in, cout, kd));
    if (ks) L.packed = c.get<float>(conv3d_ks_packed_floats(in.c, 8 * cout, 8));
    if (mfma && c.sink && c.sink->phase == kPackCollect && c.base != nullptr) c.run(launch_deconv3d_mfma(L, c.s));
    if (ks && c.sink && c.sink->phase == kPackCollect && c.base != nullptr) c.run(launch_deconv3d_ks(L, c.s));
    auto launch = [&]() {
        return cell ? launch_deconv3d_cell(L, c.s)
                    : ks ? launch_deconv3d_ks(L, c.s) : mfma ? launch_deconv3d_mfma(L, c.s) : launch_deconv_direct(L, c.s);
    };
    // partial records per (n, c): cell kernel [workgroup]; K-split / MFMA kernels [tile][parity class]; direct [d][tile]
    const int per_group = cell ? deconv3d_cell_records(in, cout)
                               : ks ? deconv3d_ks_tiles(in, cout) * 8
                                    : mfma ? deconv3d_mfma_tiles(in) * (kd == 4 ? 8 : 4) : deconv_direct_tiles(o.g) * o.g.d;
    if (norm) {
        const size_t records = (size_t)o.g.n * o.g.c * per_group;
        L.partials = c.get<double>(records * 2);
        const int groups = o.g.n * o.g.c;
        o.normed = true;
        o.scale = c.get<float>(groups);
        o.shift = c.get<float>(groups);
        o.mean = c.get<float>(groups);
        o.rstd = c.get<float>(groups);
        o.bound = c.get<float>(1);
        o.bound_n = 1;
        o.bounded = true;
        if (!c.plan) {
            bool chained = false;
            if (c.chain && ks) {
                const KsChainFold fold{P.gamma, P.beta, o.scale, o.shift, o.mean, o.rstd, o.bound, groups, per_group, o.g.c,
                                       (double)o.g.volume()};
                c.run(deconv3d_ks_chain_add(*c.chain, L, fold, &chained));
            }
            if (!chained) {
                c.flush_chain();
                c.run(launch());
                c.run(launch_in_finalize(L.partials, groups, per_group, (double)o.g.volume(), P.gamma, P.beta,
                                         o.g.c, 1, o.scale, o.shift, o.mean, o.rstd, c.s, o.bound));
            }
        }
    } else if (!c.plan) {
        c.flush_chain();
        c.run(launch());
    }
    tape_layer(c, 1, kd, kd == 4 ? 2 : 1, a, b, in, o, &P, norm);
    return o;
}

// ---- MatchingOperation after layer 0: residual blocks + last conv ---------------------------------
// x0 plain [n, F, d, h, w]; kernel depth 1, InstanceNorm statistics per (n, c, d) plane.
static void operation_tail(Ctx& c, const PdsMatchingParams& P, const Src& x0, const Geom& g, float* signature) {
    const int F = P.features;
    Src cur = x0;
    DT t2;
    // A residual sum norm(t2) + x is a plain tensor; the kernel that forms it records its largest magnitude (the range
    // certificate the fp16-split kernels conv2d_x3 / conv2d_t8 scale by).  x0 itself -- a convolution of the caller's
    // tensor, of unknown scale -- carries none: its consumer takes the range-safe form.
    for (int r = 0; r < P.residual_blocks; ++r) {
        DT t1 = conv_block(c, cur, no_src(), g, P.blocks[2 * r], F, 1, 1, 1);
        t2 = conv_block(c, t1.src(), no_src(), g, P.blocks[2 * r + 1], F, 1, 1, 1);
        if (r + 1 < P.residual_blocks) {
            DT nxt;  // plain residual sum  x_{r+1} = norm(t2) + x_r
            nxt.raw = c.get<float>(g.numel());
            nxt.g = g;
            carve_amax(c, nxt, materialize_records(g));
            if (!c.plan) c.run(launch_materialize(t2.src(), cur, g, nxt.raw, c.s, nxt.bound));
            tape_layer(c, 2, 0, 0, t2.src(), cur, g, nxt, nullptr, false);
            cur = nxt.src();
        }
    }
    if (P.residual_blocks > 0)
        conv_block(c, t2.src(), cur, g, P.last, P.signature_features, 1, 1, 1, signature);
    else
        conv_block(c, cur, no_src(), g, P.last, P.signature_features, 1, 1, 1, signature);
}

// Can the fused Matching path (layer-0 terms in the loader, residual sums as side outputs) be used?
static bool fused_matching_supported(const PdsMatchingParams& P, int batch, int h, int w, int d_count) {
    ConvLayer L{};
    L.a = plain_src(nullptr);
    L.b = no_src();
    L.in = Geom{batch, P.features, d_count, h, w};
    L.out_g = L.in;
    L.kd = 1;
    L.stride = 1;
    ConvLayer T = L;
    T.out_g.c = P.signature_features;
    return conv2d_mfma_supported(L) && conv2d_mfma_supported(T);
}

// what the layer-0 backward (pds_matching_bwd) needs from a training-route walk
struct MatchingL0 {
    const float* w3 = nullptr;   // [3 sets][F][F][3][3]: left half, right half, right half without dx = +1
};

// train: the differentiable route.  Layer 0 keeps its factorisation (the right descriptor is convolved once, no
// [D', B, 128, h, w] concat exists), x0 = A + shift_d(G) is materialised as the first tape tensor and the rest of
// MatchingOperation runs layer by layer, every output kept for the backward pass.
static void matching_pipeline(Ctx& c, const PdsMatchingParams& P, const float* left, const float* right,
                              float* signatures, int batch, int h, int w, int d_begin, int d_count, bool train = false,
                              MatchingL0* l0_out = nullptr) {
    const int F = P.features;
    const size_t wn = (size_t)F * F * 9;
    float* w3 = c.get<float>(3 * wn);       // [3 sets][F][F][3][3]: left half, right half, right half without dx=+1
    float* bias3 = c.get<float>(3 * F);
    if (l0_out) l0_out->w3 = w3;
    const Geom g{batch, F, d_count, h, w};
    const bool fused = [&]() {
        static const bool enabled = []() {  // PDS_MATCHING_FUSED=0 selects the unfused sequence (A/B, debugging)
            const char* e = debug_switch("PDS_MATCHING_FUSED");
            return !(e && e[0] == '0');
        }();
        return enabled && fused_matching_supported(P, batch, h, w, d_count);
    }();
    // Column form (misc.hip): G2 / Ha / Hb / H0 only at the columns that are read, as corrections to G / H; the
    // convolutions run over two 64-channel planes instead of three (layer 0) and five 128-channel ones (layer 1).
    // PDS_MATCHING_COLUMNS=0 keeps the whole-plane form.
    const bool columns = [&]() {
        static const bool enabled = []() {
            const char* e = debug_switch("PDS_MATCHING_COLUMNS");
            return !(e && e[0] == '0');
        }();
        return enabled && fused && !train && P.residual_blocks >= 1 && F % 8 == 0;
    }();
    const int l0_planes = columns ? 2 : 3;
    // plane 0: left, planes 1(-2): right, each behind one zero column -- two in the column form: the width is even
    // (Winograd kernel) and the output is, after zeroing two columns of A, the input of the layer-1 launch
    const int l0_pad = columns ? 2 : 1;
    const int l0_rs = w + l0_pad;
    const Geom g3{batch, F, l0_planes, h, l0_rs};
    float* x3 = c.get<float>(g3.numel());
    float* wcol0 = columns ? c.get<float>(9 * (size_t)F * F) : nullptr;   // [dx][ic][dy][oc] of the right half of conv0
    float* wcol1 = columns ? c.get<float>(9 * (size_t)F * F) : nullptr;   // ... of the first conv of block 1
    if (c.before_packing()) {
        c.run(launch_split_first_weights(P.first.weight, P.first.bias, w3, w3 + wn, w3 + 2 * wn, bias3, F, F, c.s));
        if (columns) {
            c.run(launch_column_weights(P.first.weight, 2 * F, F, F, F, wcol0, c.s));
            c.run(launch_column_weights(P.blocks[0].weight, F, 0, F, F, wcol1, c.s));
        }
    }
    if (!c.plan) c.run(launch_l0_stack_inputs(left, right, x3, (size_t)batch * F, h, w, l0_planes, l0_pad, c.s));
    // A = conv_L(left) + bias, G = conv_R(right), G2 = G without its dx = +1 taps: the planes of y3
    float* y3;
    if (fused) {
        // one launch, per-plane weight sets
        PdsConvBlockParams p3{w3, bias3, nullptr, nullptr};
        ConvExtra e3;
        e3.plane_weight_sets = l0_planes;
        // (not a tape layer: the training route differentiates layer 0 through its factorisation, matching_backward)
        Tape* tape = c.tape;
        c.tape = nullptr;
        y3 = conv_block(c, plain_src(x3), no_src(), g3, p3, F, 1, 1, 1, nullptr, true, nullptr, nullptr, &e3).raw;
        c.tape = tape;
    } else {
        // generic kernels share one weight set per launch: three launches into the planes of y3
        y3 = c.get<float>(g3.numel());
        float* tmp_in = c.get<float>((size_t)batch * F * h * (w + 1));
        float* tmp_out = c.get<float>((size_t)batch * F * h * (w + 1));
        const Geom g1{batch, F, 1, h, w + 1};
        Tape* tape = c.tape;
        c.tape = nullptr;   // (as above)
        for (int p = 0; p < 3; ++p) {
            if (!c.plan) c.run(launch_pad_left1(p == 0 ? left : right, tmp_in, (size_t)batch * F * h, w, c.s));
            PdsConvBlockParams pp{w3 + p * wn, bias3 + p * F, nullptr, nullptr};
            conv_block(c, plain_src(tmp_in), no_src(), g1, pp, F, 1, 1, 1, tmp_out);
            // scatter [B*F][h][w+1] into plane p of y3
            if (!c.plan)
                c.run((int)hipMemcpy2DAsync(y3 + (size_t)p * h * (w + 1), (size_t)3 * h * (w + 1) * sizeof(float), tmp_out,
                                        (size_t)h * (w + 1) * sizeof(float), (size_t)h * (w + 1) * sizeof(float),
                                        (size_t)batch * F, hipMemcpyDeviceToDevice, c.s));
        }
        c.tape = tape;
    }
    const size_t l0_cstride = (size_t)l0_planes * h * l0_rs;
    const float* l0A = y3 + l0_pad;                                      // column of x = 0 in plane 0
    const float* l0G = y3 + (size_t)h * l0_rs + (l0_pad - 1);            // plane 1; index u + 1 holds G[u]
    const float* l0G2 = y3 + (size_t)2 * h * l0_rs + (l0_pad - 1);       // plane 2
    float* g2buf = nullptr;
    if (columns) {
        // G2 in a buffer of its own with the channel stride of y3 (its consumers take ONE stride for A, G, G2); only
        // the columns u = w - 1 - d of the planes of this call are ever written or read
        g2buf = c.get<float>(g3.numel());
        if (!c.plan)
            c.run(launch_l0_column_fix(y3 + (size_t)h * l0_rs, right, wcol0, g2buf + (size_t)h * l0_rs, y3, l0_pad,
                                       l0_cstride, l0_rs, l0_pad, batch, F, F, h, w, d_begin, d_count, c.s));
        l0G2 = g2buf + (size_t)h * l0_rs + (l0_pad - 1);
    }
    if (!fused || train) {
        DT x0;   // plain; the kernel that forms it records its largest magnitudes (the range certificate, Src::bound)
        x0.g = g;
        x0.raw = c.get<float>(g.numel());
        carve_amax(c, x0, l0_combine_records(batch, F, d_count));
        if (!c.plan)
            c.run(launch_l0_combine(l0A, l0G, l0G2, l0_cstride, x0.raw, batch, F, h, w, d_begin, d_count, c.s, x0.bound));
        Src x0s = x0.src();
        if (c.tape) {   // tape tensor 0 of the training route: its gradient is what the layer-0 backward starts from
            TapeTensor t;
            t.raw = x0.raw;
            t.g = g;
            t.bound = x0.bound;
            t.bound_n = x0.bound_n;
            t.bounded = true;
            x0s.id = c.tape->add(t);
        }
        operation_tail(c, P, x0s, g, signatures);
        return;
    }
    // Fused: x0 = A + shift_d(G) is never stored.  The first conv forms it inside its loader; the first
    // residual sum x1 = norm(t2) + x0 is produced by one streaming kernel that re-forms x0 from the
    // cache-resident A / G (one 425 MB stream in, one out, instead of two in).
    ConvExtra l0;
    l0.l0A = l0A;
    l0.l0G = l0G;
    l0.l0G2 = l0G2;
    l0.l0_cstride = l0_cstride;
    l0.l0_rs = l0_rs;
    l0.d_begin = d_begin;
    const Src none = no_src();
    if (P.residual_blocks == 0) {
        conv_block(c, none, none, g, P.last, P.signature_features, 1, 1, 1, signatures, true, nullptr, nullptr, &l0);
        return;
    }
    // Channel-blocked activations between the 64-channel layers (round 5; conv2d_x3.hip: X3Args::in_cb8): level 1 = the
    // tensor between the two convolutions of a residual block (produced and consumed by conv2d_x3 alone); level 2 (opt-in,
    // PDS_MATCHING_CB8=2): in addition the first 64 -> 64 launch forms its input t1 = LeakyReLU(B + shift_d(H)) while it stages it, from the
    // channel-blocked layer-1 planes (misc.hip: l1_blocked_kernel) -- l1_combine only computes t1's statistics, the 425 MB
    // round trip of t1 through HBM is gone.  Bit-identical; measured NEUTRAL (l1_combine 109 -> 63 us without its stores, + 20 us
    // for the re-layout, + 11 us on the launch; 410 vs 408.5 pairs/s in a same-box A/B), so level 1 stays the default
    const int cb8_level = [&]() {
        static const int level = []() {   // PDS_MATCHING_CB8=0: planar NCDHW everywhere (A/B, tests)
            const char* e = debug_switch("PDS_MATCHING_CB8");
            return e ? atoi(e) : 1;
        }();
        if (!(F == 64 && h % 16 == 0 && w % 16 == 0)) return 0;
        ConvLayer probe;   // would conv2d_x3 serve these layers in its fp16 form (PDS_X3 / PDS_X3_FP16 may say no)?
        probe.a = plain_src(nullptr);
        probe.a.bounded = 1;
        probe.b = no_src();
        probe.in = g;
        probe.out_g = g;
        probe.kd = 1;
        probe.stride = 1;
        return conv2d_x3_cb8_ok(probe, true, true) ? level : 0;
    }();
    // Layer 1 factorised like layer 0 (misc.hip): B = conv1(A) + b1, H / Ha / Hb / H0 = conv1 of the G rows, as one
    // 5-plane launch with tap-masked weight sets; then LeakyReLU(B + shift_d(H)) + statistics in one streaming pass.
    DT t1;
    ConvExtra fly;
    bool on_the_fly = false;
    {
        DT y4;
        const float* corr = nullptr;
        const float* corr0 = nullptr;
        if (columns) {
            float* w2 = c.get<float>(2 * wn);
            float* bias2 = c.get<float>(2 * F);
            const Geom g2{batch, F, 2, h, w + 2};   // == g3: the layer-0 output is the input (A's two left columns zeroed)
            float* cr = c.get<float>((size_t)batch * F * h * d_count * 2);
            float* cr0 = c.get<float>((size_t)batch * F * h);
            if (c.before_packing())
                c.run(launch_l1_weights2(P.blocks[0].weight, P.blocks[0].bias, w2, bias2, F, F, c.s));
            if (!c.plan)
                c.run(launch_l1_column_terms(y3 + (size_t)h * l0_rs, g2buf + (size_t)h * l0_rs, wcol1, cr, cr0,
                                             l0_cstride, l0_rs, l0_pad, batch, F, F, h, w, d_begin, d_count, c.s));
            float* x2 = y3;
            PdsConvBlockParams p2{w2, bias2, nullptr, nullptr};
            ConvExtra e2;
            e2.plane_weight_sets = 2;
            y4 = conv_block(c, plain_src(x2), none, g2, p2, F, 1, 1, 1, nullptr, true, nullptr, nullptr, &e2);
            corr = cr;
            corr0 = cr0;
        } else {
            const size_t wn4 = (size_t)kL1Planes * F * 2 * F * 9;
            float* w4 = c.get<float>(wn4);
            float* bias4 = c.get<float>(kL1Planes * F);
            const Geom g4{batch, 2 * F, kL1Planes, h, w + 2};
            float* x4 = c.get<float>(g4.numel());
            if (c.before_packing())
                c.run(launch_l1_weights(P.blocks[0].weight, P.blocks[0].bias, w4, bias4, F, F, c.s));
            if (!c.plan) c.run(launch_l1_stack_inputs(y3, x4, batch, F, h, w, c.s));
            PdsConvBlockParams p4{w4, bias4, nullptr, nullptr};
            ConvExtra e4;
            e4.plane_weight_sets = kL1Planes;
            y4 = conv_block(c, plain_src(x4), none, g4, p4, F, 1, 1, 1, nullptr, true, nullptr, nullptr, &e4);
        }
        t1.g = g;
        t1.per_plane = 1;
        t1.raw = c.get<float>(g.numel());
        const int tiles = l1_combine_tiles(h, w);
        double* partials = c.get<double>((size_t)batch * F * d_count * tiles * 2);
        const int groups = batch * F * d_count;
        t1.normed = true;
        t1.scale = c.get<float>(groups);
        t1.shift = c.get<float>(groups);
        t1.mean = c.get<float>(groups);
        t1.rstd = c.get<float>(groups);
        t1.bound = c.get<float>(1);
        t1.bound_n = 1;
        t1.bounded = true;
        if (columns && cb8_level >= 2) {
            const int pad = d_begin + d_count;   // zero columns left of H: x - d + 2 + pad >= 0 for every plane of this call
            fly.l1_bstride = (unsigned)(l1_blocked_b_floats(h, w) * sizeof(float));
            fly.l1_hstride = (unsigned)(l1_blocked_h_floats(h, w, pad, d_count) * sizeof(float));
            fly.l1_edge = (unsigned)(l1_blocked_edge_offset_floats(h, w, pad) * sizeof(float));
            fly.l1_P = pad;
            fly.l1_d0 = d_begin;
            float* Bc = c.get<float>((size_t)batch * (F / 8) * l1_blocked_b_floats(h, w));
            float* Hx = c.get<float>((size_t)batch * (F / 8) * l1_blocked_h_floats(h, w, pad, d_count));
            fly.l1B = Bc;
            fly.l1H = Hx;
            on_the_fly = true;
            if (!c.plan)
                c.run(launch_l1_blocked(y4.raw, corr, corr0, Bc, Hx, batch, F, h, w, pad, d_begin, d_count, c.s));
        }
        if (!c.plan) {
            // (on the fly: statistics only -- t1 itself is never stored; its buffer is still the home of the first residual sum)
            c.run(launch_l1_combine(y4.raw, corr, corr0, on_the_fly ? nullptr : t1.raw, partials, batch, F, h, w, d_begin,
                                    d_count, c.s));
            c.run(launch_in_finalize(partials, groups, tiles, (double)h * w, P.blocks[0].gamma, P.blocks[0].beta, F,
                                     d_count, t1.scale, t1.shift, t1.mean, t1.rstd, c.s, t1.bound));
        }
    }
    DT t2 = on_the_fly ? conv_block(c, t1.src(), none, g, P.blocks[1], F, 1, 1, 1, nullptr, true, nullptr, nullptr, &fly)
                       : conv_block(c, t1.src(), none, g, P.blocks[1], F, 1, 1, 1);
    if (P.residual_blocks == 1) {
        conv_block(c, t2.src(), none, g, P.last, P.signature_features, 1, 1, 1, signatures, true, nullptr, nullptr,
                   &l0);
        return;
    }
    // This walk is inference-only (no tape), so the [B, 64, D', h, w] activations rotate through THREE buffers (the
    // most that are live at once: a block's input, its first and its second layer) instead of one per layer: t1's
    // buffer is dead once t2 exists, t2's once the residual sum is formed.
    // x_r = norm(t2) + x_{r-1} is a plain tensor: the kernel that forms it records its largest magnitudes, the range
    // certificate of the fp16-split kernels behind it (conv2d_x3, conv2d_t8)
    DT cur;
    cur.g = g;
    cur.raw = t1.raw;   // x1 = norm(t2) + x0 overwrites t1
    carve_amax(c, cur, materialize_l0_records(g));
    if (!c.plan)
        c.run(launch_materialize_l0(t2.src(), g, l0A, l0G, l0G2, l0_cstride, l0_rs, d_begin, cur.raw, c.s, cur.bound));
    float* spare_a = t2.raw;                        // free from here on
    float* spare_b = c.get<float>(g.numel());
    for (int r = 1; r < P.residual_blocks; ++r) {
        ConvExtra blocked;
        blocked.out_cb8 = cb8_level >= 1;
        t1 = conv_block(c, cur.src(), none, g, P.blocks[2 * r], F, 1, 1, 1, spare_a, true, nullptr, nullptr, &blocked);
        t2 = conv_block(c, t1.src(), none, g, P.blocks[2 * r + 1], F, 1, 1, 1, spare_b);
        if (r + 1 < P.residual_blocks) {
            DT nxt;                                 // t1 is dead: x_{r+1} = norm(t2) + x_r goes there
            nxt.g = g;
            nxt.raw = spare_a;
            carve_amax(c, nxt, materialize_records(g));
            if (!c.plan) c.run(launch_materialize(t2.src(), cur.src(), g, nxt.raw, c.s, nxt.bound));
            spare_a = cur.raw;
            cur = nxt;
        }
    }
    conv_block(c, t2.src(), cur.src(), g, P.last, P.signature_features, 1, 1, 1, signatures);
}

static void operation_pipeline(Ctx& c, const PdsMatchingParams& P, const float* concatenated, float* signature,
                               int n, int h, int w) {
    const Geom gin{n, 2 * P.features, 1, h, w};
    DT x0 = conv_block(c, external_src(c, concatenated, gin), no_src(), gin, P.first, P.features, 1, 1, 1);
    operation_tail(c, P, x0.src(), x0.g, signature);
}

// ---- Regularization (reference regularization.py:94-126) -----------------------------------------
static DT regularization_trunk(Ctx& c, const PdsRegularizationParams& P, const float* ms, const float* left,
                               int batch, int d, int h, int w) {
    const int F = P.features;
    const Geom g0{batch, F, d, h, w};
    // tape ids: 0 = signatures, 1 = left shortcut ([batch, F, h, w] broadcast along D, regularization.py:115)
    const Src ms_src = external_src(c, ms, g0);
    Src shortcut = external_src(c, left, g0, 1);
    // the K-split layers of the inner levels go through a chain: one persistent launch per run of consecutive ones
    KsChain chain;
    c.chain_sync = c.get<unsigned>(kKsChainStateWords);
    c.chain = (!c.plan && conv3d_ks_chain_enabled()) ? &chain : nullptr;
    DT out = conv_block(c, ms_src, no_src(), g0, P.smoothing, F, 3, 1, 0);
    DT pushed[4];
    for (int i = 0; i < 4; ++i) {
        pushed[i] = out;
        const int cin = out.g.c;
        // contraction_block(shortcut + output): a = output, b = shortcut (b may broadcast along D)
        DT down = conv_block(c, out.src(), shortcut, out.g, P.contraction[i][0], 2 * cin, 3, 2, 0);
        DT smooth = conv_block(c, down.src(), no_src(), down.g, P.contraction[i][1], 2 * cin, 3, 1, 0);
        shortcut = down.src();
        out = smooth;
    }
    for (int i = 0; i < 4; ++i) {
        const int cin = out.g.c;
        DT up = deconv_block(c, out.src(), no_src(), out.g, P.expansion[i][0], cin / 2, 4);
        out = conv_block(c, up.src(), pushed[3 - i].src(), up.g, P.expansion[i][1], cin / 2, 3, 1, 0);
    }
    DT half = deconv_block(c, out.src(), no_src(), out.g, P.upsample_half, F / 2, 4);
    c.flush_chain();
    c.chain = nullptr;
    return half;
}

bool upsample_full_valu_supported(int cin);
int launch_upsample_full(const float* in, const float* scale, const float* shift, const float* w, const float* bias,
                         float* cost, int batch, int cin, int d, int hi_, int wi, hipStream_t s);

static void regularization_pipeline(Ctx& c, const PdsRegularizationParams& P, const float* ms, const float* left,
                                    float* cost, int batch, int d, int h, int w) {
    DT half = regularization_trunk(c, P, ms, left, batch, d, h, w);
    if (upsample_full_valu_supported(half.g.c)) {
        // 4 -> 1 channels: the plane-sweeping VALU kernel beats the MFMA path (which wastes 12 of 16 rows)
        float* w_pairs = c.get<float>((size_t)half.g.c * 48);   // weight-derived: written in the packing walk only
        if (c.before_packing()) c.run(launch_upsample_weight_pairs(P.upsample_full.weight, w_pairs, half.g.c, c.s));
        if (!c.plan)
            c.run(launch_upsample_full(half.raw, half.scale, half.shift, w_pairs, P.upsample_full.bias,
                                       cost, batch, half.g.c, half.g.d, half.g.h, half.g.w, c.s));
        DT full;
        full.raw = cost;
        full.g = Geom{batch, 1, half.g.d, 2 * half.g.h, 2 * half.g.w};
        tape_layer(c, 1, 3, 1, half.src(), no_src(), half.g, full, &P.upsample_full, false);
        return;
    }
    deconv_block(c, half.src(), no_src(), half.g, P.upsample_full, 1, 3, cost);
}


// ---- stand-alone ContractionBlock3d / ExpansionBlock3d (reference regularization.py:28-31, 54-57) ----------
// pp[0] / pp[1]: the two conv blocks of the module.  Outputs are plain (normalised) tensors: tape ops of type 2.
static void contraction_pipeline(Ctx& c, const PdsConvBlockParams* pp, const float* x, float* down_out,
                                 float* smooth_out, const Geom& g, int* id_down_out = nullptr,
                                 int* id_smooth_out = nullptr) {
    const Src xs = external_src(c, x, g);
    DT down = conv_block(c, xs, no_src(), g, pp[0], 2 * g.c, 3, 2, 0);
    DT smooth = conv_block(c, down.src(), no_src(), down.g, pp[1], 2 * g.c, 3, 1, 0);
    if (!c.plan) {
        c.run(launch_materialize(down.src(), no_src(), down.g, down_out, c.s));
        c.run(launch_materialize(smooth.src(), no_src(), smooth.g, smooth_out, c.s));
    }
    DT od, os;
    od.raw = down_out;
    od.g = down.g;
    os.raw = smooth_out;
    os.g = smooth.g;
    tape_layer(c, 2, 0, 0, down.src(), no_src(), down.g, od, nullptr, false);
    tape_layer(c, 2, 0, 0, smooth.src(), no_src(), smooth.g, os, nullptr, false);
    if (id_down_out) *id_down_out = od.id;
    if (id_smooth_out) *id_smooth_out = os.id;
}

static void expansion_pipeline(Ctx& c, const PdsConvBlockParams* pp, const float* x, const float* shortcut,
                               float* out, const Geom& g) {
    const Src xs = external_src(c, x, g);                                                    // tape id 0
    const Geom gs{g.n, g.c / 2, 2 * g.d, 2 * g.h, 2 * g.w};
    const Src ss = external_src(c, shortcut, gs);                                            // tape id 1
    DT up = deconv_block(c, xs, no_src(), g, pp[0], g.c / 2, 4);
    DT sm = conv_block(c, up.src(), ss, up.g, pp[1], g.c / 2, 3, 1, 0);
    if (!c.plan) c.run(launch_materialize(sm.src(), no_src(), sm.g, out, c.s));
    DT o;
    o.raw = out;
    o.g = sm.g;
    tape_layer(c, 2, 0, 0, sm.src(), no_src(), sm.g, o, nullptr, false);                     // last tensor
}

// ---- Embedding (reference embedding.py:46-65) over a virtually padded image (size_adapter.py:29-43) --------
// image [batch, C0, h, w]; descriptor [batch, F, H4, W4]; shortcut [batch, S, H4, W4] with
// H2 = ceil((h + top) / 2), H4 = ceil(H2 / 2) (same for the width).
// the image head as the backward pass needs it: folded InstanceNorm coefficients of the image and the tape id of the
// space-to-depth tensor (which receives a gradient only when the caller wants d loss / d image)
struct ImageHead {
    bool want_grad = false;
    const float* scale = nullptr;
    const float* shift = nullptr;
    int id = -1;
};

static void embedding_pipeline(Ctx& c, const PdsEmbeddingParams& P, const float* image, float* descriptor,
                               float* shortcut, int batch, int h, int w, int top, int left, int* id_descriptor = nullptr,
                               int* id_shortcut = nullptr, ImageHead* head = nullptr) {
    const int C0 = P.input_features, F = P.features;
    // parameter-free InstanceNorm2d of the padded image (embedding.py:32), folded into the re-layout below
    const int chunks = image_stats_chunks(h, w);
    double* partials = c.get<double>((size_t)batch * C0 * chunks * 2);
    float* scale0 = c.get<float>(batch * C0);
    float* shift0 = c.get<float>(batch * C0);
    const Geom g1{batch, 4 * C0, 1, (h + top + 1) / 2, (w + left + 1) / 2};
    float* s0 = c.get<float>(g1.numel());
    if (!c.plan) {
        c.run(launch_image_stats(image, batch * C0, h, w, partials, c.s));
        c.run(launch_in_finalize(partials, batch * C0, chunks, (double)(h + top) * (w + left), nullptr, nullptr, C0, 1,
                                 scale0, shift0, nullptr, nullptr, c.s));
        c.run(launch_space_to_depth(Src{image, scale0, shift0, 0, 0}, batch, C0, h, w, top, left, s0, c.s));
    }
    const Src s0_src = external_src(c, s0, g1, 0, head && head->want_grad);  // tape id 0: a gradient only for d image
    if (head) {
        head->scale = scale0;
        head->shift = shift0;
        head->id = s0_src.id;
    }
    // convolutional_block_5x5_stride_2 twice (embedding.py:33-36), each as k3 s1 over space-to-depth input
    float* w1 = c.get<float>((size_t)F * 4 * C0 * 9);
    if (c.before_packing()) c.run(launch_s2d_weights(P.downsampling[0].weight, w1, F, C0, c.s));
    ConvExtra e1;
    e1.weight_used = w1;
    e1.s2d_cin = C0;
    DT t1 = conv_block(c, s0_src, no_src(), g1, P.downsampling[0], F, 1, 1, 1, nullptr, true, nullptr, nullptr, &e1);
    DT s1;
    s1.g = Geom{batch, 4 * F, 1, (t1.g.h + 1) / 2, (t1.g.w + 1) / 2};
    s1.raw = c.get<float>(s1.g.numel());
    // the re-layout of a normalised tensor is a plain tensor with the same range certificate (conv2d_x3: fp16 form)
    s1.bound = c.get<float>(1);
    s1.bound_n = 1;
    s1.bounded = true;
    if (!c.plan) c.run(launch_space_to_depth(t1.src(), batch, F, t1.g.h, t1.g.w, 0, 0, s1.raw, c.s, s1.bound));
    tape_layer(c, 3, 0, 0, t1.src(), no_src(), t1.g, s1, nullptr, false);
    float* w2 = c.get<float>((size_t)F * 4 * F * 9);
    if (c.before_packing()) c.run(launch_s2d_weights(P.downsampling[1].weight, w2, F, F, c.s));
    ConvExtra e2;
    e2.weight_used = w2;
    e2.s2d_cin = F;
    DT t2 = conv_block(c, s1.src(), no_src(), s1.g, P.downsampling[1], F, 1, 1, 1, nullptr, true, nullptr, nullptr, &e2);
    // residual blocks (embedding.py:38-41); the last sum is the descriptor
    const Geom g = t2.g;
    Src cur = t2.src();
    for (int r = 0; r < P.residual_blocks; ++r) {
        DT u1 = conv_block(c, cur, no_src(), g, P.blocks[2 * r], F, 1, 1, 1);
        DT u2 = conv_block(c, u1.src(), no_src(), g, P.blocks[2 * r + 1], F, 1, 1, 1);
        DT nxt;   // a residual sum is a plain tensor: the kernel that forms it records its largest magnitudes (Src::bound)
        nxt.g = g;
        nxt.raw = (r + 1 == P.residual_blocks) ? descriptor : c.get<float>(g.numel());
        carve_amax(c, nxt, materialize_records(g));
        if (!c.plan) c.run(launch_materialize(u2.src(), cur, g, nxt.raw, c.s, nxt.bound));
        tape_layer(c, 2, 0, 0, u2.src(), cur, g, nxt, nullptr, false);
        cur = nxt.src();
    }
    if (P.residual_blocks == 0) {
        DT d0;
        d0.g = g;
        d0.raw = descriptor;
        if (!c.plan) c.run(launch_materialize(cur, no_src(), g, descriptor, c.s));
        tape_layer(c, 2, 0, 0, cur, no_src(), g, d0, nullptr, false);
        cur = d0.src();
    }
    if (id_descriptor) *id_descriptor = cur.id;
    // _shortcut = convolutional_block_3x3(descriptor) (embedding.py:43-44, 65)
    DT v = conv_block(c, cur, no_src(), g, P.shortcut, P.shortcut_features, 1, 1, 1);
    DT so;
    so.g = v.g;
    so.raw = shortcut;
    if (!c.plan) c.run(launch_materialize(v.src(), no_src(), v.g, shortcut, c.s));
    tape_layer(c, 2, 0, 0, v.src(), no_src(), v.g, so, nullptr, false);
    if (id_shortcut) *id_shortcut = so.id;
}

// ====================================================================================================
// Backward: reverse walk over a tape.
// ====================================================================================================
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
// walk that under-estimates must surface as an error, not as a write past the caller's buffer
static thread_local size_t g_backward_arena_bytes = ~(size_t)0;
struct ArenaLimit {
    explicit ArenaLimit(size_t bytes) { g_backward_arena_bytes = bytes; }
    ~ArenaLimit() { g_backward_arena_bytes = ~(size_t)0; }
};

// dhat[i]: gradient with respect to the NORMALISED value of tensor i.  Entries preset by the caller (the
// gradient of the output, the gradient buffers of the external inputs) are used as they are; the others
// are carved from the backward arena on first use.
static void backward_walk(Ctx& c, const Tape& T, const GradMap& M, std::vector<float*>& dhat,
                          std::vector<char>& written) {
    static const bool debug_arena = getenv("PDS_DEBUG_ARENA") != nullptr;
    const std::vector<char> preset(written);  // gradients that live in the caller's tensors: never taken over
    std::vector<int> producer(T.tensors.size(), -1), consumers(T.tensors.size(), 0);
    for (size_t j = 0; j < T.layers.size(); ++j) {
        producer[T.layers[j].out] = (int)j;
        if (T.layers[j].a >= 0) ++consumers[T.layers[j].a];
        if (T.layers[j].b >= 0) ++consumers[T.layers[j].b];
    }
    for (int li = (int)T.layers.size() - 1; li >= 0; --li) {
        const TapeLayer& L = T.layers[li];
        const TapeTensor& out = T.tensors[L.out];
        float* g = dhat[L.out];
        if (debug_arena)
            fprintf(stderr, "[pds] backward %s layer %d type %d a %d b %d: arena at %zu\n", c.base ? "run " : "plan", li,
                    L.type, L.a, L.b, c.off);
        if (!written[L.out]) {
            c.run(set_error(-1, "backward: layer %d has no upstream gradient", li));
            return;
        }
        // may_adopt: grad_in is an arena buffer nobody else will read or write (a layer's fresh dx): the first
        // gradient of a tensor then simply BECOMES that buffer instead of being copied into a new one
        auto route = [&](int id, const float* grad_in, const Geom& in_g, bool may_adopt = false) {
            if (id < 0 || !T.tensors[id].needs_grad) return;
            const TapeTensor& t = T.tensors[id];
            if (may_adopt && !dhat[id] && !written[id] && !t.bcast_d) {
                dhat[id] = c.plan && !grad_in ? reinterpret_cast<float*>(8) : const_cast<float*>(grad_in);
                written[id] = 1;
                return;
            }
            if (!dhat[id]) {
                dhat[id] = c.get<float>(t.bcast_d ? (size_t)t.g.n * t.g.c * t.g.h * t.g.w : t.g.numel());
                if (!dhat[id]) dhat[id] = reinterpret_cast<float*>(8);  // plan mode: mark as carved
            }
            if (!c.plan) {
                if (t.bcast_d)
                    c.run(launch_grad_reduce_d(dhat[id], grad_in, in_g, written[id], c.s));
                else
                    c.run(launch_grad_add(dhat[id], grad_in, in_g.numel(), written[id], c.s));
            }
            written[id] = 1;
        };
        // One gradient for BOTH inputs of a layer (the terms of a sum, the two sources of a convolution).  `grad_in` is an
        // arena buffer nobody else reads after this layer (`mine`), so ONE input may take it over instead of receiving a
        // copy.  Both may even share it when one of them (`ro`) only ever reads it -- this layer is its single consumer,
        // so nothing is accumulated into it -- and is done reading before anything is accumulated into the other (`acc`):
        // ro's gradient is read when ro's producer is processed, so no layer between that producer and this one may
        // consume acc.  The residual blocks have exactly this shape (ro = the block's last convolution, acc = its input):
        // a 425 MB copy per residual sum / two-source layer of Matching.
        auto route_pair = [&](int a, int b, const float* grad_in, const Geom& in_g, bool mine) {
            // `mine` also requires that no OTHER tape tensor still owns this buffer as its gradient with its producer yet to
            // be processed (an earlier share, `dhat[b] = dhat[a]` below): taking the buffer over and accumulating into it
            // would corrupt that tensor's gradient.  The Matching / Regularization / Embedding tapes never form that shape;
            // the check turns the topological assumption into a rule -- such a buffer is copied, not adopted (run
            // walks only: planning walks carry marker pointers.  A tape that did trigger it would need more arena than planned
            // and fail loudly with the overflow error).
            if (mine && grad_in && !c.plan)
                for (size_t id = 0; id < T.tensors.size(); ++id)
                    if ((int)id != L.out && dhat[id] == grad_in && producer[id] >= 0 && producer[id] < li) mine = false;
            auto fresh = [&](int id) {
                return mine && id >= 0 && T.tensors[id].needs_grad && !T.tensors[id].bcast_d && !dhat[id] && !written[id];
            };
            auto may_share = [&](int ro, int acc) {
                if (consumers[ro] != 1 || producer[ro] < 0) return false;
                for (int j = producer[ro] + 1; j < li; ++j)
                    if (T.layers[j].a == acc || T.layers[j].b == acc) return false;
                return true;
            };
            const bool fa = fresh(a), fb = fresh(b);
            if (fa && fb && a != b && (may_share(a, b) || may_share(b, a))) {
                route(a, grad_in, in_g, true);
                dhat[b] = dhat[a];
                written[b] = 1;
                return;
            }
            route(a, grad_in, in_g, fa);
            route(b, grad_in, in_g, fb && !fa);
        };
        if (L.type == 2) {  // plain sum: the gradient flows unchanged to both terms (every consumer of the sum has
            route_pair(L.a, L.b, g, L.out_g, !preset[L.out]);   // delivered its share: the buffer is dead after this layer)
            continue;
        }
        if (L.type == 3) {  // space-to-depth: the adjoint is the inverse permutation
            if (!T.tensors[L.a].needs_grad) continue;
            float* dx = c.get<float>(L.in_g.numel());
            if (!c.plan) c.run(launch_depth_to_space(g, L.in_g.n, L.in_g.c, L.in_g.h, L.in_g.w, dx, c.s));
            route(L.a, dx, L.in_g, true);
            continue;
        }
        const PdsConvBlockParams* gp = M.find(L.P);
        if (!gp || !gp->weight || !gp->bias || (L.norm && (!gp->gamma || !gp->beta))) {
            c.run(set_error(-1, "backward: missing gradient buffers for layer %d", li));
            return;
        }
        // 1. through InstanceNorm + LeakyReLU
        const float* dz = g;
        // range certificate of dz (max |dz|, collected by the InstanceNorm backward that writes it): with it the
        // 64-channel weight and data gradients run their fp16-split kernels; a bare layer's dz (the caller's gradient)
        // gets its certificate from the bias-gradient pass below
        Src sdz = plain_src(nullptr);
        if (L.norm) {
            float* dz_amax = c.get<float>(kDzAmaxSlots);
            sdz.bound = dz_amax;
            sdz.bound_n = kDzAmaxSlots;
            sdz.bounded = 1;
            float* dzb = c.get<float>(out.g.numel());
            double* scratch = c.get<double>(in_bwd_scratch_doubles(out.g));
            const int groups = out.g.n * out.g.c * (out.per_plane ? out.g.d : 1);
            float* m1 = c.get<float>(groups);
            float* m2 = c.get<float>(groups);
            if (!c.plan)
                c.run(launch_in_bwd(g, out.raw, out.g, out.per_plane, out.mean, out.rstd, L.P->gamma, scratch, m1, m2,
                                    dzb, const_cast<float*>(gp->gamma), const_cast<float*>(gp->beta),
                                    const_cast<float*>(gp->bias), 0, c.s, dz_amax));   // (the bias gradient comes with it)
            dz = dzb;
        }
        sdz.p = dz;
        // 2. parameters
        const TapeTensor& ta = T.tensors[L.a];
        Src sa = ta.src();
        sa.bcast_d = 0;
        Src sb = no_src();
        if (L.b >= 0) sb = T.tensors[L.b].src();
        if (!L.norm) {   // a bare layer: dz is the upstream gradient itself, its channel sums need a pass of their own
            // (the same pass certifies the range of the caller's gradient)
            const int records = channel_sum_splits(out.g) * out.g.c;
            double* bias_scratch = c.get<double>((size_t)records);
            float* dz_amax = c.get<float>((size_t)records);
            sdz.bound = dz_amax;
            sdz.bound_n = records;
            sdz.bounded = 1;
            if (!c.plan)
                c.run(launch_channel_sum(dz, out.g, const_cast<float*>(gp->bias), 0, bias_scratch, c.s, dz_amax));
        }
        const float* weight = L.s2d_cin ? L.weight_used : L.P->weight;
        const int taps = L.kd * 9;
        // space-to-depth layer: the gradient of the 3x3 weights is formed in scratch, then gathered into the 5x5 one
        float* dweight = L.s2d_cin ? c.get<float>((size_t)L.out_g.c * L.in_g.c * taps) : const_cast<float*>(gp->weight);
        if (wgrad2d_mfma_supported(L.type, L.kd, L.stride, sb, L.in_g, L.out_g)) {
            float* ws = c.get<float>(wgrad2d_mfma_scratch_floats(L.in_g, L.out_g));
            if (!c.plan)
                c.run(launch_wgrad2d_mfma(sa, sb, sdz, dweight, L.in_g, L.out_g, 0, ws, c.s));
        } else if (wgrad3d_mfma_supported(L.type, L.kd, L.stride, L.in_g, L.out_g)) {
            float* ws = c.get<float>(wgrad3d_mfma_scratch_floats(L.in_g, L.out_g));
            if (!c.plan)
                c.run(launch_wgrad3d_mfma(sa, sb, dz, dweight, L.in_g, L.out_g, 0, ws, c.s));
        } else if (wgrad_up_full_mfma_supported(L.type, L.kd, sb, L.in_g, L.out_g)) {
            float* ws = c.get<float>(wgrad_up_full_mfma_scratch_floats(L.in_g));
            if (!c.plan) c.run(launch_wgrad_up_full_mfma(sa, dz, dweight, L.in_g, 0, ws, c.s));
        } else if (wgrad3d_s2_mfma_supported(L.type, L.kd, L.stride, L.in_g, L.out_g)) {
            float* ws = c.get<float>(wgrad3d_s2_mfma_scratch_floats(L.type, L.in_g, L.out_g));
            if (!c.plan)
                c.run(launch_wgrad3d_s2_mfma(L.type, sa, sb, dz, dweight, L.in_g, L.out_g, 0, ws, c.s));
        } else {
            static const bool debug_fallback = getenv("PDS_DEBUG_ARENA") != nullptr;
            if (debug_fallback && !c.plan)
                fprintf(stderr, "[pds] VALU weight gradient: type %d kd %d stride %d in [%d,%d,%d,%d,%d] out c %d two-source %d\n",
                        L.type, L.kd, L.stride, L.in_g.n, L.in_g.c, L.in_g.d, L.in_g.h, L.in_g.w, L.out_g.c, sb.p != nullptr);
            double* weight_scratch = c.get<double>(bwd_weight_scratch_doubles(L.type, L.kd, L.in_g, L.out_g));
            if (!c.plan)
                c.run(launch_bwd_weight(L.type, L.kd, L.stride, sa, sb, dz, dweight, L.in_g, L.out_g, 0, weight_scratch,
                                        c.s));
        }
        if (L.s2d_cin && !c.plan)
            c.run(launch_s2d_weights_bwd(dweight, const_cast<float*>(gp->weight), L.out_g.c, L.s2d_cin, 0, c.s));
        // 3. input
        if (!ta.needs_grad && (L.b < 0 || !T.tensors[L.b].needs_grad)) continue;
        float* dx = c.get<float>(L.in_g.numel());
        if (L.type == 0 && L.stride == 1) {
            // stride-1 convolution: dx = conv(dz, flipped weights) on the forward kernels (MFMA where supported)
            float* wf = c.get<float>((size_t)L.out_g.c * L.in_g.c * taps);
            if (!c.plan) c.run(launch_flip_weights(weight, wf, L.out_g.c, L.in_g.c, taps, c.s));
            if (L.kd == 1 && L.in_g.c > 64 && L.in_g.c % 64 == 0) {
                // more than 64 gradient channels (the 4C-channel input of a space-to-depth layer): no MFMA tiling
                // covers that as one launch, so it is cut into 64-channel blocks per batch entry, each of which
                // runs on the 64-channel (Winograd) kernel instead of the VALU fallback
                const size_t vol = (size_t)L.in_g.d * L.in_g.h * L.in_g.w;
                if (L.in_g.w % 2 == 0 && L.out_g.c % 4 == 0) {
                    // Winograd kernels: one launch per 64-channel block over the whole batch, written as a channel
                    // slice of dx (the planes of the training-mode Matching are batch entries: 96 one-plane launches
                    // that each filled a quarter of the chip became 2)
                    for (int j = 0; j < L.in_g.c / 64; ++j) {
                        PdsConvBlockParams pf{wf + (size_t)j * 64 * L.out_g.c * taps, nullptr, nullptr, nullptr};
                        ConvExtra slice;
                        slice.out_batch_channels = L.in_g.c;
                        conv_block(c, sdz, no_src(), L.out_g, pf, 64, 1, 1, 0,
                                   dx ? dx + (size_t)j * 64 * vol : nullptr, true, nullptr, nullptr, &slice);
                    }
                } else {
                    Geom one = L.out_g;
                    one.n = 1;
                    for (int i = 0; i < L.in_g.n; ++i)
                        for (int j = 0; j < L.in_g.c / 64; ++j) {
                            PdsConvBlockParams pf{wf + (size_t)j * 64 * L.out_g.c * taps, nullptr, nullptr, nullptr};
                            float* block = dx ? dx + ((size_t)i * L.in_g.c + (size_t)j * 64) * vol : nullptr;
                            conv_block(c, plain_src(dz ? dz + (size_t)i * L.out_g.c * vol : nullptr), no_src(), one, pf,
                                       64, 1, 1, 0, block);
                        }
                }
            } else {
                PdsConvBlockParams pf{wf, nullptr, nullptr, nullptr};
                conv_block(c, sdz, no_src(), L.out_g, pf, L.in_g.c, L.kd, 1, 0, dx);
            }
        } else if (!c.plan) {
            c.run(launch_bwd_data(L.type, L.kd, L.stride, dz, weight, dx, L.in_g, L.out_g, c.s));
        }
        route_pair(L.a, L.b, dx, L.in_g, true);   // dx goes to both sources of a two-source layer
    }
}

}  // namespace pds

using namespace pds;

// ====================================================================================================
extern "C" {

int pds_abi_version(void) { return PDS_ABI_VERSION; }
long long pds_nonfinite_statistics(int reset) { return nonfinite_statistics(reset); }

int pds_probe_begin(const char* kernel, int capacity) {
    PDS_REQUIRE(kernel && kernel[0] && strlen(kernel) < sizeof(g_probe_name), "probe: bad kernel name");
    PDS_REQUIRE(capacity > 0 && capacity <= kProbeMax, "probe: capacity %d outside 1..%d", capacity, kProbeMax);
    for (; g_probe_events < capacity; ++g_probe_events) {
        if (hipEventCreate(&g_probe_start[g_probe_events]) != hipSuccess ||
            hipEventCreate(&g_probe_stop[g_probe_events]) != hipSuccess)
            return set_error(-1, "probe: hipEventCreate failed");
    }
    strcpy(g_probe_name, kernel);
    g_probe_capacity = capacity;
    g_probe_count = 0;
    g_probe_armed.store(1, std::memory_order_release);
    return 0;
}

int pds_probe_end(float* ms, int* workgroups, int capacity) {
    g_probe_armed.store(0, std::memory_order_release);
    const int n = g_probe_count < capacity ? g_probe_count : capacity;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(g_probe_stop[i]) != hipSuccess) return set_error(-1, "probe: hipEventSynchronize failed");
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_probe_start[i], g_probe_stop[i]) != hipSuccess)
            return set_error(-1, "probe: hipEventElapsedTime failed");
        if (ms) ms[i] = t;
        if (workgroups) workgroups[i] = g_probe_wgs[i];
    }
    return n;
}
const char* pds_last_error(void) { return g_error; }

int pds_debug_chain_stamps(unsigned* ticks, int capacity) {
    PDS_REQUIRE(ticks && capacity > 0, "chain stamps: bad arguments");
    return ks_chain_debug_stamps(ticks, capacity);
}

int pds_subpixel_map_fwd(const float* similarities, float* disparities, int batch, int planes, int height,
                         int width, int half_support_window, int disparity_step, pds_stream_t stream) {
    PDS_REQUIRE(similarities && disparities, "subpixel_map: null pointer");
    PDS_REQUIRE(batch > 0 && planes > 0 && height > 0 && width > 0, "subpixel_map: bad shape");
    PDS_REQUIRE(disparity_step >= 1 && half_support_window >= 1 && half_support_window % disparity_step == 0,
                "subpixel_map: bad window/step");
    // Python floor division of the negated window (estimator.py:66-68)
    const int hi = half_support_window / disparity_step;
    const int lo = -((half_support_window + disparity_step - 1) / disparity_step);
    return launch_subpixel_map(similarities, disparities, batch, planes, height, width, lo, hi, disparity_step,
                               (hipStream_t)stream);
}

int pds_shift_concat_fwd(const float* left, const float* right, float* out, int batch, int channels, int h, int w,
                         int d_begin, int d_count, pds_stream_t stream) {
    PDS_REQUIRE(left && right && out, "shift_concat: null pointer");
    PDS_REQUIRE(batch > 0 && channels > 0 && h > 0 && w > 0 && d_begin >= 0 && d_count > 0,
                "shift_concat: bad shape");
    return launch_shift_concat(left, right, out, batch, channels, h, w, d_begin, d_count, (hipStream_t)stream);
}

static int check_matching_params(const PdsMatchingParams* P) {
    PDS_REQUIRE(P, "matching: null params");
    PDS_REQUIRE(P->features > 0 && P->signature_features > 0 && P->residual_blocks >= 0, "matching: bad params");
    PDS_REQUIRE(P->first.weight && P->first.bias && P->last.weight && P->last.bias, "matching: null weights");
    for (int i = 0; i < 2 * P->residual_blocks; ++i)
        PDS_REQUIRE(P->blocks && P->blocks[i].weight && P->blocks[i].bias && P->blocks[i].gamma && P->blocks[i].beta,
                    "matching: null residual-block weights");
    return 0;
}

size_t pds_matching_workspace_bytes(const PdsMatchingParams* params, int batch, int h, int w, int d_count) {
    if (check_matching_params(params)) return 0;
    Ctx c{nullptr, 0, true, nullptr};
    matching_pipeline(c, *params, nullptr, nullptr, nullptr, batch, h, w, 0, d_count);
    return c.off;
}

int pds_matching_fwd(const PdsMatchingParams* params, const float* left, const float* right, float* signatures,
                     int batch, int h, int w, int d_begin, int d_count, void* workspace, size_t workspace_bytes,
                     int weights_resident, pds_stream_t stream) {
    if (int rc = check_matching_params(params)) return rc;
    PDS_REQUIRE(left && right && signatures && workspace, "matching: null pointer");
    PDS_REQUIRE(batch > 0 && h > 0 && w > 0 && d_begin >= 0 && d_count > 0, "matching: bad shape");
    const size_t need = pds_matching_workspace_bytes(params, batch, h, w, d_count);
    PDS_REQUIRE(workspace_bytes >= need, "matching: workspace too small (%zu < %zu)", workspace_bytes, need);
    return run_with_batched_packing(workspace, (hipStream_t)stream, [&](Ctx& c) {
        matching_pipeline(c, *params, left, right, signatures, batch, h, w, d_begin, d_count);
    }, weights_resident != 0);
}

static int matching_backward(bool plan, size_t* bytes, const PdsMatchingParams* params, const PdsMatchingParams* grads,
                             const float* left, const float* right, const float* grad_signatures, float* grad_left,
                             float* grad_right, int batch, int h, int w, int d_begin, int d_count, void* fwd_workspace,
                             void* workspace, hipStream_t stream);

/* ABI v5: the differentiable route of Matching + MatchingOperation (see include/pds_hip.h) */
size_t pds_matching_train_workspace_bytes(const PdsMatchingParams* params, int batch, int h, int w, int d_count) {
    if (check_matching_params(params)) return 0;
    Ctx c{nullptr, 0, true, nullptr};
    matching_pipeline(c, *params, nullptr, nullptr, nullptr, batch, h, w, 0, d_count, true);
    return c.off;
}

int pds_matching_train_fwd(const PdsMatchingParams* params, const float* left, const float* right, float* signatures,
                           int batch, int h, int w, int d_begin, int d_count, void* workspace, size_t workspace_bytes,
                           pds_stream_t stream) {
    if (int rc = check_matching_params(params)) return rc;
    PDS_REQUIRE(left && right && signatures && workspace, "matching_train: null pointer");
    PDS_REQUIRE(batch > 0 && h > 0 && w > 0 && d_begin >= 0 && d_count > 0, "matching_train: bad shape");
    PDS_REQUIRE(params->residual_blocks >= 0, "matching_train: bad block count");
    const size_t need = pds_matching_train_workspace_bytes(params, batch, h, w, d_count);
    PDS_REQUIRE(workspace_bytes >= need, "matching_train: workspace too small (%zu < %zu)", workspace_bytes, need);
    return run_with_batched_packing(workspace, (hipStream_t)stream, [&](Ctx& c) {
        matching_pipeline(c, *params, left, right, signatures, batch, h, w, d_begin, d_count, true);
    });
}

size_t pds_matching_bwd_workspace_bytes(const PdsMatchingParams* params, int batch, int h, int w, int d_count) {
    if (check_matching_params(params)) return 0;
    size_t bytes = 0;
    if (matching_backward(true, &bytes, params, params, nullptr, nullptr, nullptr, nullptr, nullptr, batch, h, w, 0,
                          d_count, nullptr, nullptr, nullptr))
        return 0;
    return bytes + 256;
}

int pds_matching_bwd(const PdsMatchingParams* params, const PdsMatchingParams* grads, const float* left,
                     const float* right, const float* grad_signatures, float* grad_left, float* grad_right, int batch,
                     int h, int w, int d_begin, int d_count, void* fwd_workspace, size_t fwd_workspace_bytes,
                     void* workspace, size_t workspace_bytes, pds_stream_t stream) {
    if (int rc = check_matching_params(params)) return rc;
    if (int rc = check_matching_params(grads)) return rc;
    PDS_REQUIRE(left && right && grad_signatures && grad_left && grad_right && fwd_workspace && workspace,
                "matching_bwd: null pointer");
    PDS_REQUIRE(batch > 0 && h > 0 && w > 0 && d_begin >= 0 && d_count > 0, "matching_bwd: bad shape");
    PDS_REQUIRE(fwd_workspace_bytes >= pds_matching_train_workspace_bytes(params, batch, h, w, d_count),
                "matching_bwd: forward workspace too small");
    const size_t need = pds_matching_bwd_workspace_bytes(params, batch, h, w, d_count);
    PDS_REQUIRE(workspace_bytes >= need, "matching_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    ArenaLimit limit(workspace_bytes);
    return matching_backward(false, nullptr, params, grads, left, right, grad_signatures, grad_left, grad_right, batch, h,
                             w, d_begin, d_count, fwd_workspace, workspace, (hipStream_t)stream);
}

size_t pds_matching_operation_workspace_bytes(const PdsMatchingParams* params, int n, int h, int w) {
    if (check_matching_params(params)) return 0;
    Ctx c{nullptr, 0, true, nullptr};
    operation_pipeline(c, *params, nullptr, nullptr, n, h, w);
    return c.off;
}

int pds_matching_operation_fwd(const PdsMatchingParams* params, const float* concatenated, float* signature, int n,
                               int h, int w, void* workspace, size_t workspace_bytes, pds_stream_t stream) {
    if (int rc = check_matching_params(params)) return rc;
    PDS_REQUIRE(concatenated && signature && workspace, "matching_operation: null pointer");
    PDS_REQUIRE(n > 0 && h > 0 && w > 0, "matching_operation: bad shape");
    const size_t need = pds_matching_operation_workspace_bytes(params, n, h, w);
    PDS_REQUIRE(workspace_bytes >= need, "matching_operation: workspace too small (%zu < %zu)", workspace_bytes,
                need);
    return run_with_batched_packing(workspace, (hipStream_t)stream, [&](Ctx& c) {
        operation_pipeline(c, *params, concatenated, signature, n, h, w);
    });
}

static int check_block(const PdsConvBlockParams& b, bool norm, const char* name) {
    PDS_REQUIRE(b.weight && b.bias, "%s: null weight/bias", name);
    if (norm) PDS_REQUIRE(b.gamma && b.beta, "%s: null InstanceNorm affine", name);
    return 0;
}

static int check_regularization(const PdsRegularizationParams* P, int batch, int d, int h, int w) {
    PDS_REQUIRE(P, "regularization: null params");
    PDS_REQUIRE(P->features >= 2 && P->features % 2 == 0, "regularization: features must be even");
    PDS_REQUIRE(batch > 0 && d > 0 && h > 0 && w > 0, "regularization: bad shape");
    PDS_REQUIRE(d % 16 == 0 && h % 16 == 0 && w % 16 == 0,
                "regularization: D, h, w must be multiples of 16 (got %d, %d, %d)", d, h, w);
    PDS_REQUIRE((d / 16) * (h / 16) * (w / 16) > 1,
                "regularization: InstanceNorm needs more than one element at 1/16 scale");
    if (int rc = check_block(P->smoothing, true, "regularization._smoothing")) return rc;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j) {
            if (int rc = check_block(P->contraction[i][j], true, "regularization._contraction_blocks")) return rc;
            if (int rc = check_block(P->expansion[i][j], true, "regularization._expansion_blocks")) return rc;
        }
    if (int rc = check_block(P->upsample_half, true, "regularization._upsample_to_halfsize")) return rc;
    return check_block(P->upsample_full, false, "regularization._upsample_to_fullsize");
}

size_t pds_regularization_workspace_bytes(const PdsRegularizationParams* params, int batch, int d, int h, int w) {
    if (check_regularization(params, batch, d, h, w)) return 0;
    Ctx c{nullptr, 0, true, nullptr};
    regularization_pipeline(c, *params, nullptr, nullptr, nullptr, batch, d, h, w);
    // the fused eval entry point additionally stages the cost volume in the workspace
    c.get<float>((size_t)batch * 2 * d * 4 * h * 4 * w);
    return c.off;
}

int pds_regularization_fwd(const PdsRegularizationParams* params, const float* signatures,
                           const float* left_shortcut, float* cost, int batch, int d, int h, int w, void* workspace,
                           size_t workspace_bytes, int weights_resident, pds_stream_t stream) {
    if (int rc = check_regularization(params, batch, d, h, w)) return rc;
    PDS_REQUIRE(signatures && left_shortcut && cost && workspace, "regularization: null pointer");
    const size_t need = pds_regularization_workspace_bytes(params, batch, d, h, w);
    PDS_REQUIRE(workspace_bytes >= need, "regularization: workspace too small (%zu < %zu)", workspace_bytes, need);
    return run_with_batched_packing(workspace, (hipStream_t)stream, [&](Ctx& c) {
        regularization_pipeline(c, *params, signatures, left_shortcut, cost, batch, d, h, w);
    }, weights_resident != 0);
}

int pds_regularization_subpixel_map_fwd(const PdsRegularizationParams* params, const float* signatures,
                                        const float* left_shortcut, float* disparities, int batch, int d, int h,
                                        int w, int half_support_window, int disparity_step, int crop_top,
                                        int crop_left, void* workspace, size_t workspace_bytes, int weights_resident,
                                        pds_stream_t stream) {
    if (int rc = check_regularization(params, batch, d, h, w)) return rc;
    PDS_REQUIRE(signatures && left_shortcut && disparities && workspace, "regularization_subpixel_map: null pointer");
    PDS_REQUIRE(disparity_step >= 1 && half_support_window >= 1 && half_support_window % disparity_step == 0,
                "regularization_subpixel_map: bad window/step");
    PDS_REQUIRE(crop_top >= 0 && crop_top < 4 * h && crop_left >= 0 && crop_left < 4 * w,
                "regularization_subpixel_map: bad crop (%d, %d)", crop_top, crop_left);
    const size_t need = pds_regularization_workspace_bytes(params, batch, d, h, w);
    PDS_REQUIRE(workspace_bytes >= need, "regularization_subpixel_map: workspace too small (%zu < %zu)",
                workspace_bytes, need);
    Ctx c{(char*)workspace, 0, false, (hipStream_t)stream};
    const int hi = half_support_window / disparity_step;
    const int lo = -((half_support_window + disparity_step - 1) / disparity_step);
    if (upsample_estimator_supported(params->features / 2, lo, hi)) {
        // fused: the full-resolution cost volume is never materialised
        DT half;
        float* w_pairs = nullptr;
        if (int rc = run_with_batched_packing(workspace, (hipStream_t)stream, [&](Ctx& cc) {
                half = regularization_trunk(cc, *params, signatures, left_shortcut, batch, d, h, w);
                // (the same carve as regularization_pipeline, which sized the workspace)
                w_pairs = cc.get<float>((size_t)half.g.c * 48);
                if (cc.before_packing())
                    cc.run(launch_upsample_weight_pairs(params->upsample_full.weight, w_pairs, half.g.c, cc.s));
            }, weights_resident != 0))
            return rc;
        return launch_upsample_estimator(half.raw, half.scale, half.shift, w_pairs,
                                         params->upsample_full.bias, disparities, batch, half.g.c, half.g.d, half.g.h,
                                         half.g.w, lo, hi, disparity_step, crop_top, crop_left, (hipStream_t)stream);
    }
    PDS_REQUIRE(crop_top == 0 && crop_left == 0,
                "regularization_subpixel_map: the crop is only folded into the fused kernel (4 features, window <= 4 taps)");
    float* cost = c.get<float>((size_t)batch * 2 * d * 4 * h * 4 * w);
    if (int rc = run_with_batched_packing((char*)workspace + c.off, (hipStream_t)stream, [&](Ctx& cc) {
            regularization_pipeline(cc, *params, signatures, left_shortcut, cost, batch, d, h, w);
        }, weights_resident != 0))
        return rc;
    return pds_subpixel_map_fwd(cost, disparities, batch, 2 * d, 4 * h, 4 * w, half_support_window, disparity_step,
                                stream);
}

size_t pds_conv_block_workspace_bytes(int n, int cin, int cout, int d, int h, int w, int kd, int stride,
                                      int per_plane) {
    PdsConvBlockParams dummy{nullptr, nullptr, (const float*)1, (const float*)1};
    // sized for the chained form (input behind a deferred InstanceNorm): it may pick a kernel with more statistics
    // records per plane than the plain form, never fewer; with and without a range bound (the kernel choice -- and with
    // it the packed-weight scratch -- depends on it): the larger of the two
    size_t need = 0;
    for (int bounded = 0; bounded < 2; ++bounded) {
        Ctx c{nullptr, 0, true, nullptr};
        Src src = plain_src(nullptr);
        src.normed = 1;
        src.bounded = bounded;
        conv_block(c, src, no_src(), Geom{n, cin, d, h, w}, dummy, cout, kd, stride, per_plane, (float*)1, true,
                   (float*)1, (float*)1);
        if (c.off > need) need = c.off;
    }
    return need + 256;
}

int pds_conv_block_fwd(const PdsConvBlockParams* params, const float* x, float* raw, float* scale, float* shift,
                       int n, int cin, int cout, int d, int h, int w, int kd, int stride, int per_plane,
                       void* workspace, size_t workspace_bytes, pds_stream_t stream) {
    PDS_REQUIRE(params && x && raw && workspace, "conv_block: null pointer");
    PDS_REQUIRE(params->weight && params->bias, "conv_block: null weight/bias");
    PDS_REQUIRE(n > 0 && cin > 0 && cout > 0 && d > 0 && h > 0 && w > 0, "conv_block: bad shape");
    PDS_REQUIRE((kd == 1 || kd == 3) && (stride == 1 || stride == 2) && !(kd == 1 && stride == 2),
                "conv_block: unsupported kd=%d stride=%d", kd, stride);
    if (params->gamma) PDS_REQUIRE(params->beta && scale && shift, "conv_block: null InstanceNorm outputs");
    const size_t need = pds_conv_block_workspace_bytes(n, cin, cout, d, h, w, kd, stride, per_plane);
    PDS_REQUIRE(workspace_bytes >= need, "conv_block: workspace too small (%zu < %zu)", workspace_bytes, need);
    Ctx c{(char*)workspace, 0, false, (hipStream_t)stream};
    conv_block(c, plain_src(x), no_src(), Geom{n, cin, d, h, w}, *params, cout, kd, stride, per_plane, raw, true,
               scale, shift);
    return c.err;
}

// The same block behind another block: x is the producer's RAW output and the loader applies the producer's folded
// InstanceNorm, x^ = x_scale * x + x_shift (per (n, c), or per (n, c, d) when x_per_plane) -- how the blocks of
// MatchingOperation / Regularization are chained inside the modules (no normalised tensor is ever stored).  x_bound
// (one device float bounding |x^|, or null) is the range certificate of common.hpp Src::bound.
int pds_conv_block_chained_fwd(const PdsConvBlockParams* params, const float* x, const float* x_scale,
                               const float* x_shift, int x_per_plane, const float* x_bound, float* raw, float* scale,
                               float* shift, int n, int cin, int cout, int d, int h, int w, int kd, int stride,
                               int per_plane, void* workspace, size_t workspace_bytes, pds_stream_t stream) {
    PDS_REQUIRE(params && x && x_scale && x_shift && raw && workspace, "conv_block_chained: null pointer");
    PDS_REQUIRE(params->weight && params->bias, "conv_block_chained: null weight/bias");
    PDS_REQUIRE(n > 0 && cin > 0 && cout > 0 && d > 0 && h > 0 && w > 0, "conv_block_chained: bad shape");
    PDS_REQUIRE((kd == 1 || kd == 3) && (stride == 1 || stride == 2) && !(kd == 1 && stride == 2),
                "conv_block_chained: unsupported kd=%d stride=%d", kd, stride);
    if (params->gamma) PDS_REQUIRE(params->beta && scale && shift, "conv_block_chained: null InstanceNorm outputs");
    const size_t need = pds_conv_block_workspace_bytes(n, cin, cout, d, h, w, kd, stride, per_plane);
    PDS_REQUIRE(workspace_bytes >= need, "conv_block_chained: workspace too small (%zu < %zu)", workspace_bytes, need);
    Ctx c{(char*)workspace, 0, false, (hipStream_t)stream};
    Src src{x, x_scale, x_shift, x_per_plane ? 1 : 0, 0};
    src.normed = 1;
    if (x_bound) {
        src.bound = x_bound;
        src.bound_n = 1;
        src.bounded = 1;
    }
    conv_block(c, src, no_src(), Geom{n, cin, d, h, w}, *params, cout, kd, stride, per_plane, raw, true, scale, shift);
    return c.err;
}

// stand-in parameters of the planning walks: non-null marks (never dereferenced), so that the same checks that
// guard a real call pass
#define PDS_MARK reinterpret_cast<const float*>(8)
static const PdsConvBlockParams kDummyBlocks[2] = {{PDS_MARK, PDS_MARK, PDS_MARK, PDS_MARK},
                                                   {PDS_MARK, PDS_MARK, PDS_MARK, PDS_MARK}};

size_t pds_contraction_block_workspace_bytes(int batch, int c_, int d, int h, int w) {
    Ctx c{nullptr, 0, true, nullptr};
    contraction_pipeline(c, kDummyBlocks, nullptr, nullptr, nullptr, Geom{batch, c_, d, h, w});
    return c.off + 256;
}

int pds_contraction_block_fwd(const PdsConvBlockParams* downsampling, const PdsConvBlockParams* smoothing,
                              const float* x, float* down_out, float* smooth_out, int batch, int c_, int d, int h,
                              int w, void* workspace, size_t workspace_bytes, pds_stream_t stream) {
    PDS_REQUIRE(downsampling && smoothing && x && down_out && smooth_out && workspace, "contraction: null pointer");
    PDS_REQUIRE(batch > 0 && c_ > 0 && d > 0 && h > 0 && w > 0, "contraction: bad shape");
    if (int rc = check_block(*downsampling, true, "contraction._downsampling_2x")) return rc;
    if (int rc = check_block(*smoothing, true, "contraction._smoothing")) return rc;
    const size_t need = pds_contraction_block_workspace_bytes(batch, c_, d, h, w);
    PDS_REQUIRE(workspace_bytes >= need, "contraction: workspace too small (%zu < %zu)", workspace_bytes, need);
    const PdsConvBlockParams pp[2] = {*downsampling, *smoothing};
    Ctx c{(char*)workspace, 0, false, (hipStream_t)stream};
    contraction_pipeline(c, pp, x, down_out, smooth_out, Geom{batch, c_, d, h, w});
    return c.err;
}

size_t pds_expansion_block_workspace_bytes(int batch, int c_, int d, int h, int w) {
    Ctx c{nullptr, 0, true, nullptr};
    expansion_pipeline(c, kDummyBlocks, nullptr, nullptr, nullptr, Geom{batch, c_, d, h, w});
    return c.off + 256;
}

int pds_expansion_block_fwd(const PdsConvBlockParams* upsampling, const PdsConvBlockParams* smoothing, const float* x,
                            const float* shortcut, float* out, int batch, int c_, int d, int h, int w,
                            void* workspace, size_t workspace_bytes, pds_stream_t stream) {
    PDS_REQUIRE(upsampling && smoothing && x && shortcut && out && workspace, "expansion: null pointer");
    PDS_REQUIRE(batch > 0 && c_ >= 2 && c_ % 2 == 0 && d > 0 && h > 0 && w > 0, "expansion: bad shape");
    if (int rc = check_block(*upsampling, true, "expansion._upsampling_2x")) return rc;
    if (int rc = check_block(*smoothing, true, "expansion._smoothing")) return rc;
    const size_t need = pds_expansion_block_workspace_bytes(batch, c_, d, h, w);
    PDS_REQUIRE(workspace_bytes >= need, "expansion: workspace too small (%zu < %zu)", workspace_bytes, need);
    const PdsConvBlockParams pp[2] = {*upsampling, *smoothing};
    Ctx c{(char*)workspace, 0, false, (hipStream_t)stream};
    expansion_pipeline(c, pp, x, shortcut, out, Geom{batch, c_, d, h, w});
    return c.err;
}

// ----------------------------------------------------------------------------------------------------
// backward entry points
// ----------------------------------------------------------------------------------------------------
static size_t regularization_fwd_bytes(const PdsRegularizationParams* params, int batch, int d, int h, int w) {
    Ctx c{nullptr, 0, true, nullptr};
    regularization_pipeline(c, *params, nullptr, nullptr, nullptr, batch, d, h, w);
    return c.off;
}

static int regularization_backward(bool plan, size_t* bytes, const PdsRegularizationParams* params,
                                   const PdsRegularizationParams* grads, const float* signatures,
                                   const float* left_shortcut, const float* grad_cost, float* grad_signatures,
                                   float* grad_left_shortcut, int batch, int d, int h, int w, void* fwd_workspace,
                                   void* workspace, hipStream_t stream) {
    Tape tape;
    Ctx re{plan ? nullptr : (char*)fwd_workspace, 0, true, stream};  // re-walk: pointers only, no launches
    re.tape = &tape;
    regularization_pipeline(re, *params, signatures, left_shortcut, const_cast<float*>(grad_cost) /*placeholder*/,
                            batch, d, h, w);
    if (re.err) return re.err;
    std::vector<float*> dhat(tape.tensors.size(), nullptr);
    std::vector<char> written(tape.tensors.size(), 0);
    // tape order: tensor 0 = signatures, 1 = left shortcut, last = cost
    dhat[0] = grad_signatures;
    dhat[1] = grad_left_shortcut;
    dhat[tape.tensors.size() - 1] = const_cast<float*>(grad_cost);
    written[tape.tensors.size() - 1] = 1;
    GradMap M{reinterpret_cast<const char*>(params), reinterpret_cast<const char*>(grads), sizeof(PdsRegularizationParams)};
    Ctx c{plan ? nullptr : (char*)workspace, 0, plan, stream};
    if (!plan) c.limit = g_backward_arena_bytes;
    if (plan) {  // the walk dereferences nothing in plan mode, but needs non-null marks for the presets
        dhat[0] = dhat[1] = dhat[tape.tensors.size() - 1] = reinterpret_cast<float*>(8);
    }
    backward_walk(c, tape, M, dhat, written);
    if (bytes) *bytes = c.off;
    return c.err;
}

size_t pds_regularization_bwd_workspace_bytes(const PdsRegularizationParams* params, int batch, int d, int h, int w) {
    if (check_regularization(params, batch, d, h, w)) return 0;
    size_t bytes = 0;
    PdsRegularizationParams dummy = *params;
    if (regularization_backward(true, &bytes, params, &dummy, nullptr, nullptr, nullptr, nullptr, nullptr, batch, d, h,
                                w, nullptr, nullptr, nullptr))
        return 0;
    return bytes + 256;
}

int pds_regularization_bwd(const PdsRegularizationParams* params, const PdsRegularizationParams* grads,
                           const float* signatures, const float* left_shortcut, const float* grad_cost,
                           float* grad_signatures, float* grad_left_shortcut, int batch, int d, int h, int w,
                           void* fwd_workspace, size_t fwd_workspace_bytes, void* workspace, size_t workspace_bytes,
                           pds_stream_t stream) {
    if (int rc = check_regularization(params, batch, d, h, w)) return rc;
    PDS_REQUIRE(grads && signatures && left_shortcut && grad_cost && grad_signatures && grad_left_shortcut &&
                    fwd_workspace && workspace,
                "regularization_bwd: null pointer");
    PDS_REQUIRE(fwd_workspace_bytes >= regularization_fwd_bytes(params, batch, d, h, w),
                "regularization_bwd: forward workspace too small");
    const size_t need = pds_regularization_bwd_workspace_bytes(params, batch, d, h, w);
    PDS_REQUIRE(workspace_bytes >= need, "regularization_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    ArenaLimit limit(workspace_bytes);
    return regularization_backward(false, nullptr, params, grads, signatures, left_shortcut, grad_cost,
                                   grad_signatures, grad_left_shortcut, batch, d, h, w, fwd_workspace, workspace,
                                   (hipStream_t)stream);
}

static int operation_backward(bool plan, size_t* bytes, const PdsMatchingParams* params, const PdsMatchingParams* grads,
                              const float* concatenated, const float* grad_signature, float* grad_concatenated, int n,
                              int h, int w, void* fwd_workspace, void* workspace, hipStream_t stream) {
    Tape tape;
    Ctx re{plan ? nullptr : (char*)fwd_workspace, 0, true, stream};
    re.tape = &tape;
    operation_pipeline(re, *params, concatenated, const_cast<float*>(grad_signature) /*placeholder*/, n, h, w);
    if (re.err) return re.err;
    std::vector<float*> dhat(tape.tensors.size(), nullptr);
    std::vector<char> written(tape.tensors.size(), 0);
    dhat[0] = grad_concatenated;
    dhat[tape.tensors.size() - 1] = const_cast<float*>(grad_signature);
    written[tape.tensors.size() - 1] = 1;
    GradMap M{reinterpret_cast<const char*>(params), reinterpret_cast<const char*>(grads), sizeof(PdsMatchingParams)};
    M.blocks_params = params->blocks;
    M.blocks_grads = grads->blocks;
    M.blocks_count = 2 * params->residual_blocks;
    Ctx c{plan ? nullptr : (char*)workspace, 0, plan, stream};
    if (!plan) c.limit = g_backward_arena_bytes;
    if (plan) dhat[0] = dhat[tape.tensors.size() - 1] = reinterpret_cast<float*>(8);
    backward_walk(c, tape, M, dhat, written);
    if (bytes) *bytes = c.off;
    return c.err;
}

// Backward of the training route of Matching (matching_pipeline(train)): the tape walk from the signatures down to
// x0, then layer 0 through its factorisation -- ONE streaming reduction of d loss / d x0 over the disparity planes
// (l0_combine_bwd) and single-plane convolution gradients, instead of a 128 -> 64 weight / data gradient over all planes
// and the adjoint of an 850 MB concat (reference: autograd through matching.py:50-62).
static int matching_backward(bool plan, size_t* bytes, const PdsMatchingParams* params, const PdsMatchingParams* grads,
                             const float* left, const float* right, const float* grad_signatures, float* grad_left,
                             float* grad_right, int batch, int h, int w, int d_begin, int d_count, void* fwd_workspace,
                             void* workspace, hipStream_t stream) {
    Tape tape;
    MatchingL0 l0;
    Ctx re{plan ? nullptr : (char*)fwd_workspace, 0, true, stream};
    re.tape = &tape;
    matching_pipeline(re, *params, left, right, const_cast<float*>(grad_signatures) /*placeholder*/, batch, h, w, d_begin,
                      d_count, true, &l0);
    if (re.err) return re.err;
    if (tape.tensors.empty()) return set_error(-1, "matching_bwd: empty tape");
    std::vector<float*> dhat(tape.tensors.size(), nullptr);
    std::vector<char> written(tape.tensors.size(), 0);
    dhat[tape.tensors.size() - 1] = plan ? reinterpret_cast<float*>(8) : const_cast<float*>(grad_signatures);
    written[tape.tensors.size() - 1] = 1;
    GradMap M{reinterpret_cast<const char*>(params), reinterpret_cast<const char*>(grads), sizeof(PdsMatchingParams)};
    M.blocks_params = params->blocks;
    M.blocks_grads = grads->blocks;
    M.blocks_count = 2 * params->residual_blocks;
    Ctx c{plan ? nullptr : (char*)workspace, 0, plan, stream};
    if (!plan) c.limit = g_backward_arena_bytes;
    backward_walk(c, tape, M, dhat, written);
    if (c.err) return c.err;
    if (!written[0]) return set_error(-1, "matching_bwd: no gradient reached x0");
    // ---- layer 0 ----------------------------------------------------------------------------------------------------
    const int F = params->features;
    const size_t wn = (size_t)F * F * 9;
    const Geom g1{batch, F, 1, h, w + 1};
    const size_t n1 = g1.numel();
    float* gy_a = c.get<float>(n1);
    float* gy_gs = c.get<float>(n1);
    float* gy_g = c.get<float>(n1);
    float* gy_g2 = c.get<float>(n1);
    float* lp = c.get<float>(n1);   // the descriptors behind one zero column, as the forward convolved them
    float* rp = c.get<float>(n1);
    float* dwl = c.get<float>(wn);
    float* dws = c.get<float>(wn);
    float* dwg = c.get<float>(wn);
    float* wscratch = c.get<float>(wgrad2d_mfma_scratch_floats(g1, g1));
    double* bias_scratch = c.get<double>((size_t)channel_sum_splits(g1) * F);
    float* wf = c.get<float>(3 * wn);
    float* dxa = c.get<float>(n1);
    float* dxg = c.get<float>(n1);
    float* dxg2 = c.get<float>(n1);
    if (!wgrad2d_mfma_supported(0, 1, 1, no_src(), g1, g1)) return set_error(-1, "matching_bwd: unsupported feature width %d", F);
    if (!c.plan) {
        const size_t rows = (size_t)batch * F * h;
        c.run(launch_l0_combine_bwd(dhat[0], gy_a, gy_gs, gy_g, gy_g2, batch, F, h, w, d_begin, d_count, c.s));
        c.run(launch_pad_left1(left, lp, rows, w, c.s));
        c.run(launch_pad_left1(right, rp, rows, w, c.s));
        // parameters: the bias belongs to the left term; the right half takes the G + G2 gradient for its dx <= 0 taps
        // and the G gradient alone for dx = +1 (G2 = conv_R without those taps)
        c.run(launch_channel_sum(gy_a, g1, const_cast<float*>(grads->first.bias), 0, bias_scratch, c.s));
        c.run(launch_wgrad2d_mfma(plain_src(lp), no_src(), plain_src(gy_a), dwl, g1, g1, 0, wscratch, c.s));
        c.run(launch_wgrad2d_mfma(plain_src(rp), no_src(), plain_src(gy_gs), dws, g1, g1, 0, wscratch, c.s));
        c.run(launch_wgrad2d_mfma(plain_src(rp), no_src(), plain_src(gy_g), dwg, g1, g1, 0, wscratch, c.s));
        c.run(launch_first_weight_grads(dwl, dws, dwg, const_cast<float*>(grads->first.weight), F, F, c.s));
        for (int k = 0; k < 3; ++k) c.run(launch_flip_weights(l0.w3 + k * wn, wf + k * wn, F, F, 9, c.s));
    }
    // descriptors: dx = conv(dz, flipped weights) on the forward kernels, one single-plane launch per term
    const float* dz3[3] = {gy_a, gy_g, gy_g2};
    float* dx3[3] = {dxa, dxg, dxg2};
    for (int k = 0; k < 3; ++k) {
        PdsConvBlockParams pf{wf + k * wn, nullptr, nullptr, nullptr};
        conv_block(c, plain_src(dz3[k]), no_src(), g1, pf, F, 1, 1, 0, dx3[k]);
    }
    if (!c.plan) {
        const size_t rows = (size_t)batch * F * h;
        c.run(launch_crop_left1_add(dxa, nullptr, grad_left, rows, w, c.s));
        c.run(launch_crop_left1_add(dxg, dxg2, grad_right, rows, w, c.s));
    }
    if (bytes) *bytes = c.off;
    return c.err;
}

size_t pds_matching_operation_bwd_workspace_bytes(const PdsMatchingParams* params, int n, int h, int w) {
    if (check_matching_params(params)) return 0;
    size_t bytes = 0;
    if (operation_backward(true, &bytes, params, params, nullptr, nullptr, nullptr, n, h, w, nullptr, nullptr, nullptr))
        return 0;
    return bytes + 256;
}

int pds_matching_operation_bwd(const PdsMatchingParams* params, const PdsMatchingParams* grads,
                               const float* concatenated, const float* grad_signature, float* grad_concatenated, int n,
                               int h, int w, void* fwd_workspace, size_t fwd_workspace_bytes, void* workspace,
                               size_t workspace_bytes, pds_stream_t stream) {
    if (int rc = check_matching_params(params)) return rc;
    if (int rc = check_matching_params(grads)) return rc;
    PDS_REQUIRE(concatenated && grad_signature && grad_concatenated && fwd_workspace && workspace,
                "matching_operation_bwd: null pointer");
    PDS_REQUIRE(fwd_workspace_bytes >= pds_matching_operation_workspace_bytes(params, n, h, w),
                "matching_operation_bwd: forward workspace too small");
    const size_t need = pds_matching_operation_bwd_workspace_bytes(params, n, h, w);
    PDS_REQUIRE(workspace_bytes >= need, "matching_operation_bwd: workspace too small (%zu < %zu)", workspace_bytes,
                need);
    ArenaLimit limit(workspace_bytes);
    return operation_backward(false, nullptr, params, grads, concatenated, grad_signature, grad_concatenated, n, h, w,
                              fwd_workspace, workspace, (hipStream_t)stream);
}

static int block_backward(bool plan, size_t* bytes, bool expansion, const PdsConvBlockParams* pp,
                          const PdsConvBlockParams* gg, const float* x, const float* shortcut,
                          const float* grad_out0, const float* grad_out1, float* grad_x, float* grad_shortcut,
                          const Geom& g, void* fwd_workspace, void* workspace, hipStream_t stream) {
    Tape tape;
    Ctx re{plan ? nullptr : (char*)fwd_workspace, 0, true, stream};
    re.tape = &tape;
    int id0 = -1, id1 = -1;
    if (expansion) {
        expansion_pipeline(re, pp, x, shortcut, const_cast<float*>(grad_out0), g);
        id0 = (int)tape.tensors.size() - 1;
    } else {
        contraction_pipeline(re, pp, x, const_cast<float*>(grad_out0), const_cast<float*>(grad_out1), g, &id0, &id1);
    }
    if (re.err) return re.err;
    std::vector<float*> dhat(tape.tensors.size(), nullptr);
    std::vector<char> written(tape.tensors.size(), 0);
    float* mark = reinterpret_cast<float*>(8);
    dhat[0] = plan ? mark : grad_x;
    if (expansion) dhat[1] = plan ? mark : grad_shortcut;
    dhat[id0] = plan ? mark : const_cast<float*>(grad_out0);
    written[id0] = 1;
    if (id1 >= 0) {
        dhat[id1] = plan ? mark : const_cast<float*>(grad_out1);
        written[id1] = 1;
    }
    GradMap M{reinterpret_cast<const char*>(pp), reinterpret_cast<const char*>(gg), 2 * sizeof(PdsConvBlockParams)};
    Ctx c{plan ? nullptr : (char*)workspace, 0, plan, stream};
    if (!plan) c.limit = g_backward_arena_bytes;
    backward_walk(c, tape, M, dhat, written);
    if (bytes) *bytes = c.off;
    return c.err;
}

size_t pds_contraction_block_bwd_workspace_bytes(int batch, int c_, int d, int h, int w) {
    size_t bytes = 0;
    if (block_backward(true, &bytes, false, kDummyBlocks, kDummyBlocks, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr, Geom{batch, c_, d, h, w}, nullptr, nullptr, nullptr))
        return 0;
    return bytes + 256;
}

int pds_contraction_block_bwd(const PdsConvBlockParams* downsampling, const PdsConvBlockParams* smoothing,
                              const PdsConvBlockParams* grad_downsampling, const PdsConvBlockParams* grad_smoothing,
                              const float* x, const float* grad_down, const float* grad_smooth, float* grad_x,
                              int batch, int c_, int d, int h, int w, void* fwd_workspace, size_t fwd_workspace_bytes,
                              void* workspace, size_t workspace_bytes, pds_stream_t stream) {
    PDS_REQUIRE(downsampling && smoothing && grad_downsampling && grad_smoothing && x && grad_down && grad_smooth &&
                    grad_x && fwd_workspace && workspace,
                "contraction_bwd: null pointer");
    PDS_REQUIRE(batch > 0 && c_ > 0 && d > 0 && h > 0 && w > 0, "contraction_bwd: bad shape");
    PDS_REQUIRE(fwd_workspace_bytes >= pds_contraction_block_workspace_bytes(batch, c_, d, h, w),
                "contraction_bwd: forward workspace too small");
    PDS_REQUIRE(workspace_bytes >= pds_contraction_block_bwd_workspace_bytes(batch, c_, d, h, w),
                "contraction_bwd: workspace too small");
    const PdsConvBlockParams pp[2] = {*downsampling, *smoothing};
    const PdsConvBlockParams gg[2] = {*grad_downsampling, *grad_smoothing};
    ArenaLimit limit(workspace_bytes);
    return block_backward(false, nullptr, false, pp, gg, x, nullptr, grad_down, grad_smooth, grad_x, nullptr,
                          Geom{batch, c_, d, h, w}, fwd_workspace, workspace, (hipStream_t)stream);
}

size_t pds_expansion_block_bwd_workspace_bytes(int batch, int c_, int d, int h, int w) {
    size_t bytes = 0;
    if (block_backward(true, &bytes, true, kDummyBlocks, kDummyBlocks, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr, Geom{batch, c_, d, h, w}, nullptr, nullptr, nullptr))
        return 0;
    return bytes + 256;
}

int pds_expansion_block_bwd(const PdsConvBlockParams* upsampling, const PdsConvBlockParams* smoothing,
                            const PdsConvBlockParams* grad_upsampling, const PdsConvBlockParams* grad_smoothing,
                            const float* x, const float* shortcut, const float* grad_out, float* grad_x,
                            float* grad_shortcut, int batch, int c_, int d, int h, int w, void* fwd_workspace,
                            size_t fwd_workspace_bytes, void* workspace, size_t workspace_bytes, pds_stream_t stream) {
    PDS_REQUIRE(upsampling && smoothing && grad_upsampling && grad_smoothing && x && shortcut && grad_out && grad_x &&
                    grad_shortcut && fwd_workspace && workspace,
                "expansion_bwd: null pointer");
    PDS_REQUIRE(batch > 0 && c_ >= 2 && c_ % 2 == 0 && d > 0 && h > 0 && w > 0, "expansion_bwd: bad shape");
    PDS_REQUIRE(fwd_workspace_bytes >= pds_expansion_block_workspace_bytes(batch, c_, d, h, w),
                "expansion_bwd: forward workspace too small");
    PDS_REQUIRE(workspace_bytes >= pds_expansion_block_bwd_workspace_bytes(batch, c_, d, h, w),
                "expansion_bwd: workspace too small");
    const PdsConvBlockParams pp[2] = {*upsampling, *smoothing};
    const PdsConvBlockParams gg[2] = {*grad_upsampling, *grad_smoothing};
    ArenaLimit limit(workspace_bytes);
    return block_backward(false, nullptr, true, pp, gg, x, shortcut, grad_out, nullptr, grad_x, grad_shortcut,
                          Geom{batch, c_, d, h, w}, fwd_workspace, workspace, (hipStream_t)stream);
}

// ---- Embedding ------------------------------------------------------------------------------------------
static int check_embedding(const PdsEmbeddingParams* P, int batch, int h, int w, int top, int left) {
    PDS_REQUIRE(P, "embedding: null params");
    PDS_REQUIRE(P->input_features > 0 && P->features > 0 && P->shortcut_features > 0 && P->residual_blocks >= 0,
                "embedding: bad feature counts");
    PDS_REQUIRE(P->residual_blocks == 0 || P->blocks, "embedding: residual block parameters missing");
    PDS_REQUIRE(batch > 0 && h > 0 && w > 0 && top >= 0 && left >= 0, "embedding: bad shape");
    return 0;
}

static PdsEmbeddingParams plan_embedding_params(const PdsEmbeddingParams* P, std::vector<PdsConvBlockParams>& blocks) {
    // workspace planning never dereferences parameter pointers, but gamma decides whether a layer normalises
    PdsEmbeddingParams q{};
    q.input_features = P->input_features;
    q.features = P->features;
    q.shortcut_features = P->shortcut_features;
    q.residual_blocks = P->residual_blocks;
    const float* const mark = reinterpret_cast<const float*>(8);  // never dereferenced
    const PdsConvBlockParams normed{mark, mark, mark, mark};
    q.downsampling[0] = q.downsampling[1] = q.shortcut = normed;
    blocks.assign((size_t)2 * P->residual_blocks + 1, normed);
    q.blocks = blocks.data();
    return q;
}

size_t pds_embedding_workspace_bytes(const PdsEmbeddingParams* params, int batch, int h, int w, int pad_top,
                                     int pad_left) {
    if (check_embedding(params, batch, h, w, pad_top, pad_left)) return 0;
    std::vector<PdsConvBlockParams> blocks;
    const PdsEmbeddingParams q = plan_embedding_params(params, blocks);
    Ctx c{nullptr, 0, true, nullptr};
    embedding_pipeline(c, q, nullptr, nullptr, nullptr, batch, h, w, pad_top, pad_left);
    return c.off + 256;
}

static int check_embedding_blocks(const PdsEmbeddingParams* P) {
    if (int rc = check_block(P->downsampling[0], true, "embedding._embedding_modules.1")) return rc;
    if (int rc = check_block(P->downsampling[1], true, "embedding._embedding_modules.2")) return rc;
    for (int i = 0; i < 2 * P->residual_blocks; ++i)
        if (int rc = check_block(P->blocks[i], true, "embedding residual block")) return rc;
    return check_block(P->shortcut, true, "embedding._shortcut");
}

int pds_embedding_fwd(const PdsEmbeddingParams* params, const float* image, float* descriptor, float* shortcut,
                      int batch, int h, int w, int pad_top, int pad_left, void* workspace, size_t workspace_bytes,
                      int weights_resident, pds_stream_t stream) {
    if (int rc = check_embedding(params, batch, h, w, pad_top, pad_left)) return rc;
    PDS_REQUIRE(image && descriptor && shortcut && workspace, "embedding: null pointer");
    if (int rc = check_embedding_blocks(params)) return rc;
    const size_t need = pds_embedding_workspace_bytes(params, batch, h, w, pad_top, pad_left);
    PDS_REQUIRE(workspace_bytes >= need, "embedding: workspace too small (%zu < %zu)", workspace_bytes, need);
    return run_with_batched_packing(workspace, (hipStream_t)stream, [&](Ctx& c) {
        embedding_pipeline(c, *params, image, descriptor, shortcut, batch, h, w, pad_top, pad_left);
    }, weights_resident != 0);
}

static int embedding_backward(bool plan, size_t* bytes, const PdsEmbeddingParams* params, const PdsEmbeddingParams* grads,
                              const float* image, const float* descriptor, float* grad_descriptor,
                              const float* grad_shortcut, float* grad_image, int batch, int h, int w, int top, int left,
                              void* fwd_workspace, void* workspace, hipStream_t stream) {
    Tape tape;
    ImageHead head;
    head.want_grad = grad_image != nullptr;   // (a planning walk passes a non-null mark)
    Ctx re{plan ? nullptr : (char*)fwd_workspace, 0, true, stream};
    re.tape = &tape;
    int id_d = -1, id_s = -1;
    // the descriptor feeds the shortcut block, so its forward values are needed; the shortcut output is not
    embedding_pipeline(re, *params, image, const_cast<float*>(descriptor), const_cast<float*>(grad_shortcut), batch, h,
                       w, top, left, &id_d, &id_s, &head);
    if (re.err) return re.err;
    std::vector<float*> dhat(tape.tensors.size(), nullptr);
    std::vector<char> written(tape.tensors.size(), 0);
    float* mark = reinterpret_cast<float*>(8);
    dhat[id_d] = plan ? mark : grad_descriptor;   // accumulated in place: the shortcut branch adds to it
    dhat[id_s] = plan ? mark : const_cast<float*>(grad_shortcut);
    written[id_d] = written[id_s] = 1;
    GradMap M{reinterpret_cast<const char*>(params), reinterpret_cast<const char*>(grads), sizeof(PdsEmbeddingParams)};
    M.blocks_params = params->blocks;
    M.blocks_grads = grads->blocks;
    M.blocks_count = 2 * params->residual_blocks;
    Ctx c{plan ? nullptr : (char*)workspace, 0, plan, stream};
    if (!plan) c.limit = g_backward_arena_bytes;
    backward_walk(c, tape, M, dhat, written);
    if (head.want_grad && !c.err) {
        // embedding.py:32 under autograd: depth-to-space + the parameter-free InstanceNorm2d of the padded image
        if (!written[head.id]) return set_error(-1, "embedding_bwd: no gradient reached the image head");
        if (!plan)
            c.run(launch_image_grad(dhat[head.id], image, head.scale, head.shift, batch, params->input_features, h, w,
                                    top, left, grad_image, stream));
    }
    if (bytes) *bytes = c.off;
    return c.err;
}

size_t pds_embedding_bwd_workspace_bytes(const PdsEmbeddingParams* params, int batch, int h, int w, int pad_top,
                                         int pad_left) {
    if (check_embedding(params, batch, h, w, pad_top, pad_left)) return 0;
    std::vector<PdsConvBlockParams> blocks, gblocks;
    const PdsEmbeddingParams q = plan_embedding_params(params, blocks);
    const PdsEmbeddingParams gq = plan_embedding_params(params, gblocks);
    size_t bytes = 0;
    if (embedding_backward(true, &bytes, &q, &gq, nullptr, nullptr, nullptr, nullptr, nullptr, batch, h, w, pad_top,
                           pad_left, nullptr, nullptr, nullptr))
        return 0;
    return bytes + 256;
}

size_t pds_embedding_image_bwd_workspace_bytes(const PdsEmbeddingParams* params, int batch, int h, int w, int pad_top,
                                               int pad_left) {
    if (check_embedding(params, batch, h, w, pad_top, pad_left)) return 0;
    std::vector<PdsConvBlockParams> blocks, gblocks;
    const PdsEmbeddingParams q = plan_embedding_params(params, blocks);
    const PdsEmbeddingParams gq = plan_embedding_params(params, gblocks);
    size_t bytes = 0;
    if (embedding_backward(true, &bytes, &q, &gq, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<float*>(8), batch,
                           h, w, pad_top, pad_left, nullptr, nullptr, nullptr))
        return 0;
    return bytes + 256;
}

int pds_embedding_bwd(const PdsEmbeddingParams* params, const PdsEmbeddingParams* grads, const float* image,
                      const float* descriptor, float* grad_descriptor, const float* grad_shortcut, int batch, int h,
                      int w, int pad_top, int pad_left, void* fwd_workspace, size_t fwd_workspace_bytes, void* workspace,
                      size_t workspace_bytes, pds_stream_t stream) {
    if (int rc = check_embedding(params, batch, h, w, pad_top, pad_left)) return rc;
    PDS_REQUIRE(grads && image && descriptor && grad_descriptor && grad_shortcut && fwd_workspace && workspace,
                "embedding_bwd: null pointer");
    PDS_REQUIRE(params->residual_blocks == 0 || grads->blocks, "embedding_bwd: gradient blocks missing");
    if (int rc = check_embedding_blocks(params)) return rc;
    PDS_REQUIRE(fwd_workspace_bytes >= pds_embedding_workspace_bytes(params, batch, h, w, pad_top, pad_left),
                "embedding_bwd: forward workspace too small");
    PDS_REQUIRE(workspace_bytes >= pds_embedding_bwd_workspace_bytes(params, batch, h, w, pad_top, pad_left),
                "embedding_bwd: workspace too small");
    ArenaLimit limit(workspace_bytes);
    return embedding_backward(false, nullptr, params, grads, image, descriptor, grad_descriptor, grad_shortcut, nullptr,
                              batch, h, w, pad_top, pad_left, fwd_workspace, workspace, (hipStream_t)stream);
}

int pds_embedding_image_bwd(const PdsEmbeddingParams* params, const PdsEmbeddingParams* grads, const float* image,
                            const float* descriptor, float* grad_descriptor, const float* grad_shortcut,
                            float* grad_image, int batch, int h, int w, int pad_top, int pad_left, void* fwd_workspace,
                            size_t fwd_workspace_bytes, void* workspace, size_t workspace_bytes, pds_stream_t stream) {
    if (int rc = check_embedding(params, batch, h, w, pad_top, pad_left)) return rc;
    PDS_REQUIRE(grads && image && descriptor && grad_descriptor && grad_shortcut && grad_image && fwd_workspace &&
                    workspace,
                "embedding_image_bwd: null pointer");
    PDS_REQUIRE(params->residual_blocks == 0 || grads->blocks, "embedding_image_bwd: gradient blocks missing");
    if (int rc = check_embedding_blocks(params)) return rc;
    PDS_REQUIRE(fwd_workspace_bytes >= pds_embedding_workspace_bytes(params, batch, h, w, pad_top, pad_left),
                "embedding_image_bwd: forward workspace too small");
    PDS_REQUIRE(workspace_bytes >= pds_embedding_image_bwd_workspace_bytes(params, batch, h, w, pad_top, pad_left),
                "embedding_image_bwd: workspace too small");
    ArenaLimit limit(workspace_bytes);
    return embedding_backward(false, nullptr, params, grads, image, descriptor, grad_descriptor, grad_shortcut,
                              grad_image, batch, h, w, pad_top, pad_left, fwd_workspace, workspace, (hipStream_t)stream);
}

// ---- evaluation metrics (errors.py:9-74) ---------------------------------------------------------------
size_t pds_disparity_errors_workspace_bytes(size_t count) {
    return disparity_errors_partial_doubles(count) * sizeof(double) + 256;
}

int pds_disparity_errors_fwd(const float* estimated, const float* ground_truth, size_t count, float n,
                             float* pixelwise_absolute_error, float* pixelwise_n_pixels_error, double* stats,
                             void* workspace, size_t workspace_bytes, pds_stream_t stream) {
    PDS_REQUIRE(estimated && ground_truth && stats && workspace, "disparity_errors: null pointer");
    PDS_REQUIRE(count > 0, "disparity_errors: empty input");
    PDS_REQUIRE(workspace_bytes >= pds_disparity_errors_workspace_bytes(count), "disparity_errors: workspace too small");
    return launch_disparity_errors(estimated, ground_truth, count, n, pixelwise_absolute_error,
                                   pixelwise_n_pixels_error, stats, reinterpret_cast<double*>(workspace),
                                   (hipStream_t)stream);
}

size_t pds_subpixel_cross_entropy_workspace_bytes(int n, int h, int w) {
    return sce_partial_doubles((size_t)n * h * w) * sizeof(double) + 256;
}

int pds_subpixel_cross_entropy_fwd(const float* similarities, const float* ground_truth, const float* weights,
                                   float* loss, float* lse, float* stats, int n, int planes, int h, int w,
                                   float diversity, int disparity_step, void* workspace, size_t workspace_bytes,
                                   pds_stream_t stream) {
    PDS_REQUIRE(similarities && ground_truth && loss && lse && stats && workspace, "subpixel_cross_entropy: null pointer");
    PDS_REQUIRE(n > 0 && planes > 0 && h > 0 && w > 0, "subpixel_cross_entropy: bad shape");
    PDS_REQUIRE(diversity > 0.f && disparity_step >= 1, "subpixel_cross_entropy: bad diversity / step");
    PDS_REQUIRE(workspace_bytes >= pds_subpixel_cross_entropy_workspace_bytes(n, h, w),
                "subpixel_cross_entropy: workspace too small");
    return launch_sce_fwd(similarities, ground_truth, weights, loss, lse, stats, (double*)workspace, n, planes, h, w,
                          diversity, disparity_step, (hipStream_t)stream);
}

int pds_subpixel_cross_entropy_bwd(const float* similarities, const float* ground_truth, const float* weights,
                                   const float* lse, const float* stats, const float* grad_loss,
                                   float* grad_similarities, int n, int planes, int h, int w, float diversity,
                                   int disparity_step, pds_stream_t stream) {
    PDS_REQUIRE(similarities && ground_truth && lse && stats && grad_loss && grad_similarities,
                "subpixel_cross_entropy_bwd: null pointer");
    PDS_REQUIRE(n > 0 && planes > 0 && h > 0 && w > 0, "subpixel_cross_entropy_bwd: bad shape");
    return launch_sce_bwd(similarities, ground_truth, weights, lse, stats, grad_loss, grad_similarities, n, planes, h,
                          w, diversity, disparity_step, (hipStream_t)stream);
}

int pds_subpixel_cross_entropy_weights_bwd(const float* similarities, const float* ground_truth, const float* lse,
                                           const float* stats, const float* grad_loss, float* grad_weights, int n,
                                           int planes, int h, int w, float diversity, int disparity_step,
                                           pds_stream_t stream) {
    PDS_REQUIRE(similarities && ground_truth && lse && stats && grad_loss && grad_weights,
                "subpixel_cross_entropy_weights_bwd: null pointer");
    PDS_REQUIRE(n > 0 && planes > 0 && h > 0 && w > 0, "subpixel_cross_entropy_weights_bwd: bad shape");
    return launch_sce_weights_bwd(similarities, ground_truth, lse, stats, grad_loss, grad_weights, n, planes, h, w,
                                  diversity, disparity_step, (hipStream_t)stream);
}

int pds_shift_concat_bwd(const float* grad_out, float* grad_left, float* grad_right, int batch, int channels, int h,
                         int w, int d_begin, int d_count, pds_stream_t stream) {
    PDS_REQUIRE(grad_out && grad_left && grad_right, "shift_concat_bwd: null pointer");
    PDS_REQUIRE(batch > 0 && channels > 0 && h > 0 && w > 0 && d_begin >= 0 && d_count > 0,
                "shift_concat_bwd: bad shape");
    return launch_shift_concat_bwd(grad_out, grad_left, grad_right, batch, channels, h, w, d_begin, d_count,
                                   (hipStream_t)stream);
}

}  // extern "C"
