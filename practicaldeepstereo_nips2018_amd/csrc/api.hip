// C ABI of libpds_hip.so (include/pds_hip.h): argument checks, workspace carving and the
// per-module launch sequences.  No allocation, no synchronisation: everything is enqueued on the
// caller's stream into caller-owned memory.
#include "api_internal.hpp"

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


// a plain tensor whose producer writes `records` per-workgroup maxima of |value|
void carve_amax(Ctx& c, DT& t, int records) {
    t.bound = c.get<float>((size_t)records);
    t.bound_n = records;
    t.bounded = true;
}

// a caller-provided plain tensor as a source, registered on the tape when one is being recorded
Src external_src(Ctx& c, const float* p, const Geom& g, int bcast_d, bool needs_grad) {
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

void tape_layer(Ctx& c, int type, int kd, int stride, const Src& a, const Src& b, const Geom& in_g, DT& o,
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

Geom conv_out_geom(const Geom& in, int cout, int kd, int stride) {
    Geom o = in;
    o.c = cout;
    if (stride == 2) {
        if (kd == 3) o.d = (in.d + 1) / 2;
        o.h = (in.h + 1) / 2;
        o.w = (in.w + 1) / 2;
    }
    return o;
}


// conv (+ LeakyReLU + deferred InstanceNorm when P.gamma) ; out_raw may be caller-provided
DT conv_block(Ctx& c, const Src& a, const Src& b, const Geom& in, const PdsConvBlockParams& P, int cout, int kd, int stride,
              int per_plane, float* out_raw, bool allow_mfma, float* scale_out, float* shift_out, const ConvExtra* extra) {
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
    else if (allow_mfma && conv3d_nx_supported(L)) kind = 10;  // 16-channel quarter-resolution layers: fp16-split operands
    else if (allow_mfma && conv3d_mfma_supported(L)) kind = 3;
    if (kind == 2)
        L.packed = c.get<float>(conv2d_mfma_packed_floats(in.c, cout) * (L.plane_weight_sets > 0 ? L.plane_weight_sets : 1));
    if (kind == 4)
        L.packed = c.get<float>(conv2d_wino_packed_floats(in.c, cout) * (L.plane_weight_sets > 0 ? L.plane_weight_sets : 1));
    if (kind == 3) L.packed = c.get<float>(conv3d_mfma_packed_floats(o.g, in.c, stride));
    if (kind == 7) L.packed = c.get<float>(conv3d_ks_packed_floats(in.c, cout, 27));
    if (kind == 10) L.packed = c.get<float>(conv3d_nx_packed_floats(in.c, cout));
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
                                                             : kind == 10 ? launch_conv3d_nx(L, c.s)
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
                                    : kind == 7 ? conv3d_ks_tiles(o.g)
                                    : kind == 10 ? conv3d_nx_tiles(o.g) : conv_direct_tiles_for(o.g, stride);
        const bool volume_records = kind == 3 || kind == 6 || kind == 7 || kind == 10;   // [(n, c)][record] instead of [(n, c, d)][tile]
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

DT deconv_block(Ctx& c, const Src& a, const Src& b, const Geom& in, const PdsConvBlockParams& P, int cout, int kd,
                float* out_raw) {
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


int check_block(const PdsConvBlockParams& b, bool norm, const char* name) {
    PDS_REQUIRE(b.weight && b.bias, "%s: null weight/bias", name);
    if (norm) PDS_REQUIRE(b.gamma && b.beta, "%s: null InstanceNorm affine", name);
    return 0;
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
