// Regularization / ContractionBlock3d / ExpansionBlock3d: module walks and C ABI entry points (reference
// regularization.py:11-126).
#include "api_internal.hpp"

namespace pds {

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

}  // namespace pds

using namespace pds;

extern "C" {

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

}  // extern "C"
