// Matching / MatchingOperation: module walks and C ABI entry points (reference matching.py:12-112).
#include "api_internal.hpp"

namespace pds {

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
    // round trip of t1 through HBM is gone.  Bit-identical (l1_combine 109 -> 63 us without its stores, + 20 us for the re-layout,
    // + 11 us on the launch).  Round 5 measured it neutral with 20-step timed regions (410 vs 408.5 pairs/s); with the 400-step
    // regions of round 6 it is a small, repeatable gain -- 435.7 / 436.2 against 433.6 / 432.3 pairs/s and 2.603 / 2.609 against
    // 2.627 / 2.635 ms per sequential pair, interleaved on one box -- so level 2 is the default now
    const int cb8_level = [&]() {
        static const int level = []() {   // PDS_MATCHING_CB8=0: planar NCDHW everywhere (A/B, tests)
            const char* e = debug_switch("PDS_MATCHING_CB8");
            return e ? atoi(e) : 2;   // (round 6: level 2 by default, see the note above)
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

}  // namespace pds

using namespace pds;

extern "C" {

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

}  // extern "C"
