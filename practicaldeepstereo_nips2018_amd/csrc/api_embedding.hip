// Embedding: module walk and C ABI entry points (reference embedding.py:11-65, size_adapter.py:29-43).
#include "api_internal.hpp"

namespace pds {

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

}  // namespace pds

using namespace pds;

extern "C" {

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

}  // extern "C"
