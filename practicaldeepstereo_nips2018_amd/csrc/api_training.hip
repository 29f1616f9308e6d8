// The reverse walk over a backward tape (include/pds_hip.h: every pds_*_bwd entry point ends here).
#include "api_internal.hpp"

namespace pds {

thread_local size_t g_backward_arena_bytes = ~(size_t)0;

// ====================================================================================================
// Backward: reverse walk over a tape.
// ====================================================================================================
void backward_walk(Ctx& c, const Tape& T, const GradMap& M, std::vector<float*>& dhat,
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
