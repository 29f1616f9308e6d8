/*
 * pds_hip.h -- C ABI of libpds_hip.so: the MI355X (gfx950) implementation of the
 * Practical Deep Stereo cost-volume hot path (Matching -> Regularization -> SubpixelMap).
 *
 * The reference has no FFI on this path: the seam is Python constructor injection
 * (reference practical_deep_stereo/network.py:17-24) and the modules are torch.nn code.
 * The entry points below are therefore what a ctypes/cffi binding of each reference
 * module's forward would call; every prototype cites the reference code it replaces.
 * INTEGRATION.md shows the reference-side stub.
 *
 * Conventions (all entry points):
 *  - plain C: raw device pointers, ints, one opaque stream handle (a hipStream_t).
 *    No torch types.  All tensors are contiguous fp32, NCHW / NCDHW like the reference.
 *  - the library never allocates, frees or synchronises: every buffer (inputs, outputs,
 *    workspace, packed weights) belongs to the caller; all work is enqueued on `stream`;
 *    calls are re-entrant and hipGraph-capturable.
 *  - return value 0 = enqueued; non-zero = error (negative: bad argument, positive:
 *    hipError_t).  pds_last_error() gives a thread-local message.  Nothing throws.
 *  - `weights_resident` (the forward entry points that re-lay weights out): pass 0 unless this
 *    very workspace was last used by the same entry point with the same shapes and the same
 *    parameter VALUES; then 1 skips the weight re-layout launches (the packed weights a module
 *    owns are immutable between optimizer steps).  The Python mirror tracks this
 *    through the parameters' version counters.
 *  - argument validation that the reference does in Python (the ValueErrors of
 *    estimator.py:34-41, network.py:28-31) stays in the Python mirror.
 */
#ifndef PDS_HIP_H
#define PDS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDS_ABI_VERSION 6

typedef void* pds_stream_t; /* hipStream_t */

int pds_abi_version(void);
const char* pds_last_error(void);

/* ABI v5.  Non-finite InstanceNorm statistics.  The reference propagates NaN / inf silently (network_blocks.py:47-85
 * have no checks); here every statistics kernel that finds a group whose mean or variance is not finite (a NaN / inf
 * in the caller's tensors or parameters, or an overflow) bumps a counter in host-mapped memory, so a caller can
 * tell "garbage in" from a number WITHOUT a device synchronisation in the hot path.  Returns the count reported by
 * kernels that have completed so far (all devices of the process; synchronise the stream first for an exact answer)
 * and resets it when `reset` is non-zero; -1 when the counter could not be set up. */
long long pds_nonfinite_statistics(int reset);

/* ABI v5.  Launch probe (measurement only; bench.py's roofline): times named kernels IN SITU -- inside whatever
 * sequence of launches the caller enqueues -- with a HIP event pair on the launch stream around each of them, instead
 * of a micro-benchmark of the isolated kernel.  pds_probe_begin arms the probe for launches whose name contains
 * `kernel` ("conv2d_x3", "conv2d_t8w"), at most `capacity` of them (<= 256; events are created on first use and
 * re-used); pds_probe_end disarms it, waits for the recorded events and writes the durations in launch order
 * (milliseconds) and the launch grids (workgroups) to ms[] / workgroups[] (either may be NULL); returns the number of
 * launches recorded, or a negative error code.  Not thread-safe and not for production paths: events between
 * launches serialise them. */
int pds_probe_begin(const char* kernel, int capacity);
int pds_probe_end(float* ms, int* workgroups, int capacity);

/* ABI v6.  Measurement only: the inner levels of the Regularization hourglass (regularization.py:22-26, 48-52: the
 * layers the K-split kernel serves) run as ONE persistent launch whose workgroups walk the layer list (conv3d_ks.hip:
 * conv3d_ks_chain_kernel).  Synchronises the device and writes, for every layer of the LAST such launch of this
 * process, the time at which it was complete (InstanceNorm folded) in ticks of the 100 MHz device clock since the
 * first ticket of the launch was drawn; returns the number of layers (0: no such launch yet) or a negative error code.
 * (The chain kernel never hangs the GPU: a workgroup whose producer layer does not report within 2 s stops waiting and
 * adds 2^20 to the counter behind pds_nonfinite_statistics -- the results of that launch are then garbage, and visible as such.) */
int pds_debug_chain_stamps(unsigned* ticks, int capacity);

/* ------------------------------------------------------------------------------------
 * Layer parameters in the reference's own (PyTorch) layouts.
 *   conv   weight [Cout, Cin, kD, kH, kW]  (Conv2d: kD == 1)   network_blocks.py:9-24
 *   deconv weight [Cin, Cout, kD, kH, kW]                      network_blocks.py:37-44, 75-85
 *   gamma/beta: InstanceNorm affine (NULL for a bare conv)     network_blocks.py:58,72,85
 * ---------------------------------------------------------------------------------- */
typedef struct PdsConvBlockParams {
    const float* weight;
    const float* bias;
    const float* gamma;
    const float* beta;
} PdsConvBlockParams;

/* ------------------------------------------------------------------------------------
 * SubpixelMap.__call__                      reference estimator.py:45-91
 * similarities [batch, planes, height, width] -> disparities [batch, height, width].
 * First-occurrence arg-max, taps j in range(-hw // step, hw // step + 1) (Python floor
 * division), invalid taps get probability 0, result = sum softmax * step * index.
 * ---------------------------------------------------------------------------------- */
int pds_subpixel_map_fwd(const float* similarities, float* disparities,
                         int batch, int planes, int height, int width,
                         int half_support_window, int disparity_step,
                         pds_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Matching.forward, generic-operation path  reference matching.py:12-13, 50-61
 * Builds cat([left, S_d(right)], dim=1) for disparities d_begin .. d_begin+d_count-1:
 * out [d_count, batch, 2*channels, h, w]; S_d(R)[x] = R[x-d] for x >= d else 0.
 * The Python mirror then applies the user's arbitrary `operation` per plane.
 * ---------------------------------------------------------------------------------- */
int pds_shift_concat_fwd(const float* left, const float* right, float* out,
                         int batch, int channels, int h, int w,
                         int d_begin, int d_count, pds_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Matching(maximum_disparity, MatchingOperation()).forward, fused fast path
 *                                           reference matching.py:34-63 + :66-112
 * left/right [batch, features, h, w] -> signatures [batch, sig, d_count, h, w] holding
 * disparities d_begin .. d_begin+d_count-1 (d_begin/d_count shard the disparity axis
 * across GPUs; the whole range is d_begin=0, d_count=maximum_disparity+1).
 * `first` is the bare conv 2*features->features, `blocks` holds 2*residual_blocks
 * conv+LeakyReLU+InstanceNorm2d blocks (ResidualBlock, network_blocks.py:134-144),
 * `last` the bare conv features->sig.  InstanceNorm statistics are per (batch, channel,
 * disparity plane), as in the reference's per-disparity calls.
 * ---------------------------------------------------------------------------------- */
typedef struct PdsMatchingParams {
    int features;            /* 64  */
    int signature_features;  /* 8   */
    int residual_blocks;     /* 2   */
    PdsConvBlockParams first;
    const PdsConvBlockParams* blocks; /* [2 * residual_blocks] */
    PdsConvBlockParams last;
} PdsMatchingParams;

size_t pds_matching_workspace_bytes(const PdsMatchingParams* params, int batch, int h, int w,
                                    int d_count);
int pds_matching_fwd(const PdsMatchingParams* params,
                     const float* left, const float* right, float* signatures,
                     int batch, int h, int w, int d_begin, int d_count,
                     void* workspace, size_t workspace_bytes, int weights_resident,
                     pds_stream_t stream);

/* MatchingOperation.forward on an already concatenated tensor [n, 2*features, h, w]
 * -> [n, sig, h, w]                          reference matching.py:97-112 */
size_t pds_matching_operation_workspace_bytes(const PdsMatchingParams* params, int n, int h, int w);
int pds_matching_operation_fwd(const PdsMatchingParams* params,
                               const float* concatenated, float* signature,
                               int n, int h, int w,
                               void* workspace, size_t workspace_bytes, pds_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Regularization.forward                    reference regularization.py:94-126
 * signatures [batch, F, D, h, w] + left shortcut [batch, F, h, w] -> cost
 * [batch, 2*D, 4*h, 4*w].  D, h, w must be multiples of 16 (four stride-2 levels).
 * ---------------------------------------------------------------------------------- */
typedef struct PdsRegularizationParams {
    int features;                        /* 8 */
    PdsConvBlockParams smoothing;        /* regularization.py:77-78 */
    PdsConvBlockParams contraction[4][2];/* [level][0=_downsampling_2x, 1=_smoothing]  :79-82 */
    PdsConvBlockParams expansion[4][2];  /* [level][0=_upsampling_2x,   1=_smoothing]  :83-86 */
    PdsConvBlockParams upsample_half;    /* :87-89 */
    PdsConvBlockParams upsample_full;    /* bare deconv (3,4,4)/(1,2,2), :90-92 */
} PdsRegularizationParams;

size_t pds_regularization_workspace_bytes(const PdsRegularizationParams* params,
                                          int batch, int d, int h, int w);
int pds_regularization_fwd(const PdsRegularizationParams* params,
                           const float* signatures, const float* left_shortcut, float* cost,
                           int batch, int d, int h, int w,
                           void* workspace, size_t workspace_bytes, int weights_resident,
                           pds_stream_t stream);

/* Eval-mode fusion of Regularization's last layer with SubpixelMap (network.py:50-51):
 * the full-resolution cost volume is never written.  SizeAdapter.unpad (size_adapter.py:45-52)
 * is folded into the store: disparities is the contiguous [batch, 4*h - crop_top,
 * 4*w - crop_left] image without the rows / columns SizeAdapter.pad added on top / left
 * (crop 0, 0: the padded size). */
int pds_regularization_subpixel_map_fwd(const PdsRegularizationParams* params,
                                        const float* signatures, const float* left_shortcut,
                                        float* disparities,
                                        int batch, int d, int h, int w,
                                        int half_support_window, int disparity_step,
                                        int crop_top, int crop_left,
                                        void* workspace, size_t workspace_bytes, int weights_resident,
                                        pds_stream_t stream);

/* ContractionBlock3d.forward                reference regularization.py:28-31
 * x [batch, C, D, H, W] -> down, smooth [batch, 2C, D/2, H/2, W/2] (ceil for odd sizes). */
size_t pds_contraction_block_workspace_bytes(int batch, int c, int d, int h, int w);
int pds_contraction_block_fwd(const PdsConvBlockParams* downsampling, const PdsConvBlockParams* smoothing,
                              const float* x, float* down, float* smooth,
                              int batch, int c, int d, int h, int w,
                              void* workspace, size_t workspace_bytes, pds_stream_t stream);

/* ExpansionBlock3d.forward                  reference regularization.py:54-57
 * x [batch, C, D, H, W], shortcut [batch, C/2, 2D, 2H, 2W] -> out like shortcut. */
size_t pds_expansion_block_workspace_bytes(int batch, int c, int d, int h, int w);
int pds_expansion_block_fwd(const PdsConvBlockParams* upsampling, const PdsConvBlockParams* smoothing,
                            const float* x, const float* shortcut, float* out,
                            int batch, int c, int d, int h, int w,
                            void* workspace, size_t workspace_bytes, pds_stream_t stream);

/* ------------------------------------------------------------------------------------
 * One convolution block of network_blocks.py:47-72 on a plain input tensor x [n, cin, d, h, w]:
 * raw = LeakyReLU(conv(x) + bias) [n, cout, d', h', w'] plus the folded InstanceNorm coefficients
 * (normalised = scale * raw + shift, scale/shift [n*cout] or [n*cout*d'] when per_plane).
 * kd = 1 is Conv2d applied to every d-plane (Matching), kd = 3 is Conv3d; stride 1 or 2.
 * With params->gamma == NULL the block is a bare convolution (scale/shift untouched).
 * This is the launch bench.py times for the roofline of the dominant kernel.
 * ---------------------------------------------------------------------------------- */
size_t pds_conv_block_workspace_bytes(int n, int cin, int cout, int d, int h, int w, int kd, int stride,
                                      int per_plane);
int pds_conv_block_fwd(const PdsConvBlockParams* params, const float* x, float* raw, float* scale,
                       float* shift, int n, int cin, int cout, int d, int h, int w, int kd, int stride,
                       int per_plane, void* workspace, size_t workspace_bytes, pds_stream_t stream);
/* ABI v3 (x_bound: v5).  The same block chained behind another one, as inside the modules (network_blocks.py:47-72:
 * Conv -> LeakyReLU -> InstanceNorm, then the next Conv): x is the producer's RAW output and the loader applies the
 * producer's folded InstanceNorm, x^ = x_scale * x + x_shift ([n*cin], or [n*cin*d] when x_per_plane).  Same workspace
 * size as pds_conv_block_fwd.  This is the form in which the 64 -> 64 layers of MatchingOperation (matching.py:85-88)
 * run in the hot path, and the launch bench.py times for the roofline of the dominant kernel.
 * x_bound: device pointer to ONE float that bounds |x^| (inside the modules in_finalize writes
 * max_c |gamma_c| sqrt(group size) + |beta_c|, which is rigorous), or NULL.  The fp16-split kernels scale their
 * operands by a power of two derived from it, so the bound may be loose by orders of magnitude but must hold; with NULL
 * nothing is assumed about the range and the range-safe forms run (three-way bf16 split / exact fp32). */
int pds_conv_block_chained_fwd(const PdsConvBlockParams* params, const float* x, const float* x_scale,
                               const float* x_shift, int x_per_plane, const float* x_bound, float* raw, float* scale,
                               float* shift, int n, int cin, int cout, int d, int h, int w, int kd, int stride,
                               int per_plane, void* workspace, size_t workspace_bytes, pds_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Backward (training: loss.backward() in reference pds_trainer.py:40-46 reaches these modules through
 * autograd).  Each *_bwd re-derives the forward's intermediates from the forward workspace, which the
 * caller must have kept untouched since the matching *_fwd call (the arena layout is deterministic; the
 * library keeps no state).  `grads` mirrors `params`: every pointer is the gradient buffer of that
 * parameter, written (not accumulated).  Input gradients are written too.
 * ---------------------------------------------------------------------------------- */
size_t pds_regularization_bwd_workspace_bytes(const PdsRegularizationParams* params, int batch, int d, int h, int w);
int pds_regularization_bwd(const PdsRegularizationParams* params, const PdsRegularizationParams* grads,
                           const float* signatures, const float* left_shortcut, const float* grad_cost,
                           float* grad_signatures, float* grad_left_shortcut, int batch, int d, int h, int w,
                           void* fwd_workspace, size_t fwd_workspace_bytes, void* workspace, size_t workspace_bytes,
                           pds_stream_t stream);

size_t pds_matching_operation_bwd_workspace_bytes(const PdsMatchingParams* params, int n, int h, int w);
int pds_matching_operation_bwd(const PdsMatchingParams* params, const PdsMatchingParams* grads,
                               const float* concatenated, const float* grad_signature, float* grad_concatenated,
                               int n, int h, int w, void* fwd_workspace, size_t fwd_workspace_bytes,
                               void* workspace, size_t workspace_bytes, pds_stream_t stream);

/* ABI v5.  Matching(MatchingOperation) with gradients (reference matching.py:34-63 under autograd, driven by
 * pds_trainer.py:40-46).  pds_matching_train_fwd is pds_matching_fwd on the differentiable route: layer 0 keeps its
 * factorisation (conv_L(left) + shift_d(conv_R(right)): the [D', B, 128, h, w] concat of matching.py:50-62 never
 * exists), x0 is materialised and every layer output is kept in `workspace`, which the caller preserves until
 * pds_matching_bwd.  pds_matching_bwd walks back from grad_signatures [batch, 8, d_count, h, w] to the parameter
 * gradients (`grads` mirrors `params`, written) and to grad_left / grad_right [batch, features, h, w] (written); layer 0
 * is differentiated through its factorisation: one streaming reduction of d loss / d x0 over the disparity planes, then
 * single-plane convolution gradients.  With d_begin / d_count the gradients are the partial sums of that plane range. */
size_t pds_matching_train_workspace_bytes(const PdsMatchingParams* params, int batch, int h, int w, int d_count);
int pds_matching_train_fwd(const PdsMatchingParams* params, const float* left, const float* right, float* signatures,
                           int batch, int h, int w, int d_begin, int d_count, void* workspace, size_t workspace_bytes,
                           pds_stream_t stream);
size_t pds_matching_bwd_workspace_bytes(const PdsMatchingParams* params, int batch, int h, int w, int d_count);
int pds_matching_bwd(const PdsMatchingParams* params, const PdsMatchingParams* grads, const float* left,
                     const float* right, const float* grad_signatures, float* grad_left, float* grad_right, int batch,
                     int h, int w, int d_begin, int d_count, void* fwd_workspace, size_t fwd_workspace_bytes,
                     void* workspace, size_t workspace_bytes, pds_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Embedding                                 reference embedding.py:11-65 (producer of the path's inputs)
 *   image [batch, input_features, h, w] -> descriptor [batch, features, H4, W4], shortcut [batch, shortcut_features, H4, W4]
 *   H4 = ceil(ceil((h + pad_top) / 2) / 2), W4 likewise.  pad_top / pad_left are the zero rows / columns
 *   SizeAdapter.pad (size_adapter.py:29-43) would have prepended: they are applied virtually, the padded image is
 *   never materialised (the InstanceNorm2d of embedding.py:32 still sees them, as in the reference).
 *   `downsampling` = _embedding_modules.1 / .2 (k5 s2 blocks), `blocks` = 2 * residual_blocks conv blocks of
 *   _embedding_modules.3.., `shortcut` = _shortcut.
 * ---------------------------------------------------------------------------------- */
typedef struct PdsEmbeddingParams {
    int input_features;     /* 3  */
    int features;           /* 64 */
    int shortcut_features;  /* 8  */
    int residual_blocks;    /* 2  */
    PdsConvBlockParams downsampling[2];
    const PdsConvBlockParams* blocks; /* [2 * residual_blocks] */
    PdsConvBlockParams shortcut;
} PdsEmbeddingParams;

size_t pds_embedding_workspace_bytes(const PdsEmbeddingParams* params, int batch, int h, int w, int pad_top,
                                     int pad_left);
int pds_embedding_fwd(const PdsEmbeddingParams* params, const float* image, float* descriptor, float* shortcut,
                      int batch, int h, int w, int pad_top, int pad_left, void* workspace, size_t workspace_bytes,
                      int weights_resident, pds_stream_t stream);
/* backward (pds_trainer.py:40-46): needs the untouched forward workspace and the descriptor the forward call
 * returned; grad_descriptor is used as scratch (the shortcut branch's contribution is added to it in place) */
size_t pds_embedding_bwd_workspace_bytes(const PdsEmbeddingParams* params, int batch, int h, int w, int pad_top,
                                         int pad_left);
int pds_embedding_bwd(const PdsEmbeddingParams* params, const PdsEmbeddingParams* grads, const float* image,
                      const float* descriptor, float* grad_descriptor, const float* grad_shortcut, int batch, int h,
                      int w, int pad_top, int pad_left, void* fwd_workspace, size_t fwd_workspace_bytes, void* workspace,
                      size_t workspace_bytes, pds_stream_t stream);
/* ABI v4.  The same backward pass continued to the image (embedding.py:32,46-65 under autograd): grad_image
 * [batch, input_features, h, w] = d loss / d image through the first convolution and the parameter-free
 * InstanceNorm2d of the padded image (the virtual pad pixels take part in its statistics and receive no gradient). */
size_t pds_embedding_image_bwd_workspace_bytes(const PdsEmbeddingParams* params, int batch, int h, int w, int pad_top,
                                               int pad_left);
int pds_embedding_image_bwd(const PdsEmbeddingParams* params, const PdsEmbeddingParams* grads, const float* image,
                            const float* descriptor, float* grad_descriptor, const float* grad_shortcut,
                            float* grad_image, int batch, int h, int w, int pad_top, int pad_left, void* fwd_workspace,
                            size_t fwd_workspace_bytes, void* workspace, size_t workspace_bytes, pds_stream_t stream);

/* backward of the stand-alone blocks (regularization.py:28-31, 54-57 under autograd); grad_* param structs hold
 * the gradient buffers of the two conv blocks, written */
size_t pds_contraction_block_bwd_workspace_bytes(int batch, int c, int d, int h, int w);
int pds_contraction_block_bwd(const PdsConvBlockParams* downsampling, const PdsConvBlockParams* smoothing,
                              const PdsConvBlockParams* grad_downsampling, const PdsConvBlockParams* grad_smoothing,
                              const float* x, const float* grad_down, const float* grad_smooth, float* grad_x,
                              int batch, int c, int d, int h, int w, void* fwd_workspace, size_t fwd_workspace_bytes,
                              void* workspace, size_t workspace_bytes, pds_stream_t stream);
size_t pds_expansion_block_bwd_workspace_bytes(int batch, int c, int d, int h, int w);
int pds_expansion_block_bwd(const PdsConvBlockParams* upsampling, const PdsConvBlockParams* smoothing,
                            const PdsConvBlockParams* grad_upsampling, const PdsConvBlockParams* grad_smoothing,
                            const float* x, const float* shortcut, const float* grad_out, float* grad_x,
                            float* grad_shortcut, int batch, int c, int d, int h, int w, void* fwd_workspace,
                            size_t fwd_workspace_bytes, void* workspace, size_t workspace_bytes, pds_stream_t stream);

/* backward of pds_shift_concat_fwd: grad_out [d_count, batch, 2*channels, h, w] -> grad_left, grad_right
 * [batch, channels, h, w]   (reference matching.py:50-61 under autograd) */
int pds_shift_concat_bwd(const float* grad_out, float* grad_left, float* grad_right, int batch, int channels,
                         int h, int w, int d_begin, int d_count, pds_stream_t stream);

/* ------------------------------------------------------------------------------------
 * SubpixelCrossEntropy.forward and its gradient      reference loss.py:16-78
 * similarities [n, planes, h, w]; ground_truth [n, h, w] (inf = unknown); weights [n, h, w] or NULL.
 * fwd writes loss[1], lse[n*h*w] (log-sum-exp per pixel, kept for bwd) and stats[2] = {sum w*entropy,
 * denominator}; bwd writes d loss / d similarities scaled by the device scalar grad_loss[1];
 * weights_bwd (ABI v4) writes d loss / d weights [n, h, w] = grad_loss * (entropy - loss) / denominator at known
 * pixels, 0 elsewhere (loss.py:74-77 under autograd; only meaningful when fwd ran with weights).
 * ---------------------------------------------------------------------------------- */
size_t pds_subpixel_cross_entropy_workspace_bytes(int n, int h, int w);
int pds_subpixel_cross_entropy_fwd(const float* similarities, const float* ground_truth, const float* weights,
                                   float* loss, float* lse, float* stats, int n, int planes, int h, int w,
                                   float diversity, int disparity_step, void* workspace, size_t workspace_bytes,
                                   pds_stream_t stream);
int pds_subpixel_cross_entropy_bwd(const float* similarities, const float* ground_truth, const float* weights,
                                   const float* lse, const float* stats, const float* grad_loss,
                                   float* grad_similarities, int n, int planes, int h, int w, float diversity,
                                   int disparity_step, pds_stream_t stream);
int pds_subpixel_cross_entropy_weights_bwd(const float* similarities, const float* ground_truth, const float* lse,
                                           const float* stats, const float* grad_loss, float* grad_weights, int n,
                                           int planes, int h, int w, float diversity, int disparity_step,
                                           pds_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Evaluation metrics                         reference errors.py:9-74 (pds_trainer.py:48-58)
 *   estimated / ground_truth: `count` floats each (any shape, contiguous); unknown ground truth is +-inf.
 *   pixelwise_absolute_error[count]  = known ? |est - gt| : 0            (may be NULL)
 *   pixelwise_n_pixels_error[count]  = known && |est - gt| > n ? 1 : 0   (may be NULL)
 *   stats[3] (fp64) = { sum of absolute errors over known pixels, known pixels, pixels with error > n }
 *   => mean absolute error = stats[0] / stats[1], n-pixels error [%] = 100 * stats[2] / stats[1] (0 if no pixel known)
 * ---------------------------------------------------------------------------------- */
size_t pds_disparity_errors_workspace_bytes(size_t count);
int pds_disparity_errors_fwd(const float* estimated, const float* ground_truth, size_t count, float n,
                             float* pixelwise_absolute_error, float* pixelwise_n_pixels_error, double* stats,
                             void* workspace, size_t workspace_bytes, pds_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PDS_HIP_H */
