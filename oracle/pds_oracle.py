"""CPU oracle for the Practical Deep Stereo cost-volume hot path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the reported CPU baseline.  The product package
(``practicaldeepstereo_nips2018_amd``) never imports it and has no CPU fallback.

It is a functional PyTorch-CPU fp32 (or fp64, as arbiter) restatement of the
reference's arithmetic for

  * ``Matching`` / ``MatchingOperation``   reference practical_deep_stereo/matching.py:12-112
  * ``ContractionBlock3d`` / ``ExpansionBlock3d`` / ``Regularization``
                                           reference practical_deep_stereo/regularization.py:11-126
  * ``SubpixelMap``                        reference practical_deep_stereo/estimator.py:10-91
  * the layer factories those use          reference practical_deep_stereo/network_blocks.py:19-144
  * ``SubpixelCrossEntropy``               reference practical_deep_stereo/loss.py:16-78 (consumer in training)
  * ``Embedding``                          reference practical_deep_stereo/embedding.py:11-65 (producer of the path's
                                           inputs) and ``SizeAdapter.pad`` size_adapter.py:29-43

Parity pinning: ``tests/golden/make_golden.py`` (run in the build container,
where /root/reference is importable) checks every function below against the
reference modules on the reference's own known-answer tests
(test/test_matching.py:17-32, test/test_estimator.py:14-27) and on seeded
random tensors, then writes the committed fixtures under ``tests/golden``.
``tests/test_oracle_golden.py`` re-checks the oracle against those fixtures
without the reference being present.

Weights are taken from a flat ``dict`` whose keys are the reference's
state-dict names, so the same dict drives the oracle, the
reference (``load_state_dict``) and the HIP modules.
"""
import torch
import torch.nn.functional as F

LEAKY_SLOPE = 0.1  # network_blocks.py:57,71,84
IN_EPS = 1e-5      # torch.nn.InstanceNorm{2,3}d default, network_blocks.py:58,72,85


# --------------------------------------------------------------------------------------
# layer blocks (network_blocks.py)
# --------------------------------------------------------------------------------------
def _act_norm(x, gamma, beta):
    """LeakyReLU(0.1) then affine InstanceNorm with statistics of ``x`` itself
    (network_blocks.py:57-58, 71-72, 84-85: conv -> lrelu -> IN)."""
    x = F.leaky_relu(x, LEAKY_SLOPE)
    return F.instance_norm(x, weight=gamma, bias=beta, eps=IN_EPS)


def conv_block_2d(p, prefix, x):
    """convolutional_block_3x3, network_blocks.py:97-103 -> :47-58."""
    x = F.conv2d(x, p[prefix + '.0.weight'], p[prefix + '.0.bias'], padding=1)
    return _act_norm(x, p[prefix + '.2.weight'], p[prefix + '.2.bias'])


def conv_block_3d(p, prefix, x, stride=1):
    """convolutional_block_3x3x3[_stride_2], network_blocks.py:106-121 -> :61-72."""
    x = F.conv3d(x, p[prefix + '.0.weight'], p[prefix + '.0.bias'], stride=stride, padding=1)
    return _act_norm(x, p[prefix + '.2.weight'], p[prefix + '.2.bias'])


def deconv_block_3d(p, prefix, x):
    """transposed_convolutional_block_4x4x4_stride_2, network_blocks.py:124-131 -> :75-85."""
    x = F.conv_transpose3d(x, p[prefix + '.0.weight'], p[prefix + '.0.bias'], stride=2, padding=1)
    return _act_norm(x, p[prefix + '.2.weight'], p[prefix + '.2.bias'])


def conv_block_5x5_stride_2(p, prefix, x):
    """convolutional_block_5x5_stride_2, network_blocks.py:86-92 -> :47-58 (kernel 5, stride 2, padding 2)."""
    x = F.conv2d(x, p[prefix + '.0.weight'], p[prefix + '.0.bias'], stride=2, padding=2)
    return _act_norm(x, p[prefix + '.2.weight'], p[prefix + '.2.bias'])


def residual_block_2d(p, prefix, x):
    """ResidualBlock, network_blocks.py:134-144: convs(x) + x, no activation after the add."""
    y = conv_block_2d(p, prefix + '.convolutions.0', x)
    y = conv_block_2d(p, prefix + '.convolutions.1', y)
    return y + x


# --------------------------------------------------------------------------------------
# matching.py
# --------------------------------------------------------------------------------------
def shift_right(right, disparity):
    """S_d(R)[..., x] = R[..., x - d] for x >= d else 0 (matching.py:12-13, 50-51, 57-59)."""
    if disparity == 0:
        return right
    w = right.shape[-1]
    return F.pad(right, (disparity, 0, 0, 0))[..., :w]


def matching(left, right, maximum_disparity, operation):
    """Matching.forward, matching.py:34-63.  ``operation`` is any callable on
    cat([left, S_d(right)], 1); results stacked on dim 2 in order d = 0..max.
    Kept in the reference's looped form so the CPU baseline is not handicapped
    (the batched form is 2.9x slower on CPU)."""
    planes = []
    for d in range(maximum_disparity + 1):
        planes.append(operation(torch.cat([left, shift_right(right, d)], dim=1)))
    return torch.stack(planes, dim=2)


def matching_operation(p, prefix, x, number_of_residual_blocks=2):
    """MatchingOperation.forward, matching.py:97-112: conv3x3(128->64), residual
    blocks, conv3x3(64->8); the first and last conv are bare (matching.py:80-93)."""
    m = prefix + '._matching_operation_modules'
    x = F.conv2d(x, p[m + '.0.weight'], p[m + '.0.bias'], padding=1)
    for i in range(number_of_residual_blocks):
        x = residual_block_2d(p, '%s.%d' % (m, 1 + i), x)
    last = 1 + number_of_residual_blocks
    return F.conv2d(x, p['%s.%d.weight' % (m, last)], p['%s.%d.bias' % (m, last)], padding=1)


def matching_with_operation(p, prefix, left, right, maximum_disparity):
    """Matching(maximum_disparity, MatchingOperation()) with parameters under
    ``prefix + '._operation'`` (network.py:60-61)."""
    return matching(left, right, maximum_disparity,
                    lambda x: matching_operation(p, prefix + '._operation', x))


# --------------------------------------------------------------------------------------
# regularization.py
# --------------------------------------------------------------------------------------
def contraction_block_3d(p, prefix, x):
    """ContractionBlock3d.forward, regularization.py:28-31 -> (down, smooth(down))."""
    down = conv_block_3d(p, prefix + '._downsampling_2x', x, stride=2)
    return down, conv_block_3d(p, prefix + '._smoothing', down)


def expansion_block_3d(p, prefix, x, shortcut_from_contraction):
    """ExpansionBlock3d.forward, regularization.py:54-57."""
    up = deconv_block_3d(p, prefix + '._upsampling_2x', x)
    return conv_block_3d(p, prefix + '._smoothing', up + shortcut_from_contraction)


def regularization(p, prefix, matching_signatures, shortcut_from_left_image):
    """Regularization.forward, regularization.py:94-126."""
    shortcuts = []
    shortcut = shortcut_from_left_image.unsqueeze(2)
    output = conv_block_3d(p, prefix + '._smoothing', matching_signatures)
    for i in range(4):
        shortcuts.append(output)
        shortcut, output = contraction_block_3d(
            p, '%s._contraction_blocks.%d' % (prefix, i), shortcut + output)
    for i in range(4):
        output = expansion_block_3d(
            p, '%s._expansion_blocks.%d' % (prefix, i), output, shortcuts.pop())
    half = deconv_block_3d(p, prefix + '._upsample_to_halfsize', output)
    full = F.conv_transpose3d(half, p[prefix + '._upsample_to_fullsize.weight'],
                              p[prefix + '._upsample_to_fullsize.bias'],
                              stride=(1, 2, 2), padding=(1, 1, 1))
    return full.squeeze(1)


# --------------------------------------------------------------------------------------
# estimator.py
# --------------------------------------------------------------------------------------
def check_subpixel_map_arguments(half_support_window, disparity_step):
    """The three ValueErrors of SubpixelMap.__init__, estimator.py:34-41."""
    if disparity_step < 1:
        raise ValueError('"disparity_step" should be positive integer.')
    if half_support_window < 1:
        raise ValueError('"half_support_window" should be positive integer.')
    if half_support_window % disparity_step != 0:
        raise ValueError('"half_support_window" should be multiple of the'
                         '"disparity_step"')


def subpixel_map(similarities, half_support_window=4, disparity_step=2):
    """SubpixelMap.__call__, estimator.py:45-91, restated per pixel:
    m = first arg-max over dim 1; taps k = m + j for j in
    range(-hw // step, hw // step + 1) (Python floor division of the NEGATED
    window, estimator.py:66-68); taps outside [0, Dh) get probability 0;
    disparity = sum softmax(taps) * step * k."""
    check_subpixel_map_arguments(half_support_window, disparity_step)
    n_planes = similarities.shape[1]
    best = similarities.argmax(dim=1, keepdim=True)  # first occurrence on CPU
    shifts = range(-half_support_window // disparity_step,
                   half_support_window // disparity_step + 1)
    taps, values = [], []
    for j in shifts:
        k = best + j
        valid = (k >= 0) & (k < n_planes)
        kc = k.clamp(0, n_planes - 1)
        s = torch.gather(similarities, 1, kc)
        taps.append(torch.where(valid, s, torch.full_like(s, float('-inf'))))
        # estimator.py:71,79-82: the index is zeroed where invalid before it
        # is turned into a disparity value, so invalid taps carry disparity 0.
        values.append(torch.where(valid, k, torch.zeros_like(k)).to(similarities.dtype)
                      * disparity_step)
    taps = torch.stack(taps, dim=1)
    values = torch.stack(values, dim=1)
    prob = torch.softmax(taps, dim=1)
    return (prob * values).sum(1).squeeze(1)


# --------------------------------------------------------------------------------------
# loss.py (consumer of the path in training)
# --------------------------------------------------------------------------------------
def subpixel_cross_entropy(similarities, ground_truth_disparities, weights=None, diversity=1.0,
                           disparity_step=2):
    """SubpixelCrossEntropy.forward, loss.py:30-78, restated without the Python loop over planes:
    cross-entropy between softmax(similarities) and the unnormalised Laplace distribution
    exp(-|gt - k*step| / diversity) / (2*diversity) centred at the ground truth, normalised by the
    target's mass, averaged over the pixels whose ground truth is not inf (optionally weighted)."""
    planes = similarities.shape[1]
    known = ground_truth_disparities != float('inf')
    log_p = F.log_softmax(similarities, dim=1)
    levels = (torch.arange(planes, dtype=similarities.dtype, device=similarities.device)
              * disparity_step).view(1, planes, 1, 1)
    target = torch.exp(-torch.abs(ground_truth_disparities.unsqueeze(1) - levels) / diversity) / (2 * diversity)
    sum_target = target.sum(1)
    sum_target_log_p = (log_p * target).sum(1)
    entropy = -sum_target_log_p[known] / sum_target[known]
    if weights is not None:
        w = weights[known]
        return (w * entropy).sum() / (w.sum() + 1e-15)
    return entropy.mean()


# --------------------------------------------------------------------------------------
# errors.py (evaluation metrics)
# --------------------------------------------------------------------------------------
def absolute_error(estimated, ground_truth, use_mean=True):
    """compute_absolute_error, errors.py:9-44: |est - gt| with unknown (inf) ground truth shown as 0 and left
    out of the average; 0.0 when nothing is known."""
    difference = (estimated - ground_truth).abs()
    unknown = torch.isinf(ground_truth)
    pixelwise = torch.where(unknown, torch.zeros_like(difference), difference)
    known = difference[~unknown]
    if known.numel() == 0:
        return pixelwise, 0.0
    return pixelwise, (known.mean() if use_mean else known.median()).item()


def n_pixels_error(estimated, ground_truth, n=3.0):
    """compute_n_pixels_error, errors.py:47-74: 1 where |est - gt| > n (unknown pixels: 0), and the percentage of
    such pixels among the known ones."""
    unknown = torch.isinf(ground_truth)
    beyond = (estimated - ground_truth).abs().gt(n).float()
    pixelwise = torch.where(unknown, torch.zeros_like(beyond), beyond)
    known = beyond[~unknown]
    if known.numel() == 0:
        return pixelwise, 0.0
    return pixelwise, known.mean().item() * 100


# --------------------------------------------------------------------------------------
# embedding.py / size_adapter.py (producers of the path's inputs)
# --------------------------------------------------------------------------------------
def pad_to_multiple(image, minimum_size=64):
    """SizeAdapter.pad, size_adapter.py:29-43: zero rows on top and zero columns on the left up to the
    next multiple of ``minimum_size``.  Returns (padded, rows, columns)."""
    height, width = image.shape[-2:]
    rows = -height % minimum_size
    columns = -width % minimum_size
    return F.pad(image, (columns, 0, rows, 0)), rows, columns


def embedding(p, prefix, image, number_of_residual_blocks=2):
    """Embedding.forward, embedding.py:46-65: InstanceNorm2d without affine parameters (:32), two
    5x5 stride-2 blocks (:33-36), residual blocks (:38-41); returns (descriptor, shortcut) with
    shortcut = convolutional_block_3x3(descriptor) (:43-44, 65)."""
    m = prefix + '._embedding_modules'
    x = F.instance_norm(image, eps=IN_EPS)
    x = conv_block_5x5_stride_2(p, m + '.1', x)
    x = conv_block_5x5_stride_2(p, m + '.2', x)
    for i in range(number_of_residual_blocks):
        x = residual_block_2d(p, '%s.%d' % (m, 3 + i), x)
    return x, conv_block_2d(p, prefix + '._shortcut', x)


# --------------------------------------------------------------------------------------
# whole hot path + helpers used by tests / bench
# --------------------------------------------------------------------------------------
def hot_path(p, left_descriptor, right_descriptor, shortcut_from_left, maximum_disparity,
             matching_prefix='_matching', regularization_prefix='_regularization',
             half_support_window=4, disparity_step=2, return_stages=False):
    """Matching -> Regularization -> SubpixelMap exactly as network.py:38-52 chains
    them in eval mode.  ``maximum_disparity`` is the image-level value (191, 63,
    ...); Matching runs (max+1)//4 planes (network.py:36)."""
    n = (maximum_disparity + 1) // 4 - 1
    ms = matching_with_operation(p, matching_prefix, left_descriptor, right_descriptor, n)
    cost = regularization(p, regularization_prefix, ms, shortcut_from_left)
    disparity = subpixel_map(cost, half_support_window, disparity_step)
    if return_stages:
        return ms, cost, disparity
    return disparity


def network_training_output(p, left_image, right_image, maximum_disparity, embedding_prefix='_embedding',
                            matching_prefix='_matching', regularization_prefix='_regularization'):
    """PdsNetwork.forward in TRAINING mode (network.py:38-52): pad both images (size_adapter.py:29-43), descriptor
    network on each, Matching, Regularization, then -- the estimator being skipped (network.py:50-52) -- the matching
    cost cropped back to the image (size_adapter.py:45-52).  -> [batch, (max + 1) // 2, H, W]."""
    left_padded, rows, columns = pad_to_multiple(left_image)
    right_padded = pad_to_multiple(right_image)[0]
    left_descriptor, shortcut = embedding(p, embedding_prefix, left_padded)
    right_descriptor = embedding(p, embedding_prefix, right_padded)[0]
    n = (maximum_disparity + 1) // 4 - 1
    ms = matching_with_operation(p, matching_prefix, left_descriptor, right_descriptor, n)
    cost = regularization(p, regularization_prefix, ms, shortcut)
    return cost[..., rows:, columns:]


def cast_params(p, dtype):
    return {k: v.to(dtype) for k, v in p.items()}
