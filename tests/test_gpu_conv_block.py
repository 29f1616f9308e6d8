"""GPU (-m gpu): the single-layer entry point pds_conv_block_fwd against an fp64 PyTorch-CPU convolution.

One block of reference network_blocks.py:47-72 (Conv -> LeakyReLU(0.1) -> InstanceNorm with affine parameters) on a
plain tensor.  The shapes walk every kernel behind the entry point: the Winograd-domain 64-channel kernel
(conv2d_wino.hip: even widths, partial 64-column tiles, 12 / 64 input channels, several planes and batch entries,
per-plane and per-volume statistics, bare convolution), the direct MFMA kernel (odd widths, 8 / 16 output channels),
the 3-D MFMA kernel (stride 1 and 2) and the VALU fallback (channel counts no MFMA tiling covers).  Since round 3 the
Cin % 16 == 0 -> 64 layers run on conv2d_x3.hip (fp32 operands split into 16-bit parts on the 16-bit matrix pipe): a
plain input of unknown scale takes the range-safe form (three bf16 parts, six partial products), an input behind a
deferred InstanceNorm -- pds_conv_block_chained_fwd, the way the hot path chains its blocks -- the fp16 form (two parts,
three products).  Same cases and the same tolerance, plus cases that walk the persistent tile queues.
Tolerance (stated): max-abs <= 2e-5 on the O(1) activations, and on the normalised output scale * raw + shift."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from practicaldeepstereo_nips2018_amd import _lib

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope='module')
def dev(hip_library):
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def run_block(dev, x, weight, bias, gamma, beta, kd, stride, per_plane):
    lib = _lib.load()
    n, cin, d, h, w = x.shape
    cout = weight.shape[0]
    od = (d + 1) // 2 if (stride == 2 and kd == 3) else d
    oh, ow = ((h + 1) // 2, (w + 1) // 2) if stride == 2 else (h, w)
    xg = x.to(dev).contiguous()
    tensors = [t.to(dev).contiguous() if t is not None else None for t in (weight, bias, gamma, beta)]
    params = _lib.ConvBlockParams()
    params.weight, params.bias = tensors[0].data_ptr(), tensors[1].data_ptr()
    params.gamma = tensors[2].data_ptr() if gamma is not None else None
    params.beta = tensors[3].data_ptr() if beta is not None else None
    raw = torch.full((n, cout, od, oh, ow), float('nan'), device=dev)
    groups = n * cout * (od if per_plane else 1)
    scale = torch.zeros(groups, device=dev)
    shift = torch.zeros(groups, device=dev)
    nbytes = lib.pds_conv_block_workspace_bytes(n, cin, cout, d, h, w, kd, stride, per_plane)
    ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
    _lib.check(lib.pds_conv_block_fwd(ctypes.byref(params), _lib.ptr(xg), _lib.ptr(raw), _lib.ptr(scale),
                                      _lib.ptr(shift), n, cin, cout, d, h, w, kd, stride, per_plane, _lib.ptr(ws),
                                      ws.numel(), _lib.stream_handle(dev)), 'pds_conv_block_fwd')
    torch.cuda.synchronize()
    return raw.cpu(), scale.cpu(), shift.cpu()


def reference(x, weight, bias, gamma, beta, kd, stride, per_plane):
    x, weight, bias = x.double(), weight.double(), bias.double()
    if kd == 1:
        n, cin, d, h, w = x.shape
        planes = x.permute(0, 2, 1, 3, 4).reshape(n * d, cin, h, w)
        y = F.conv2d(planes, weight, bias, stride=stride, padding=1)
        y = y.reshape(n, d, -1, y.shape[-2], y.shape[-1]).permute(0, 2, 1, 3, 4)
    else:
        y = F.conv3d(x, weight, bias, stride=stride, padding=1)
    if gamma is None:
        return y, None
    raw = F.leaky_relu(y, 0.1)
    dims = (3, 4) if per_plane else (2, 3, 4)
    mean = raw.mean(dim=dims, keepdim=True)
    var = raw.var(dim=dims, unbiased=False, keepdim=True)
    shape = (1, -1, 1, 1, 1)
    normed = (raw - mean) / torch.sqrt(var + 1e-5) * gamma.double().view(shape) + beta.double().view(shape)
    return raw, normed


CASES = [
    # n, cin, cout, d, h, w, kd, stride, per_plane, affine
    (1, 64, 64, 3, 12, 64, 1, 1, 1, True),     # Winograd kernel, one full tile column
    (2, 64, 64, 2, 9, 50, 1, 1, 1, True),      # Winograd: batch 2, partial tile, ragged rows
    (1, 12, 64, 1, 7, 130, 1, 1, 0, True),     # Winograd: 12 input channels (space-to-depth layer), per-volume stats
    (1, 64, 64, 8, 6, 16, 1, 1, 1, False),     # Winograd: bare convolution, 8 planes (XCD re-mapping active)
    (1, 64, 64, 2, 8, 33, 1, 1, 1, True),      # odd width: direct MFMA kernel
    (2, 64, 64, 2, 40, 72, 1, 1, 1, True),     # 16x16-tile Winograd kernel (15 tiles vs 20 wide ones), ragged on both axes
    (1, 64, 64, 8, 16, 48, 1, 1, 1, False),    # 16x16-tile Winograd kernel, exact tiling, bare, XCD re-mapping active
    (1, 128, 64, 1, 24, 20, 1, 1, 0, True),    # 16x16 tiles, 128 input channels, width 20 (one partial tile column)
    (2, 64, 64, 8, 32, 16, 1, 1, 1, True),     # 16x16 tiles, batch 2 x 8 planes: XCD re-mapping with a batch axis
    (1, 64, 64, 48, 48, 80, 1, 1, 1, True),    # conv2d_x3: 432 tiles, more than one per persistent workgroup (queues, stealing)
    (3, 64, 64, 5, 17, 47, 1, 1, 1, True),     # conv2d_x3: 15 planes (uneven queues), ragged rows, right-half-empty tile column
    (1, 48, 64, 2, 20, 36, 1, 1, 0, True),     # conv2d_x3: three K-steps (the fewest it takes), per-volume statistics
    (1, 64, 8, 3, 10, 40, 1, 1, 1, False),     # 8 output channels: direct MFMA kernel, one channel block
    (2, 64, 16, 1, 5, 24, 1, 1, 1, True),
    (1, 8, 8, 6, 10, 20, 3, 1, 0, True),       # conv3d MFMA
    (1, 8, 16, 8, 12, 20, 3, 2, 0, True),      # conv3d MFMA stride 2
    (1, 6, 6, 4, 6, 10, 3, 1, 0, True),        # channel counts without an MFMA tiling: VALU kernel
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'n%d_%dto%d_d%d_%dx%d_k%d_s%d_pp%d_%s' % (
    c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], 'in' if c[9] else 'bare'))
def test_conv_block_against_fp64(dev, case):
    n, cin, cout, d, h, w, kd, stride, per_plane, affine = case
    g = torch.Generator().manual_seed(1000 + cin * 7 + cout * 3 + w)
    x = torch.randn(n, cin, d, h, w, generator=g)
    fan_in = cin * 9 * (3 if kd == 3 else 1)
    wshape = (cout, cin, 3, 3) if kd == 1 else (cout, cin, 3, 3, 3)
    weight = torch.randn(*wshape, generator=g) / fan_in ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    gamma = torch.rand(cout, generator=g) + 0.5 if affine else None
    beta = torch.randn(cout, generator=g) * 0.2 if affine else None
    raw, scale, shift = run_block(dev, x, weight, bias, gamma, beta, kd, stride, per_plane)
    want_raw, want_normed = reference(x, weight, bias, gamma, beta, kd, stride, per_plane)
    assert raw.shape == want_raw.shape
    assert not torch.isnan(raw).any(), 'output positions left unwritten'
    err = float((raw.double() - want_raw).abs().max())
    assert err <= TOL, err
    if affine:
        groups = (n, cout, raw.shape[2] if per_plane else 1, 1, 1)
        normed = raw.double() * scale.double().view(groups) + shift.double().view(groups)
        err_n = float((normed - want_normed).abs().max())
        assert err_n <= 5 * TOL, err_n


CHAINED = [
    # n, cin, d, h, w, x_per_plane, cout: conv2d_x3 in its fp16 form (the input sits behind a deferred InstanceNorm)
    (1, 64, 48, 48, 80, 1, 64),    # many tiles per persistent workgroup
    (3, 64, 5, 17, 47, 1, 64),     # uneven queues, ragged rows, right-half-empty tile column
    (1, 48, 2, 20, 36, 0, 64),     # three K-steps, per-volume input statistics
    (2, 128, 1, 24, 20, 0, 64),    # 128 input channels
    # conv2d_t8 on the fp16-split MFMA: full-width form (rows of 240 / 320 / 100 columns, ragged heights) and 16 x 32 tiles
    (1, 64, 3, 20, 240, 1, 8),
    (2, 64, 2, 13, 320, 1, 8),
    (1, 64, 5, 9, 100, 0, 8),
    (1, 64, 2, 21, 36, 1, 8),
]


@pytest.mark.parametrize('case', CHAINED, ids=lambda c: 'n%d_%dto%d_d%d_%dx%d_xpp%d' % (c[0], c[1], c[6], c[2], c[3], c[4], c[5]))
def test_chained_conv_block_against_fp64(dev, case):
    """pds_conv_block_chained_fwd (ABI v3): the loader applies the producer's folded InstanceNorm, x^ = s * x + h.  The
    64-output-channel layers then run the fp16 two-way-split form of conv2d_x3 (three products per multiply): same 2e-5
    bound as the exact-fp32 kernels.  The raw producer output is deliberately far from unit scale (x 37, offset 5)."""
    n, cin, d, h, w, xpp, cout = case
    lib = _lib.load()
    g = torch.Generator().manual_seed(77 + cin + w)
    x = torch.randn(n, cin, d, h, w, generator=g) * 37.0 + 5.0
    groups_in = (n, cin, d if xpp else 1, 1, 1)
    x_scale = (torch.rand(groups_in, generator=g) + 0.5) / 37.0
    x_shift = torch.randn(groups_in, generator=g) * 0.2 - 5.0 * x_scale
    weight = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    # the 64 -> 8 layer of MatchingOperation is a bare convolution (matching.py:89-93): no InstanceNorm behind it
    affine = cout == 64
    gamma = torch.rand(cout, generator=g) + 0.5 if affine else None
    beta = torch.randn(cout, generator=g) * 0.2 if affine else None
    tensors = [t.to(dev).contiguous() if t is not None else None for t in (weight, bias, gamma, beta)]
    params = _lib.ConvBlockParams()
    params.weight, params.bias = tensors[0].data_ptr(), tensors[1].data_ptr()
    params.gamma = tensors[2].data_ptr() if affine else None
    params.beta = tensors[3].data_ptr() if affine else None
    raw = torch.full((n, cout, d, h, w), float('nan'), device=dev)
    scale = torch.zeros(n * cout * d, device=dev)
    shift = torch.zeros(n * cout * d, device=dev)
    ws = torch.empty(int(lib.pds_conv_block_workspace_bytes(n, cin, cout, d, h, w, 1, 1, 1)), dtype=torch.uint8, device=dev)
    xg, sg, hg = x.to(dev), x_scale.reshape(-1).to(dev).contiguous(), x_shift.reshape(-1).to(dev).contiguous()
    # the reference sees the fp32 normalised input the loader forms (one fma per element)
    xhat = torch.addcmul(x_shift.expand_as(x), x_scale.expand_as(x), x)
    # ABI v5: the range certificate of the input (inside the modules in_finalize writes a rigorous, much looser one)
    bound = xhat.abs().max().reshape(1).to(dev)
    _lib.check(lib.pds_conv_block_chained_fwd(ctypes.byref(params), _lib.ptr(xg), _lib.ptr(sg), _lib.ptr(hg), xpp,
                                              _lib.ptr(bound), _lib.ptr(raw), _lib.ptr(scale), _lib.ptr(shift), n, cin, cout,
                                              d, h, w, 1, 1, 1, _lib.ptr(ws), ws.numel(), _lib.stream_handle(dev)),
               'pds_conv_block_chained_fwd')
    torch.cuda.synchronize()
    want_raw, want_normed = reference(xhat, weight, bias, gamma, beta, 1, 1, 1)
    err = float((raw.cpu().double() - want_raw).abs().max())
    assert not torch.isnan(raw).any() and err <= TOL, err
    # "as accurate as an fp32 fma chain" (tools/ubench/fp16x2_probe.hip: mean 2.0e-7 at K = 576): the mean error too
    mean_err = float((raw.cpu().double() - want_raw).abs().mean())
    assert mean_err <= 6e-7, mean_err
    if affine:
        normed = raw.cpu().double() * scale.cpu().double().view(n, cout, d, 1, 1) + shift.cpu().double().view(n, cout, d, 1, 1)
        err_n = float((normed - want_normed).abs().max())
        assert err_n <= 5 * TOL, err_n
