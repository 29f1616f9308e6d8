"""Shared helpers of the test-suite: seeded modules, golden fixtures, error metrics."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz')) as z:
        return {k: torch.from_numpy(np.array(z[k])) for k in z.files if z[k].dtype.kind not in 'US'}   # (not name lists)


def checksum(state_dict):
    return float(sum(v.double().sum() for v in state_dict.values()))


def prefixed(state_dict, prefix):
    return {prefix + '.' + k: v.detach().cpu() for k, v in state_dict.items()}


def seeded(factory, seed=0):
    torch.manual_seed(seed)
    return factory()


def images(batch, height, width):
    """Input recipe of SURVEY.md 8c: left first, then right, uniform [0, 255)."""
    g = torch.Generator().manual_seed(1)
    left = torch.rand(batch, 3, height, width, generator=g) * 255
    right = torch.rand(batch, 3, height, width, generator=g) * 255
    return left, right


def maxdiff(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def meandiff(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().mean())


def disparity_report(gpu, cpu):
    delta = (gpu.detach().double().cpu() - cpu.detach().double().cpu()).abs()
    smooth = delta[delta <= 0.5]
    return {'mae': float(delta.mean()), 'max': float(delta.max()),
            'flips': float((delta > 0.5).double().mean()),
            'mae_noflip': float(smooth.mean()) if smooth.numel() else 0.0}


def host_descriptors(net, image):
    """(descriptor, shortcut) of a padded image on the host through the oracle's restatement of the descriptor
    network (the package's Embedding runs on the GPU only): hot-path tests feed these bit-identical inputs to
    both the HIP path and the CPU oracle."""
    from oracle import pds_oracle
    params = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    padded = pds_oracle.pad_to_multiple(image)[0]
    with torch.no_grad():
        return pds_oracle.embedding(params, '_embedding', padded)


def flip_allowance(params, ld, rd, shortcut, maximum_disparity, disparity_fp32, slack=2):
    """How many arg-max flips a result may show against the oracle's fp32 disparity, DERIVED from this run instead of a
    constant: the oracle is run once more in fp64 on the same inputs; its own fp32 run flips F pixels
    against that truth, and a result that is no further from the truth than the reference (F + slack flips against fp64)
    can differ from the fp32 run at no more than F + (F + slack) pixels (a pixel differs from the fp32 run only if one of
    the two differs from the truth).  Returns (allowance against the fp32 oracle, fp64 disparity, F)."""
    from oracle import pds_oracle
    with torch.no_grad():
        truth = pds_oracle.hot_path(pds_oracle.cast_params(params, torch.float64), ld.double(), rd.double(),
                                    shortcut.double(), maximum_disparity)
    reference_flips = int(((disparity_fp32.detach().double().cpu() - truth).abs() > 0.5).sum())
    return 2 * reference_flips + slack, truth, reference_flips
