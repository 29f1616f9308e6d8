"""Parameter containers for the layer stacks of the hot path.

Mirrors the factories of reference practical_deep_stereo/network_blocks.py (:19-24, :37-44,
:47-85, :97-144) only in what a checkpoint sees: the same sub-module indices (0 = conv,
1 = LeakyReLU, 2 = InstanceNorm) and therefore the same state-dict keys, and the same
construction order so ``torch.manual_seed`` yields the reference's initial weights.

These modules are never executed by PyTorch: Embedding, Matching and Regularization hand their
parameters to the HIP library (libpds_hip.so).  They are torch modules only so that parameters are
registered, moved and (de)serialised the way the reference's are.
"""
from torch import nn

LEAKY_SLOPE = 0.1


def _norm_for(conv):
    return {nn.Conv2d: nn.InstanceNorm2d}.get(type(conv), nn.InstanceNorm3d)(
        conv.out_channels, affine=True)


class ConvActNorm(nn.Sequential):
    """conv -> LeakyReLU(0.1) -> InstanceNorm(affine) (network_blocks.py:47-85)."""

    def __init__(self, conv):
        super(ConvActNorm, self).__init__(
            conv, nn.LeakyReLU(negative_slope=LEAKY_SLOPE, inplace=True), _norm_for(conv))

    @property
    def conv(self):
        return self[0]

    @property
    def norm(self):
        return self[2]


def convolution_3x3(cin, cout):
    return nn.Conv2d(cin, cout, kernel_size=3, padding=1)


def transposed_convolution_3x4x4_stride_122(cin, cout):
    return nn.ConvTranspose3d(cin, cout, kernel_size=(3, 4, 4), stride=(1, 2, 2), padding=(1, 1, 1))


def convolutional_block_5x5_stride_2(cin, cout):
    return ConvActNorm(nn.Conv2d(cin, cout, kernel_size=5, stride=2, padding=2))


def convolutional_block_3x3(cin, cout):
    return ConvActNorm(nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1))


def convolutional_block_3x3x3(cin, cout):
    return ConvActNorm(nn.Conv3d(cin, cout, kernel_size=3, stride=1, padding=1))


def convolutional_block_3x3x3_stride_2(cin, cout):
    return ConvActNorm(nn.Conv3d(cin, cout, kernel_size=3, stride=2, padding=1))


def transposed_convolutional_block_4x4x4_stride_2(cin, cout):
    return ConvActNorm(nn.ConvTranspose3d(cin, cout, kernel_size=4, stride=2, padding=1))


class ResidualBlock(nn.Module):
    """Two conv blocks and a skip sum without activation (network_blocks.py:134-144)."""

    def __init__(self, number_of_features):
        super(ResidualBlock, self).__init__()
        self.convolutions = nn.Sequential(
            convolutional_block_3x3(number_of_features, number_of_features),
            convolutional_block_3x3(number_of_features, number_of_features))

    def forward(self, block_input):
        return block_input + self.convolutions(block_input)
