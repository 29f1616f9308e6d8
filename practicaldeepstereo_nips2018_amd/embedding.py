"""Siamese descriptor network (off the hot path: stays on PyTorch-ROCm / MIOpen by design,
BASELINE.json north_star).  Same layer stack, outputs and state-dict keys as
reference practical_deep_stereo/embedding.py:11-65 so one checkpoint serves both."""
from torch import nn

from practicaldeepstereo_nips2018_amd import network_blocks


class Embedding(nn.Module):
    def __init__(self,
                 number_of_input_features=3,
                 number_of_embedding_features=64,
                 number_of_shortcut_features=8,
                 number_of_residual_blocks=2):
        super(Embedding, self).__init__()
        stack = [nn.InstanceNorm2d(number_of_input_features),
                 network_blocks.convolutional_block_5x5_stride_2(number_of_input_features,
                                                                 number_of_embedding_features),
                 network_blocks.convolutional_block_5x5_stride_2(number_of_embedding_features,
                                                                 number_of_embedding_features)]
        stack.extend(network_blocks.ResidualBlock(number_of_embedding_features)
                     for _ in range(number_of_residual_blocks))
        self._embedding_modules = nn.ModuleList(stack)
        self._shortcut = network_blocks.convolutional_block_3x3(number_of_embedding_features,
                                                                number_of_shortcut_features)

    def forward(self, image):
        """image [batch, 3, H, W] -> (descriptor [batch, 64, H/4, W/4], shortcut [batch, 8, H/4, W/4])."""
        descriptor = image
        for layer in self._embedding_modules:
            descriptor = layer(descriptor)
        return descriptor, self._shortcut(descriptor)
