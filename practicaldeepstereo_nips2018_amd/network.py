"""PdsNetwork: the dependency-injected pipeline that consumes the hot-path modules.

Same constructor, ``set_maximum_disparity``, ``pass_through_network``, ``forward`` and
``default`` as reference practical_deep_stereo/network.py:14-65, so a ``PdsTrainer``-style loop
can use it unchanged; ``default()`` wires the MI355X modules of this package instead of the
reference classes.  In eval mode, when the injected modules are this package's Regularization and
SubpixelMap, the last regularization layer and the estimator run fused.
"""
from torch import nn

from practicaldeepstereo_nips2018_amd import embedding
from practicaldeepstereo_nips2018_amd import estimator
from practicaldeepstereo_nips2018_amd import matching
from practicaldeepstereo_nips2018_amd import regularization
from practicaldeepstereo_nips2018_amd import size_adapter


class PdsNetwork(nn.Module):
    def __init__(self, size_adapter_module, embedding_module, matching_module,
                 regularization_module, estimator_module):
        super(PdsNetwork, self).__init__()
        self._size_adapter = size_adapter_module
        self._embedding = embedding_module
        self._matching = matching_module
        self._regularization = regularization_module
        self._estimator = estimator_module
        self.fuse_estimator = True

    def set_maximum_disparity(self, maximum_disparity):
        if (maximum_disparity + 1) % 64 != 0:
            raise ValueError(
                '"maximum_disparity" + 1 should be multiple of 64, e.g.,'
                '"maximum disparity" can be equal to 63, 191, 255, 319...')
        self._maximum_disparity = maximum_disparity
        # the embedding downsamples 4x, so Matching covers (max + 1) / 4 planes (network.py:33-36)
        self._matching.set_maximum_disparity((maximum_disparity + 1) // 4 - 1)

    def _signatures(self, left_image, right_image):
        left_descriptor, shortcut_from_left = self._embedding(left_image)
        right_descriptor = self._embedding(right_image)[0]
        return self._matching(left_descriptor, right_descriptor), shortcut_from_left

    def pass_through_network(self, left_image, right_image):
        signatures, shortcut_from_left = self._signatures(left_image, right_image)
        return self._regularization(signatures, shortcut_from_left), shortcut_from_left

    def _can_fuse(self):
        return (self.fuse_estimator and isinstance(self._regularization, regularization.Regularization)
                and isinstance(self._estimator, estimator.SubpixelMap))

    def forward(self, left_image, right_image):
        """Sub-pixel disparity in eval mode, matching cost in training mode (network.py:45-52)."""
        left = self._size_adapter.pad(left_image)
        right = self._size_adapter.pad(right_image)
        if not self.training and self._can_fuse():
            signatures, shortcut_from_left = self._signatures(left, right)
            output = self._regularization.forward_with_estimator(signatures, shortcut_from_left,
                                                                 self._estimator)
        else:
            output = self.pass_through_network(left, right)[0]
            if not self.training:
                output = self._estimator(output)
        return self._size_adapter.unpad(output)

    @staticmethod
    def default(maximum_disparity=255):
        network = PdsNetwork(
            size_adapter_module=size_adapter.SizeAdapter(),
            embedding_module=embedding.Embedding(),
            matching_module=matching.Matching(operation=matching.MatchingOperation(),
                                              maximum_disparity=0),
            regularization_module=regularization.Regularization(),
            estimator_module=estimator.SubpixelMap())
        network.set_maximum_disparity(maximum_disparity)
        return network
