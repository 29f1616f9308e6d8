"""PdsNetwork: the dependency-injected pipeline that consumes the hot-path modules.

Same constructor, ``set_maximum_disparity``, ``pass_through_network``, ``forward`` and
``default`` as reference practical_deep_stereo/network.py:14-65, so a ``PdsTrainer``-style loop
can use it unchanged; ``default()`` wires the MI355X modules of this package instead of the
reference classes.  In eval mode, when the injected modules are this package's Regularization and
SubpixelMap, the last regularization layer and the estimator run fused.
"""
import torch
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

    def freeze_weights(self):
        """Inference deployment: promise that the parameters stay untouched, so the modules keep their re-laid-out
        weights between calls (``_lib.FrozenWeightsMixin``; not in the reference).  ``train()`` undoes it;
        ``.to()`` and ``load_state_dict`` invalidate the kept weights once (the network stays frozen); after editing
        parameters through ``.data`` call ``invalidate_weights()`` yourself."""
        for module in self.modules():
            if module is not self and hasattr(module, 'freeze_weights'):
                module.freeze_weights()
        return self

    def invalidate_weights(self):
        for module in self.modules():
            if module is not self and hasattr(module, 'invalidate_weights'):
                module.invalidate_weights()
        return self

    def _signatures(self, left_image, right_image):
        left_descriptor, shortcut_from_left = self._embedding(left_image)
        right_descriptor = self._embedding(right_image)[0]
        return self._matching(left_descriptor, right_descriptor), shortcut_from_left

    def _can_fuse_padding(self, left_image, right_image):
        return (isinstance(self._embedding, embedding.Embedding)
                and isinstance(self._size_adapter, size_adapter.SizeAdapter)
                and left_image.shape == right_image.shape and left_image.is_cuda)

    def _signatures_from_unpadded(self, left_image, right_image):
        """Both images through ONE embedding call (InstanceNorm statistics are per image, so batching them is
        the same arithmetic as network.py:38-40) with SizeAdapter.pad applied inside its loader."""
        pad_top, pad_left = self._size_adapter.measure(left_image)
        batch = left_image.size(0)
        descriptors, shortcuts = self._embedding.forward_padded(
            torch.cat([left_image, right_image], 0), pad_top, pad_left)
        return self._matching(descriptors[:batch], descriptors[batch:]), shortcuts[:batch]

    def pass_through_network(self, left_image, right_image):
        signatures, shortcut_from_left = self._signatures(left_image, right_image)
        return self._regularization(signatures, shortcut_from_left), shortcut_from_left

    def _can_fuse(self):
        return (self.fuse_estimator and isinstance(self._regularization, regularization.Regularization)
                and isinstance(self._estimator, estimator.SubpixelMap))

    def forward(self, left_image, right_image):
        """Sub-pixel disparity in eval mode, matching cost in training mode (network.py:45-52)."""
        if self._can_fuse_padding(left_image, right_image):
            signatures, shortcut_from_left = self._signatures_from_unpadded(left_image, right_image)
        else:
            signatures, shortcut_from_left = self._signatures(self._size_adapter.pad(left_image),
                                                              self._size_adapter.pad(right_image))
        if not self.training and self._can_fuse():
            crop = self._size_adapter.padding() if hasattr(self._size_adapter, 'padding') else None
            if crop is not None and self._regularization.can_fold_crop(self._estimator):
                # SizeAdapter.unpad (size_adapter.py:45-52) folded into the estimator's store: the result is the
                # contiguous [batch, H, W] image, not a view of the padded one
                return self._regularization.forward_with_estimator(signatures, shortcut_from_left, self._estimator,
                                                                   crop=crop)
            output = self._regularization.forward_with_estimator(signatures, shortcut_from_left,
                                                                 self._estimator)
        else:
            output = self._regularization(signatures, shortcut_from_left)
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
