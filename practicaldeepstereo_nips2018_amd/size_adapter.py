"""Pads network inputs to multiples of a minimum size and crops the output back.

Off the hot path; same behaviour as reference
practical_deep_stereo/size_adapter.py:11-52: zero rows are added on TOP and zero columns on the
LEFT, and the amounts of the last ``pad`` call are remembered for ``unpad``.
"""
import torch.nn.functional as F


class SizeAdapter(object):
    def __init__(self, minimum_size=64):
        self._minimum_size = minimum_size
        self._pixels_pad_to_width = None
        self._pixels_pad_to_height = None

    def _padding_for(self, size):
        return (-size) % self._minimum_size

    def measure(self, network_input):
        """Records (and returns) the padding ``pad`` would apply, without building the padded tensor: the HIP
        embedding applies it in its loader."""
        height, width = network_input.shape[-2:]
        self._pixels_pad_to_height = self._padding_for(height)
        self._pixels_pad_to_width = self._padding_for(width)
        return self._pixels_pad_to_height, self._pixels_pad_to_width

    def padding(self):
        """(rows added on top, columns added on the left) by the last ``pad`` / ``measure``."""
        return self._pixels_pad_to_height, self._pixels_pad_to_width

    def pad(self, network_input):
        self.measure(network_input)
        return F.pad(network_input, (self._pixels_pad_to_width, 0, self._pixels_pad_to_height, 0))

    def unpad(self, network_output):
        return network_output[..., self._pixels_pad_to_height:, self._pixels_pad_to_width:]
