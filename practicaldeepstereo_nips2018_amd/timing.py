"""Time-per-image protocol of the reference's test loop.

Reference ``Trainer._test`` (practical_deep_stereo/trainer.py:229-252) takes an example from the loader on the
HOST, moves its tensors to the GPU (``_move_tensors_to_cuda``, :241-242), then ``_run_network_and_measure_time``
(:141-148) brackets the whole ``network(left_image, right_image)`` call -- pad, descriptor network on both images,
Matching, Regularization, SubpixelMap, crop (network.py:45-52, pds_trainer.py:35-38) -- with
``cuda.synchronize(); time.time()`` on both sides; the errors are computed afterwards (pds_trainer.py:48-58) and the
times averaged (pds_trainer.py:63-64).  The README's 0.62 s per 960x540 image is that average.

``time_per_image`` reproduces it step for step with the HIP modules (metrics stay on the device,
``errors.compute_*``), and additionally reports the same measurement with the host->device copy INSIDE the bracket
(what a caller holding host buffers pays; inputs already resident in HBM are the bench's headline)."""
import time

import torch

from . import errors


def run_network_and_measure_time(network, left_image, right_image):
    """trainer.py:141-148: synchronize, time, network(left, right), synchronize -> (output, seconds)."""
    torch.cuda.synchronize(left_image.device)
    start_time = time.time()
    output = network(left_image, right_image)
    torch.cuda.synchronize(left_image.device)
    return output, float(time.time() - start_time)


def time_per_image(network, examples, device, warmup=2):
    """examples: iterable of dicts {'left': host image [1,3,H,W], 'right': ..., 'disparity': optional host ground
    truth [1,H,W]} (the loader's items, trainer.py:235).  Returns averages over the examples after ``warmup``."""
    network.eval()
    times, times_with_copy, maes, three_px = [], [], [], []
    with torch.no_grad():
        for index, example in enumerate(examples):
            torch.cuda.synchronize(device)
            copy_start = time.time()
            left = example['left'].to(device, non_blocking=False)       # trainer.py:241-242
            right = example['right'].to(device, non_blocking=False)
            output, seconds = run_network_and_measure_time(network, left, right)
            with_copy = float(time.time() - copy_start)
            if example.get('disparity') is not None:                    # pds_trainer.py:48-58, on the device
                truth = example['disparity'].to(device)
                maes.append(float(errors.compute_absolute_error(output, truth)[1]))
                three_px.append(float(errors.compute_n_pixels_error(output, truth)[1]))
            if index >= warmup:
                times.append(seconds)
                times_with_copy.append(with_copy)
    result = {'time_per_image_ms': 1e3 * sum(times) / max(len(times), 1),
              'time_per_image_with_host_copy_ms': 1e3 * sum(times_with_copy) / max(len(times_with_copy), 1),
              'time_per_image_median_ms': 1e3 * sorted(times)[len(times) // 2] if times else 0.0,
              'per_example_ms': [round(1e3 * t, 3) for t in times],
              'examples': len(times),
              'protocol': 'trainer.py:141-148,241-242: host tensors -> .cuda() -> synchronize, time, network(left, right) '
                          '[pad, descriptor network x2, Matching, Regularization, SubpixelMap, crop], synchronize; one '
                          'example at a time'}
    if maes:
        result['mean_absolute_error'] = sum(maes) / len(maes)
        result['three_pixels_error'] = sum(three_px) / len(three_px)
    return result
