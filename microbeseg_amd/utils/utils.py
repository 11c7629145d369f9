"""Boundary glue of the hot path: frame padding to the tested shapes, min-max normalisation, checkpoint naming and
the training-info JSON.  Mirror of the hot-path subset of ``src/utils/utils.py`` (reference):
``min_max_normalization`` (:50-74), ``unique_path`` (:77-91), ``write_train_info`` (:94-107),
``zero_pad_model_input`` (:124-163).  Pure host code (numpy / json); same signatures and return values."""
import json

import numpy as np

# model input sizes the reference pads frames up to (utils.py:137-138); all divisible by 64
TESTED_IMG_SHAPES = (64, 128, 256, 320, 512, 768, 1024, 1280, 1408, 1600, 1920, 2048, 2240, 2560, 3200, 4096, 4480,
                     6080, 8192)


def plane_gen(img):
    """One-plane generator over a 2-D image, the form omero's ``createImageFromNumpySeq`` consumes (reference utils.py:5-8)."""
    yield img


def write_inference_results(results, path):
    """``path / 'results.json'`` <- the results dict (reference utils.py:110-121)."""
    (path / 'results.json').write_text(json.dumps(results, ensure_ascii=False, indent=2), encoding='utf-8')


def get_nucleus_ids(img):
    """Sorted instance ids (> 0) of an intensity-coded label image (reference utils.py:11-22)."""
    ids = np.unique(img)
    return ids[np.searchsorted(ids, 0, side='right'):]


def border_correction(mask, border_width=10):
    """Zero, IN PLACE, every instance that has no pixel inside the field of interest (the frame minus a rim of
    ``border_width`` pixels) and return the mask (contract of reference utils.py:25-47).  One table lookup instead of a
    pass over the image per dropped instance; the evaluation itself runs the fused device kernel
    (evaluation/stats_utils.py: relabel_device)."""
    h, w = mask.shape[:2]
    inner = mask[border_width:h - border_width, border_width:w - border_width]
    gone = np.setdiff1d(get_nucleus_ids(mask), get_nucleus_ids(inner), assume_unique=True)
    if gone.size:
        mask[np.isin(mask, gone)] = 0
    return mask


def min_max_normalization(img, min_value=None, max_value=None):
    """ Clip to [min_value, max_value] and map to [-1, 1] (float32). """
    if max_value is None:
        max_value = img.max()
    if min_value is None:
        min_value = img.min()
    img = np.clip(img, min_value, max_value)
    img = 2 * (img.astype(np.float32) - min_value) / (max_value - min_value) - 1
    return img.astype(np.float32)


def unique_path(directory, name_pattern):
    """ First ``directory / name_pattern.format(k)``, k = 1, 2, ..., that does not exist yet. """
    counter = 0
    while True:
        counter += 1
        path = directory / name_pattern.format(counter)
        if not path.exists():
            return path


def write_train_info(configs, path):
    """ ``<run_name>.json`` next to the checkpoint (read back by the inference code: infer.py:83-84,119-127). """
    with open(path / (configs['run_name'] + '.json'), 'w', encoding='utf-8') as outfile:
        json.dump(configs, outfile, ensure_ascii=False, indent=2)
    return None


def _pad_to_tested(n):
    """padding of one frame edge up to the next tested shape, or None beyond 8192"""
    k = int(np.searchsorted(TESTED_IMG_SHAPES, n))
    return None if k == len(TESTED_IMG_SHAPES) else TESTED_IMG_SHAPES[k] - n


def pad_amounts(shape):
    """[pad_y, pad_x] up to the next tested shape; raises for frames larger than 8192 (reference utils.py:147-155)."""
    pads = [_pad_to_tested(shape[0]), _pad_to_tested(shape[1])]
    if None in pads:
        raise Exception('Image too big to pad. Use sliding windows')
    return pads


def zero_pad_model_input(img, pad_val=0):
    """Pad the TOP and LEFT edge of a frame (H, W) or (H, W, C) with ``pad_val`` up to the next tested model input size
    (contract of reference utils.py:124-163).

    :return: padded img, [rows padded, columns padded]
    """
    # the reference transposes a 3-D frame to (C, W, H) and reads its first two axes, i.e. (C, W): kept, because the
    # returned pad amounts are part of the contract its callers crop with (tests/golden/host_contract.npz)
    probe = img.shape[:2] if img.ndim == 2 else (img.shape[2], img.shape[1])
    pads = [v for v in (_pad_to_tested(probe[0]), _pad_to_tested(probe[1])) if v is not None]
    if not pads:
        raise Exception('Image too big to pad. Use sliding windows')
    if len(pads) < 2:                                  # one edge beyond 8192: the reference fails on pads[1] here as well
        raise IndexError('list index out of range')
    if img.ndim == 3:
        t = np.transpose(img, (2, 1, 0))
        t = np.pad(t, ((pads[0], 0), (pads[1], 0), (0, 0)), mode='constant', constant_values=pad_val)
        return np.transpose(t, (2, 1, 0)), pads
    return np.pad(img, ((pads[0], 0), (pads[1], 0)), mode='constant', constant_values=pad_val), pads
