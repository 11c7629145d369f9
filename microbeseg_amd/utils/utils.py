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
    """generator yielding the planes of a 2-D image (what omero's createImageFromNumpySeq wants; reference utils.py:5-8)"""
    for p in [img]:
        yield p


def write_inference_results(results, path):
    """Write inference results (dict) to ``path / 'results.json'`` (reference utils.py:110-121)."""
    with open(path / 'results.json', 'w', encoding='utf-8') as outfile:
        json.dump(results, outfile, ensure_ascii=False, indent=2)


def get_nucleus_ids(img):
    """ Ids (> 0) present in an intensity-coded label image (reference utils.py:11-22). """
    values = np.unique(img)
    return values[values > 0]


def border_correction(mask, border_width=10):
    """ Delete (in place, like the reference: utils.py:25-47) the instances that are not visible inside the field of
    interest = the mask minus a border of ``border_width`` pixels.  Host helper with the reference's signature; the
    evaluation itself uses the fused device kernel (evaluation/stats_utils.py: relabel_device). """
    ids_prediction = get_nucleus_ids(mask)
    foi = mask[border_width:mask.shape[0] - border_width, border_width:mask.shape[1] - border_width]
    ids_foi = get_nucleus_ids(foi)
    for id_prediction in ids_prediction:
        if id_prediction not in ids_foi:
            mask[mask == id_prediction] = 0
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


def pad_amounts(shape):
    """[pad_y, pad_x] up to the next tested shape; raises for frames larger than 8192 (reference utils.py:147-155)."""
    pads = []
    for i in range(2):
        for s in TESTED_IMG_SHAPES:
            if shape[i] <= s:
                pads.append(s - shape[i])
                break
    if len(pads) < 2:
        raise Exception('Image too big to pad. Use sliding windows')
    return pads


def zero_pad_model_input(img, pad_val=0):
    """ Pad the TOP and LEFT of a frame with ``pad_val`` up to the next tested model input size.

    :return: padded img, [rows padded, columns padded]
    """
    if len(img.shape) == 3:  # 3D image (z-dimension needs no pads)
        img = np.transpose(img, (2, 1, 0))
    # NB: like the reference, a frame exceeding 8192 in only ONE dimension yields a single pad entry and fails below
    pads = []
    for i in range(2):
        for s in TESTED_IMG_SHAPES:
            if img.shape[i] <= s:
                pads.append(s - img.shape[i])
                break
    if not pads:
        raise Exception('Image too big to pad. Use sliding windows')
    if len(img.shape) == 3:
        img = np.pad(img, ((pads[0], 0), (pads[1], 0), (0, 0)), mode='constant', constant_values=pad_val)
        img = np.transpose(img, (2, 1, 0))
    else:
        img = np.pad(img, ((pads[0], 0), (pads[1], 0)), mode='constant', constant_values=pad_val)
    return img, [pads[0], pads[1]]
