"""Test-set / inference data set of the evaluation (mirror of ``src/inference/inference_dataset.py``: InferenceDataset
:13-41, pre_processing_transforms :44-62, Normalization :80-91, Padding :94-104, Scaling :107-125, ToTensor :128-140).

Host code (numpy): per image ``2 * (f32(img) - min) / (max - min) - 1``, top/left padding with the minimum up to the
next tested model input size, tensor (1, H, W).  CLAHE and down-scaling need scikit-image, which this stack does not
ship; the evaluation (eval.py:121-122) uses neither (apply_clahe=False, scale_factor=1) and asking for them raises.
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from ..utils import tiffio
from ..utils.utils import zero_pad_model_input


class InferenceDataset(Dataset):
    """ Images ``img*.tif`` of a directory, sorted by name. """

    def __init__(self, data_dir, transform=lambda x: x):
        self.img_ids = sorted(data_dir.glob('img*.tif'))
        self.transform = transform

    def __len__(self):
        return len(self.img_ids)

    def __getitem__(self, idx):
        img_id = self.img_ids[idx]
        img = tiffio.imread(str(img_id))
        sample = {'image': img, 'id': img_id.stem}
        return self.transform(sample)


class _Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, sample):
        for t in self.ts:
            sample = t(sample)
        return sample


def pre_processing_transforms(apply_clahe=False, scale_factor=1):
    """ ContrastEnhancement -> Normalization -> Scaling -> Padding -> ToTensor, like the reference. """
    return _Compose([ContrastEnhancement(apply_clahe), Normalization(), Scaling(scale_factor), Padding(), ToTensor()])


class ContrastEnhancement(object):
    def __init__(self, apply_clahe):
        self.apply_clahe = apply_clahe

    def __call__(self, sample):
        if self.apply_clahe:
            raise RuntimeError("CLAHE pre-processing needs scikit-image (equalize_adapthist), which is outside the "
                               "MI355X hot path; the evaluation runs with apply_clahe=False")
        return sample


class Normalization(object):
    def __call__(self, sample):
        img = sample['image']
        sample['image'] = 2 * (img.astype(np.float32) - img.min()) / (img.max() - img.min()) - 1
        return sample


class Padding(object):
    def __call__(self, sample):
        img = sample['image']
        img, pads = zero_pad_model_input(img=img, pad_val=np.min(img))
        sample['image'] = img
        sample['pads'] = pads
        return sample


class Scaling(object):
    def __init__(self, scale):
        self.scale = scale

    def __call__(self, sample):
        sample['original_size'] = sample['image'].shape
        if self.scale < 1:
            raise RuntimeError("down-scaling needs scikit-image (transform.rescale); the evaluation uses scale 1")
        return sample


class ToTensor(object):
    def __call__(self, sample):
        img = sample['image']
        if len(img.shape) == 2:
            img = img[None, :, :]
        img = torch.from_numpy(np.ascontiguousarray(img)).to(torch.float)
        return img, sample['id'], sample['pads'], sample['original_size']
