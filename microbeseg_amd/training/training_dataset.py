"""Crop tensor contract of the training hot path.

Mirror of ``src/training/training_dataset.py`` (reference, :6-63) and of the ``ToTensor`` transform
(``src/training/mytransforms.py``:365-406): an item is ``(img, border_label, cell_label)`` for the distance method
(fp32, shape (1, H, W); image normalised from uint16 [0, 65535] to [-1, 1]) or ``(img, label)`` for the boundary
method (label int64 (H, W) in {0, 1, 2}).  On-disk layout: ``<root>/<mode>/img_<id>.tif`` with
``cell_dist_<id>.tif`` + ``neighbor_dist_<id>.tif`` (distance) or ``boundary_<id>.tif`` (boundary).

The reference's CPU augmentation pipeline (flip / rotate / scale / blur / noise / contrast, mytransforms.py:12-362)
runs on the device here (``training/device_augment.py``): for the 'train' phase ``augmentors`` returns
``RawToTensor`` — the un-normalised crop and its labels — and ``TrainWorker`` hands the batch to ``DeviceAugment``, whose
last stage is ToTensor's normalisation.  The 'val' phase is plain ``ToTensor`` as in the reference.
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from ..utils import tiffio as tiff
from ..utils.utils import min_max_normalization


class ToTensor(object):
    """ Normalise the image and convert image + labels to tensors (reference mytransforms.py:365-406). """

    def __init__(self, label_type, min_value, max_value):
        self.min_value = min_value
        self.max_value = max_value
        self.label_type = label_type

    def __call__(self, sample):
        sample['image'] = min_max_normalization(sample['image'], min_value=self.min_value, max_value=self.max_value)
        for key in sample:
            if key != 'id':
                sample[key] = np.transpose(sample[key], (2, 0, 1))     # (H, W, C) -> (C, H, W)
        img = torch.from_numpy(np.ascontiguousarray(sample['image'])).to(torch.float)
        if self.label_type == 'boundary':
            label = torch.from_numpy(np.ascontiguousarray(sample['label']))[0, :, :].to(torch.long)
            return img, label
        elif self.label_type == 'distance':
            cell_label = torch.from_numpy(np.ascontiguousarray(sample['cell_label'])).to(torch.float)
            border_label = torch.from_numpy(np.ascontiguousarray(sample['border_label'])).to(torch.float)
            return img, border_label, cell_label
        raise Exception('Unknown label type')

    def __repr__(self):
        return f"ToTensor(label_type={self.label_type!r}, min_value={self.min_value}, max_value={self.max_value})"


class RawToTensor(object):
    """ 'train' phase: the raw crop for the device augmentation — image as int32 (uint16 values, no normalisation),
    labels as in ToTensor.  ``min_value`` / ``max_value`` travel along for DeviceAugment's final normalisation. """
    device_augment = True

    def __init__(self, label_type, min_value, max_value):
        self.min_value = min_value
        self.max_value = max_value
        self.label_type = label_type

    def __call__(self, sample):
        for key in sample:
            if key != 'id':
                sample[key] = np.transpose(sample[key], (2, 0, 1))     # (H, W, C) -> (C, H, W)
        img = torch.from_numpy(np.ascontiguousarray(sample['image']).astype(np.int32))
        if self.label_type == 'boundary':
            label = torch.from_numpy(np.ascontiguousarray(sample['label']))[0, :, :].to(torch.long)
            return img, label
        elif self.label_type == 'distance':
            cell_label = torch.from_numpy(np.ascontiguousarray(sample['cell_label'])).to(torch.float)
            border_label = torch.from_numpy(np.ascontiguousarray(sample['border_label'])).to(torch.float)
            return img, border_label, cell_label
        raise Exception('Unknown label type')

    def __repr__(self):
        return ("Compose(Flip(p=1.0), Contrast(p=0.45), Scaling(p=0.25), Rotate(p=0.25), Blur(p=0.3), Noise(p=0.3), "
                f"ToTensor(label_type={self.label_type!r}, min_value={self.min_value}, max_value={self.max_value})) "
                "[on device]")


def augmentors(label_type, min_value, max_value, device_augmentation=True):
    """ Transforms per phase (reference mytransforms.py:12-35: augmentation + ToTensor for 'train', ToTensor for 'val').
    ``device_augmentation=False`` gives the un-augmented ToTensor for both phases (parity tests, benchmarks). """
    t = ToTensor(label_type=label_type, min_value=min_value, max_value=max_value)
    if not device_augmentation:
        return {'train': t, 'val': t}
    return {'train': RawToTensor(label_type=label_type, min_value=min_value, max_value=max_value), 'val': t}


class TrainingDataset(Dataset):
    """ Pytorch data set for instance segmentation crops (same constructor / item contract as the reference). """

    def __init__(self, root_dir, label_type, mode='train', transform=lambda x: x):
        self.img_ids = sorted((root_dir / mode).glob('img*.tif'))
        self.mode = mode
        self.root_dir = root_dir
        self.transform = transform
        self.label_type = label_type

    def __len__(self):
        return len(self.img_ids)

    def __getitem__(self, idx):
        img_id = self.img_ids[idx]
        img = tiff.imread(str(img_id))[..., None]
        suffix = img_id.name.split('img')[-1]
        if self.label_type == 'distance':
            cell = tiff.imread(str(img_id.parent / "cell_dist{}".format(suffix))).astype(np.float32)[..., None]
            border = tiff.imread(str(img_id.parent / "neighbor_dist{}".format(suffix))).astype(np.float32)[..., None]
            sample = {'image': img, 'cell_label': cell, 'border_label': border, 'id': img_id.stem}
        elif self.label_type == 'boundary':
            label = tiff.imread(str(img_id.parent / "boundary{}".format(suffix))).astype(np.uint8)[..., None]
            sample = {'image': img, 'label': label, 'id': img_id.stem}
        else:
            raise Exception('Unknown label type')
        return self.transform(sample)
