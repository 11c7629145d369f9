from microbeseg_amd.inference.inference_dataset import InferenceDataset, pre_processing_transforms  # noqa: F401
