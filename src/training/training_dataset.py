from microbeseg_amd.training.training_dataset import TrainingDataset  # noqa: F401
