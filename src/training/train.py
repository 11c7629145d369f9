from microbeseg_amd.training.train import CreateLabelsWorker, TrainWorker, get_max_epochs, seed_worker  # noqa: F401
