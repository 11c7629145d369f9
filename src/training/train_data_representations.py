from microbeseg_amd.training.train_data_representations import boundary_label, border_label, get_label  # noqa: F401
