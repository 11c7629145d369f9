from microbeseg_amd.training.train_data_representations import (boundary_label, border_label, cell_distance_label,  # noqa: F401
                                                                distance_label, distance_label_batch, get_label, j4_label,
                                                                max_major_axis_length)
