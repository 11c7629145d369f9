from microbeseg_amd.inference.postprocessing import distance_postprocessing, boundary_postprocessing  # noqa: F401
