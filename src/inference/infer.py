from microbeseg_amd.inference.infer import InferWorker  # noqa: F401
