from microbeseg_amd.evaluation.eval import EvalWorker  # noqa: F401
