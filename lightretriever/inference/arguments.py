from lightretriever_amd.inference import InferenceArguments  # noqa: F401
