from lightretriever_amd.inference import PytorchRPCExactSearchModel  # noqa: F401
