from lightretriever_amd.retriever import FaissIndex  # noqa: F401
