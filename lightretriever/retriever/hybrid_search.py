from lightretriever_amd.retriever import HybridSearch  # noqa: F401
