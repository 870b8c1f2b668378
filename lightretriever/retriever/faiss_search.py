from lightretriever_amd.retriever import DenseRetrievalFaissSearch, FlatIPFaissSearch  # noqa: F401
