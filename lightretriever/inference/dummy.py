class DummyModel:
    """BM25 passthrough model of the reference (inference/dummy.py): sparse path, out of scope."""

    def __init__(self, *a, **k):
        raise NotImplementedError("DummyModel (BM25/Anserini path) is outside the accelerated dense path")
