class AnseriniSearch:
    """Lucene/Anserini sparse searcher: out of scope (no JVM path on the accelerated route)."""

    def __init__(self, *a, **k):
        raise NotImplementedError("AnseriniSearch (sparse path) is outside the accelerated dense path")
