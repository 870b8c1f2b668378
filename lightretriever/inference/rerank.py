class RerankerModel:
    """Out of scope of the MI355X dense path: importing works, using it raises."""

    def __init__(self, *a, **k):
        raise NotImplementedError("RerankerModel is outside the accelerated dense path")
