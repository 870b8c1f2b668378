from lightretriever_amd.score_fuse_utils import fuse_scores_linear, fuse_scores_rrf  # noqa: F401
