#!/usr/bin/env python3
"""Golden vectors for hit-list fusion from the REAL reference functions
retriever/score_fuse_utils.py:{fuse_scores_rrf, fuse_scores_linear}.  Build container only.
Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_fusion_goldens.py"""
import importlib.util
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_fuse", "/root/reference/src/lightretriever/retriever/score_fuse_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def system(rng, qids, n_docs, k, scale, integer=False):
    out = {}
    for q in qids:
        docs = rng.choice(n_docs, size=k, replace=False)
        sc = rng.standard_normal(k).astype(np.float32) * scale
        if integer:
            sc = np.round(np.abs(sc) * 50) + rng.permutation(k) * 1e-3       # impact-like, all distinct
        out[q] = {"d%d" % d: float(s) for d, s in zip(docs, sc)}
    return out


def main():
    rng = np.random.default_rng(11)
    qids = ["q%d" % i for i in range(7)]
    dense = system(rng, qids, 60, 25, 0.3)                     # heavy overlap with the sparse list
    sparse = system(rng, qids[:-1] + ["only_sparse"], 60, 40, 3.0, integer=True)
    third = system(rng, qids[:3], 60, 10, 1.0)
    dense["single"] = {"d1": 0.5}                               # max == min -> eps path
    sparse["single"] = {"d1": 7.0, "d2": 3.0}
    cases = {"dense": dense, "sparse": sparse, "third": third,
             "rrf": ref.fuse_scores_rrf([dense, sparse]), "rrf_k10": ref.fuse_scores_rrf([dense, sparse], k=10),
             "rrf_three": ref.fuse_scores_rrf([dense, sparse, third]),
             "linear": ref.fuse_scores_linear([dense, sparse], weights=[0.7, 0.3]),
             "linear_5050": ref.fuse_scores_linear([dense, sparse], weights=[0.5, 0.5], eps=1e-6),
             "linear_three": ref.fuse_scores_linear([dense, sparse, third], weights=[0.5, 0.3, 0.2])}
    with open(os.path.join(HERE, "fusion.json"), "w") as f:
        json.dump(cases, f)
    print({k: len(v) for k, v in cases.items()})


if __name__ == "__main__":
    main()
