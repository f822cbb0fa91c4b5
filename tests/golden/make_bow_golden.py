#!/usr/bin/env python3
"""Generate tests/golden/bow_golden.npz with the REFERENCE's own DBoW2::BowVector / DBoW2::FeatureVector, compiled in place
from /root/reference/src/dbow2/DBoW2/{BowVector,FeatureVector}.cpp by oracle/ref_dbow2/Makefile (outputs in oracle/_ref/,
never committed).

Runs only in the build container.  Inputs are seeded per-feature results of a vocabulary descent (word id, word weight with
some zero = stopped words, node id); the fixture stores the inputs and, for every weighting (TF_IDF, TF, IDF, BINARY) x
scoring family (L1-normalised, L2-normalised, not normalised), the BowVector and FeatureVector the reference classes
produce when driven like TemplatedVocabulary::transform (oracle/ref_dbow2/driver.cpp).
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

CASES = [dict(n=2000, n_words=900, n_nodes=90, seed=1), dict(n=300, n_words=20, n_nodes=4, seed=2),
         dict(n=1, n_words=1, n_nodes=1, seed=3), dict(n=0, n_words=1, n_nodes=1, seed=4)]
SCORINGS = (0, 1, 5)   # L1_NORM, L2_NORM, DOT_PRODUCT (ScoringType): L1-normalised, L2-normalised, no normalisation


def case_inputs(n, n_words, n_nodes, seed):
    rng = np.random.default_rng(seed)
    word = rng.integers(0, n_words, n).astype(np.int32) * 7 + 3          # sparse, unordered ids
    node = (word // 70).astype(np.int32) % max(n_nodes, 1) + 11
    idf = rng.uniform(0.05, 6.0, n_words * 7 + 10)
    idf[rng.uniform(size=len(idf)) < 0.05] = 0.0                           # stopped words (weight 0 is skipped)
    weight = idf[word].astype(np.float64)
    return word, weight, node


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_dbow2")])
    ref = g.load_oracle().RefDBoW2()
    out = {}
    for ci, kw in enumerate(CASES):
        word, weight, node = case_inputs(**kw)
        out[f"c{ci}_word"], out[f"c{ci}_weight"], out[f"c{ci}_node"] = word, weight, node
        for weighting in range(4):
            w = weight if weighting in (0, 2) else (weight > 0).astype(np.float64)   # TF / BINARY: weight 1 (0 = stopped)
            for scoring in SCORINGS:
                (bid, bval), (fnode, fstart, fidx) = ref.assemble(word, w, node, weighting, scoring)
                k = f"c{ci}_w{weighting}_s{scoring}"
                out[k + "_bow_id"], out[k + "_bow_val"] = bid, bval
                out[k + "_fv_node"], out[k + "_fv_start"], out[k + "_fv_idx"] = fnode, fstart, fidx
    out["n_cases"] = np.array(len(CASES))
    path = os.path.join(ROOT, "tests", "golden", "bow_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
