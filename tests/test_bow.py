"""DBoW2 transform for float descriptors (TemplatedVocabulary<FSift>::transform, Frame::ComputeBoW):
oracle known-answer checks on CPU, HIP-vs-oracle parity on the GPU through the C ABI."""
import os

import numpy as np
import pytest

from tests.test_matcher import BOUNDS, make_frame, perturbed_descriptors


def _ancestor(voc, node, level):
    parent = np.zeros(voc["n_nodes"], np.int32)
    for i in range(voc["n_nodes"]):
        parent[voc["child_ids"][voc["child_start"][i]:voc["child_start"][i + 1]]] = i
    while voc["level"][node] > level:
        node = parent[node]
    return node


def _queries(voc, n, sigma, seed):
    rng = np.random.default_rng(seed)
    leaves = np.nonzero(voc["word_id"] >= 0)[0]
    src = leaves[rng.integers(0, len(leaves), n)]
    return perturbed_descriptors(voc["desc"][src], sigma, seed + 1), src


# ------------------------------------------------------------------ CPU: oracle known answers
def test_oracle_descent_reaches_source_leaf(oracle, synth):
    voc = synth.vocabulary(k=6, L=3, seed=1, stop_frac=0.0)
    V = oracle.vocabulary(voc)
    q, src = _queries(voc, 500, 0.01, 2)
    word, node, weight = V.descend(q, levelsup=2)
    assert (word == voc["word_id"][src]).mean() > 0.97
    hit = word == voc["word_id"][src]
    exp_nodes = np.array([_ancestor(voc, s, voc["L"] - 2) for s in src])
    np.testing.assert_array_equal(node[hit], exp_nodes[hit])
    np.testing.assert_array_equal(weight[hit], voc["weight"][src][hit])
    # levelsup >= L -> root
    assert (V.descend(q, levelsup=3)[1] == 0).all()


def test_oracle_descent_matches_float64_argmin_and_tie_rule(oracle, synth):
    voc = synth.vocabulary(k=5, L=2, seed=3)
    # duplicate a child's descriptor: the earlier child must win the tie (strict `<`, TemplatedVocabulary.h:1241-1246)
    c = voc["child_ids"][voc["child_start"][0]:voc["child_start"][1]]
    voc["desc"][c[3]] = voc["desc"][c[1]]
    V = oracle.vocabulary(voc)
    q = voc["desc"][c[1]][None].copy()
    _, node, _ = V.descend(q, levelsup=1)
    assert node[0] == c[1]
    rng = np.random.default_rng(4)
    q = rng.standard_normal((300, 128)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    word, _, _ = V.descend(q, levelsup=0)
    for i in range(len(q)):
        cur = 0
        while voc["child_start"][cur + 1] > voc["child_start"][cur]:
            ch = voc["child_ids"][voc["child_start"][cur]:voc["child_start"][cur + 1]]
            d = ((q[i].astype(np.float64) - voc["desc"][ch].astype(np.float64)) ** 2).sum(1)
            order = np.sort(d)
            if order[1] - order[0] < 1e-6:
                cur = -1
                break
            cur = ch[np.argmin(d)]
        if cur >= 0:
            assert word[i] == voc["word_id"][cur]


@pytest.mark.parametrize("weighting,scoring", [(0, 0), (0, 1), (1, 5), (2, 0), (3, 5)])
def test_oracle_transform_matches_python_maps(oracle, synth, weighting, scoring):
    voc = synth.vocabulary(k=4, L=3, seed=5, stop_frac=0.2)
    V = oracle.vocabulary(voc, weighting, scoring)
    q, _ = _queries(voc, 400, 0.05, 6)
    word, node, weight = V.descend(q, levelsup=1)
    (bid, bval), (fnode, fstart, fidx) = V.transform(q, levelsup=1)
    bow, fv = {}, {}
    for i in range(len(q)):
        if weight[i] > 0:
            if weighting in (0, 1):
                bow[word[i]] = bow.get(word[i], 0.0) + weight[i] if word[i] in bow else weight[i]
            else:
                bow.setdefault(word[i], weight[i])
            fv.setdefault(node[i], []).append(i)
    ids = sorted(bow)
    vals = np.array([bow[i] for i in ids])
    if scoring == 5:
        if weighting in (0, 1):
            vals = vals / float(len(vals))
    elif scoring == 1:
        nrm = 0.0
        for v in vals:
            nrm += v * v
        vals = vals / np.sqrt(nrm)
    else:
        nrm = 0.0
        for v in vals:
            nrm += abs(v)
        vals = vals / nrm
    np.testing.assert_array_equal(bid, ids)
    np.testing.assert_array_equal(bval, vals)
    np.testing.assert_array_equal(fnode, sorted(fv))
    for k, nid in enumerate(sorted(fv)):
        np.testing.assert_array_equal(fidx[fstart[k]:fstart[k + 1]], fv[nid])
    assert (weight == 0).any()                       # stop words exist and are dropped from both containers
    assert fstart[-1] == (weight > 0).sum()


# ------------------------------------------------------------------ GPU parity through the C ABI
@pytest.mark.gpu
@pytest.mark.parametrize("k,L,ragged,levelsup", [(10, 3, False, 2), (10, 4, False, 4), (7, 5, True, 3), (3, 6, True, 4),
                                                 (20, 2, False, 1), (64, 1, False, 0), (33, 2, True, 1)])
def test_bow_descend_parity(hip, oracle, synth, k, L, ragged, levelsup):
    voc = synth.vocabulary(k=k, L=L, seed=10 + k, ragged=ragged)
    hip.voc_load(voc)
    V = oracle.vocabulary(voc)
    rng = np.random.default_rng(11)
    q1, _ = _queries(voc, 1500, 0.2, 12)
    q2 = rng.standard_normal((533, 128)).astype(np.float32)
    q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
    q = np.concatenate([q1, q2])
    gw, gn, gwt = hip.bow_descend(q, levelsup=levelsup)
    ew, en, ewt = V.descend(q, levelsup=levelsup)
    np.testing.assert_array_equal(gw, ew)
    np.testing.assert_array_equal(gn, en)
    np.testing.assert_array_equal(gwt, ewt)
    assert len(np.unique(gw)) > 1


@pytest.mark.gpu
def test_bow_descend_ties_and_nan(hip, oracle, synth):
    voc = synth.vocabulary(k=10, L=2, seed=20)
    c = voc["child_ids"][voc["child_start"][0]:voc["child_start"][1]]
    voc["desc"][c[7]] = voc["desc"][c[2]]
    voc["desc"][c[9]] = voc["desc"][c[2]]
    hip.voc_load(voc)
    V = oracle.vocabulary(voc)
    q = np.concatenate([voc["desc"][c[2]][None], np.zeros((1, 128), np.float32), np.full((1, 128), np.nan, np.float32)])
    g = hip.bow_descend(q, levelsup=1)
    e = V.descend(q, levelsup=1)
    assert g[1][0] == c[2]
    for a, b in zip(g, e):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("weighting,scoring", [(0, 0), (0, 1), (1, 5), (2, 0), (3, 5), (0, 5)])
def test_compute_bow_parity(hip, oracle, synth, weighting, scoring):
    voc = synth.vocabulary(k=10, L=3, seed=30, stop_frac=0.1)
    hip.voc_load(voc, weighting, scoring)
    V = oracle.vocabulary(voc, weighting, scoring)
    kps, desc = make_frame(2000, 31)
    (gb, gv), (gn, gs, gi) = hip.compute_bow(desc, levelsup=2)
    (eb, ev), (en, es, ei) = V.transform(desc, levelsup=2)
    np.testing.assert_array_equal(gb, eb)
    np.testing.assert_array_equal(gv, ev)
    np.testing.assert_array_equal(gn, en)
    np.testing.assert_array_equal(gs, es)
    np.testing.assert_array_equal(gi, ei)
    # same through a resident frame slot
    hip.frame_set(3, kps, desc, BOUNDS)
    (sb, sv), (sn, ss, si) = hip.compute_bow(slot=3, n=len(kps), levelsup=2)
    np.testing.assert_array_equal(sb, eb)
    np.testing.assert_array_equal(sv, ev)
    np.testing.assert_array_equal(si, ei)
    if scoring == 0:
        assert abs(gv.sum() - 1.0) < 1e-12


@pytest.mark.gpu
def test_compute_bow_feeds_search_by_bow(hip, oracle, synth, pkg):
    """ComputeBoW -> SearchByBoW end to end: the FeatureVectors produced on the device drive M3 and give the
    same matches as the oracle pipeline"""
    voc = synth.vocabulary(k=10, L=3, seed=40, stop_frac=0.0)
    hip.voc_load(voc)
    V = oracle.vocabulary(voc)
    k1, d1 = make_frame(1500, 41)
    rng = np.random.default_rng(42)
    perm = rng.permutation(len(k1))
    k2 = k1[perm].copy()
    d2 = perturbed_descriptors(d1[perm], 0.03, 43)
    hip.frame_set(0, k1, d1, BOUNDS)
    hip.frame_set(1, k2, d2, BOUNDS)
    _, (n1, s1, i1) = hip.compute_bow(slot=0, n=len(k1), levelsup=2)
    _, (n2, s2, i2) = hip.compute_bow(slot=1, n=len(k2), levelsup=2)
    def per_kp(nodes, start, idx, n):
        out = np.full(n, -1, np.int32)
        for k in range(len(nodes)):
            out[idx[start[k]:start[k + 1]]] = nodes[k]
        return out
    has = np.ones(len(k1), np.uint8)
    gm, gn = hip.match_bow(0, 1, len(k2), per_kp(n1, s1, i1, len(k1)), per_kp(n2, s2, i2, len(k2)), has, 0.7, True)
    node1 = V.descend(d1, levelsup=2)[1]
    node2 = V.descend(d2, levelsup=2)[1]
    em, en = oracle.match_bow(oracle.frame(k1, d1, BOUNDS), oracle.frame(k2, d2, BOUNDS), node1, node2, has, 0.7, True)
    np.testing.assert_array_equal(gm, em)
    assert gn == en and gn > 0.5 * len(k1)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    ok = gm >= 0
    assert (gm[ok] == perm[np.nonzero(ok)[0]]).mean() > 0.95


@pytest.mark.gpu
def test_voc_load_rejects_bad_trees(hip, synth, pkg):
    voc = synth.vocabulary(k=4, L=2, seed=50)
    bad = dict(voc)
    bad["child_ids"] = voc["child_ids"].copy()
    bad["child_ids"][-1] = bad["child_ids"][0]            # a node with two parents / one unreachable
    with pytest.raises(Exception):
        hip.voc_load(bad)
    bad = dict(voc)
    bad["word_id"] = np.full_like(voc["word_id"], -1)       # leaves without words
    with pytest.raises(Exception):
        hip.voc_load(bad)
    hip.voc_load(voc)                                       # a good one still loads afterwards
    assert len(hip.bow_descend(voc["desc"][1:5], levelsup=1)[0]) == 4


# ------------------------------------------------------------------ pinned: the reference's own BowVector / FeatureVector
def _bow_golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bow_golden.npz"))


def test_oracle_bow_assembly_equals_reference_classes(oracle):
    """BowVector::addWeight / addIfNotExist / normalize and FeatureVector::addFeature (src/dbow2/DBoW2/BowVector.cpp,
    FeatureVector.cpp), driven like TemplatedVocabulary::transform: the oracle's assembly against fixtures produced by the
    reference's own sources compiled in place (tests/golden/make_bow_golden.py) -- word ids, node lists and the f64
    weights bit for bit, for every weighting x normalisation."""
    G = _bow_golden()
    for ci in range(int(G["n_cases"])):
        word, weight, node = G[f"c{ci}_word"], G[f"c{ci}_weight"], G[f"c{ci}_node"]
        for weighting in range(4):
            w = weight if weighting in (0, 2) else (weight > 0).astype(np.float64)
            for scoring in (0, 1, 5):
                (bid, bval), (fnode, fstart, fidx) = oracle.bow_assemble(word, w, node, weighting, scoring)
                k = f"c{ci}_w{weighting}_s{scoring}"
                np.testing.assert_array_equal(bid, G[k + "_bow_id"])
                np.testing.assert_array_equal(bval.view(np.uint64), G[k + "_bow_val"].view(np.uint64))
                np.testing.assert_array_equal(fnode, G[k + "_fv_node"])
                np.testing.assert_array_equal(fstart, G[k + "_fv_start"])
                np.testing.assert_array_equal(fidx, G[k + "_fv_idx"])


def test_reference_dbow2_live_when_built(oracle, oracle_mod):
    """where oracle/_ref/libdbow2_ref.so exists (build container, GPU box): the same comparison live on fresh inputs"""
    if not oracle_mod.RefDBoW2.available():
        pytest.skip("oracle/_ref/libdbow2_ref.so not built")
    ref = oracle_mod.RefDBoW2()
    rng = np.random.default_rng(77)
    for n in (0, 5, 1500):
        word = rng.integers(0, 400, n).astype(np.int32)
        node = (word // 13).astype(np.int32)
        weight = np.where(rng.uniform(size=n) < 0.1, 0.0, rng.uniform(0.1, 5.0, n))
        for weighting in range(4):
            for scoring in range(6):
                a = oracle.bow_assemble(word, weight, node, weighting, scoring)
                b = ref.assemble(word, weight, node, weighting, scoring)
                for x, y in zip(a[0] + a[1], b[0] + b[1]):
                    np.testing.assert_array_equal(x.view(np.uint64) if x.dtype == np.float64 else x,
                                                  y.view(np.uint64) if y.dtype == np.float64 else y)
