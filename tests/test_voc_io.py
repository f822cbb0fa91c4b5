"""DBoW2 vocabulary file (cv::FileStorage YAML of TemplatedVocabulary::save) <-> flat arrays for asd_voc_load
(asd-slam_amd/host/voc_io.hpp).  The reference's vocabulary file is not in its tree: round trips only.  CPU."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "asd-slam_amd", "host", "test_replay_io")


def _to_bin(voc, path, scoring=0, weighting=0):
    with open(path, "wb") as f:
        np.array([voc["k"], voc["L"], scoring, weighting, voc["n_nodes"], len(voc["child_ids"])], np.int32).tofile(f)
        voc["child_start"].astype(np.int32).tofile(f)
        voc["child_ids"].astype(np.int32).tofile(f)
        voc["word_id"].astype(np.int32).tofile(f)
        voc["weight"].astype(np.float64).tofile(f)
        voc["desc"].astype(np.float32).tofile(f)


def _from_bin(path):
    raw = open(path, "rb").read()
    hdr = np.frombuffer(raw, np.int32, 6)
    n, nc = int(hdr[4]), int(hdr[5])
    o = 24
    out = {"k": int(hdr[0]), "L": int(hdr[1]), "scoring": int(hdr[2]), "weighting": int(hdr[3]), "n_nodes": n}
    out["child_start"] = np.frombuffer(raw, np.int32, n + 1, o); o += 4 * (n + 1)
    out["child_ids"] = np.frombuffer(raw, np.int32, nc, o); o += 4 * nc
    out["word_id"] = np.frombuffer(raw, np.int32, n, o); o += 4 * n
    out["weight"] = np.frombuffer(raw, np.float64, n, o); o += 8 * n
    out["desc"] = np.frombuffer(raw, np.float32, n * 128, o).reshape(n, 128)
    return out


@pytest.fixture(scope="module")
def probe():
    if not os.path.exists(PROBE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "asd-slam_amd", "csrc"), "../host/test_replay_io"])
    return lambda *a: subprocess.run([PROBE, *map(str, a)], capture_output=True, text=True)


@pytest.mark.parametrize("k,L,ragged", [(4, 3, False), (5, 4, True)])
def test_vocabulary_yaml_round_trip(probe, synth, tmp_path, k, L, ragged):
    voc = synth.vocabulary(k=k, L=L, seed=9, ragged=ragged)
    _to_bin(voc, tmp_path / "a.bin", scoring=0, weighting=0)
    assert probe("vocwrite", tmp_path / "a.bin", tmp_path / "voc.yml").returncode == 0
    text = (tmp_path / "voc.yml").read_text()
    assert text.startswith("%YAML:1.0") and "nodeId:1, parentId:0" in text
    r = probe("vocdump", tmp_path / "voc.yml", tmp_path / "b.bin")
    assert r.returncode == 0, r.stderr
    got = _from_bin(tmp_path / "b.bin")
    assert (got["k"], got["L"], got["n_nodes"]) == (k, L, voc["n_nodes"])
    # the tree comes back with the children in file order; the depth-first writer keeps each parent's children in order
    np.testing.assert_array_equal(got["child_start"], voc["child_start"])
    np.testing.assert_array_equal(got["child_ids"], voc["child_ids"])
    np.testing.assert_array_equal(got["word_id"], voc["word_id"])
    np.testing.assert_array_equal(got["weight"], voc["weight"])          # %.17g round-trips doubles
    np.testing.assert_array_equal(got["desc"][1:], voc["desc"][1:])      # %.9g round-trips floats


def test_vocabulary_reader_tolerates_filestorage_layout(probe, tmp_path):
    """whitespace / line wraps as cv::FileStorage emits them; the child order is the FILE order, not the id order"""
    d = lambda v: " ".join([repr(float(v))] * 128) + " "
    yml = ("%YAML:1.0\n---\nvocabulary:\n   k: 2\n   L: 1\n   scoringType: 0\n   weightingType: 0\n   nodes:\n"
           "      - { nodeId:2, parentId:0, weight:1.5000000000000000e+00,\n          descriptor:\"" + d(0.25) + "\" }\n"
           "      - { nodeId:1, parentId:0,\n          weight:2.,\n          descriptor:\"" + d(-0.5) + "\" }\n"
           "   words:\n      - { wordId:0, nodeId:2 }\n      - { wordId:1, nodeId:1 }\n")
    (tmp_path / "v.yml").write_text(yml)
    assert probe("vocdump", tmp_path / "v.yml", tmp_path / "v.bin").returncode == 0
    got = _from_bin(tmp_path / "v.bin")
    assert got["n_nodes"] == 3
    assert got["child_ids"].tolist() == [2, 1]                            # m_nodes[0].children in file order
    assert got["word_id"].tolist() == [-1, 1, 0]
    assert got["weight"].tolist() == [0.0, 2.0, 1.5]
    assert (got["desc"][2] == np.float32(0.25)).all() and (got["desc"][1] == np.float32(-0.5)).all()
    # broken files are refused
    (tmp_path / "bad.yml").write_text(yml.replace("descriptor:\"" + d(0.25), "descriptor:\"1 2 3 "))
    assert probe("vocdump", tmp_path / "bad.yml", tmp_path / "x.bin").returncode == 1
    assert probe("vocdump", tmp_path / "missing.yml", tmp_path / "x.bin").returncode == 1
