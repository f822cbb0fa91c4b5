"""ASDNet descriptor: oracle vs the reference's own outputs; HIP vs oracle + golden."""
import numpy as np
import pytest

# descriptors are unit-norm f32 rows; the reference accumulates in f32 (libtorch), we accumulate in f32
# on the matrix cores in a different order with BN folded: |d| <= 2e-5 abs per component.
DESC_ATOL = 2e-5


def test_oracle_matches_reference_golden(oracle, synth, asdnet_golden):
    g = asdnet_golden
    layers = synth.asdnet_weights(int(g["weight_seed"]))
    out, l6 = oracle.asdnet_forward(layers, g["patches"], want_l6=True)
    assert out.shape == (64, 128)
    np.testing.assert_allclose(out, g["desc"], atol=5e-6, rtol=0)
    np.testing.assert_allclose(l6[:4], g["act_l6"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)


def test_oracle_constant_patch_is_finite(oracle, synth):
    layers = synth.asdnet_weights(0)
    p = np.full((2, 32, 32), 200, np.uint8)
    out = oracle.asdnet_forward(layers, p)
    assert np.isfinite(out).all()
    np.testing.assert_array_equal(out[0], out[1])


@pytest.mark.gpu
def test_hip_matches_golden(hip, asdnet_golden):
    g = asdnet_golden
    out = hip.describe(g["patches"])
    np.testing.assert_allclose(out, g["desc"], atol=DESC_ATOL, rtol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 31, 33, 257])
def test_hip_matches_oracle_ragged(hip, oracle, synth, n):
    layers = synth.asdnet_weights(0)
    patches = synth.random_patches(n, seed=100 + n)
    ref = oracle.asdnet_forward(layers, patches)
    out = hip.describe(patches)
    np.testing.assert_allclose(out, ref, atol=DESC_ATOL, rtol=0)


@pytest.mark.gpu
def test_hip_empty_and_capacity(hip, pkg):
    assert hip.describe(np.zeros((0, 32, 32), np.uint8)).shape == (0, 128)
    with pytest.raises(pkg.AsdError) as ei:
        hip.describe(np.zeros((5000, 32, 32), np.uint8))
    assert ei.value.code == -5


@pytest.mark.gpu
def test_hip_full_size_properties(hip, synth):
    """N = 4000 (the initialisation extractor's 2 x nFeatures, Tracking.cc:85): size-independent
    properties -- unit norm, batch-position independence, determinism."""
    patches = synth.random_patches(4000, seed=7)
    out = hip.describe(patches)
    assert np.isfinite(out).all()
    np.testing.assert_allclose(np.linalg.norm(out.astype(np.float64), axis=1), 1.0, atol=1e-5)
    perm = np.random.default_rng(0).permutation(4000)
    out_p = hip.describe(patches[perm])
    np.testing.assert_array_equal(out_p, out[perm])
    np.testing.assert_array_equal(hip.describe(patches), out)
