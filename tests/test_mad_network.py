"""The denoiser kernel's "median getter" network (path-tracer_amd/csrc/mad_network.h), compiled
for the host, against the literal two-pass loop of the CPU oracle (DenoiserNN.cpp:190-256):
bit-identical results on every nine-tuple tried -- exhaustively over small alphabets (which
cover every ordering with ties, zeros and negative values that nine slots can see), and on
random and adversarial tuples (denormals, infinities, NaN, signed zeros, near-ties)."""
import itertools

import numpy as np


def _same(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def _check(oracle, windows):
    import ptk
    w = np.ascontiguousarray(windows, dtype=np.float32).reshape(-1, 9)
    got = ptk.mad_network_host(w)
    ref = oracle.mad_procedure(w)
    bad = np.nonzero(~_same(got, ref))[0]
    assert len(bad) == 0, "first mismatch: %r -> network %r, loop %r" % (
        w[bad[0]].tolist(), got[bad[0]], ref[bad[0]])
    return len(w)


def test_exhaustive_small_alphabets(oracle):
    # every 9-tuple over {-1, 0, 1, 2}: 262 144 tuples; all orderings of up to four levels
    vals = np.float32([-1.0, 0.0, 1.0, 2.0])
    idx = np.array(list(itertools.product(range(4), repeat=9)), dtype=np.int8)
    n = _check(oracle, vals[idx])
    # every 9-tuple over three positive levels and over three mixed levels
    idx3 = np.array(list(itertools.product(range(3), repeat=9)), dtype=np.int8)
    n += _check(oracle, np.float32([0.25, 0.5, 0.75])[idx3])
    n += _check(oracle, np.float32([-0.5, 0.125, 3.0])[idx3])
    assert n == 4 ** 9 + 2 * 3 ** 9


def test_all_orderings_of_distinct_values(oracle):
    # all 9! = 362 880 permutations of nine distinct values, positive and shifted negative
    perms = np.array(list(itertools.permutations(range(9))), dtype=np.int8)
    base = np.float32([0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9])
    _check(oracle, base[perms])
    _check(oracle, (base - np.float32(0.45))[perms])


def test_random_and_adversarial(oracle):
    rng = np.random.default_rng(99)
    n = 400000
    _check(oracle, rng.normal(size=(n, 9)))
    _check(oracle, rng.uniform(0, 1, size=(n, 9)))
    # few distinct levels per window: ties everywhere
    lv = rng.normal(size=(n, 3)).astype(np.float32)
    _check(oracle, np.take_along_axis(lv, rng.integers(0, 3, size=(n, 9)), axis=1))
    # near-ties: one ulp apart
    base = rng.uniform(0.5, 1.0, size=(n, 1)).astype(np.float32)
    steps = rng.integers(-2, 3, size=(n, 9))
    near = (base.view(np.int32) + steps).view(np.float32)
    _check(oracle, near)
    # special values mixed in
    special = np.float32([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3.4e38, -3.4e38,
                          1.0, -1.0])
    w = rng.normal(size=(n, 9)).astype(np.float32)
    mask = rng.random((n, 9)) < 0.4
    w[mask] = special[rng.integers(0, len(special), size=int(mask.sum()))]
    _check(oracle, w)
    # constant windows: the result is zero whatever the level
    c = rng.normal(size=(1000, 1)).astype(np.float32) * np.ones((1, 9), np.float32)
    import ptk
    assert not ptk.mad_network_host(c).any()


def test_nine_equal_finite_values_give_plus_zero(oracle):
    """What the feature kernels' wave-wide shortcut relies on: a window whose nine values compare
    equal and are finite gives +0 -- the bits, not just the value -- in the literal loop and in
    the network alike: two million random bit patterns, every power of two, the smallest and
    largest magnitudes, and every mixture of +0 and -0 (which compare equal)."""
    import ptk
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 1 << 32, size=2_000_000, dtype=np.uint64).astype(np.uint32)
    vals = bits.view(np.float32)
    vals = vals[np.isfinite(vals)]
    powers = np.float32(2.0) ** np.arange(-149, 128, dtype=np.float32)
    edge = np.float32([1e-45, 1.1754942e-38, 1.17549435e-38, 3.4028235e38, 1.0, 0.0])
    vals = np.concatenate([vals, powers, -powers, edge, -edge]).astype(np.float32)
    w = np.repeat(vals[:, None], 9, axis=1)
    for got in (ptk.mad_network_host(w), oracle.mad_procedure(w)):
        assert (got.view(np.uint32) == 0).all()
    zeros = np.float32([0.0, -0.0])[np.array(list(itertools.product(range(2), repeat=9)))]
    for got in (ptk.mad_network_host(zeros), oracle.mad_procedure(zeros)):
        assert (got.view(np.uint32) == 0).all()
