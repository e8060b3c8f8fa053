"""CPU tests of the CBS oracle (oracle/cbs_oracle.py) and of the host-only part of the product's CBS
(the sequential boundary, wcx_cbs_getbdry).  DNAcopy itself is not available (SURVEY.md §8c shim
2): the oracle is pinned on the one DNAcopy run the reference ships (last test) and otherwise
checked against the MATHEMATICS it restates: the tail
approximation against a Monte-Carlo estimate, the sequential boundary against its defining error
probability, the permutation stream against uniformity."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import cbs_oracle as CO


@pytest.fixture(scope="module")
def bdry():
    g = np.load(os.path.join(GOLDEN, "cbs_bdry.npz"))
    assert float(g["eta"]) == 0.05 and int(g["nperm"]) == 10000 and int(g["max_ones"]) == 101
    return g["table"]


def test_feistel_is_a_keyed_bijection_and_looks_uniform():
    for n in (4, 5, 17, 200, 201, 1000, 4097):
        key = CO.test_key(7, 3, 0, n, 0)
        p = CO.feistel_perm(n, key, 11)
        assert sorted(p.tolist()) == list(range(n))
        assert not np.array_equal(p, CO.feistel_perm(n, key, 12))
        assert not np.array_equal(p, CO.feistel_perm(n, CO.test_key(8, 3, 0, n, 0), 11))
    # every position receives every element about equally often
    n, reps = 50, 4000
    key = CO.test_key(1, 0, 0, n, 0)
    counts = np.zeros((n, n))
    for p in range(reps):
        counts[np.arange(n), CO.feistel_perm(n, key, p)] += 1
    chi2 = ((counts - reps / n) ** 2 / (reps / n)).sum()          # ~ chi2 with (n-1)^2 = 2401 dof
    assert 2401 - 5 * np.sqrt(2 * 2401) < chi2 < 2401 + 5 * np.sqrt(2 * 2401)


def test_boundary_fixture_is_the_oracles_own_derivation(bdry):
    """The first 20 blocks re-derived here (scipy hypergeometric CDF) equal the fixture; the
    derivation is sequential in the block index, so this pins the generator."""
    again = CO.getbdry(0.05, 10000, 20)
    assert again == bdry[:len(again)].tolist()
    assert bdry[0] == 9500                                         # nperm - int(nperm eta)


def test_product_boundary_equals_oracle(bdry):
    """wcx_cbs_getbdry (C++, pmf recurrence) == the oracle's table (scipy) for max.ones = 101, i.e.
    every alpha <= 0.01 at nperm = 10000 -- 5151 stopping points."""
    from wisecondorx_amd import _lib
    lib = _lib.load()
    out = np.zeros(len(bdry), dtype=np.int32)
    _lib.check(lib.wcx_cbs_getbdry(0.05, 10000, 101, out.ctypes.data))
    assert np.array_equal(out, bdry)
    for j in (1, 2, 11, 50, 101):                                  # blocks ascend and end by nperm
        b = out[j * (j - 1) // 2: j * (j + 1) // 2]
        assert np.all(np.diff(b) > 0) and b[-1] <= 10000


@pytest.mark.parametrize("j", [1, 2, 5, 11])
def test_boundary_stops_a_just_not_significant_test_with_probability_eta(bdry, j):
    """[VO07] section 2.2: a test with exactly j exceedances among the 10000 permutations (just NOT
    significant at budget j - 1) must be declared significant early with probability ~ eta = 0.05."""
    rng = np.random.default_rng(j)
    block = bdry[j * (j - 1) // 2: j * (j + 1) // 2]
    trials, early = 20000, 0
    for _ in range(trials):
        pos = np.sort(rng.choice(10000, size=j, replace=False)) + 1   # 1-based permutation numbers
        # significant early <=> for some r < j: fewer than r + 1 exceedances by permutation block[r]
        early += any(np.searchsorted(pos, block[r], side="right") <= r for r in range(j))
    assert 0.035 < early / trials < 0.065


def test_tail_probability_against_monte_carlo():
    """What tailp() IS, measured: for Gaussian noise (known variance) the probability that the
    maximum over arcs holding a fraction in [delta, 1 - delta] of the points of the ONE-SIDED
    statistic exceeds b -- (1/4) b^3 phi(b) int nu^2 / (t (1 - t))^2 over [delta, 1/2], Siegmund's
    approximation as DNAcopy's tailp codes it (recalled: constant 9.973557e-2, half range).  The
    TWO-SIDED statistic CBS maximises (|Z|, here d^2) exceeds b twice as often in the tail: the
    simulation pins that factor, so the formula is neither mis-scaled nor mis-integrated."""
    rng = np.random.default_rng(5)
    n, reps, kmin = 300, 4000, 26
    x = rng.normal(size=(reps, n))
    x -= x.mean(axis=1, keepdims=True)
    S = np.concatenate((np.zeros((reps, 1)), np.cumsum(x, axis=1)), axis=1)
    best = np.zeros(reps)
    for a in range(kmin, n - kmin + 1):
        d = S[:, a:] - S[:, :-a]
        best = np.maximum(best, (d * d).max(axis=1) / (a * (n - a) / n))
    t = np.sqrt(best)
    for b in (3.5, 4.0):
        emp = float((t >= b).mean())
        approx = CO.tailp(b, kmin / n, n)
        assert 0.85 < 2.0 * approx / emp < 1.5, (b, emp, approx)


def test_cheap_lower_bound_of_nu():
    """cbs_seg.hip skips the nu(x) series of clearly-null segments with nu(x) >= exp(-0.583 x) / 2:
    checked against the series on a dense grid (the bound is in fact within [0.5, 0.62] of nu up to
    x = 3 and ever looser beyond, where nu ~ 2 / x^2 decays polynomially)."""
    import math
    for x in np.concatenate((np.linspace(0.0101, 3.0, 120), np.linspace(3.0, 40.0, 75))):
        assert 0.5 * math.exp(-0.583 * x) <= CO.nu(float(x)) <= 1.0 + 1e-12, x


def test_hybrid_statistic_covers_exactly_the_short_arcs():
    rng = np.random.default_rng(2)
    n = 230
    w = rng.uniform(0.5, 2.0, n)
    rw = np.sqrt(w)
    y = rng.normal(size=n) * rw
    Wp = np.concatenate(([0.0], np.cumsum(w)))
    perm = np.stack([CO.feistel_perm(n, 99, p) for p in range(3)])
    got = CO.perm_stats(y, rw, Wp, 5000.0, perm, True)
    guarded = CO.perm_stats(y, rw, Wp, 1.0, perm, True)     # tss <= bss + 1e-4 -> bss + 1
    W = Wp[-1]
    for q in range(3):
        v = rw * y[perm[q]]
        T = v.sum()
        v = v - T / W * w                                      # re-centred on its weighted mean
        S = np.concatenate(([0.0], np.cumsum(v)))
        best = 0.0
        for i in range(n):
            for j in range(i + 2, n + 1):
                a = j - i
                if n - a < 2 or not (a <= 25 or n - a <= 25):
                    continue
                wa = Wp[j] - Wp[i]
                best = max(best, (S[j] - S[i]) ** 2 / (wa * (W - wa) / W))
        assert 1.0 < best < 5000.0
        tp = 5000.0 - T * T / W
        np.testing.assert_allclose(got[q], best / ((tp - best) / (n - 2.0)), rtol=1e-10)
        np.testing.assert_allclose(guarded[q], best * (n - 2.0), rtol=1e-10)


def test_sequential_rule():
    block = [10, 20, 30]                                           # budget 2
    sig, nrej, np_ = CO.sequential_decision(iter([0] * 100), 2, block, 100)
    assert (sig, nrej, np_) == (True, 0, 10)
    sig, nrej, np_ = CO.sequential_decision(iter([1, 0, 0, 1] + [0] * 96), 2, block, 100)
    assert (sig, nrej, np_) == (True, 2, 30)
    sig, nrej, np_ = CO.sequential_decision(iter([1, 1, 1] + [0] * 97), 2, block, 100)
    assert (sig, nrej, np_) == (False, 3, 3)
    sig, nrej, np_ = CO.sequential_decision(iter([0] * 100), 0, [1000], 100)   # boundary beyond nperm
    assert (sig, nrej, np_) == (True, 0, 100)


def test_oracle_segments_planted_changes_and_leaves_noise_alone(bdry):
    CO.load_boundary_table(bdry)
    rng = np.random.default_rng(4)
    x = rng.normal(0, 0.05, 700)
    w = rng.uniform(0.5, 2.0, 700)
    assert CO.changepoints(x, w, 1e-4, 3, 0) == [700]
    x[250:300] += 0.12
    x[520:] -= 0.1
    tr = []
    ends = CO.changepoints(x, w, 1e-4, 3, 0, tr)
    assert len(ends) == 4 and abs(ends[0] - 250) <= 2 and abs(ends[1] - 300) <= 2 and abs(ends[2] - 520) <= 2
    assert any(r.get("why") == "t>=7" for r in tr)
    # the shortcuts do not change the outcome here
    assert CO.changepoints(x, w, 1e-4, 3, 0, strict=True) == ends
    # constant and tiny series
    assert CO.changepoints(np.full(50, 0.3), np.ones(50), 1e-2, 1, 0) == [50]
    assert CO.changepoints(np.array([0.1, 0.5, 0.2]), np.ones(3), 1e-2, 1, 0) == [3]


def example_case():
    """The reference's shipped DNAcopy run (tests/golden/example_bed.npz, see make_golden.py example):
    per-chromosome ratios in CBS.R's input convention (0 = blacklisted), unit weights (the run's
    weights are not shipped), and the segments it produced."""
    g = np.load(os.path.join(GOLDEN, "example_bed.npz"))
    off = np.concatenate(([0], np.cumsum(g["bins_per_chr"])))
    r = np.nan_to_num(g["ratios"], nan=0.0)
    results_r = [r[off[c]:off[c + 1]] for c in range(len(off) - 1)]
    results_w = [np.ones(len(v)) for v in results_r]
    return results_r, results_w, int(g["binsize"]), g["segments"]


def test_oracle_reproduces_the_references_shipped_dnacopy_segments(bdry):
    """THE pin against DNAcopy itself: docs/include/example.bed of the reference is a real run of
    CBS.R (DNAcopy, alpha = 1e-4, 100 kb bins, 30 321 bins of a trisomy-21 NIPT sample).  With unit
    weights the oracle reproduces every one of its 50 segments bin for bin: 44 chromosomes / arms
    left unsplit, the NA-gap splits of CBS.R:84-113, the change-points DNAcopy placed across short
    (<= 20-bin) NA gaps (chr1 143.3 | 145.3 Mb, chr2 92.1 | 94.1 Mb, chr8, chr16) and the
    chr21 13.1 Mb change-point of the trisomy.  The segment ratios agree to the 4 decimals printed
    only approximately (3e-3): that run's weights are unknown."""
    from oracle import wcx_oracle as O
    CO.load_boundary_table(bdry)
    results_r, results_w, binsize, want = example_case()
    got = O.cbs_r_wrapper(results_r, results_w, "F", 1e-4, binsize, 1, CO.cbs_segment)
    assert [tuple(s[:3]) for s in got] == [tuple(int(v) for v in s[:3]) for s in want]
    assert np.max(np.abs(np.array([s[3] for s in got]) - want[:, 3])) < 3e-3
    t21 = [s for s in got if s[0] == 20 and s[1] == 131][0]
    assert abs(t21[3] - 0.0923) < 2e-4                      # ID_aberrations.bed: 21 gain 0.0923
    # the outcome does not hinge on the permutation stream
    again = O.cbs_r_wrapper(results_r, results_w, "F", 1e-4, binsize, 77, CO.cbs_segment)
    assert [s[:3] for s in again] == [s[:3] for s in got]


def test_oracle_shipped_segments_under_random_weights(bdry):
    """The DNAcopy pin under weights: the shipped run's weights are unknown (unit weights reproduce its
    50 segments).  Eight random get_weights-like weight vectors (uniform 0.5 ... 2): 49 of the 50
    segments never move; only the chr21 stub's change-point (two bins before the centromere's NA gap)
    slides or merges away.  (The device runs the same check with 32 vectors:
    tests/test_gpu_cbs_oracle.py.)"""
    from oracle import wcx_oracle as O
    from test_gpu_cbs_oracle import weight_robustness
    CO.load_boundary_table(bdry)
    results_r, results_w, binsize, want = example_case()
    want3 = [tuple(int(v) for v in s[:3]) for s in want]
    rng = np.random.default_rng(2024)
    runs = []
    for _ in range(8):
        ws = [rng.uniform(0.5, 2.0, len(v)) for v in results_w]
        runs.append([tuple(s[:3]) for s in O.cbs_r_wrapper(results_r, ws, "F", 1e-4, binsize, 1, CO.cbs_segment)])
    exact, moved, bad = weight_robustness(runs, want3)
    assert not bad and exact >= 3 and exact + moved == 8
