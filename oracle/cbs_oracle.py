"""CPU ORACLE for SURVEY.md §8a row a16 (circular binary segmentation) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (wisecondorx_amd/) never does.

What this restates.  The reference delegates the segmentation to Bioconductor DNAcopy 1.76.0
(`segment(CNA.object, alpha=..., verbose=1, weights=...)`, /root/reference/src/wisecondorx/include/
CBS.R:70-73, called from predict_tools.py:242-257 / main.py:279; version pin conda.yml:14).  DNAcopy's
source is NOT under /root/reference and R cannot be installed here, so there is nothing to run.
PIN: the ONE DNAcopy output the reference repository ships -- docs/include/example.bed, a real 100 kb
NIPT trisomy-21 run: 30 321 bin ratios in, 50 segments out -- is reproduced bin for bin, all 50
segments (tests/test_oracle_cbs.py::test_oracle_reproduces_the_references_shipped_dnacopy_segments,
fixture tests/golden/example_bed.npz made by tests/golden/make_golden.py example).  That run's
weights are not shipped (unit weights are used), and one sample exercises few borderline decisions,
so beyond that example the breakpoints remain PARITY UNPINNED.  This file is an independent plain
NumPy statement of the algorithm DNAcopy implements, written from

  [O04]  Olshen, Venkatraman, Lucito, Wigler, "Circular binary segmentation for the analysis of
         array-based DNA copy number data", Biostatistics 5 (2004): the max-arc t statistic, the
         permutation reference distribution, the recursion;
  [VO07] Venkatraman, Olshen, "A faster circular binary segmentation algorithm for the analysis of
         array CGH data", Bioinformatics 23 (2007): the hybrid p-value (tail approximation for long
         arcs, Siegmund 1988 / Yao 1989 + permutations of the short-arc maximum), the sequential
         stopping boundary for the permutations, the edge-effect (two-sample) tests of a ternary split;

and, for everything the papers leave open (weights, constants, order of the decisions), from the
structure of DNAcopy's R / Fortran code AS RECALLED by the author (segment(), changepoints(),
wfindcpt, wtmaxo, whtmaxp, wtpermp, tailp / nu / it1tsq, getbdry / etabdry / pexceed, getmncwt).
Each such item is marked [DNAcopy, recalled] below: it cannot be verified offline.  DNAcopy defaults
used by CBS.R's call: nperm = 10000, p.method = "hybrid", min.width = 2, kmax = 25, nmin = 200,
eta = 0.05, undo.splits = "none" (trim only feeds the undo step, which is off).

The ONE deliberate, unavoidable difference from DNAcopy: the random permutations.  DNAcopy draws
Fisher-Yates shuffles from R's Mersenne-Twister after set.seed(seed) (CBS.R:67-69), one stream for
the whole run.  A data-parallel implementation needs a counter-based stream, so the SPECIFICATION
shared by this oracle and the device code (wisecondorx_amd/csrc/cbs_seg.hip) -- shared as a
specification, written twice independently -- is:

  key of a test   K = mix64(mix64(mix64(mix64(seed) ^ chrom) ^ (lo << 32 | hi)) ^ kind)
                  chrom = 0-based chromosome index, [lo, hi) = the tested segment in the NA-free
                  series of that chromosome, kind = 0 segmentation test, 1 / 2 left / right edge test;
                  mix64 = the splitmix64 finaliser (constants in mix64() below);
  permutation p   s0 = mix64(K ^ (p * 0xd1342543de82ef95 mod 2^64)), s1 = mix64(s0 + 1),
                  round keys rk = lo32(s0), hi32(s0), lo32(s1), hi32(s1);
  bijection       of [0, n): a = ceil(sqrt(n)); x -> (L, R) = divmod(x, a); four Feistel rounds
                  (L, R) <- (R, (L + floor(fmix32(R + rk[r] mod 2^32) * a / 2^32)) mod a),
                  fmix32 = murmur3's 32-bit finaliser; x' = L a + R; repeat while x' >= n (cycle
                  walking).  The permuted series has element pi(i) at position i.

The result therefore depends on (seed, chromosome, segment) only -- not on the sample's position in
a batch, not on scheduling.
"""
import math

import numpy as np

M64 = (1 << 64) - 1
NPERM, KMAX, NMIN, MINW, ETA, NGRID = 10000, 25, 200, 2, 0.05, 100


# --------------------------------------------------------------------------- permutation stream

def mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def test_key(seed, chrom, lo, hi, kind):
    k = mix64(int(seed) & M64)
    k = mix64(k ^ int(chrom))
    k = mix64(k ^ ((int(lo) << 32) | int(hi)))
    return mix64(k ^ int(kind))


def _fmix32(h):
    h = h ^ (h >> np.uint64(16))
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h = h ^ (h >> np.uint64(13))
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    return h ^ (h >> np.uint64(16))


def feistel_perm(n, key, p):
    """pi[0..n): the p-th permutation of the test with key `key` (see the module docstring)."""
    a = int(math.isqrt(n))
    if a * a < n:
        a += 1
    s0 = mix64(key ^ ((p * 0xD1342543DE82EF95) & M64))
    s1 = mix64((s0 + 1) & M64)
    rk = [s0 & 0xFFFFFFFF, s0 >> 32, s1 & 0xFFFFFFFF, s1 >> 32]
    ua = np.uint64(a)
    x = np.arange(n, dtype=np.uint64)
    todo = np.arange(n)
    out = np.empty(n, dtype=np.int64)
    while todo.size:
        L, R = x // ua, x % ua
        for r in range(4):
            h = _fmix32((R + np.uint64(rk[r])) & np.uint64(0xFFFFFFFF))
            t = (L + ((h * ua) >> np.uint64(32))) % ua
            L, R = R, t
        x = L * ua + R
        done = x < np.uint64(n)
        out[todo[done]] = x[done].astype(np.int64)
        todo, x = todo[~done], x[~done]
    return out


# --------------------------------------------------------------------------- sequential boundary

def _etabdry(nperm, eta0, n1s):
    """[DNAcopy, recalled: etabdry] boundary for a test that is just NOT significant with n1s
    exceedances in nperm permutations: ibdry[r] = the smallest number of permutations i at which
    seeing at most r exceedances has probability <= eta0 when the n1s exceedances are scattered
    uniformly over the nperm permutations (hypergeometric; [VO07] section 2.2)."""
    from scipy.stats import hypergeom
    out = []
    i = 1
    for r in range(n1s):
        # P(X_i <= r), X_i ~ Hypergeom(population nperm, n1s marked, i drawn): decreasing in i
        lo_i, hi_i = i, nperm
        # smallest i >= current i with cdf <= eta0 (it exists: the cdf at i = nperm is 0 for r < n1s)
        while lo_i < hi_i:
            mid = (lo_i + hi_i) // 2
            if hypergeom.cdf(r, nperm, n1s, mid) <= eta0:
                hi_i = mid
            else:
                lo_i = mid + 1
        out.append(lo_i)
        i = lo_i      # the Fortran loop tests the next level at the next i at the earliest
        i += 1
    return out


def _pexceed(nperm, n1s, b):
    """[DNAcopy, recalled: pexceed] (approximate) probability that the exceedance path of a test with
    exactly n1s exceedances stays below the boundary b[0..n1s) somewhere, i.e. is stopped early.
    DNAcopy sums exp(lchoose ...) terms in double precision; the search in getbdry() divides by
    differences of two such values ~1e-5 apart, so the last digits of lgamma decide single stopping
    points.  Here the binomial terms are evaluated EXACTLY (integers) and rounded once."""
    from fractions import Fraction
    from math import comb
    tot = comb(nperm, n1s)
    num = comb(nperm - b[0], n1s)
    if n1s >= 2:
        num += b[0] * comb(nperm - b[1], n1s - 1)
    if n1s >= 3:
        t = comb(nperm - b[2], n1s - 2)
        num += (b[0] * (b[0] - 1) // 2) * t
        num += b[0] * (b[1] - b[0]) * t
    for i in range(4, n1s + 1):
        n1, n2, n3 = b[i - 4], b[i - 3], b[i - 2]
        t = comb(nperm - b[i - 1], n1s - i + 1)
        num += comb(n1, i - 1) * t
        num += comb(n1, i - 2) * (n3 - n1) * t
        num += comb(n1, i - 3) * (n2 - n1) * (n3 - n2) * t
        num += comb(n1, i - 3) * ((n2 - n1) * (n2 - n1 - 1) // 2) * t
    return float(Fraction(num, tot))


def getbdry(eta, nperm, max_ones, tol=1e-2):
    """[DNAcopy, recalled: getbdry] the triangular boundary table: block j (1-based, length j, at
    offset j (j - 1) / 2) is used by a test that tolerates j - 1 exceedances.  Block 1 is
    nperm - int(nperm eta); for block j the per-level error eta0 is searched (regula falsi between
    1.1 and 0.25 times the previous block's value, relative tolerance tol) so that the probability
    of stopping a just-not-significant test early is eta."""
    bdry = [nperm - int(nperm * eta)]
    eta0 = eta
    for j in range(2, max_ones + 1):
        etahi = eta0 * 1.1
        b = _etabdry(nperm, etahi, j)
        phi = _pexceed(nperm, j, b)
        etalo = eta0 * 0.25
        b = _etabdry(nperm, etalo, j)
        plo = _pexceed(nperm, j, b)
        it = 0
        while (etahi - etalo) / etalo > tol and it < 200 and phi != plo:
            it += 1
            eta0 = etalo + (etahi - etalo) * (eta - plo) / (phi - plo)
            b = _etabdry(nperm, eta0, j)
            pexcd = _pexceed(nperm, j, b)
            if pexcd > eta:
                etahi, phi = eta0, pexcd
            else:
                etalo, plo = eta0, pexcd
        bdry.extend(b)
    return bdry


_BDRY_CACHE = {}


def boundary_block(alpha, nrejc, nperm=NPERM, eta=ETA):
    """The nrejc + 1 stopping points of a test with budget nrejc ([DNAcopy, recalled: segment():
    max.ones = floor(nperm alpha) + 1; sbdry = getbdry(eta, nperm, max.ones); wfindcpt starts at
    k = nrejc (nrejc + 1) / 2 + 1)."""
    max_ones = int(math.floor(nperm * alpha)) + 1
    key = (eta, nperm, max_ones)
    if key not in _BDRY_CACHE:
        _BDRY_CACHE[key] = getbdry(eta, nperm, max_ones)
    t = _BDRY_CACHE[key]
    o = nrejc * (nrejc + 1) // 2
    return t[o:o + nrejc + 1]


# --------------------------------------------------------------------------- tail probability

def _phi(x):
    return 0.5 * math.erfc(-x / math.sqrt(2.0))


def nu(x):
    """Siegmund's nu(x) = 2 x^-2 exp(-2 sum_k Phi(-x sqrt(k) / 2) / k) ([VO07] eq. for the tail
    approximation; DNAcopy's nu() truncates the series at a relative step of 1e-6, here it is
    summed until the terms vanish in double precision)."""
    if x <= 0.01:
        return math.exp(-0.583 * x)
    kk = int((17.0 / x) ** 2) + 1
    kk = min(kk, 40000000)
    k = np.arange(1, kk + 1, dtype=np.float64)
    from scipy.special import erfc
    s = float(np.sum(0.5 * erfc(x * np.sqrt(k) * 0.5 / math.sqrt(2.0)) / k))
    return math.exp(math.log(2.0) - 2.0 * math.log(x) - 2.0 * s)


def _it1tsq(x, a):
    """integral of (t (1 - t))^-2 over [x, x + a]."""
    def f(t):
        y = t - 0.5
        return 8.0 * y / (1.0 - 4.0 * y * y) + 2.0 * math.log((1.0 + 2.0 * y) / (1.0 - 2.0 * y))
    return f(x + a) - f(x)


def tailp(b, delta, m, ngrid=NGRID):
    """P(max over arcs holding a fraction in [delta, 1 - delta] of the points of the binary-
    segmentation statistic >= b) for m Gaussian points: b^3 phi(b) int nu(b / sqrt(m t (1 - t)))^2
    / (t (1 - t))^2 dt over [delta, 1/2] (doubled by symmetry), midpoint rule in nu on ngrid cells
    with the rational factor integrated exactly ([VO07]; [DNAcopy, recalled: tailp])."""
    dincr = (0.5 - delta) / ngrid
    bsqrtm = b / math.sqrt(m)
    tl = 0.5 - dincr
    acc = 0.0
    for i in range(ngrid):
        t = 0.5 - 0.5 * dincr - i * dincr
        x = bsqrtm / math.sqrt(t * (1.0 - t))
        v = nu(x)
        acc += v * v * _it1tsq(tl, dincr)
        tl -= dincr
    return 9.973557e-2 * b ** 3 * math.exp(-b * b / 2.0) * acc


# --------------------------------------------------------------------------- statistics

def _prefix(v):
    return np.concatenate(([0.0], np.cumsum(v)))


def max_arc(x, w, minw=MINW):
    """Observed statistic ([O04] eq. 2 with weights; [DNAcopy, recalled: wtmaxo]): over all arcs
    (i, j], minw <= j - i <= n - minw, of the series centred on its weighted mean,
    bss = (S_j - S_i)^2 / (w_a (W - w_a) / W), S / w_a = weighted partial sums / arc weight.
    Returns (t^2 = bss / ((tss - bss) / (n - 2)), i, j, tss); ties -> smallest (i, j).
    [DNAcopy, recalled]: tss <= bss + 1e-4 is replaced by bss + 1."""
    n = len(x)
    W = float(np.sum(w))
    mean = float(np.sum(w * x)) / W
    c = x - mean
    tss = float(np.sum(w * c * c))
    S, Wp = _prefix(w * c), _prefix(w)
    best = (-1.0, 0, 0)
    for a in range(minw, n - minw + 1):
        d = S[a:] - S[:-a]
        wa = Wp[a:] - Wp[:-a]
        b = d * d / (wa * (W - wa) / W)
        i = int(np.argmax(b))
        bv = float(b[i])
        if bv > best[0] or (bv == best[0] and (i, i + a) < (best[1], best[2])):
            best = (bv, i, i + a)
    bss, bi, bj = best
    t = tss
    if t <= bss + 1e-4:
        t = bss + 1.0
    return bss / ((t - bss) / (n - 2.0)), bi, bj, tss


def _stat_from_bss(bss, tss, n):
    t = np.where(tss <= bss + 1e-4, bss + 1.0, tss)
    return bss / ((t - bss) / (n - 2.0))


def perm_stats(y, rw, Wp, tss, perms, hybrid, minw=MINW, kmax=KMAX):
    """Statistic of a block of permuted series ([VO07] section 2.1; [DNAcopy, recalled: wxperm +
    whtmaxp / wtmaxp]).  y = sqrt(w) (x - mean) is what is exchangeable under H0 when the variance
    of point i is sigma^2 / w_i; the permuted value at position i is y[pi(i)] / sqrt(w_i), so its
    weighted values are v_i = sqrt(w_i) y[pi(i)].  Unlike the unweighted case their total T is not
    0, and an arc and its complement only carry the same statistic for a centred series: the
    permuted series is re-centred on ITS weighted mean T / W (v_i - (T / W) w_i) and its total sum of
    squares is tss - T^2 / W (sum v_i^2 / w_i = sum y^2 = tss is permutation invariant).  [Whether
    DNAcopy's whtmaxp re-centres is not recalled with certainty; without it the arcs whose
    complement is short pick up T^2, which grows with n var(sqrt w) and would swamp the statistic.]
    hybrid: only arcs with at most kmax points, or whose complement has at most kmax points;
    otherwise all arcs."""
    n = len(y)
    W = Wp[-1]
    w = np.diff(Wp)
    V = rw[None, :] * y[perms]                       # (P, n)
    T = V.sum(axis=1)
    V = V - (T / W)[:, None] * w[None, :]
    S = np.concatenate((np.zeros((V.shape[0], 1)), np.cumsum(V, axis=1)), axis=1)
    bmax = np.zeros(V.shape[0])
    if hybrid:
        a_hi = min(kmax, n - minw)
        lens = list(range(minw, a_hi + 1)) + list(range(max(n - kmax, a_hi + 1), n - minw + 1))
    else:
        lens = range(minw, n - minw + 1)
    for a in lens:
        d = S[:, a:] - S[:, :-a]
        wa = Wp[a:] - Wp[:-a]
        b = d * d / (wa * (W - wa) / W)[None, :]
        bmax = np.maximum(bmax, b.max(axis=1))
    return _stat_from_bss(bmax, tss - T * T / W, n)


def weighted_delta(Wp, kmax=KMAX):
    """[DNAcopy, recalled: getmncwt] with weights the short-arc limit of the tail approximation is
    the smallest weight fraction of any arc of kmax + 1 points (wrap-around arcs included);
    (kmax + 1) / n for unit weights."""
    n = len(Wp) - 1
    j = kmax + 1
    W = Wp[-1]
    m = np.min(Wp[j:] - Wp[:-j])
    nmj = n - j                                       # wrap-around arcs = complements of arcs of n - j points
    if nmj >= 1:
        m = min(m, float(np.min(W - (Wp[nmj:] - Wp[:-nmj]))))
    return float(m) / W


# --------------------------------------------------------------------------- one test (wfindcpt)

def sequential_decision(exceed_iter, nrejc, block, nperm=NPERM):
    """[VO07] section 2.2 / [DNAcopy, recalled: wfindcpt loop].  exceed_iter yields the exceedance
    indicator of permutation 1, 2, ...  Returns (significant, nrej, np_used)."""
    nrej = 0
    np_ = 0
    for e in exceed_iter:
        np_ += 1
        if e:
            nrej += 1
        if nrej > nrejc:
            return False, nrej, np_
        if np_ >= block[nrej]:
            return True, nrej, np_
        if np_ >= nperm:
            break
    return True, nrej, np_


def _exceed_stream(y, rw, Wp, tss, key, thr, hybrid, nperm, chunk=64):
    n = len(y)
    for p0 in range(0, nperm, chunk):
        ps = range(p0, min(nperm, p0 + chunk))
        perms = np.stack([feistel_perm(n, key, p) for p in ps])
        st = perm_stats(y, rw, Wp, tss, perms, hybrid)
        for v in st:
            yield bool(thr <= v)


def edge_test(xc, w, n1, n12, key, alpha, nperm=NPERM, strict=False):
    """Two-sample test of one change-point of a ternary split ([VO07] section 2.3; [DNAcopy,
    recalled: wtpermp]).  xc = the n12 points of both sides, centred on the mean of the SEGMENT
    under test (not re-centred here).  Statistic: |weighted mean of the shorter side - weighted mean
    of all n12|; a permutation puts sqrt(w) x [pi(i)] at position i and sums sqrt(w_i) * that over
    the LAST m1 positions, divided by the shorter side's weight [DNAcopy, recalled -- including that
    the last m1 positions are used whichever side is shorter].  t^2 > 25 with m1 >= 10 is accepted
    without permutations [DNAcopy, recalled]; strict=True runs them anyway.
    Returns (keep, nrej or -1, tstat)."""
    n2 = n12 - n1
    if n1 == 1 or n2 == 1:
        return False, nperm, 0.0
    rw = np.sqrt(w)
    rn1, rn2 = float(np.sum(w[:n1])), float(np.sum(w[n1:]))
    xs1, xs2 = float(np.sum(w[:n1] * xc[:n1])), float(np.sum(w[n1:] * xc[n1:]))
    rn = rn1 + rn2
    xbar = (xs1 + xs2) / rn
    tss = float(np.sum(w * xc * xc)) - rn * xbar * xbar
    if n1 <= n2:
        m1, rm1, dm = n1, rn1, abs(xs1 / rn1 - xbar)
        tstat = dm * dm * rn1 * rn / rn2
    else:
        m1, rm1, dm = n2, rn2, abs(xs2 / rn2 - xbar)
        tstat = dm * dm * rn2 * rn / rn1
    ostat = 0.99999 * dm
    tstat = tstat / ((tss - tstat) / (n12 - 2.0))
    if tstat > 25.0 and m1 >= 10 and not strict:
        return True, -1, tstat
    yy = xc * rw
    tail = np.arange(n12 - m1, n12)
    nrej = 0
    for p in range(nperm):
        pi = feistel_perm(n12, key, p)
        xsum = float(np.sum(rw[tail] * yy[pi[tail]]))
        if ostat <= abs(xsum / rm1 - xbar):
            nrej += 1
            if nrej / float(nperm) > alpha:
                return False, nrej, tstat            # cannot come back under alpha
    return nrej / float(nperm) <= alpha, nrej, tstat


def find_cpt(x, w, alpha, seed, chrom, lo, hi, trace=None, nperm=NPERM, strict=False):
    """One call of DNAcopy's wfindcpt on the segment [lo, hi) of a chromosome's NA-free series
    (x, w = that segment).  Returns the list of change-points (offsets inside the segment)."""
    n = len(x)
    rec = {"chr": chrom, "lo": lo, "hi": hi, "n": n, "cpt": []}
    if trace is not None:
        trace.append(rec)
    if n < 2 * MINW:
        rec["why"] = "short"
        return []
    if float(np.max(x) - np.min(x)) <= 1.4901161193847656e-08:     # isTRUE(all.equal(diff(range), 0))
        rec["why"] = "constant"
        return []
    hybrid = n > NMIN
    ostat_full, bi, bj, tss = max_arc(x, w)
    rec.update(ostat=ostat_full, bi=bi, bj=bj, tss=tss, hybrid=hybrid)
    ostat1 = math.sqrt(ostat_full)
    ostat = ostat_full * 0.99999
    W = float(np.sum(w))
    mean = float(np.sum(w * x)) / W
    xc = x - mean
    significant = None
    if ostat1 <= 0.1:
        rec["why"] = "t<=0.1"
        return []
    arc = min(bj - bi, n - bj + bi)
    if ostat1 >= 7.0 and arc >= 10 and not strict:
        significant = True
        rec["why"] = "t>=7"
    else:
        rw = np.sqrt(w)
        y = xc * rw
        Wp = _prefix(w)
        if hybrid:
            delta = weighted_delta(Wp)
            pval1 = tailp(ostat1, delta, n)
            rec.update(pval1=pval1, delta=delta)
            if pval1 > alpha:
                rec["why"] = "tailp"
                return []
            pval2 = alpha - pval1
        else:
            pval2 = alpha
        nrejc = int(pval2 * float(nperm))
        block = boundary_block(alpha, nrejc, nperm)
        key = test_key(seed, chrom, lo, hi, 0)
        significant, nrej, np_used = sequential_decision(
            _exceed_stream(y, rw, Wp, tss, key, ostat, hybrid, nperm), nrejc, block, nperm)
        rec.update(nrejc=nrejc, nrej=nrej, np=np_used, why="perm")
        if not significant:
            return []
    rec["significant"] = True
    if bj == n:
        cpt = [bi]
    elif bi == 0:
        cpt = [bj]
    else:
        cpt = []
        keep, nrej1, t1 = edge_test(xc[:bj], w[:bj], bi, bj, test_key(seed, chrom, lo, hi, 1), alpha,
                                    nperm, strict)
        if keep:
            cpt.append(bi)
        keep2, nrej2, t2 = edge_test(xc[bi:], w[bi:], bj - bi, n - bi, test_key(seed, chrom, lo, hi, 2),
                                     alpha, nperm, strict)
        if keep2:
            cpt.append(bj)
        rec["edge"] = [(bool(keep), nrej1), (bool(keep2), nrej2)]
    rec["cpt"] = list(cpt)
    return cpt


def changepoints(x, w, alpha, seed, chrom, trace=None, nperm=NPERM, strict=False):
    """[DNAcopy, recalled: changepoints()] the stack of segment ends; always the LAST segment is
    tested next ([O04] section 3: recursive application).  Returns the sorted segment ends."""
    n = len(x)
    seg_end = [0, n]
    change_loc = []
    while len(seg_end) > 1:
        lo, hi = seg_end[-2], seg_end[-1]
        cpt = []
        if hi - lo >= 2 * MINW:
            cpt = find_cpt(x[lo:hi], w[lo:hi], alpha, seed, chrom, lo, hi, trace, nperm, strict)
        if not cpt:
            change_loc.append(hi)
            seg_end.pop()
        else:
            seg_end[-1:-1] = [lo + c for c in cpt]
    return sorted(change_loc)


def cbs_segment(c, y, w, alpha, state):
    """segment_fn of wcx_oracle.cbs_r_wrapper: y with NaN = missing (DNAcopy drops non-finite rows,
    segment(): ina <- which(is.finite(genomdati))), w aligned; returns [(start, end)] 1-based
    inclusive bin positions (loc.start / loc.end in the x = 1-based bin index coordinate, CBS.R:49)."""
    ok = np.flatnonzero(np.isfinite(y))
    ends = changepoints(y[ok], w[ok], alpha, state.get("seed") or 0, c, state.get("trace"),
                        state.get("nperm", NPERM), state.get("strict", False))
    out, prev = [], 0
    for e in ends:
        out.append((int(ok[prev]) + 1, int(ok[e - 1]) + 1))
        prev = e
    return out


# --------------------------------------------------------------------------- helpers for the tests

def load_boundary_table(table, eta=ETA, nperm=NPERM):
    """Install a precomputed getbdry() table (tests/golden/cbs_bdry.npz, generated by THIS module's
    getbdry through tests/golden/make_golden.py bdry; tests/test_oracle_cbs.py re-derives its first
    blocks).  Saves the 40+ s the scipy-based derivation takes for alpha = 0.01."""
    table = [int(v) for v in table]
    max_ones = int(round((math.sqrt(8 * len(table) + 1) - 1) / 2))
    assert max_ones * (max_ones + 1) // 2 == len(table)
    _BDRY_CACHE[(eta, nperm, max_ones)] = table
    # the derivation is sequential in the block index: a longer table holds every shorter one
    for m in range(1, max_ones):
        _BDRY_CACHE.setdefault((eta, nperm, m), table[:m * (m + 1) // 2])


def segment_series(job):
    """Process-pool worker: (chrom, y with NaN, w, alpha, seed, strict, boundary table or None) ->
    (chrom, [(start, end)] 1-based inclusive, trace records)."""
    c, y, w, alpha, seed, strict, table = job
    if table is not None:
        load_boundary_table(table)
    trace = []
    segs = cbs_segment(c, np.asarray(y, dtype=float), np.asarray(w, dtype=float), alpha,
                       {"seed": seed, "trace": trace, "strict": strict})
    return c, segs, trace
