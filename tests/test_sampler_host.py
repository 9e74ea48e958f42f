"""Host-side sampler pieces against a line-by-line Python restatement of the reference's Rust
(/root/reference/src/engine/sampling.rs:41-86,197-256,270-369,464-480; /root/reference/src/engine/mirostat.rs).  The reference has no tests for
these functions, so the restatement below IS the checker: it follows the Rust statement by statement in f32."""
import ctypes as C
import math

import numpy as np
import pytest

from blazr_amd import _lib as L

f32 = np.float32


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- Python restatements -------------------------------------------------------------------------------------------------------------------
def py_dynatemp(logits, base, rng, exponent):                      # sampling.rs:41-86
    mx = max(logits)
    probs = [f32(math.exp(f32(l - mx))) for l in logits]
    s = f32(0)
    for p in probs:
        s = f32(s + p)
    ent = f32(0)
    for p in probs:
        n = f32(p / s)
        if n > 0:
            ent = f32(ent - f32(n * f32(math.log(n))))
    max_ent = f32(math.log(f32(len(logits))))
    ne = min(max(f32(ent / max_ent), f32(0)), f32(1)) if max_ent > 0 else f32(0.5)
    mapped = f32(float(ne) ** float(exponent))
    return max(f32(f32(base) - f32(rng) + f32(f32(2.0) * f32(rng)) * mapped), f32(0.01))


def py_dry(logits, recent, multiplier, base, allowed):              # sampling.rs:270-320
    logits = list(logits)
    hist = recent[len(recent) - allowed:] if 0 < allowed < len(recent) else recent
    if len(hist) < base:
        return logits
    ss = max(len(hist) - (base - 1), 0)
    suffix = hist[ss:]
    for start in range(ss):
        end = start + len(suffix)
        if end >= len(hist):
            break
        if hist[start:end] == suffix:
            ml, s, e = len(suffix), start, ss
            while s > 0 and e > 0 and hist[s - 1] == hist[e - 1]:
                ml, s, e = ml + 1, s - 1, e - 1
            nt = hist[end]
            if nt < len(logits):
                logits[nt] = f32(logits[nt] - f32(multiplier) * f32(ml))
    return logits


def py_typical(logits, typical_p):                                  # sampling.rs:322-369
    mx = max(logits)
    probs = [f32(math.exp(f32(l - mx))) for l in logits]
    s = f32(0)
    for p in probs:
        s = f32(s + p)
    ent = f32(0)
    for p in probs:
        n = f32(p / s)
        if n > 0:
            ent = f32(ent - f32(n * f32(math.log(n))))
    dev = []
    for i, p in enumerate(probs):
        n = f32(p / s)
        info = f32(-math.log(n)) if n > 0 else f32(np.inf)
        dev.append((i, abs(f32(info - ent)), n))
    dev.sort(key=lambda t: t[1])                                    # stable, like Rust's sort_by
    keep, cum = set(), f32(0)
    for i, _, n in dev:
        if cum >= typical_p and cum > 0:
            break
        keep.add(i)
        cum = f32(cum + n)
    return keep


def py_logprobs(logits, chosen, top_n):                             # sampling.rs:197-256
    mx = max(logits)
    s = f32(0)
    for l in logits:
        s = f32(s + f32(math.exp(f32(l - mx))))
    lse = f32(f32(math.log(s)) + mx)
    lps = [f32(l - lse) for l in logits]
    n = min(top_n, len(logits), 20)
    order = sorted(range(len(logits)), key=lambda i: -lps[i])[:n]
    return lps[chosen], order, [lps[i] for i in order]


# ---- tests ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_dynamic_temperature(seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(257) * rng.uniform(0.2, 6)).astype(np.float32)
    for base, r, ex in ((0.8, 0.5, 1.0), (1.0, 0.9, 2.0), (0.3, 0.4, 0.5)):
        got = L.lib().bz_compute_dynamic_temperature(_p(x), len(x), base, r, ex)
        assert abs(got - py_dynatemp(list(x), base, r, ex)) <= 2e-5
        assert max(base - r, 0.01) - 1e-6 <= got <= base + r + 1e-6       # documented range [base - range, base + range], floor 0.01
    flat = np.zeros(64, np.float32)                                   # uniform distribution: normalised entropy 1 -> base + range
    assert abs(L.lib().bz_compute_dynamic_temperature(_p(flat), 64, 0.7, 0.2, 1.0) - 0.9) < 1e-5


def test_dry_penalty_known_answers():
    V = 50
    base = np.zeros(V, np.float32)
    # history ... 7 8 9 | 7 8 : the suffix (7, 8) occurred before and was followed by 9 -> logit[9] -= multiplier * match_len
    recent = np.array([1, 7, 8, 9, 3, 7, 8], dtype=np.uint32)
    x = base.copy()
    L.check(L.lib().bz_apply_dry_penalty(_p(x), V, _p(recent), len(recent), 0.8, 3, 0))
    want = np.zeros(V, np.float32)
    want[9] = -0.8 * 2
    assert np.array_equal(x, want)
    rng = np.random.default_rng(5)
    for _ in range(40):
        n = int(rng.integers(0, 40))
        recent = rng.integers(0, 6, size=n).astype(np.uint32)      # small alphabet: many repeated n-grams
        x = rng.standard_normal(V).astype(np.float32)
        b, allowed, mult = int(rng.integers(1, 5)), int(rng.choice([0, 5, 12, 100])), float(rng.uniform(0.1, 2))
        want = np.array(py_dry(list(x), recent.tolist(), mult, b, allowed), dtype=np.float32)
        L.check(L.lib().bz_apply_dry_penalty(_p(x), V, _p(recent), n, mult, b, allowed))
        assert np.array_equal(x, want)


@pytest.mark.parametrize("seed", range(6))
def test_typical_filter(seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(120) * 2).astype(np.float32)
    for tp in (0.2, 0.5, 0.9, 1.0):
        y = x.copy()
        L.check(L.lib().bz_apply_typical_filter(_p(y), len(y), tp))
        keep = py_typical(list(x), tp)
        got = {i for i in range(len(y)) if y[i] != -np.inf}
        assert got == keep and all(y[i] == x[i] for i in got)
        p = np.exp(x - x.max()); p /= p.sum()
        assert p[list(got)].sum() >= min(tp, 1.0) - 1e-5            # kept mass reaches typical_p


def test_logit_bias_and_logprobs():
    rng = np.random.default_rng(3)
    x = rng.standard_normal(300).astype(np.float32)
    ids = np.array([5, 7, 7, 999], dtype=np.uint32)                   # duplicate id: the later entry wins; out-of-vocab ignored (sampling.rs:472-475)
    bias = np.array([1.5, -2.0, 3.0, 100.0], dtype=np.float32)
    y = x.copy()
    L.check(L.lib().bz_apply_logit_bias(_p(y), len(y), _p(ids), _p(bias), 4))
    want = x.copy(); want[5] += 1.5; want[7] += 3.0
    assert np.array_equal(y, want)
    clp, tid, tlp, n = C.c_float(), np.zeros(20, np.uint32), np.zeros(20, np.float32), C.c_int()
    L.check(L.lib().bz_compute_logprobs(_p(x), len(x), 17, 5, C.byref(clp), _p(tid), _p(tlp), C.byref(n)))
    wl, order, lps = py_logprobs(list(x), 17, 5)
    assert n.value == 5 and tid[:5].tolist() == order and abs(clp.value - wl) <= 2e-6 and np.abs(tlp[:5] - np.array(lps)).max() <= 2e-6
    L.check(L.lib().bz_compute_logprobs(_p(x), len(x), 17, 50, C.byref(clp), _p(tid), _p(tlp), C.byref(n)))
    assert n.value == 20                                            # capped at 20 (sampling.rs:229)


def test_mirostat_v2():
    rng = np.random.default_rng(11)
    x = (rng.standard_normal(400) * 3).astype(np.float32)
    h = C.c_void_p()
    L.check(L.lib().bz_mirostat_create(3.0, 0.2, 42, C.byref(h)))
    assert abs(L.lib().bz_mirostat_mu(h) - 6.0) < 1e-6                # mu = 2 tau (mirostat.rs:27-33)
    p = np.exp((x - x.max()).astype(np.float64)); p /= p.sum()
    mus, toks = [], []
    for _ in range(300):
        mu_before = L.lib().bz_mirostat_mu(h)
        tok, lp = C.c_uint32(), C.c_float()
        L.check(L.lib().bz_mirostat_sample(h, _p(x), len(x), 1.0, C.byref(tok), C.byref(lp)))
        surprise = -math.log2(p[tok.value])
        assert surprise <= mu_before + 1e-3 or tok.value == int(p.argmax())     # truncation: surprise <= mu, or the top-1 fallback
        assert abs(lp.value - math.log(p[tok.value])) < 1e-4
        assert abs(L.lib().bz_mirostat_mu(h) - (mu_before - 0.2 * (surprise - 3.0))) < 1e-3   # mu -= eta (surprise - tau)
        mus.append(L.lib().bz_mirostat_mu(h)); toks.append(tok.value)
    assert len(set(toks)) > 5
    # the controller holds the observed surprise near tau
    assert abs(np.mean([-math.log2(p[t]) for t in toks[100:]]) - 3.0) < 0.6
    L.lib().bz_mirostat_free(h)
    # same seed -> same draws
    seqs = []
    for _ in range(2):
        L.check(L.lib().bz_mirostat_create(3.0, 0.2, 7, C.byref(h)))
        s = []
        for _ in range(20):
            tok = C.c_uint32()
            L.check(L.lib().bz_mirostat_sample(h, _p(x), len(x), 0.8, C.byref(tok), None))
            s.append(tok.value)
        seqs.append(s)
        L.lib().bz_mirostat_free(h)
    assert seqs[0] == seqs[1]
