"""Measured margins of the TRSV parity grid (test infrastructure).

Every TRSV parity check calls `check(key, err, blanket)`: the error is kept as
the maximum seen for `key` = (kernel family, Ar, St, uplo, oracle level), and
compared with 4 x the maximum that `profiles/r05_trsv_parity_margins.json`
(tracked; written by an earlier run of the same tests on an MI355X) holds for
that key - never with more than `blanket`, SURVEY.md 8d D4's starting value
x 100 (a triangular solve amplifies rounding by the factor's condition number),
which is also what a key without a recorded maximum is held to. The kernels are
deterministic (same bits on every run and box), so 4 x is slack for changes of
the summation order, not for noise. At the end of a GPU session the maxima of
THAT session are written to gpurun_out/trsv_parity_margins.json (conftest.py);
copying that file over the tracked one is how the margins are refreshed.
"""
import json
import os

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
TRACKED = os.path.join(REPO, "profiles", "r05_trsv_parity_margins.json")
SESSION = os.path.join(REPO, "gpurun_out", "trsv_parity_margins.json")
FACTOR = 4.0
NAMES = {0: "f16", 1: "f32", 2: "f64"}

_seen = {}
_tracked = None


def _recorded():
    global _tracked
    if _tracked is None:
        _tracked = {}
        if os.path.exists(TRACKED):
            with open(TRACKED) as f:
                _tracked = json.load(f).get("max_rel_error", {})
    return _tracked


def key(family, ar, st, upper, level):
    return f"{family}|Ar={NAMES[ar]}|St={NAMES[st]}|{'upper' if upper else 'lower'}|{level}"


def tolerance(k, blanket):
    recorded = _recorded().get(k)
    if recorded is None or recorded <= 0.0:
        return blanket
    return min(blanket, FACTOR * recorded)


def check(k, err, blanket):
    """Keeps `err` for the session's record and asserts the tolerance."""
    _seen[k] = max(_seen.get(k, 0.0), float(err))
    tol = tolerance(k, blanket)
    assert err <= tol, (k, err, tol)


def dump():
    if not _seen:
        return
    os.makedirs(os.path.dirname(SESSION), exist_ok=True)
    out = {"metric": "max over the parity grid (sizes x diag) of sum|oracle - hip| / sum|oracle| "
                     "per kernel family | Ar | St | uplo | oracle level (exact_trsv = semantic "
                     "oracle, ref_trsv = order-faithful emulation of the reference's schedule)",
           "tolerance_rule": f"min(100 x SURVEY D4, {FACTOR:g} x the maximum recorded here)",
           "max_rel_error": dict(sorted(_seen.items()))}
    with open(SESSION, "w") as f:
        json.dump(out, f, indent=1)
