"""Comparison helpers shared by the oracle tests (CPU) and the parity tests (GPU)."""
import numpy as np


def canonical(ll):
    """eig() row order is implementation-defined: sort by frequency, then 1/T2."""
    ll = np.asarray(ll)
    if ll.size == 0:
        return ll.reshape(0, 4)
    with np.errstate(all="ignore"):
        return ll[np.lexsort((1.0 / ll[:, 1], ll[:, 2]))]


def keep_mask(ll, tol=1e-6):
    return (ll[:, 0] > tol) & (ll[:, 1] > 0)


def match_to_truth(ll, truth, amp_floor=1e-4):
    """Pick the lines matching the analytic peaks (reference test_kbdm.py:26-29 does the same)."""
    sel = ll[ll[:, 0] > amp_floor]
    return sel[np.argsort(sel[:, 2])], truth[np.argsort(truth[:, 2])]


def assert_lines_close(got, want, rel=1e-8, phase_abs=1e-8, what=""):
    """(A, T2, F) relative, phase absolute (mod 2*pi)."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    for c, name in ((0, "A"), (1, "T2"), (2, "F")):
        err = np.abs(got[:, c] - want[:, c]) / np.maximum(np.abs(want[:, c]), 1e-300)
        assert err.max() <= rel, f"{what}: {name} rel err {err.max():.3e} > {rel} at row {err.argmax()}"
    dph = np.angle(np.exp(1j * (got[:, 3] - want[:, 3])))
    assert np.abs(dph).max() <= phase_abs, f"{what}: phase err {np.abs(dph).max():.3e} > {phase_abs}"


def genuine_rows(canon_ll, truth, freq_tol=0.5):
    """Rows of a canonical line list that sit on a true peak (by frequency) with the largest amplitude."""
    rows = []
    for a, t2, f, ph in truth:
        cand = np.where(np.abs(canon_ll[:, 2] - f) < freq_tol)[0]
        assert cand.size, f"no line near {f} Hz"
        rows.append(cand[np.argmax(canon_ll[cand, 0])])
    return canon_ll[np.array(rows)]


def resolved_genuine_rows(want, truth, freq_tol=0.5):
    """Row indices (into a canonical line list) of the true peaks that the REFERENCE side resolves: for every true
    frequency the line of largest amplitude within `freq_tol` Hz, if there is one (weak or broad peaks drop out at
    small m, under heavy noise or with q > 0).  Canonical lists of equal length correspond row by row, so the same
    indices address the other side."""
    rows = []
    for f in np.asarray(truth)[:, 2]:
        cand = np.where(np.abs(want[:, 2] - f) < freq_tol)[0]
        if cand.size:
            rows.append(int(cand[np.argmax(want[cand, 0])]))
    return np.array(sorted(set(rows)), dtype=np.int64)
