"""GPU: cooperative panels (llckbdm_amd/csrc/kb_team.hpp, kb_panel_team.hpp) - T workgroups share the panel kernels of the
blocked bidiagonalisation (reference kbdm.py:166) and of the blocked Hessenberg reduction (kbdm.py:192) of ONE member.
  * a member gets the same BITS from a team of 1, 2, 4, 8 workgroups (full pipeline through the C ABI, and the two stage
    entry points), with teams on lane 0 only and on every lane;
  * the team size a launch gets depends on the batch (count x T must fit the context's budget of resident workgroups), the
    results do not: a member solved alone == the same member inside a batch, bit for bit;
  * no member reports a status (a team that gives up sets SVD_NOCONV / EIG_NOCONV)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DWELL = 5e-4
FIELDS = ("lines", "sv", "mu", "keep", "status")


def _engine(monkeypatch, T, budget=256, all_lanes=0):
    from llckbdm_amd.engine import Engine
    monkeypatch.setenv("KBDM_PANEL_T", str(T))
    monkeypatch.setenv("KBDM_PANEL_BUDGET", str(budget))
    monkeypatch.setenv("KBDM_PANEL_T_ALL", str(all_lanes))
    e = Engine(0, in_flight=1)
    e.wide_solve = False          # (these tests set the team size of the pool's context themselves)
    return e


def _members():
    from llckbdm_amd import datasets
    sig, _, _ = datasets.config2(seed=7)
    # panels exist from m = 96 on; 400 has eleven of them; sizes around the chunk boundaries of the slot decomposition
    ms = np.array([400, 399, 385, 384, 383, 321, 320, 257, 256, 255, 193, 161, 129, 128, 127, 97, 96, 95, 64, 33], dtype=np.int32)
    return sig, ms


def test_team_sizes_give_the_same_bits(monkeypatch):
    sig, ms = _members()
    idx = np.zeros(len(ms), np.int32)
    base = None
    for T, all_lanes in ((1, 0), (2, 0), (4, 1), (8, 0), (8, 1), (5, 1)):
        eng = _engine(monkeypatch, T, all_lanes=all_lanes)
        res = eng.solve(sig, idx, ms, dwell=DWELL)
        eng.close()
        assert not res.status.any(), (T, res.status)
        if base is None:
            base = res
        else:
            for f in FIELDS:
                assert np.array_equal(getattr(res, f), getattr(base, f)), f"{f}: team of {T} differs from a team of one"


def test_member_alone_equals_member_in_a_batch_whatever_team_it_gets(monkeypatch):
    """budget 64: the 20-member batch gets teams of 2 (24 x 2 <= 64), a lone member a team of 8."""
    sig, ms = _members()
    eng = _engine(monkeypatch, 8, budget=64)
    batch = eng.solve(sig, np.zeros(len(ms), np.int32), ms, dwell=DWELL)
    for i in (0, 4, 9, 15):
        one = eng.solve(sig, np.zeros(1, np.int32), ms[i:i + 1], dwell=DWELL)
        assert np.array_equal(one.lines, batch.line_list(i)) and np.array_equal(one.sv, batch.singular_values(i))
        assert np.array_equal(one.mu, batch.eigenvalues(i)) and one.status[0] == batch.status[i] == 0
    eng.close()


def test_stage_entry_points_with_teams(monkeypatch):
    rng = np.random.default_rng(11)
    mats = [rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)) for n in (300, 200, 130, 96)]
    out = {}
    for T in (1, 4, 8):
        eng = _engine(monkeypatch, T)
        svd, st1 = eng.svd(mats)
        eig, st2 = eng.eig(mats)
        eng.close()
        assert not st1.any() and not st2.any()
        out[T] = (svd, eig)
    for (L, s, R), A in zip(out[1][0], mats):
        n = A.shape[0]
        assert np.abs((L * s) @ R.conj().T - A).max() < 1e-12 * n
        assert np.abs(s - np.linalg.svd(A, compute_uv=False)).max() < 1e-12 * n
    for (mu, Pm), A in zip(out[1][1], mats):
        assert np.abs(A @ Pm - Pm * mu).max() < 1e-10 * A.shape[0]
    for T in (4, 8):
        for a, b in zip(out[1][0], out[T][0]):
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), f"svd stage: team of {T}"
        for a, b in zip(out[1][1], out[T][1]):
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), f"eig stage: team of {T}"


def test_synchronous_calls_run_on_the_wide_context_with_the_same_bits():
    """Engine.solve (kbdm, sample_kbdm, llc_kbdm) runs on the context with panel teams, Engine.submit on the pool's contexts
    without: same bits, and the wide context never runs beside the pool."""
    from llckbdm_amd.engine import Engine
    sig, ms = _members()
    idx = np.zeros(len(ms), np.int32)
    eng = Engine(0, in_flight=2)
    a = eng.solve(sig, idx, ms, dwell=DWELL)
    assert eng._wide is not None
    p1 = eng.submit(sig, idx, ms, dwell=DWELL)
    p2 = eng.submit(sig, idx, ms, dwell=DWELL)
    c = eng.solve(sig, idx, ms, dwell=DWELL)            # the pool is busy: this one goes to the pool as well
    b1, b2 = p1.result(), p2.result()
    for r in (b1, b2, c):
        for f in FIELDS:
            assert np.array_equal(getattr(r, f), getattr(a, f)), f
    eng.close()
