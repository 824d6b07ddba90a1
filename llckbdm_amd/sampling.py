"""Drop-in for ``llckbdm.sampling`` (reference llckbdm/sampling.py).  The reference's serial
``for m in m_range: kbdm(...)`` loop (sampling.py:52-70) becomes ONE batched GPU call."""
import logging

import numpy as np

from .engine import default_engine
from .kbdm import KbdmInfo, _check_finite, _resolve_m_l

logger = logging.getLogger(__name__)


def sample_kbdm(data, dwell, m_range, p, l, q=0, filter_invalid_features=True, engine=None):
    """Ensemble over ``m_range`` of one signal.  Reference: sampling.py:8-72.

    Returns ``(line_lists, infos)`` for the members whose (filtered) line list is not empty,
    in ``m_range`` order, exactly like the reference.
    """
    data = np.asarray(data)
    ms, ls = [], []
    kbdm_logger = logging.getLogger(__package__ + ".kbdm")
    for m in m_range:
        logger.info(f'Computing KBDM with m = {m}')                     # sampling.py:53
        mm, ll = _resolve_m_l(data.size, m, p, l)
        if q > 0:                                                       # kbdm.py:180, once per member as there
            kbdm_logger.debug('Using Tikhonov Regularization with q=%f', q)
        ms.append(mm)
        ls.append(ll)
    if not ms:
        return [], []
    _check_finite(data, max(ms), p)          # the first member whose window holds a NaN / Inf raises in the reference's loop
    eng = engine or default_engine()
    # check=True: a member the solver flags as not converged is retried once, then numpy.linalg.LinAlgError is raised,
    # as scipy.linalg.svd / eig would inside the reference's kbdm() (kbdm.py:166,192)
    res = eng.solve(data.reshape(1, -1), np.zeros(len(ms), dtype=np.int32), ms, ls, p=p, q=q, dwell=dwell, check=True)
    line_lists, infos = [], []
    for i, (m, ll) in enumerate(zip(ms, ls)):
        line_list = res.line_list(i)
        if filter_invalid_features:
            line_list = line_list[res.keep_mask(i)]                      # == filter_samples(line_list)
        if len(line_list) > 0:                                           # sampling.py:67-70
            line_lists.append(line_list.copy())
            infos.append(KbdmInfo(m=m, l=ll, p=p, q=q, singular_values=res.singular_values(i).copy()))
    return line_lists, infos


def sample_kbdm_signals(signals, dwell, sig_idx, m_list, p=1, l=None, q=0, filter_invalid_features=True,
                        engine=None):
    """Generalised ensemble: item i = (signals[sig_idx[i]], m_list[i]).

    Covers the pseudo-noise ensemble (README.md:10 of the reference: one noisy copy of the
    signal per member, fixed m) and multi-voxel grids with a single batched call.
    Returns (line_lists, infos, item_index) for non-empty members.
    """
    signals = np.atleast_2d(np.asarray(signals))
    ms, ls = [], []
    kbdm_logger = logging.getLogger(__package__ + ".kbdm")
    for m in m_list:
        mm, ll = _resolve_m_l(signals.shape[1], m, p, l)
        if q > 0:
            kbdm_logger.debug('Using Tikhonov Regularization with q=%f', q)
        ms.append(mm)
        ls.append(ll)
    sig_idx = np.asarray(sig_idx, dtype=np.int32)
    if len(ms):                                  # (per signal: the widest window any of its members uses)
        wid = np.zeros(signals.shape[0], dtype=np.int64)
        np.maximum.at(wid, sig_idx, np.asarray(ms, dtype=np.int64))
        for k in np.nonzero(wid)[0]:
            _check_finite(signals[k], int(wid[k]), p)
    eng = engine or default_engine()
    res = _solve_in_flight(eng, signals, sig_idx, ms, ls, p, q, dwell)
    line_lists, infos, index = [], [], []
    for i, (m, ll) in enumerate(zip(ms, ls)):
        line_list = res.line_list(i)
        if filter_invalid_features:
            line_list = line_list[res.keep_mask(i)]
        if len(line_list) > 0:
            line_lists.append(line_list.copy())
            infos.append(KbdmInfo(m=m, l=ll, p=p, q=q, singular_values=res.singular_values(i).copy()))
            index.append(i)
    return line_lists, infos, index


def _solve_in_flight(eng, signals, sig_idx, ms, ls, p, q, dwell, min_members=48):
    """One batch, or - for a grid of several signals - one batch per group of signals with up to `eng.in_flight`
    of them on the GPU at once (`Engine.submit`): the members of different signals are independent
    (sampling.py:52-62), and a batch that does not fill the chip (a few hundred members) leaves most CUs idle during
    its one-CU-per-member stages.  Returns a BatchResult in item order either way."""
    from .engine import BatchResult
    sig_idx = np.asarray(sig_idx, dtype=np.int32)
    ms, ls = np.asarray(ms, dtype=np.int32), np.asarray(ls, dtype=np.int32)
    used = np.unique(sig_idx)
    nfl = getattr(eng, "in_flight", 1)
    # groups: whole signals, enough members each to be worth a launch sequence, at most a chip-filling ~2k per batch
    if nfl < 2 or not hasattr(eng, "submit") or len(used) < 2 or len(ms) < 2 * min_members or len(ms) > 6000:
        return eng.solve(signals, sig_idx, ms, ls, p=p, q=q, dwell=dwell, check=True)
    per_sig = {s: np.nonzero(sig_idx == s)[0] for s in used}
    groups, cur = [], []
    target = max(min_members, min(2048, -(-len(ms) // (2 * nfl))))
    for s in used:
        cur.append(s)
        if sum(len(per_sig[x]) for x in cur) >= target:
            groups.append(cur)
            cur = []
    if cur:
        groups.append(cur)
    pend = []
    for g in groups:
        idx = np.concatenate([per_sig[s] for s in g])
        remap = {s: k for k, s in enumerate(g)}
        sub_sig = np.array([remap[s] for s in sig_idx[idx]], dtype=np.int32)
        pend.append((idx, eng.submit(signals[np.asarray(g)], sub_sig, ms[idx], ls[idx], p=p, q=q, dwell=dwell)))
    line_off = np.concatenate([[0], np.cumsum(ls, dtype=np.int64)])
    sv_off = np.concatenate([[0], np.cumsum(ms, dtype=np.int64)])
    lines = np.empty((int(line_off[-1]), 4))
    sv = np.empty(int(sv_off[-1]))
    mu = np.empty(int(line_off[-1]), dtype=np.complex128)
    keep = np.empty(int(line_off[-1]), dtype=np.uint8)
    status = np.empty(len(ms), dtype=np.int32)
    for idx, h in pend:
        r = h.result(check=True)
        for k, i in enumerate(idx):
            lines[line_off[i]:line_off[i + 1]] = r.line_list(k)
            mu[line_off[i]:line_off[i + 1]] = r.eigenvalues(k)
            keep[line_off[i]:line_off[i + 1]] = r.keep[r.line_off[k]:r.line_off[k + 1]]
            sv[sv_off[i]:sv_off[i + 1]] = r.singular_values(k)
            status[i] = r.status[k]
    return BatchResult(lines, sv, mu, keep, status, line_off, sv_off)


def filter_samples(samples, amplitude_tol=1e-6):
    """Keep amplitude > tol and T2 > 0.  Reference: sampling.py:75-97 (host-side, O(k))."""
    if len(samples) == 0:
        return samples
    amplitude_filter = samples[:, 0] > amplitude_tol
    T2_filter = samples[:, 1] > 0
    return samples[amplitude_filter & T2_filter]
