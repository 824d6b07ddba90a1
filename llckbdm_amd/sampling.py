"""Drop-in for ``llckbdm.sampling`` (reference llckbdm/sampling.py).  The reference's serial
``for m in m_range: kbdm(...)`` loop (sampling.py:52-70) becomes ONE batched GPU call."""
import logging

import numpy as np

from .engine import default_engine
from .kbdm import KbdmInfo, _resolve_m_l

logger = logging.getLogger(__name__)


def sample_kbdm(data, dwell, m_range, p, l, q=0, filter_invalid_features=True, engine=None):
    """Ensemble over ``m_range`` of one signal.  Reference: sampling.py:8-72.

    Returns ``(line_lists, infos)`` for the members whose (filtered) line list is not empty,
    in ``m_range`` order, exactly like the reference.
    """
    data = np.asarray(data)
    ms, ls = [], []
    for m in m_range:
        logger.info(f'Computing KBDM with m = {m}')                     # sampling.py:53
        mm, ll = _resolve_m_l(data.size, m, p, l)
        ms.append(mm)
        ls.append(ll)
    if not ms:
        return [], []
    if q > 0:
        logging.getLogger(__package__ + ".kbdm").debug('Using Tikhonov Regularization with q=%f', q)
    eng = engine or default_engine()
    res = eng.solve(data.reshape(1, -1), np.zeros(len(ms), dtype=np.int32), ms, ls, p=p, q=q, dwell=dwell)
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
    for m in m_list:
        mm, ll = _resolve_m_l(signals.shape[1], m, p, l)
        ms.append(mm)
        ls.append(ll)
    eng = engine or default_engine()
    res = eng.solve(signals, np.asarray(sig_idx, dtype=np.int32), ms, ls, p=p, q=q, dwell=dwell)
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


def filter_samples(samples, amplitude_tol=1e-6):
    """Keep amplitude > tol and T2 > 0.  Reference: sampling.py:75-97 (host-side, O(k))."""
    if len(samples) == 0:
        return samples
    amplitude_filter = samples[:, 0] > amplitude_tol
    T2_filter = samples[:, 1] > 0
    return samples[amplitude_filter & T2_filter]
