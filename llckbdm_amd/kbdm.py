"""Drop-in for ``llckbdm.kbdm`` (reference llckbdm/kbdm.py): same names, arguments, return
types and error strings; the numerics run on the GPU through libkbdm_hip.so."""
import logging

import attr
import numpy as np

from .engine import default_engine

logger = logging.getLogger(__name__)


@attr.s
class KbdmInfo:
    """Same record as the reference's KbdmInfo (kbdm.py:10-16)."""
    m = attr.ib()
    l = attr.ib()
    p = attr.ib()
    q = attr.ib()
    singular_values = attr.ib()


def _resolve_m_l(data_size, m, p, l):
    """Argument defaulting and checks, string for string as reference kbdm.py:50-62."""
    if m is None and l is None:
        raise ValueError("l or m must be specified")
    elif m is None:
        m = l
    elif l is None:
        l = m
    elif l > m:
        raise ValueError("l can't be greater than m")
    m_max = (data_size + 1 - p) / 2
    if m > m_max or l > m_max:
        raise ValueError("m or l can't be greater than (n + 1 - p)/2.")
    return int(m), int(l)


def _check_finite(data, m, p):
    """scipy.linalg.svd / eig (kbdm.py:166,192) are called with check_finite=True: a NaN or Inf among the samples that
    enter U^{p-1} or U^p - data[p-1 : 2 m + p - 1] - raises ValueError there, with this message, before any result exists."""
    lo, hi = max(int(p) - 1, 0), 2 * int(m) + int(p) - 1
    if not np.isfinite(np.asarray(data).reshape(-1)[lo:hi]).all():
        raise ValueError("array must not contain infs or NaNs")


def kbdm(data, dwell, m=None, p=1, l=None, q=0, engine=None):
    """One KBDM solve on the GPU.  Reference: kbdm.py:19-92.

    :return: (line_list[l, 4] float64 with columns (amplitude, T2, frequency, phase), KbdmInfo)
    """
    data = np.asarray(data)
    m, l = _resolve_m_l(data.size, m, p, l)
    _check_finite(data, m, p)
    if q > 0:
        logger.debug('Using Tikhonov Regularization with q=%f', q)      # reference kbdm.py:180
    eng = engine or default_engine()
    res = eng.solve(data.reshape(1, -1), [0], [m], [l], p=p, q=q, dwell=dwell, check=True)   # LinAlgError as kbdm.py:166,192
    info = KbdmInfo(m=m, p=p, l=l, q=q, singular_values=res.singular_values(0).copy())
    return res.line_list(0).copy(), info


def _compute_U_matrices(data, m, p, engine=None):
    """Hankel U^0, U^{p-1}, U^p (reference kbdm.py:95-130), assembled by the k_hankel kernel."""
    data = np.asarray(data)
    eng = engine or default_engine()
    (U0, Up_1, Up), = eng.hankel(data.reshape(1, -1), [0], [m], p)
    return U0.copy(), Up_1.copy(), Up.copy()
