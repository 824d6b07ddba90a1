"""`llckbdm.min_rmse_kbdm` with the reference's names (min_rmse_kbdm.py:12-55): the ensemble comes from the
batched GPU sampler and every member is scored in ONE call of the RMSE kernel."""
import logging

import attr
import numpy as np

from .engine import default_engine
from .sampling import sample_kbdm

logger = logging.getLogger(__name__)


@attr.s
class MinRmseKbdmResult:
    line_list = attr.ib()
    min_rmse = attr.ib()
    min_index = attr.ib()
    samples = attr.ib()
    rmses_list = attr.ib()


def min_rmse_kbdm(data, dwell, m_range=None, l=None, samples=None, engine=None):
    """Pick the ensemble member with the smallest frequency-domain RMSE.  Reference: min_rmse_kbdm.py:21-55
    (same defaults p=1, q=0, filtered samples; an empty line list scores inf; no samples -> None)."""
    eng = engine or default_engine()
    if samples is None:
        samples, _ = sample_kbdm(data=data, dwell=dwell, m_range=m_range, l=l, q=0, p=1,
                                 filter_invalid_features=True, engine=eng)
    if len(samples) == 0:
        return None
    scores = eng.rmse_batch(data, dwell, [np.asarray(s, dtype=np.float64).reshape(-1, 4) for s in samples])
    rmses = [float(r) for r in scores]
    for i, rmse in enumerate(rmses):
        logger.debug('RMSE for sample #%d: %f', i, rmse)
    min_index = int(np.argmin(rmses))
    return MinRmseKbdmResult(line_list=samples[min_index], min_rmse=rmses[min_index], min_index=min_index,
                             samples=samples, rmses_list=rmses)
