"""`llckbdm.metrics` with the reference's names (metrics.py:7-17), scored on the GPU.

The reference synthesises the estimate with `sig_gen.multi_fid`, takes two FFTs and the RMSE of their real
parts.  The HIP kernel `k_rmse` evaluates the same number without a transform (the identity is stated in
llckbdm_amd/csrc/kbdm_next.hpp) for any number of candidates in one launch."""
import numpy as np

from .engine import default_engine


def calculate_freq_domain_rmse(data, params_est, dwell, engine=None):
    """RMSE between the real parts of the normalised FFTs of `data` and of the signal synthesised from
    `params_est` (rows (amplitude, t2, frequency, phase)).  Reference: metrics.py:7-17."""
    params_est = np.asarray(params_est, dtype=np.float64).reshape(-1, 4)
    if len(params_est) == 0:
        raise ValueError("params_est is empty")           # the reference fails inside numpy's fft for an empty list
    for a, t2, _, _ in params_est:                        # sig_gen._validate_parameters (sig_gen.py:140-169) via multi_fid
        if t2 <= 0:
            raise ValueError("T2 must be positive.")
        if a < 0:
            raise ValueError("Amplitude can't be negative.")
    eng = engine or default_engine()
    return float(eng.rmse_batch(data, dwell, [params_est])[0])
