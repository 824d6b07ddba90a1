"""Signal synthesis helpers with the reference's names (llckbdm/sig_gen.py).  Host-side numpy:
they only produce the synthetic inputs of the hot path (O(N x peaks))."""
import logging

import numpy as np

logger = logging.getLogger(__name__)


def gen_t_freq_arrays(N, dwell):
    """Time and (shifted) frequency axes.  Reference: sig_gen.py:8-24.

    The reference uses a float-step ``np.arange(0, N*dwell, dwell)`` whose length is fragile;
    ``arange(N) * dwell`` has the same values and always N points.
    """
    t_array = np.arange(N) * dwell
    freq_array = np.fft.fftshift(np.fft.fftfreq(N, dwell))
    return t_array, freq_array


def _validate_parameters(a, t2, f, phase):
    """Reference: sig_gen.py:140-169."""
    if t2 <= 0:
        raise ValueError("T2 must be positive.")
    if a < 0:
        raise ValueError("Amplitude can't be negative.")
    if np.abs(phase) > 2 * np.pi:
        logger.warning('Phase is greater than 2 * pi and phase must be given in rad/s. '
                       'Check whether the correct unit is being used.')


def fid(t_array, a, t2, f, phase=0.):
    """One Free Induction Decay.  Reference: sig_gen.py:27-54."""
    _validate_parameters(a, t2, f, phase)
    return a * np.exp(-t_array / t2) * np.exp(1j * (2 * np.pi * f * t_array + phase))


def multi_fid(t_array, params):
    """Sum of FIDs; params rows are (amplitude, t2, frequency, phase).  Reference: sig_gen.py:57-71."""
    return np.sum([fid(t_array, *param) for param in params], axis=0)


def fft(data):
    """Normalised, shifted FFT.  Reference: sig_gen.py:74-88."""
    return np.fft.fftshift(np.fft.fft(data)) / np.sqrt(len(data))


def lorentzian_peak(freq_array, a, t2, f, phase=0):
    """Reference: sig_gen.py:91-121."""
    _validate_parameters(a, t2, f, phase)
    return a * np.exp(1j * phase) / ((1. / t2) + 2j * np.pi * (freq_array - f))


def spec(freq_array, params):
    """Reference: sig_gen.py:124-137."""
    return np.sum([lorentzian_peak(freq_array, *param) for param in params], axis=0)
