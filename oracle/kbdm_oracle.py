"""CPU restatement (numpy/scipy) of the reference KBDM hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, function by function, what the reference computes on the path
``sampling.sample_kbdm -> kbdm.kbdm`` so that the HIP pipeline can be checked against it
on the GPU box (where /root/reference does not exist).  It is NOT part of the product:
``llckbdm_amd`` never imports it and fails loudly when the HIP library is missing.

Parity pin: ``tests/golden/*.npz`` were produced by importing the reference itself
(``tests/golden/make_golden.py``, run in the build container); ``tests/test_oracle.py``
checks every function here against those vectors.

Third-party arithmetic the reference delegates to (not under /root/reference):
``scipy.linalg.svd`` (LAPACK zgesdd) and ``scipy.linalg.eig`` (LAPACK zgeev), scipy
unpinned ``>=1.3.0`` (reference ``requirements.txt:2``); fixtures were made with
scipy 1.15.3 / numpy 2.2.6 / OpenBLAS 0.3.28.
"""
import numpy as np
from scipy.linalg import svd, eig


# --------------------------------------------------------------------------- sig_gen
def fid(t_array, a, t2, f, phase=0.0):
    """One damped complex exponential.  Reference: llckbdm/sig_gen.py:27-54."""
    if t2 <= 0:
        raise ValueError("T2 must be positive.")            # sig_gen.py:160-161
    if a < 0:
        raise ValueError("Amplitude can't be negative.")    # sig_gen.py:163-164
    harmonic = np.exp(1j * (2 * np.pi * f * t_array + phase))
    return a * np.exp(-t_array / t2) * harmonic


def multi_fid(t_array, params):
    """Sum of FIDs, params rows = (amplitude, t2, frequency, phase).  sig_gen.py:57-71."""
    return np.sum([fid(t_array, *param) for param in params], axis=0)


# Ground-truth 16-peak "brain sim" table.  Values are DATA restated from the reference
# fixture data/params_brain_sim_1_5T.csv:1-16 (amplitude, t2, frequency, phase).
BRAIN_SIM_PARAMS = np.array([
    [0.824383, 0.0087950748, 525.30768, 0.0],
    [0.299129, 0.04, 503.388, 0.0],
    [0.105428, 0.0456621005, 482.3604, 0.0],
    [0.0150218, 0.2222222222, 464.5188, 0.0],
    [0.0411176, 0.1470588235, 455.08824, 0.0],
    [0.0201887, 0.1052631579, 414.94464, 0.0],
    [0.0777794, 0.1136363636, 410.22936, 0.0],
    [0.202612, 0.0925925926, 386.7804, 0.0],
    [0.0427286, 0.1162790698, 299.99376, 0.0],
    [0.0450798, 0.0833333333, 290.43576, 0.0],
    [0.0184325, 0.0909090909, 269.5356, 0.0],
    [0.0290276, 0.0066489362, 268.8984, 0.0],
    [0.428882, 0.0735294118, 255.5172, 0.0],
    [0.291727, 0.0199203187, 246.46896, 0.0],
    [0.11611, 0.0138504155, 160.06464, 0.0],
    [1.0, 0.002712968, 75.31704, 0.0],
])


def brain_sim_params_sorted():
    """The table sorted by frequency, as the reference fixture does (_tests/fixtures.py:29-35)."""
    return BRAIN_SIM_PARAMS[np.argsort(BRAIN_SIM_PARAMS[:, 2], kind="stable")]


def brain_sim_signal(N=2048, dwell=5e-4, params=None):
    """Noiseless fixture signal.  Reference: _tests/fixtures.py:9-47 (linspace time axis)."""
    t = np.linspace(0, dwell * N, N, endpoint=False)
    return multi_fid(t, brain_sim_params_sorted() if params is None else params)


# --------------------------------------------------------------------------- kbdm.py
class KbdmInfo:
    """Mirror of the attrs record kbdm.py:10-16 (plain class: the oracle needs no attrs)."""

    def __init__(self, m, l, p, q, singular_values):
        self.m, self.l, self.p, self.q, self.singular_values = m, l, p, q, singular_values


def compute_U_matrices(data, m, p):
    """Hankel U0, U^{p-1}, U^p.  Reference: kbdm.py:95-130."""
    data = np.asarray(data)
    U0 = np.empty((m, m), dtype=complex)
    Up_1 = np.empty((m, m), dtype=complex)
    Up = np.empty((m, m), dtype=complex)
    for i in range(m):
        U0[i] = data[i:i + m]                       # kbdm.py:117
        Up[i] = data[i + p:i + m + p]               # kbdm.py:120
    if p == 1:
        Up_1 = np.copy(U0)                          # kbdm.py:125
    else:
        for i in range(m):
            Up_1[i] = data[i + p - 1:i + m + p - 1]  # kbdm.py:128
    return U0, Up_1, Up


def normalize_eigenvectors(B, U0, normalizer="einsum"):
    """B_k (B_k^T U0 B_k)^{-1/2}, bilinear (no conjugate).  Reference: kbdm.py:215-240.

    ``normalizer="einsum"`` evaluates exactly the reference's einsum('jk,ij,ik->k')
    (kbdm.py:232; an unblocked O(m^2 l) loop).  ``"gemm"`` computes the same contraction as
    N_k = sum_i B[i,k] (U0 B)[i,k] through BLAS-3 (identical up to summation order); it is
    the fast form used when the oracle is timed as the CPU baseline.
    """
    if normalizer == "einsum":
        N_inv_sqrt = np.einsum('jk,ij,ik->k', B, U0, B)
    else:
        N_inv_sqrt = np.einsum('ik,ik->k', B, U0 @ B)
    with np.errstate(all="ignore"):
        N_sqrt = np.sqrt(1.0 / N_inv_sqrt)
    return B * N_sqrt


def solve_gep_svd(U0, Up_1, Up, l, q=0.0, lapack_driver="gesdd", normalizer="einsum"):
    """SVD of U^{p-1}, reduced eig, back-transform, normalise.  Reference: kbdm.py:133-212."""
    L, s, R_h = svd(Up_1, lapack_driver=lapack_driver)   # kbdm.py:166 (zgesdd, full matrices)
    L_ = L[:, :l]
    R_ = R_h[:l, :].conj().T                             # kbdm.py:172-177
    if normalizer == "einsum":
        # as shipped: dense diagonal matrices, np.linalg.inv, left-to-right matmul chain
        S_ = np.diag(s)[:l, :l]                          # kbdm.py:168-171
        if q > 0:
            S = S_ + q * q * np.linalg.inv(S_)           # kbdm.py:182
            Dsqi_ = np.linalg.inv(np.sqrt(S))            # kbdm.py:184
        else:
            Dsqi_ = np.linalg.inv(np.sqrt(S_))           # kbdm.py:186
        U = Dsqi_ @ L_.conj().T @ Up @ R_ @ Dsqi_        # kbdm.py:189
        mu, P = eig(U)                                   # kbdm.py:192 (zgeev)
        B = R_ @ Dsqi_ @ P                               # kbdm.py:198
    else:
        # same algebra with the diagonal matrices applied as row/column scalings
        s_ = s[:l]
        dsqi = 1.0 / np.sqrt(s_ + q * q / s_) if q > 0 else 1.0 / np.sqrt(s_)
        U = (dsqi[:, None] * (L_.conj().T @ Up @ R_)) * dsqi[None, :]
        mu, P = eig(U)
        B = (R_ * dsqi[None, :]) @ P
    B_norm = normalize_eigenvectors(B, U0, normalizer)   # kbdm.py:202
    return mu, B_norm, {'singular_values': s, 'q': q, 'l': l}


def kbdm(data, dwell, m=None, p=1, l=None, q=0, lapack_driver="gesdd", return_mu=False,
         normalizer="einsum"):
    """One ensemble member.  Reference: kbdm.py:19-92 (validation :50-62, epilogue :70-92)."""
    data = np.asarray(data)
    if m is None and l is None:
        raise ValueError("l or m must be specified")
    elif m is None:
        m = l
    elif l is None:
        l = m
    elif l > m:
        raise ValueError("l can't be greater than m")
    m_max = (data.size + 1 - p) / 2
    if m > m_max or l > m_max:
        raise ValueError("m or l can't be greater than (n + 1 - p)/2.")

    U0, Up_1, Up = compute_U_matrices(data, m, p)
    mu, B_norm, svd_info = solve_gep_svd(U0, Up_1, Up, l, q, lapack_driver=lapack_driver,
                                         normalizer=normalizer)
    info = KbdmInfo(m=m, p=p, l=l, q=q, singular_values=svd_info['singular_values'])

    with np.errstate(all="ignore"):
        D_sqrt = data[:m] @ B_norm                       # kbdm.py:71
        D = D_sqrt * D_sqrt
        A = np.abs(D)
        PH = np.angle(D)
        Omega = -1j * np.log(mu) / dwell                 # kbdm.py:82
        F = np.real(Omega) / (2 * np.pi)
        T2 = 1.0 / np.imag(Omega)
    line_list = np.column_stack((A, T2, F, PH))          # kbdm.py:88-90
    if return_mu:
        return line_list, info, mu
    return line_list, info


# --------------------------------------------------------------------------- sampling.py
def filter_samples(samples, amplitude_tol=1e-6):
    """Keep A > tol and T2 > 0.  Reference: sampling.py:75-97."""
    if len(samples) == 0:
        return samples
    keep = (samples[:, 0] > amplitude_tol) & (samples[:, 1] > 0)
    return samples[keep]


def sample_kbdm(data, dwell, m_range, p, l, q=0, filter_invalid_features=True,
                normalizer="einsum"):
    """Serial ensemble loop.  Reference: sampling.py:8-72 (loop :52-70)."""
    line_lists, infos = [], []
    for m in m_range:
        line_list, info = kbdm(data=data, dwell=dwell, m=m, p=p, l=l, q=q, normalizer=normalizer)
        if filter_invalid_features:
            line_list = filter_samples(line_list)
        if len(line_list) > 0:
            line_lists.append(line_list)
            infos.append(info)
    return line_lists, infos


# --------------------------------------------------------------------------- helpers
def canonical_order(line_list):
    """Row order of eig() is LAPACK-defined; compare line lists after sorting by (F, 1/T2)."""
    ll = np.asarray(line_list)
    if ll.size == 0:
        return ll
    with np.errstate(all="ignore"):
        key2 = 1.0 / ll[:, 1]
    return ll[np.lexsort((key2, ll[:, 2]))]


def make_noisy(signal, sigma, seed):
    """Seeded complex white noise: sigma*(randn + i randn), numpy default_rng."""
    rng = np.random.default_rng(seed)
    noise = rng.standard_normal(signal.shape[0]) + 1j * rng.standard_normal(signal.shape[0])
    return signal + sigma * noise
