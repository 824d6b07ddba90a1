"""Synthetic inputs for the benchmark configurations of BASELINE.json.

The 16-peak table is DATA restated from the reference fixture
``data/params_brain_sim_1_5T.csv`` (amplitude, t2, frequency, phase), sorted by frequency as
the reference's own fixtures do (llckbdm/_tests/fixtures.py:29-35)."""
import numpy as np

from . import sig_gen

BRAIN_SIM_PARAMS = np.array([
    [1.0, 0.002712968, 75.31704, 0.0],
    [0.11611, 0.0138504155, 160.06464, 0.0],
    [0.291727, 0.0199203187, 246.46896, 0.0],
    [0.428882, 0.0735294118, 255.5172, 0.0],
    [0.0290276, 0.0066489362, 268.8984, 0.0],
    [0.0184325, 0.0909090909, 269.5356, 0.0],
    [0.0450798, 0.0833333333, 290.43576, 0.0],
    [0.0427286, 0.1162790698, 299.99376, 0.0],
    [0.202612, 0.0925925926, 386.7804, 0.0],
    [0.0777794, 0.1136363636, 410.22936, 0.0],
    [0.0201887, 0.1052631579, 414.94464, 0.0],
    [0.0411176, 0.1470588235, 455.08824, 0.0],
    [0.0150218, 0.2222222222, 464.5188, 0.0],
    [0.105428, 0.0456621005, 482.3604, 0.0],
    [0.299129, 0.04, 503.388, 0.0],
    [0.824383, 0.0087950748, 525.30768, 0.0],
])

DWELL = 5e-4


def brain_sim_signal(N=2048, dwell=DWELL, params=BRAIN_SIM_PARAMS):
    t = np.linspace(0, dwell * N, N, endpoint=False)
    return sig_gen.multi_fid(t, params)


def add_noise(signal, sigma, seed):
    rng = np.random.default_rng(seed)
    n = rng.standard_normal(signal.shape[0]) + 1j * rng.standard_normal(signal.shape[0])
    return signal + sigma * n


def config2(seed=0):
    """C2: N=2048, 16 peaks, sigma=1e-3, m = 100..400 step 2 (151 members)."""
    sig = add_noise(brain_sim_signal(2048), 1e-3, seed)
    m = np.arange(100, 401, 2, dtype=np.int32)
    return sig.reshape(1, -1), np.zeros(len(m), dtype=np.int32), m


def north_star(seed=0, step=2):
    """The ensemble BASELINE.json's north_star quotes its target on: N=2048, 16 peaks, sigma=1e-3,
    m = 100..500 (step 2: 201 members; the sum of m^3 is 2.44 x that of C2: 7.86e9 against 3.22e9)."""
    sig = add_noise(brain_sim_signal(2048), 1e-3, seed)
    m = np.arange(100, 501, step, dtype=np.int32)
    return sig.reshape(1, -1), np.zeros(len(m), dtype=np.int32), m


def config3(count=1024, m=512, seed0=0):
    """C3: pseudo-noise ensemble, fixed m, one sigma=1e-6 noise draw per member."""
    base = brain_sim_signal(2048)
    sigs = np.stack([add_noise(base, 1e-6, seed0 + k) for k in range(count)])
    return sigs, np.arange(count, dtype=np.int32), np.full(count, m, dtype=np.int32)


def config4(mmin=200, mmax=1200, step=1):
    """C4: N=4096, the 16 table peaks + 16 seeded extra peaks, sigma=1e-3, m = 200..1200 (1001 members)."""
    rng = np.random.default_rng(1)
    extra = np.column_stack([rng.uniform(0.01, 1, 16), rng.uniform(0.005, 0.2, 16), rng.uniform(50, 950, 16),
                             np.zeros(16)])
    params = np.vstack([BRAIN_SIM_PARAMS, extra])
    t = np.arange(4096) * DWELL
    sig = add_noise(sig_gen.multi_fid(t, params), 1e-3, 4)
    m = np.arange(mmin, mmax + 1, step, dtype=np.int32)
    return sig.reshape(1, -1), np.zeros(len(m), dtype=np.int32), m


def config5(voxels=64, mmin=128, mmax=383):
    """C5: multi-voxel grid, `voxels` signals (table peaks with amplitudes scaled by U(0.5, 1.5), sigma=1e-3,
    N=2048) x an m-ensemble m = 128..383 each (64 x 256 = 16384 members)."""
    t = np.linspace(0, DWELL * 2048, 2048, endpoint=False)
    sigs = []
    for v in range(voxels):
        p = BRAIN_SIM_PARAMS.copy()
        p[:, 0] *= np.random.default_rng(100 + v).uniform(0.5, 1.5, len(p))
        sigs.append(add_noise(sig_gen.multi_fid(t, p), 1e-3, 50 + v))
    ms = np.arange(mmin, mmax + 1, dtype=np.int32)
    sig_idx = np.repeat(np.arange(voxels, dtype=np.int32), len(ms))
    return np.stack(sigs), sig_idx, np.tile(ms, voxels)
