"""`llckbdm.llckbdm` with the reference's names (llckbdm.py): LLC-KBDM = sample an m-range ensemble (GPU,
batched), pool and filter the lines, sweep `min_samples` of a density clusterer over the pooled lines in the
(Re mu, Im mu, A, 0) feature space, summarise every clustering into one line list, keep the one with the
smallest frequency-domain RMSE (GPU, one call for all candidates).

What runs where: the ensemble, the RMSE scoring, the silhouettes and the density clustering (HDBSCAN*, every value of
`min_samples` of the sweep in ONE call: `Engine.hdbscan_sweep`) are HIP kernels plus host tree code in the C-ABI
library; the feature transform and the cluster summaries are O(n) numpy.  The reference's clusterer is the
third-party `hdbscan.HDBSCAN` (llckbdm.py:280), unpinned upstream and absent here; cluster labels are not pinned by
the reference.  `clusterer="hdbscan"` / `"sklearn"` select those packages instead of the built-in
one (they cost seconds per fit at the pooled-line counts of a C2 ensemble)."""
import logging

import attr
import numpy as np

from .engine import default_engine
from .metrics import calculate_freq_domain_rmse
from .min_rmse_kbdm import min_rmse_kbdm
from .sampling import filter_samples, sample_kbdm
from .sig_gen import gen_t_freq_arrays, multi_fid

logger = logging.getLogger(__name__)


@attr.s(auto_attribs=True)
class LlcKbdmResult:
    line_list: np.ndarray = np.array([])
    rmse: float = None
    silhouette: np.ndarray = np.array([])


@attr.s(auto_attribs=True)
class IterativeLlcKbdmResult:
    line_list: np.ndarray = np.array([])
    line_lists: np.ndarray = np.array([])
    rmse: float = None
    silhouettes: np.ndarray = np.array([])


@attr.s(auto_attribs=True)
class ClusteringResult:
    num_clusters: int = 0
    labels: np.ndarray = np.array([])
    clustered: np.ndarray = np.array([])
    non_clustered: np.ndarray = np.array([])
    summarized_line_list: np.ndarray = np.array([])
    clustered_silhouettes: np.ndarray = np.array([])


MIN_CLUSTER_SIZE = 5        # hdbscan.HDBSCAN's default, which the reference does not change (llckbdm.py:280)


def _fit_labels(transformed_samples, min_samples, clusterer, engine):
    """Cluster labels of one HDBSCAN fit.  "gpu": the built-in implementation; "hdbscan": the reference's package
    (llckbdm.py:280) if installed; "sklearn": scikit-learn's HDBSCAN."""
    if clusterer == "gpu":
        eng = engine or default_engine()
        return eng.hdbscan_sweep(transformed_samples, [min_samples], MIN_CLUSTER_SIZE)[0][0]
    if clusterer == "hdbscan":
        import hdbscan
        return np.asarray(hdbscan.HDBSCAN(min_samples=min_samples).fit(transformed_samples).labels_)
    if clusterer == "sklearn":
        from sklearn.cluster import HDBSCAN
        return np.asarray(HDBSCAN(min_samples=min_samples, min_cluster_size=MIN_CLUSTER_SIZE, copy=True)
                          .fit(transformed_samples).labels_)
    raise ValueError("clusterer must be 'gpu', 'hdbscan' or 'sklearn'")


def llc_kbdm(data, dwell, m_range, p=1, l=None, q=0.0, engine=None, clusterer="gpu"):
    """Line List Clustering KBDM.  Reference: llckbdm.py:41-141 (same arguments, result type and error)."""
    if len(m_range) < 2:
        raise ValueError("size of 'm_range' must be greater than 2.")
    eng = engine or default_engine()
    line_lists, infos = sample_kbdm(data=data, dwell=dwell, m_range=m_range, p=p, l=l, q=q, engine=eng)
    if len(line_lists) == 0:                                             # llckbdm.py:86-91
        return LlcKbdmResult(line_list=np.array([]), rmse=None, silhouette=np.array([]))
    samples = np.concatenate(line_lists)
    samples = filter_samples(samples)
    transformed_line_list = _transform_line_lists(samples, dwell)
    m_range_size = len(m_range)
    clustering_results = []
    sweep = list(range(int(np.ceil(0. * m_range_size + 1)), m_range_size))           # llckbdm.py:104
    # min_samples cannot exceed the number of pooled lines (the packages raise there; such fits have no clusters)
    labels_all = None
    sil_all = {}
    if clusterer == "gpu" and len(transformed_line_list) >= 2:
        fits = [k for k in sweep if k <= len(transformed_line_list)]
        labels_all = {}
        if fits:                     # every fit on the GPU, whatever min_samples (no host clusterer behind the caller's back)
            got, _ = eng.hdbscan_sweep(transformed_line_list, fits, MIN_CLUSTER_SIZE)
            labels_all.update(zip(fits, got))
            # ... and the silhouettes of every fit in one call as well (an engine without the batched entry scores fit by fit)
            sweep_sil = getattr(eng, "silhouette_sweep", None)
            if sweep_sil is not None:
                sil, ok = sweep_sil(transformed_line_list, np.asarray(got))
                sil_all = {k: (sil[f] if ok[f] else None) for f, k in enumerate(fits)}
    for min_samples in sweep:
        logger.debug('HDBSCAN with min_samples = %d', min_samples)
        if clusterer == "gpu" and (labels_all is None or min_samples not in labels_all):
            continue
        clustering_result = _cluster_line_lists(samples=samples, transformed_samples=transformed_line_list,
                                                min_samples=min_samples, engine=eng, clusterer=clusterer,
                                                labels=None if labels_all is None else labels_all[min_samples],
                                                sil=sil_all.get(min_samples))
        if clustering_result.num_clusters > 0:
            clustering_results.append(clustering_result)
    summarized_line_lists = [cl_result.summarized_line_list for cl_result in clustering_results]
    min_rmse_kbdm_results = min_rmse_kbdm(data=data, dwell=dwell, samples=summarized_line_lists, engine=eng)
    if min_rmse_kbdm_results is None:
        return LlcKbdmResult(line_list=np.array([]), rmse=None, silhouette=np.array([]))
    silhouette = np.array(clustering_results[min_rmse_kbdm_results.min_index].clustered_silhouettes)
    return LlcKbdmResult(line_list=min_rmse_kbdm_results.line_list, rmse=min_rmse_kbdm_results.min_rmse,
                         silhouette=silhouette)


def iterative_llc_kbdm(data, dwell, m_range, p=1, l=None, q=0.0, max_iterations=5, silhouette_threshold=0.6,
                       engine=None, clusterer="gpu"):
    """Residual peeling: fit the lines LLC-KBDM is confident about, subtract their model signal, fit the rest
    again with a lower confidence bar.  Behaviour of the reference's llckbdm.py:144-199: the bar of round k is the
    `np.percentile` of that round's cluster silhouettes at rank  threshold * (1 - k / (rounds - 1))  (a 0..0.6
    PERCENT rank, as upstream), strictly-greater selection, the final RMSE is the model against itself, progress is
    printed.  (One quirk is not kept: a round in which no cluster clears the bar ends the peeling here, where the
    reference would crash inside `multi_fid` on an empty table.)"""
    if max_iterations < 1:
        raise ValueError("'max_iterations must be greater than zero")
    eng = engine or default_engine()
    observed = np.asarray(data)
    t_axis = gen_t_freq_arrays(N=len(observed), dwell=dwell)[0]
    model = np.zeros_like(observed)                  # signal of everything accepted so far
    accepted = []                                    # one (lines, silhouettes) pair per productive round
    percent_ranks = np.linspace(silhouette_threshold, 0, max_iterations)
    for round_no, rank in enumerate(percent_ranks):
        print(f'Iteration #{round_no}')
        fit = llc_kbdm(data=observed - model, dwell=dwell, m_range=m_range, p=p, l=l, q=q, engine=eng,
                       clusterer=clusterer)
        if len(fit.line_list) == 0:
            logging.info('No more peaks can be fitted. Stopping.')      # root logger, as upstream (llckbdm.py:168)
            break
        confident = fit.silhouette > np.percentile(fit.silhouette, rank)
        if not confident.any():
            logging.info('No peak passed the silhouette threshold. Stopping.')
            break
        lines = fit.line_list[confident]
        model = model + multi_fid(t_array=t_axis, params=lines)
        accepted.append((lines, fit.silhouette[confident]))
        print(f'Found {len(lines)} peaks. Total: {sum(len(a) for a, _ in accepted)} peaks.')
    if not accepted:
        return IterativeLlcKbdmResult(line_list=np.array([]), line_lists=[], silhouettes=[], rmse=None)
    per_round_lines = [a for a, _ in accepted]
    everything = np.concatenate(per_round_lines)
    rmse = calculate_freq_domain_rmse(data=model, params_est=everything, dwell=dwell, engine=eng)
    return IterativeLlcKbdmResult(line_list=everything, line_lists=per_round_lines,
                                  silhouettes=[s for _, s in accepted], rmse=rmse)


def _transform_line_lists(line_lists, dwell):
    """(A, T2, F, PH) -> (Re mu, Im mu, A, 0) with mu = exp(i dwell (2 pi F + i / T2)).  Reference: llckbdm.py:202-230
    (the phase feature is zeroed there, :219)."""
    A = line_lists[:, 0]
    T2 = line_lists[:, 1]
    F = line_lists[:, 2]
    PH = line_lists[:, 3] * 0
    OMEGA = 2 * np.pi * F + 1j / T2
    mu = np.exp(1j * dwell * OMEGA)
    return np.column_stack((np.real(mu), np.imag(mu), A, PH))


def _inverse_transform_line_lists(transformed_line_lists, dwell):
    """Inverse of `_transform_line_lists`.  Reference: llckbdm.py:233-261."""
    MU = transformed_line_lists[:, 0] + 1j * transformed_line_lists[:, 1]
    A = transformed_line_lists[:, 2]
    PH = transformed_line_lists[:, 3]
    OMEGA = -1j * np.log(MU) / dwell
    T2 = 1. / np.imag(OMEGA)
    F = np.real(OMEGA) / (2 * np.pi)
    return np.column_stack((A, T2, F, PH))


def _cluster_line_lists(samples, transformed_samples, min_samples, engine=None, clusterer="gpu", labels=None, sil=None):
    """One density clustering of the pooled lines, the mean silhouette of every cluster and the summarised line
    list.  Same result type and field meaning as the reference's llckbdm.py:264-321; the work is organised around
    ONE stable sort of the labels (cluster index sets are slices of it, per-cluster means are segmented sums).  The
    silhouettes come from the GPU kernel (`Engine.silhouette_samples`); like `sklearn.metrics.silhouette_samples`
    they are undefined for fewer than 2 or more than n-1 label values - the reference would raise there, here
    such a clustering is reported as having no clusters."""
    if labels is None:
        labels = _fit_labels(transformed_samples, min_samples, clusterer, engine)
    labels = np.asarray(labels)
    # noise (-1) first, then cluster 0, 1, ...; STABLE: inside a label the sample indices ascend.  (16-bit keys take numpy's
    # radix sort: several times faster on the 20 000 pooled lines of a C2 sweep, the same permutation.)
    small = labels.size > 0 and -32768 <= labels.min() and labels.max() <= 32767
    order = np.argsort(labels.astype(np.int16) if small else labels, kind="stable")
    sorted_labels = labels[order]
    values = np.unique(sorted_labels)
    num_clusters = int(np.count_nonzero(values >= 0))          # len(set(labels) - {-1}), llckbdm.py:285
    if num_clusters == 0 or not (2 <= len(values) <= len(labels) - 1):
        return ClusteringResult(num_clusters=0, labels=labels, clustered=np.array([]), non_clustered=np.array([]),
                                summarized_line_list=[], clustered_silhouettes=np.array([]))
    # cluster k = the samples labelled k for k = 0 .. num_clusters - 1, as the reference's loop (llckbdm.py:294-295);
    # every clusterer used here numbers its clusters contiguously, and a label value that is absent gives an EMPTY
    # index set (searchsorted: start == end), never a wrong segment
    starts = np.searchsorted(sorted_labels, np.arange(num_clusters + 1))
    members = [(order[starts[k]:starts[k + 1]].copy(),) for k in range(num_clusters)]     # np.nonzero-style tuples (ascending: stable sort)
    if sil is None:                                            # (llc_kbdm hands over the sweep's batched silhouettes)
        sil = (engine or default_engine()).silhouette_samples(transformed_samples, labels)
    # llckbdm.py:299-301: np.average of every cluster's silhouettes - a cluster is a slice of the sorted view, all the means
    # are one segmented sum over it (another order of the additions than np.average: the same means to a few ulp)
    sil_sorted = np.ascontiguousarray(np.asarray(sil, dtype=np.float64)[order])
    counts = np.diff(starts)
    mean_sil = np.full(num_clusters, np.nan)
    nonempty = np.flatnonzero(counts)
    if len(nonempty):
        mean_sil[nonempty] = np.add.reduceat(sil_sorted[:starts[-1]], starts[:-1][nonempty]) / counts[nonempty]
    clustered = np.empty(num_clusters, dtype=object)
    for k, idx in enumerate(members):
        clustered[k] = idx
    return ClusteringResult(num_clusters=num_clusters, labels=labels, clustered=clustered,
                            non_clustered=np.array((np.sort(order[:np.searchsorted(sorted_labels, 0)]),)),
                            summarized_line_list=_summarize_clusters(samples=samples, clusters=members),
                            clustered_silhouettes=mean_sil)


def _summarize_clusters(samples, clusters, summarizer=np.average):
    """One line per cluster (reference llckbdm.py:324-353): `summarizer(rows, axis=0)` of the cluster's rows with the
    T2 column replaced by the decay RATE 1/T2 - i.e. T2 is summarised harmonically - and turned back afterwards.
    `samples` is not modified; only the rows of clustered samples are inverted (noise rows may hold T2 = 0)."""
    samples = np.asarray(samples, dtype=np.float64)
    index_sets = [np.asarray(c[0] if isinstance(c, tuple) else c).ravel() for c in clusters]
    if not index_sets:
        return np.array([])
    if summarizer is None:
        summarizer = np.average
    out = np.empty((len(index_sets), samples.shape[1]), dtype=np.float64)
    if summarizer is np.average:
        # the default reduction for all clusters at once: a segmented sum over the concatenated rows, divided by the cluster
        # sizes.  np.add.reduceat adds a segment's rows in another order than np.average(rows, axis=0) does: the same means
        # to a few ulp of the summands (tests/test_next_rows.py), 25 000 numpy calls fewer on a C2 sweep
        counts = np.array([len(ix) for ix in index_sets])
        nonempty = np.flatnonzero(counts)
        out[:] = np.nan
        if len(nonempty):
            rows = samples[np.concatenate([index_sets[k] for k in nonempty])]
            rows[:, 1] = 1.0 / rows[:, 1]
            starts = np.concatenate(([0], np.cumsum(counts[nonempty])[:-1]))
            out[nonempty] = np.add.reduceat(rows, starts, axis=0) / counts[nonempty][:, None]
        out[:, 1] = 1.0 / out[:, 1]
        return out
    for k, ix in enumerate(index_sets):
        rows = samples[ix]                                   # a copy (fancy index): only clustered rows are inverted
        rows[:, 1] = 1.0 / rows[:, 1]
        out[k] = summarizer(rows, axis=0) if len(ix) else np.nan      # the reference's own reduction: llckbdm.py:346-349
    out[:, 1] = 1.0 / out[:, 1]
    return out
