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
    if clusterer == "gpu" and len(transformed_line_list) >= 2:
        fits = [k for k in sweep if k <= len(transformed_line_list)]
        if fits:
            got, _ = eng.hdbscan_sweep(transformed_line_list, fits, MIN_CLUSTER_SIZE)
            labels_all = dict(zip(fits, got))
    for min_samples in sweep:
        logger.debug('HDBSCAN with min_samples = %d', min_samples)
        if clusterer == "gpu" and (labels_all is None or min_samples not in labels_all):
            continue
        clustering_result = _cluster_line_lists(samples=samples, transformed_samples=transformed_line_list,
                                                min_samples=min_samples, engine=eng, clusterer=clusterer,
                                                labels=None if labels_all is None else labels_all[min_samples])
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
    """Residual peeling driver.  Reference: llckbdm.py:144-199."""
    if max_iterations < 1:
        raise ValueError("'max_iterations must be greater than zero")
    eng = engine or default_engine()
    data = np.asarray(data)
    curr_data_est = np.zeros_like(data)
    line_lists = []
    silhouettes = []
    t_array, _ = gen_t_freq_arrays(N=len(data), dwell=dwell)
    n_peaks = 0
    silhouette_thresholds = np.linspace(silhouette_threshold, 0, max_iterations)
    for i in range(max_iterations):
        logger.info('Iteration #%d', i)
        curr_res = data - curr_data_est
        results = llc_kbdm(data=curr_res, dwell=dwell, m_range=m_range, p=p, l=l, q=q, engine=eng, clusterer=clusterer)
        if len(results.line_list) == 0:
            logger.info('No more peaks can be fitted. Stopping.')
            break
        filtered_index = np.nonzero(
            (results.silhouette > np.percentile(results.silhouette, silhouette_thresholds[i])))
        line_list = results.line_list[filtered_index]
        if len(line_list) == 0:
            # the reference would call multi_fid with no peaks here and fail inside numpy; nothing is left to fit
            logger.info('No peak passed the silhouette threshold. Stopping.')
            break
        curr_data_est_i = multi_fid(t_array=t_array, params=line_list)
        curr_data_est = curr_data_est + curr_data_est_i
        line_lists.append(line_list)
        silhouettes.append(results.silhouette[filtered_index])
        n_peaks += len(line_list)
        logger.info('Found %d peaks. Total: %d peaks.', len(line_list), n_peaks)
    if not line_lists:
        return IterativeLlcKbdmResult(line_list=np.array([]), line_lists=[], silhouettes=[], rmse=None)
    line_list = np.concatenate(line_lists)
    rmse = calculate_freq_domain_rmse(data=curr_data_est, params_est=line_list, dwell=dwell, engine=eng)
    return IterativeLlcKbdmResult(line_list=line_list, line_lists=line_lists, silhouettes=silhouettes, rmse=rmse)


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


def _cluster_line_lists(samples, transformed_samples, min_samples, engine=None, clusterer="gpu", labels=None):
    """One density clustering of the pooled lines + per-cluster mean silhouettes + summarised line list.
    Reference: llckbdm.py:264-321.  The silhouettes come from the GPU kernel (`Engine.silhouette_samples`);
    like `sklearn.metrics.silhouette_samples` they are undefined for fewer than 2 or more than n-1 label
    values - the reference would raise there, here such a clustering is reported as having no clusters."""
    if labels is None:
        labels = _fit_labels(transformed_samples, min_samples, clusterer, engine)
    labels = np.asarray(labels)
    num_clusters = len(set(labels.tolist()) - {-1})
    n_labels = len(set(labels.tolist()))
    clustered = []
    if num_clusters > 0 and 2 <= n_labels <= len(labels) - 1:
        eng = engine or default_engine()
        sample_silhouette_values = eng.silhouette_samples(transformed_samples, labels)
        clustered_silhouettes = []
        for cluster_label in range(num_clusters):
            cluster = np.nonzero(labels == cluster_label)
            clustered.append(cluster)
            clustered_silhouettes.append(np.average(sample_silhouette_values[cluster]))
        non_clustered = np.nonzero(labels == -1)
        summarized_line_list = _summarize_clusters(samples=samples, clusters=clustered)
        clustered_arr = np.empty(len(clustered), dtype=object)
        for i, c in enumerate(clustered):
            clustered_arr[i] = c
    else:
        num_clusters = 0
        non_clustered = []
        summarized_line_list = []
        clustered_silhouettes = []
        clustered_arr = np.array([])
    return ClusteringResult(num_clusters=num_clusters, labels=labels, clustered=clustered_arr,
                            non_clustered=np.array(non_clustered), summarized_line_list=summarized_line_list,
                            clustered_silhouettes=np.array(clustered_silhouettes))


def _summarize_clusters(samples, clusters, summarizer=np.average):
    """One line per cluster: the average of (A, F, PH) and the HARMONIC average of T2 (the cluster's T2 column is
    inverted before and after the summary).  Reference: llckbdm.py:324-353."""
    line_list = []
    if summarizer is None:
        summarizer = np.average
    for cluster in clusters:
        cluster_samples = samples[cluster]                 # fancy indexing: a copy, as in the reference
        cluster_samples[:, 1] = 1 / cluster_samples[:, 1]
        summarized_cluster = summarizer(cluster_samples, axis=0)
        summarized_cluster[1] = 1 / summarized_cluster[1]
        line_list.append(summarized_cluster)
    return np.array(line_list)
