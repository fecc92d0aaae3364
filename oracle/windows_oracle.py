"""TEST INFRASTRUCTURE ONLY (never imported by the product path): NumPy restatement of the reference's window
loop, aggregate scaling and weak labels, the CPU oracle of ``camal_build_windows``.

  * window loop + NaN filter + 4-D layout: ``UKDALE_DataBuilder.get_nilm_dataset``,
    ``src/helpers/data_processing.py:377-400`` (the REFIT / IDEAL builders repeat it).  The builders need the
    datasets' CSV files, which are not in the container, so this loop is restated and NOT pinned by a run of
    the reference ("parity unpinned" for the loop itself);
  * ``NILMscaler.fit / transform`` and ``nilmdataset_to_clfdataset``: the fixtures
    ``tests/golden/windows_ref.npz`` hold the reference's own outputs on the 4-D array built here
    (``tests/golden/make_golden.py::windows_case``), and ``tests/test_oracle_golden.py`` checks this file
    against them.
"""
import numpy as np


def nilm_dataset(stems, window_size, window_stride):
    """data_processing.py:377-400 for one house.  stems [1 + 2 M][T] -> X [n][1 + M][2][W] float64, kept indices."""
    n_rows, T = stems.shape
    m_app = (n_rows - 1) // 2
    if window_size == window_stride:
        n_wins = T // window_stride
    else:
        n_wins = 1 + ((T - window_size) // window_stride)
    X = np.empty((n_wins, 1 + m_app, 2, window_size))
    kept = []
    cpt = 0
    for i in range(n_wins):
        tmp = stems[:, i * window_stride:i * window_stride + window_size]
        if not np.isnan(tmp).any():  # _check_anynan
            kept.append(i)
            X[cpt, 0, 0, :] = tmp[0, :]
            X[cpt, 0, 1, :] = (tmp[0, :] > 0).astype(dtype=int)
            key = 1
            for j in range(1, 1 + m_app):
                X[cpt, j, 0, :] = tmp[key, :]
                X[cpt, j, 1, :] = tmp[key + 1, :]
                key += 2
            cpt += 1
    return X[:cpt], np.array(kept, np.int32)


def scale_aggregate(X, stat1, divisor):
    """torch_dataset.py:108-111 on the aggregate channel, float64."""
    return (X[:, 0, 0, :] - stat1) / divisor


def weak_labels(X, appliance=1):
    """data_processing.py:26-28."""
    return np.array([1 if X[i, appliance, 1, :].any() > 0 else 0 for i in range(len(X))], np.int8)
