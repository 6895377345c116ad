"""Checkpoint I/O in the reference's format: np.savez(path, *get_all_param_values(net)), keys
`arr_0..arr_N`, each layer contributing W then b in DFS order (writer
train_dae.py:436-437, readers models/DAE_h.py:52-57, models/fcn8.py:178-180)."""
import numpy as np


def load_param_list(path, order):
    """Reads an `arr_%d` .npz into {name: (W, b)} following `order`."""
    with np.load(path) as f:
        vals = [f['arr_%d' % i] for i in range(len(f.files))]
    if len(vals) != 2 * len(order):
        raise ValueError('%s holds %d arrays, expected %d (%d layers x (W, b))'
                         % (path, len(vals), 2 * len(order), len(order)))
    return {name: (vals[2 * i], vals[2 * i + 1]) for i, name in enumerate(order)}


def save_param_list(path, params, order):
    arrs = []
    for name in order:
        arrs.extend(params[name])
    np.savez(path, *arrs)
