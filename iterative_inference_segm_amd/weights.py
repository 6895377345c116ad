"""Checkpoint I/O in the reference's format: np.savez(path, *get_all_param_values(net)), keys
`arr_0..arr_N`, each layer contributing W then b in DFS order (writer
train_dae.py:436-437, readers models/DAE_h.py:52-57, models/fcn8.py:178-180)."""
import numpy as np


def _arity(name):
    # BatchNormLayer contributes beta, gamma, mean, inv_std (lasagne get_all_param_values includes
    # the non-trainable running averages); every other layer W, b
    return 4 if name.endswith('_bn') else 2


def load_param_list(path, order):
    """Reads an `arr_%d` .npz into {name: (W, b)} (or (beta, gamma, mean, inv_std) for `*_bn`
    entries) following `order`."""
    with np.load(path) as f:
        vals = [f['arr_%d' % i] for i in range(len(f.files))]
    want = sum(_arity(n) for n in order)
    if len(vals) != want:
        raise ValueError('%s holds %d arrays, expected %d for %d layers'
                         % (path, len(vals), want, len(order)))
    out, i = {}, 0
    for name in order:
        out[name] = tuple(vals[i:i + _arity(name)])
        i += _arity(name)
    return out


def save_param_list(path, params, order):
    arrs = []
    for name in order:
        arrs.extend(params[name])
    np.savez(path, *arrs)
