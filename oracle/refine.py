"""Oracle: the iterative-inference loop (reference iterative_inference.py:258-284).
TEST INFRASTRUCTURE.
"""
import numpy as np

EPSILON = 1e-3  # iterative_inference.py:53 (overrides :30)


def refine_image(dae_fn, h_im, y_im, step, num_iter, eps=EPSILON, trace=None):
    """One image (batch dim 1), exactly the reference schedule (F1, F3):

        for it in range(num_iter):                        # :265
            grad = de_fn(h, y) = -(r(y,h) - y)            # :203-204,267
            y = clip(y - step*grad, 0, 1)                 # :270,273
            norm = np.linalg.norm(grad, axis=1).mean()    # :275
            if norm < eps: break                          # :276-277  (AFTER the update)

    `dae_fn(h_list, y) -> r`.  Returns (y, iterations_executed).
    """
    y = y_im
    iters = 0
    for _ in range(num_iter):
        r = dae_fn(h_im, y)
        grad = -(r - y)
        y = np.clip(y - step * grad, 0.0, 1.0)
        iters += 1
        norm = np.linalg.norm(grad, axis=1).mean()
        if trace is not None:
            trace.append(float(norm))
        if norm < eps:
            break
    return y, iters


def refine_batch(dae_fn, H, Y, step, num_iter, eps=EPSILON):
    """iterative_inference.py:257-284: per-image loop over the batch, results concatenated.
    Returns (Y_ii, iters[B])."""
    outs, iters = [], []
    for im in range(Y.shape[0]):
        h_im = [el[np.newaxis, im] for el in H]          # :260
        y_im = Y[np.newaxis, im]                         # :261
        y_im, it = refine_image(dae_fn, h_im, y_im, step, num_iter, eps)
        outs.append(y_im)
        iters.append(it)
    return np.concatenate(outs, axis=0), np.asarray(iters, dtype=np.int64)
