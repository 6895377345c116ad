"""Oracle: val_fn metrics (reference metrics.py:11-65,144-156; iterative_inference.py:206-210;
helpers.py:172-177).  TEST INFRASTRUCTURE.
"""
import numpy as np


def _to_2d(t):
    # dimshuffle (0,2,3,1) + reshape to (B*H*W, C): iterative_inference.py:193-200
    return np.transpose(t, (0, 2, 3, 1)).reshape(-1, t.shape[1])


def jaccard(y_pred, y_true, n_classes):
    """metrics.py:11-37 with one_hot=True: argmax both, cm[i,j] = #(pred==i & true==j) for
    i,j < n_classes (rows = prediction); returns stack([TP, TP+FP+FN]) of shape (2, C).
    np.argmax and T.argmax both return the FIRST maximal index."""
    p = np.argmax(_to_2d(y_pred), axis=1)
    t = np.argmax(_to_2d(y_true), axis=1)
    cm = np.zeros((n_classes, n_classes), dtype=np.float64)
    for i in range(n_classes):
        for j in range(n_classes):
            cm[i, j] = np.sum((p == i) & (t == j))
    tp = cm.diagonal()
    fp = cm.sum(1) - tp
    fn = cm.sum(0) - tp
    return np.stack([tp, tp + fp + fn], axis=0)


def accuracy(y_pred, y_true, void_labels):
    """metrics.py:40-65 with one_hot=True: void-masked mean of (argmax pred == argmax true)."""
    p = np.argmax(_to_2d(y_pred), axis=1)
    t = np.argmax(_to_2d(y_true), axis=1)
    acc = (p == t).astype(np.float64)
    mask = np.ones_like(acc)
    for el in void_labels:
        mask[t == el] = 0.0
    return float((acc * mask).sum() / mask.sum())


def squared_error(y_pred, y_true, void):
    """metrics.py:144-156, int `void` branch: per-pixel mean over channels of
    (y - t[:, :void])^2, masked by t[:, :void].sum(1), normalised by the mask sum."""
    t = y_true[:, :void]
    loss = ((y_pred - t) ** 2).mean(axis=1)
    mask = t.sum(axis=1)
    return float((loss * mask).sum() / mask.sum())


def val_fn(y, t, n_classes, void_labels):
    """iterative_inference.py:206-210 -> [acc, jacc(2,C), mse].  `void` = n_classes when the
    dataset has void labels (:125)."""
    void = n_classes if any(void_labels) else n_classes + 1
    return (accuracy(y, t, void_labels), jaccard(y, t, n_classes), squared_error(y, t, void))


def summarize(rec, acc, jacc, nbatches):
    """helpers.py:172-177 print_results arithmetic: (loss, acc, mean jaccard).  IoU is
    sum-then-divide with nanmean over classes; loss/acc are means of per-batch means."""
    with np.errstate(invalid='ignore', divide='ignore'):
        jacc_mean = float(np.nanmean(jacc[0, :] / jacc[1, :]))
    return rec / nbatches, acc / nbatches, jacc_mean
