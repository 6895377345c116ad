"""Dataset iterators with the contract the reference's loop relies on (reference
data_loader.py:7-121 wraps the un-vendored `dataset_loaders` package; iterative_inference.py
uses `.next()`, `.nbatches`, `.non_void_nclasses`, `.void_labels`, `.data_shape`, `.cmap`,
`.mask_labels`, :120-125,233,316).

`dataset_loaders` and the datasets are not reachable here, so the default is a seeded
synthetic iterator with CamVid's shape contract (11 classes + void, one-hot labels with the void
channel last, float32 RGB in [0,1]).  If `dataset_loaders` is importable it is used exactly as
the reference does.
"""
import numpy as np

from . import synthetic as S

CAMVID_LABELS = ['sky', 'building', 'column_pole', 'road', 'sidewalk', 'tree', 'sign', 'fence',
                 'car', 'pedestrian', 'byciclist', 'void']


class SyntheticSegmentationIterator:
    """CamVid-shaped synthetic split: `n_images` images of `image_size`, batches of `batch_size`."""

    def __init__(self, n_images=20, image_size=(224, 224), batch_size=10, n_classes=11,
                 one_hot=True, return_0_255=False, seed=1234):
        self.n_images, self.batch_size = int(n_images), int(batch_size)
        self.h, self.w = image_size
        self.non_void_nclasses = n_classes
        self.void_labels = [n_classes]
        self.data_shape = (3, self.h, self.w)
        self.one_hot, self.return_0_255 = one_hot, return_0_255
        self.nbatches = (self.n_images + self.batch_size - 1) // self.batch_size
        self.mask_labels = CAMVID_LABELS[:n_classes] + ['void'] if n_classes != 11 else CAMVID_LABELS
        rng = np.random.default_rng(7)
        self.cmap = rng.random((n_classes + 1, 3)).tolist()
        self.seed = seed
        self._i = 0

    def batch(self, i):
        """Batch `i` (deterministic in (seed, i): ranks can pick their own shard)."""
        n = min(self.batch_size, self.n_images - i * self.batch_size)
        X = S.make_images(n, self.h, self.w, seed=self.seed + 1000 * i)
        L = S.make_labels(n, self.h, self.w, n_classes=self.non_void_nclasses,
                          seed=self.seed + 1000 * i + 1)
        if self.return_0_255:
            X = X * 255.0
        if not self.one_hot:
            L = L.argmax(1).astype(np.int32)
        return X, L

    def next(self):
        out = self.batch(self._i % self.nbatches)
        self._i += 1
        return out

    __next__ = next

    def __iter__(self):
        return self


def load_data(dataset, data_augm_kwargs={}, one_hot=False, batch_size=[10, 10, 10],
              shuffle_train=True, return_0_255=False, which_set='all', synthetic=None,
              n_images=20, image_size=(224, 224), seed=1234):
    """Same positional/keyword arguments as reference data_loader.py:7-9 (+ synthetic knobs).
    Returns the iterator of `which_set` ('train' | 'val'/'valid' | 'test'), or all three."""
    if dataset not in ('camvid', 'polyps912', 'em'):
        raise ValueError('Unknown dataset')                      # data_loader.py:110-111
    if synthetic is None:
        try:
            import dataset_loaders  # noqa: F401
            synthetic = False
        except ImportError:
            synthetic = True
    if not synthetic:
        raise NotImplementedError(
            'real datasets need the external `dataset_loaders` package and data on disk '
            '(reference data_loader.py:1-4); pass synthetic=True')
    idx = {'train': 0, 'val': 1, 'valid': 1, 'test': 2}
    mk = lambda k: SyntheticSegmentationIterator(n_images, image_size, batch_size[k], 11, one_hot,
                                                 return_0_255, seed + 17 * k)
    if which_set == 'all':
        return mk(0), mk(1), mk(2)
    if which_set not in idx:
        raise ValueError('Unknown set requested')                # data_loader.py:119-120
    return mk(idx[which_set])
