"""Host-side helpers mirrored from the reference's helpers.py (name builder, result printer)."""
import numpy as np


def build_experiment_name(segm_net='fcn8', kind='fcn8', concat_h=[], optimizer='rmsprop',
                          training_loss=['crossentropy'], learning_rate=0.0001, lr_anneal=0.99,
                          data_aug=False, weight_decay=0.0001, dropout=0.5, noise=0.0,
                          from_gt=False, temperature=1.0, n_filters=64, conv_before_pool=1,
                          skip=True, additional_pool=0, unpool_type='standard', ae_h=False,
                          path_weights='', layer='probs_dimshuffle', exp_name='', bn=0):
    """Checkpoint directory name; same keyword set and string as reference helpers.py:118-169
    (golden string: plots.ipynb:84)."""
    name = exp_name + segm_net + '_' + kind + '_' + '_'.join(concat_h)
    if kind == 'standard':
        name += '_f' + str(n_filters) + 'c' + str(conv_before_pool) + 'p' + str(additional_pool)
        name += '_skip' if skip else ''
        name += '_' + unpool_type
    name += ('_dropout' + str(dropout)) if dropout > 0. else ''
    name += '_' + '_'.join(training_loss)
    name += ('_fromgt' if from_gt else '_fromfcn8') + '_z' + str(noise)
    name += '_data_aug' if bool(data_aug) else ''
    name += ('_T' + str(temperature)) if not from_gt else ''
    name += '_' + optimizer + '_lr' + str(learning_rate) + '_anneal' + str(lr_anneal) + \
        '_decay' + str(weight_decay)
    name += '_pretrained' if len(path_weights) > 0 else ''
    name += '_PlugPlay' if ae_h else ''
    name += '_' + layer
    name += '_bn' if bn else ''
    return name


def results_line(rec, acc, jacc, nbatches):
    """(loss, acc, jaccard) exactly as reference helpers.py:172-177 aggregates them: loss/acc
    are sums of per-batch means divided by nbatches, IoU is nanmean(sum num / sum denom)."""
    jacc = np.asarray(jacc, dtype=np.float64)
    with np.errstate(invalid='ignore', divide='ignore'):
        jacc_mean = float(np.nanmean(jacc[0, :] / jacc[1, :]))
    return float(rec) / nbatches, float(acc) / nbatches, jacc_mean


def print_results(st, rec, acc, jacc, nbatches):
    loss, a, j = results_line(rec, acc, jacc, nbatches)
    print(st)
    print('    Loss: ' + str(loss))
    print('    Acc: ' + str(a))
    print('    Jaccard: ' + str(j))
