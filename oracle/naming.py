"""Oracle: checkpoint-directory name builder (reference helpers.py:118-169).
TEST INFRASTRUCTURE.  Pinned by the one golden vector the reference holds: the string
recorded at plots.ipynb:84 (tests/test_oracle_naming.py).
"""


def build_experiment_name(segm_net='fcn8', kind='fcn8', concat_h=(), optimizer='rmsprop',
                          training_loss=('crossentropy',), learning_rate=0.0001,
                          lr_anneal=0.99, data_aug=False, weight_decay=0.0001, dropout=0.5,
                          noise=0.0, from_gt=False, temperature=1.0, n_filters=64,
                          conv_before_pool=1, skip=True, additional_pool=0,
                          unpool_type='standard', ae_h=False, path_weights='',
                          layer='probs_dimshuffle', exp_name='', bn=0):
    parts = [exp_name + segm_net, kind, '_'.join(concat_h)]          # helpers.py:145
    s = '_'.join(parts)
    if kind == 'standard':                                            # :147-152
        s += '_f%sc%sp%s' % (n_filters, conv_before_pool, additional_pool)
        s += '_skip' if skip else ''
        s += '_' + unpool_type
    if dropout > 0.:                                                  # :154
        s += '_dropout' + str(dropout)
    s += '_' + '_'.join(training_loss)                                # :155
    s += ('_fromgt' if from_gt else '_fromfcn8') + '_z' + str(noise)  # :156-157
    if bool(data_aug):                                                # :158
        s += '_data_aug'
    if not from_gt:                                                   # :159
        s += '_T' + str(temperature)
    s += '_%s_lr%s_anneal%s_decay%s' % (optimizer, learning_rate, lr_anneal, weight_decay)
    if len(path_weights) > 0:                                         # :164
        s += '_pretrained'
    if ae_h:                                                          # :165
        s += '_PlugPlay'
    s += '_' + layer                                                  # :166
    if bn:                                                            # :168
        s += '_bn'
    return s
