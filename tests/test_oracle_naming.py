"""CPU: the one golden vector the reference holds for this path -- the experiment name recorded
at plots.ipynb:84 for the inputs of plots.ipynb cell 2 -- pins both the oracle's and the
product's `build_experiment_name` (reference helpers.py:118-169)."""
import pytest

from oracle.naming import build_experiment_name as oracle_name
from iterative_inference_segm_amd.helpers import build_experiment_name as product_name

GOLDEN = ('fcn8_standard_pool4_f32c1p2_skip_trackind_dropout0.5_crossentropy_fromfcn8_z0_data_aug'
          '_T1.0_rmsprop_lr0.0001_anneal0.99_decay0.0001_probs_dimshuffle')
KW = dict(kind='standard', dropout=0.5, skip=True, unpool_type='trackind', n_filters=32,
          conv_before_pool=1, additional_pool=2, concat_h=['pool4'], noise=0, from_gt=False,
          temperature=1.0, layer='probs_dimshuffle', exp_name='', data_aug=True,
          training_loss=['crossentropy'], learning_rate=0.0001, lr_anneal=0.99,
          weight_decay=0.0001, optimizer='rmsprop')


@pytest.mark.parametrize('fn', [oracle_name, product_name])
def test_golden_string(fn):
    assert fn(segm_net='fcn8', **KW) == GOLDEN


@pytest.mark.parametrize('fn', [oracle_name, product_name])
def test_branches(fn):
    kw = dict(KW, kind='fcn8', dropout=0.0, from_gt=True, data_aug=False, path_weights='x',
              ae_h=True, bn=1, exp_name='flip_', concat_h=['input', 'pool3'])
    assert fn(segm_net='densenet', **kw) == (
        'flip_densenet_fcn8_input_pool3_crossentropy_fromgt_z0_rmsprop_lr0.0001_anneal0.99'
        '_decay0.0001_pretrained_PlugPlay_probs_dimshuffle_bn')
