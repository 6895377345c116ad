"""GPU: the BASELINE configs that are parity-test cases rather than bench lines, AT WORKLOAD SIZE.

  configs[2]  FC-DenseNet103 + standard DAE (padding 0, h = pool4 stack, 464 ch @14x14), 224x224,
              batch 32, 10 steps                     (models/FCDenseNet.py:61-146,196-219)
  configs[3]  FCN-8 + standard DAE on full 360x480 CamVid frames, batch 32 per GPU, fp32 path
  configs[4]  50 refinement steps at 224x224 with (i) the reference-exact context-module DAE
              (concat_h=['input'], models/contextmod_dae.py:74-105) and (ii) the build-defined
              generalised standard DAE with concat_h=['pool3','pool4'] (SURVEY A9')

Where the oracle (float64, CPU) finishes in seconds it is the checker; at the full batch sizes
the checks are fp32-vs-float64 agreement of the two HIP paths (the float64 path is pinned to the
oracle) and size-independent bit-identity properties.  Free-running fp32 refinement through the
DePool2D equality masks is compared statistically, with the reason stated in DESIGN.md section 4.
"""
import numpy as np
import pytest
import torch

from oracle import contextmod as octx, dae as odae, densenet as oden, fcn8 as ofcn8, refine as orefine
from iterative_inference_segm_amd import synthetic as S
from _parity_helpers import (TOL, agreement, assert_within_reference_sensitivity, host,
                             teacher_forced_mask_check, to64)

pytestmark = pytest.mark.gpu
F32, F64 = torch.float32, torch.float64


def _ii(fcn, dae, dtype):
    from iterative_inference_segm_amd.api import IterativeInference
    return IterativeInference(fcn, dae, 11, [11], dtype=dtype)


def _p64(params):
    return [{k: (np.asarray(v, np.float64) if k != 'kind' else v) for k, v in p.items()}
            for p in params]


# ---------------------------------------------------------------------------------------------
# configs[2]
# ---------------------------------------------------------------------------------------------
def test_config3_densenet103_dae_batch32(built_lib):
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan
    plan = layer_plan()
    # architecture identity: 103 convolutions, 256-channel final stack,
    # pool stacks 112/192/304/464/656 channels
    assert len(plan) == 103 and plan[-1] == ('softmax', 256, 11)
    assert [c for k, c, _ in plan if k == 'td'] == [112, 192, 304, 464, 656]
    params = S.make_densenet_params(plan)
    dp = S.make_dae_params(h_channels=(464,))
    assert dp['conv5_1'][0].shape == (1024, 976, 3, 3)
    B = 32
    X = S.make_images(B, 224, 224, seed=301)

    def make(dtype):
        return _ii(FCDenseNet(params, 11, layer=['pool4'], dtype=dtype),
                   StandardDAE(dp, 11, padding=0, dtype=dtype), dtype)
    ii32, ii64 = make(F32), make(F64)
    o32, o64 = ii32.pred_fcn_fn(X), ii64.pred_fcn_fn(X)
    h32, y32, h64, y64 = host(o32[0]), host(o32[1]), host(o64[0]), host(o64[1])
    assert h32.shape == (B, 464, 14, 14) and y32.shape == (B, 11, 224, 224)
    eh, ey = np.abs(h32 - h64).max() / (1 + np.abs(h64).max()), np.abs(y32 - y64).max()
    print('DenseNet103 batch 32: fp32 vs float64 HIP  h rel %.2e  y abs %.2e' % (eh, ey))
    assert eh <= TOL and ey <= TOL
    assert np.abs(y32.sum(1) - 1).max() <= 1e-5
    # batch-statistics BatchNorm (iterative_inference.py:187, P10) couples the images of a batch:
    # the same 4 images alone give DIFFERENT maps -- a reference batch must stay on one GPU
    y_sub = host(ii32.pred_fcn_fn(X[:4])[1])
    assert np.abs(y_sub - y32[:4]).max() > 1e-3
    # the refinement loop at full size (batch 32, 10 steps) stays a probability-like map ...
    r32 = ii32.refine(o32[:-1], o32[-1], 0.1, 10, early_stop=False)
    a32 = host(r32[0])
    assert list(host(r32[1])) == [10] * B and a32.min() >= 0 and a32.max() <= 1
    # ... and free-running the fp32 path is as close to the float64 path as that path is to itself
    # under a float32-level perturbation of y0 (the loop is chaotic with random weights)
    sub32, sub64 = [t[:4].contiguous() for t in o32], [t[:4].contiguous() for t in o64]
    assert_within_reference_sensitivity(ii32, ii64, sub32, sub64, (1, 2, 10), label='DenseNet103+DAE')
    # teacher-forced arithmetic parity of the DAE at this geometry (padding 0, 464-channel h)
    bits, flips, err = teacher_forced_mask_check(
        _ii(None, ii32.dae, F32), [o32[0][:2].contiguous()], o32[1][:2].contiguous(), to64(dp), 6,
        dae_kw=dict(padding=0))
    print('DAE on the DenseNet host: mask bits %d, near-tie flips %d, forced err %.2e'
          % (bits, flips, err))


def test_config3_densenet103_vs_oracle_batch2(built_lib):
    """The full 103-conv network + DAE against the oracle on a batch of 2 (batch statistics over
    both images): float64 strictly incl. 2 refinement steps, fp32 within 1e-4 on h and y."""
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan
    params = S.make_densenet_params(layer_plan())
    dp = S.make_dae_params(h_channels=(464,))
    X = S.make_images(2, 224, 224, seed=302)
    h_ref, y_ref = oden.densenet_forward(_p64(params), X.astype(np.float64), layer=['pool4'])
    ii64 = _ii(FCDenseNet(params, 11, layer=['pool4'], dtype=F64),
               StandardDAE(dp, 11, padding=0, dtype=F64), F64)
    o64 = ii64.pred_fcn_fn(X)
    assert np.abs(host(o64[0]) - h_ref).max() <= 1e-9 * (1 + np.abs(h_ref).max())
    assert np.abs(host(o64[1]) - y_ref).max() <= 1e-10
    dp64 = to64(dp)
    yii_ref, it_ref = orefine.refine_batch(lambda hh, yy: odae.dae_forward(dp64, hh, yy, padding=0),
                                           [h_ref], y_ref, 0.1, 2)
    Yii, iters, _ = ii64.refine(o64[:-1], o64[-1], 0.1, 2)
    assert list(host(iters)) == list(it_ref)
    assert np.abs(host(Yii) - yii_ref).max() <= 1e-9
    o32 = FCDenseNet(params, 11, layer=['pool4'])(torch.from_numpy(X).cuda())
    assert np.abs(host(o32[0]) - h_ref).max() <= TOL * (1 + np.abs(h_ref).max())
    assert np.abs(host(o32[1]) - y_ref).max() <= TOL


# ---------------------------------------------------------------------------------------------
# configs[3]
# ---------------------------------------------------------------------------------------------
def test_config4_360x480_fp32_refine_vs_oracle(built_lib):
    """One full CamVid frame through the fp32 path: FCN-8 within 1e-4, the DAE teacher-forced
    within 1e-4 with every mask disagreement a verified near-tie, 3 free-running steps
    statistically (558x678 / 279x339 ... maps: the Winograd / halo window geometry of this size)."""
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp, dp = S.make_fcn8_params(), S.make_dae_params()
    ii = _ii(FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle']), StandardDAE(dp, 11), F32)
    X = S.make_images(1, 360, 480, seed=77)
    out = ii.pred_fcn_fn(X)
    H, Y = out[:-1], out[-1]
    h_ref, y_ref = ofcn8.fcn8_forward(to64(fp), X.astype(np.float64), layer=['pool4', 'probs_dimshuffle'])
    assert h_ref.shape == (1, 512, 34, 42)
    assert np.abs(host(H[0]) - h_ref).max() <= TOL * (1 + np.abs(h_ref).max())
    assert np.abs(host(Y) - y_ref).max() <= TOL
    dp64 = to64(dp)
    bits, flips, err = teacher_forced_mask_check(ii, H, Y, dp64, 6)
    print('360x480: mask bits %d, near-tie flips %d, teacher-forced err %.2e' % (bits, flips, err))
    # 1 free-running step against the oracle (iteration bookkeeping; the few near-tie flips move
    # skip-sized values by one pixel, so the bound is statistical), then 3 steps against the
    # float64 HIP path under the reference-sensitivity criterion
    yii_ref, it_ref = orefine.refine_batch(lambda hh, yy: odae.dae_forward(dp64, hh, yy), [h_ref],
                                           y_ref, 0.1, 1)
    Yii, iters, _ = ii.refine(H, Y, 0.1, 1)
    agree, mean_e, max_e, frac = agreement(host(Yii), yii_ref)
    print('360x480 free-running 1 step vs oracle: argmax agreement %.5f mean err %.2e max %.2e '
          'within-1e-4 %.4f' % (agree, mean_e, max_e, frac))
    assert list(host(iters)) == list(it_ref)
    assert agree >= 0.999 and mean_e <= 1e-4
    ii64 = _ii(FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'], dtype=F64),
               StandardDAE(dp, 11, dtype=F64), F64)
    o64 = ii64.pred_fcn_fn(X)
    assert np.abs(host(o64[-1]) - y_ref).max() <= 1e-10
    assert_within_reference_sensitivity(ii, ii64, out, o64, (1, 3), label='360x480')


def test_config4_360x480_batch32_properties(built_lib):
    """configs[3] per-GPU shard (batch 32 of 360x480 frames, 10 steps) on the fp32 path: an image
    refined inside the batch of 32 and the same image in a batch of 2 give BIT-IDENTICAL maps
    (pure data parallelism: how the 256-image global batch shards over 8 GPUs cannot change a
    result), with the work eliminations on and with everything recomputed in full."""
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp, dp = S.make_fcn8_params(), S.make_dae_params()

    def make(elim):
        fcn, dae = FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle']), StandardDAE(dp, 11)
        fcn.fold_border = dae.fold_border = dae.dce = dae.licm = elim
        return _ii(fcn, dae, F32)
    B = 32
    X = S.make_images(B, 360, 480, seed=78)
    ii = make(True)
    ii.prepare(B, 360, 480)
    out = ii.pred_fcn_fn(X)
    Yii, iters, norms = ii.refine(out[:-1], out[-1], 0.1, 10, early_stop=False)
    y = host(Yii)
    assert y.shape == (B, 11, 360, 480) and y.min() >= 0 and y.max() <= 1
    assert list(host(iters)) == [10] * B and np.isfinite(host(norms)).all()
    sub = [3, 30]
    for elim in (True, False):
        jj = make(elim)
        o2 = jj.pred_fcn_fn(X[sub])
        assert np.array_equal(host(o2[-1]), host(out[-1])[sub])
        Y2, _, n2 = jj.refine(o2[:-1], o2[-1], 0.1, 10, early_stop=False)
        assert np.array_equal(host(Y2), y[sub]), 'eliminations %s' % elim
        assert np.array_equal(host(n2), host(norms)[sub])


# ---------------------------------------------------------------------------------------------
# configs[4]
# ---------------------------------------------------------------------------------------------
def test_config5_contextmod_50_steps(built_lib):
    """Variant (i), reference-exact: FCN-8 host, h = the image, context-module DAE, 50 steps at
    224x224.  No pooling masks on this path, so strict parity holds for the fp32 path too: one image
    against the oracle for all 50 steps, a batch of 8 fp32 vs float64 HIP."""
    from iterative_inference_segm_amd.contextmod import ContextModDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp, cp = S.make_fcn8_params(), S.make_contextmod_params()

    def make(dtype):
        return _ii(FCN8(fp, 11, layer=['input', 'probs_dimshuffle'], dtype=dtype),
                   ContextModDAE(cp, 11, dtype=dtype), dtype)
    ii32, ii64 = make(F32), make(F64)
    B = 8
    X = S.make_images(B, 224, 224, seed=401)
    o32, o64 = ii32.pred_fcn_fn(X), ii64.pred_fcn_fn(X)
    r32 = ii32.refine(o32[:-1], o32[-1], 0.1, 50)
    r64 = ii64.refine(o64[:-1], o64[-1], 0.1, 50)
    a32, a64 = host(r32[0]), host(r64[0])
    e = np.abs(a32 - a64).max()
    print('contextmod 50 steps, batch 8: fp32 vs float64 HIP max-abs %.2e, iterations %s'
          % (e, sorted(set(host(r64[1]).tolist()))))
    assert e <= TOL and list(host(r32[1])) == list(host(r64[1]))
    # oracle: image 0, all 50 steps (h = the image itself)
    cp64 = to64(cp)
    _, y_ref = ofcn8.fcn8_forward(to64(fp), X[:1].astype(np.float64), layer=['input', 'probs_dimshuffle'])
    yii_ref, it_ref = orefine.refine_batch(lambda hh, yy: octx.contextmod_forward(cp64, hh, yy),
                                           [X[:1].astype(np.float64)], y_ref, 0.1, 50)
    assert int(host(r64[1])[0]) == int(it_ref[0]) == int(host(r32[1])[0])
    assert np.abs(a64[:1] - yii_ref).max() <= 1e-9
    assert np.abs(a32[:1] - yii_ref).max() <= TOL


@pytest.mark.parametrize('size', [(224, 224), (37, 150)])
def test_contextmod_fused_tail_is_bit_identical(built_lib, size, monkeypatch):
    """dilconv6 + dilconv7 + softmax + update as ONE launch (csrc/conv_small.hip ctx_tail_kernel,
    models/contextmod_dae.py:98-105 + iterative_inference.py:270-277) against the three separate launches:
    the refined map bit for bit, the same iteration counts (stop test on), the norms to rounding (they are
    summed per tile instead of per 256 pixels); eager loop and graph replay, two batches through one engine
    (the second one replays the first one's captured step on the kept session buffers)."""
    from iterative_inference_segm_amd import ops
    from iterative_inference_segm_amd.contextmod import ContextModDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp, cp = S.make_fcn8_params(), S.make_contextmod_params()
    H, W = size
    B = 3
    Xs = [S.make_images(B, H, W, seed=411 + i) for i in range(2)]

    from iterative_inference_segm_amd import contextmod as CM

    def run(fused, graph, hsplit=True):
        monkeypatch.setattr(ops, 'CTX_TAIL', fused)
        # (the cached image half of the first layer -- the y half continues its FMA chain -- on and off as well)
        monkeypatch.setattr(CM, 'CTX_HSPLIT', hsplit)
        ii = _ii(FCN8(fp, 11, layer=['input', 'probs_dimshuffle']), ContextModDAE(cp, 11), F32)
        outs = []
        for X in Xs:
            o = ii.pred_fcn_fn(X)
            # eps chosen inside the range the norms pass through, so that images do stop on the way
            r = ii.refine(o[:-1], o[-1], 0.5, 12, eps=0.02, graph=graph, first_reconstruction=True)
            outs.append([host(t) for t in r])
        return outs
    ref = run(False, False, hsplit=False)
    for fused, graph, hsplit in ((True, False, True), (True, True, True), (False, False, True), (True, True, False)):
        got = run(fused, graph, hsplit)
        for (y0, it0, n0, r0), (y1, it1, n1, r1) in zip(ref, got):
            assert np.array_equal(y0, y1), np.abs(y0 - y1).max()
            assert np.array_equal(r0, r1)
            assert list(it0) == list(it1)
            assert np.allclose(n0, n1, rtol=1e-9, atol=0)
    print('contextmod fused tail %dx%d: iterations %s' % (H, W, [list(o[1]) for o in ref]))


def test_config5_multi_concat_standard_dae_50_steps(built_lib):
    """Variant (ii), build-defined (no reference parity possible, SURVEY A9'): standard DAE with
    concat_h=['pool3','pool4'] (256 + 512 channels of the FCN-8 host), pad-100, 50 steps at
    224x224.  Oracle (same generalisation): one image, float64 strictly for 3 steps and the fp32
    DAE teacher-forced; full length: the long chain through the border stores is BIT-IDENTICAL to
    full recomputation and independent of the batch composition; fp32 vs float64 statistically."""
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    concat_h = ['pool3', 'pool4']
    fp = S.make_fcn8_params()
    dp = S.make_dae_params(h_channels=(256, 512), concat_h=concat_h)
    kw = dict(concat_h=concat_h, pad_multi_concat=True)

    def make(dtype, elim=True):
        fcn = FCN8(fp, 11, layer=concat_h + ['probs_dimshuffle'], dtype=dtype)
        dae = StandardDAE(dp, 11, dtype=dtype, **kw)
        fcn.fold_border = dae.fold_border = dae.dce = dae.licm = elim
        return _ii(fcn, dae, dtype)
    B = 4
    X = S.make_images(B, 224, 224, seed=402)
    ii32 = make(F32)
    o32 = ii32.pred_fcn_fn(X)
    assert [tuple(h.shape[1:]) for h in o32[:-1]] == [(256, 52, 52), (512, 26, 26)]
    Y32, it32, n32 = ii32.refine(o32[:-1], o32[-1], 0.1, 50, early_stop=False)
    a32 = host(Y32)
    assert list(host(it32)) == [50] * B and a32.min() >= 0 and a32.max() <= 1
    # 50 steps with every elimination off, image 2 alone: same bits
    jj = make(F32, elim=False)
    o1 = jj.pred_fcn_fn(X[2:3])
    Y1, _, n1 = jj.refine(o1[:-1], o1[-1], 0.1, 50, early_stop=False)
    assert np.array_equal(host(Y1), a32[2:3]) and np.array_equal(host(n1), host(n32)[2:3])
    # float64 HIP: the fp32 path vs the reference's own sensitivity along the 50-step chain
    ii64 = make(F64)
    o64 = ii64.pred_fcn_fn(X[:2])
    assert_within_reference_sensitivity(make(F32), ii64, [t[:2].contiguous() for t in o32], o64,
                                        (1, 2, 10, 50), label='multi-concat DAE')
    # oracle on image 0: float64 strict (3 steps), fp32 teacher-forced
    dp64 = to64(dp)
    ref = ofcn8.fcn8_forward(to64(fp), X[:1].astype(np.float64), layer=concat_h + ['probs_dimshuffle'])
    h_ref, y_ref = ref[:-1], ref[-1]
    dae_fn = lambda hh, yy: odae.dae_forward(dp64, hh, yy, **kw)
    yii_ref, it_ref = orefine.refine_batch(dae_fn, h_ref, y_ref, 0.1, 3)
    k64 = make(F64)
    ok = k64.pred_fcn_fn(X[:1])
    Yk, itk, _ = k64.refine(ok[:-1], ok[-1], 0.1, 3)
    assert list(host(itk)) == list(it_ref) and np.abs(host(Yk) - yii_ref).max() <= 1e-9
    k32 = make(F32)
    o = k32.pred_fcn_fn(X[:1])
    bits, flips, err = teacher_forced_mask_check(k32, o[:-1], o[-1], dp64, 6, dae_kw=kw)
    print('multi-concat DAE: mask bits %d, near-tie flips %d, forced err %.2e' % (bits, flips, err))


# ---------------------------------------------------------------------------------------------
# dae kind 'fcn8' at full size (the DEFAULT kind of inference(), iterative_inference.py:64)
# ---------------------------------------------------------------------------------------------
def test_fcn8_kind_dae_full_size(built_lib):
    """`buildFCN8_DAE` (models/fcn8_dae.py:19-271) at real widths and 224x224: an FCN-8 on y (11
    channels) with h concatenated at the input (the image, 3 channels) and after pool3 / pool4 (the
    host FCN-8's 256- / 512-channel maps), fc6 / fc7 at 4096 channels -- ~120 GFLOP per image and
    step, the heaviest DAE of the repository.  One image, two refinement steps:
      * float64 HIP path against the float64 oracle, strict (1e-10; r, the refined map, iteration
        counts);
      * fp32 HIP path against the oracle <= 1e-4 on r and on the refined map -- a FIXED tolerance:
        this kind has no DePool2D equality masks (transposed-conv upsampling, fcn8_dae.py:128-160),
        hence no discontinuity;
      * the bf16-operand mode runs it (finite, inside [0, 1], argmax agreement with float64 printed).
    Timing of this DAE: scripts/bench_configs.py row `fcn8dae`."""
    from iterative_inference_segm_amd.fcn8 import FCN8, FCN8DAE
    concat_h = ['input', 'pool3', 'pool4']
    fp = S.make_fcn8_params(seed=61)
    dp = S.make_fcn8_dae_params(concat_h=concat_h, h_channels=(3, 256, 512), seed=62)
    X = S.make_images(1, 224, 224, seed=63)
    dp64, fp64 = to64(dp), to64(fp)
    res = {}
    for key, dtype, mma in (('f64', F64, None), ('f32', F32, None), ('bf16', F32, 'bf16')):
        ii = _ii(FCN8(fp, 11, layer=concat_h + ['probs_dimshuffle'], dtype=dtype, mma=mma),
                 FCN8DAE(dp, 11, concat_h=concat_h, dtype=dtype, mma=mma), dtype)
        out = ii.pred_fcn_fn(X)
        H, Y = out[:-1], out[-1]
        r = ii.pred_dae_fn(*(list(H) + [Y]))
        Yii, iters, _ = ii.refine(H, Y, 0.2, 2)
        res[key] = (H, Y, host(r).astype(np.float64), host(Yii).astype(np.float64), list(host(iters)))
        del ii
        torch.cuda.empty_cache()
    h_ref = ofcn8.fcn8_forward(fp64, X.astype(np.float64), layer=concat_h + ['probs_dimshuffle'])
    y_ref = h_ref[-1]
    dae_fn = lambda hh, yy: ofcn8.fcn8_forward(dp64, yy, concat_h=concat_h, h_list=hh)[0]
    assert np.abs(host(res['f64'][1]) - y_ref).max() <= 1e-10
    r_ref = dae_fn(list(h_ref[:-1]), y_ref)
    yii_ref, it_ref = orefine.refine_batch(dae_fn, list(h_ref[:-1]), y_ref, 0.2, 2)
    e64 = (np.abs(res['f64'][2] - r_ref).max(), np.abs(res['f64'][3] - yii_ref).max())
    e32 = (np.abs(res['f32'][2] - r_ref).max(), np.abs(res['f32'][3] - yii_ref).max())
    a16 = float((res['bf16'][3].argmax(1) == yii_ref.argmax(1)).mean())
    print('fcn8-kind DAE, 224x224, real widths: float64 HIP vs oracle r %.2e refined %.2e | fp32 r %.2e '
          'refined %.2e | bf16 refined argmax agreement %.4f' % (e64 + e32 + (a16,)))
    assert max(e64) <= 1e-10 and res['f64'][4] == list(it_ref)
    assert max(e32) <= TOL and res['f32'][4] == list(it_ref)
    b = res['bf16'][3]
    assert np.isfinite(b).all() and b.min() >= 0 and b.max() <= 1


# ---------------------------------------------------------------------------------------------
# whole batches in flight (api.EnginePool) on the other model families
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('family', ['densenet_c8', 'contextmod', 'fcn8_f64'])
def test_engine_pool_other_families_bit_identical(built_lib, family):
    """Batches round-robin over three engines / HIP streams against one engine in order, bit for bit: the
    FC-DenseNet host on bf16 C8 stacks (per-geometry level buffers, batch-statistics BatchNorm scratch) with a
    pad-0 DAE, the context-module DAE (kept sessions, fused tail, replayed graph) and the float64 path."""
    from iterative_inference_segm_amd.api import EnginePool
    from iterative_inference_segm_amd.contextmod import ContextModDAE
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan
    from iterative_inference_segm_amd.fcn8 import FCN8
    if family == 'densenet_c8':
        nl = [2, 3, 2, 2, 2, 2, 2, 2, 2, 3, 2]
        dparams = S.make_densenet_params(layer_plan(n_layers_per_block=nl, n_first=16, growth=16), seed=7)
        hch = 16 + 16 * (2 + 3 + 2 + 2)
        dp = S.make_dae_params(h_channels=(hch,), n_filters=16, seed=9)
        size, dtype = (64, 96), F32

        def engine():
            net = FCDenseNet(dparams, 11, layer=['pool4'], n_layers_per_block=nl, growth=16, mma='bf16c8')
            return _ii(net, StandardDAE(dp, 11, n_filters=16, padding=0, mma='bf16c8'), F32)
    elif family == 'contextmod':
        fp, cp = S.make_fcn8_params(width_div=16, fc_channels=64, seed=1), S.make_contextmod_params()
        size, dtype = (48, 70), F32

        def engine():
            return _ii(FCN8(fp, 11, layer=['input', 'probs_dimshuffle']), ContextModDAE(cp, 11), F32)
    else:
        fp = S.make_fcn8_params(width_div=16, fc_channels=64, seed=1)
        dp = S.make_dae_params(h_channels=(fp['conv4_3'][0].shape[0],), n_filters=4, seed=2)
        size, dtype = (64, 48), F64

        def engine():
            return _ii(FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'], dtype=F64),
                       StandardDAE(dp, 11, n_filters=4, dtype=F64), F64)
    Xs = [torch.from_numpy(S.make_images(3, size[0], size[1], seed=70 + i)).to(dtype).cuda() for i in range(7)]

    def batch(ii, X):
        out = ii.pred_fcn_fn(X)
        return ii.refine(out[:-1], out[-1], 0.1, 6, early_stop=False)[:3]

    single = engine()
    want = [batch(single, X) for X in Xs]
    torch.cuda.synchronize()
    pool = EnginePool([engine() for _ in range(3)])
    got = []
    for X in Xs:
        with pool.lane(X) as ii:
            got.append(batch(ii, X))
    pool.join()
    torch.cuda.synchronize()
    for k, (g, w) in enumerate(zip(got, want)):
        for a, b in zip(g, w):
            assert torch.equal(a, b), (family, k)


def test_c8_dae_without_records_replays_one_captured_step_across_batches(built_lib):
    """An h without a provenance record (the FC-DenseNet host; any h the caller made himself): the C8 DAE hands out
    the SAME session buffers unprimed for every batch of a geometry (StandardDAE.new_session), so the refinement
    step captured on the first batch is replayed for the next ones -- and every batch's result is the eager
    loop's, bit for bit (nothing of a previous batch survives: different h and y each time, one batch repeated)."""
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan
    nl = [2, 3, 2, 2, 2, 2, 2, 2, 2, 3, 2]
    dparams = S.make_densenet_params(layer_plan(n_layers_per_block=nl, n_first=16, growth=16), seed=7)
    dp = S.make_dae_params(h_channels=(16 + 16 * (2 + 3 + 2 + 2),), n_filters=16, seed=9)

    def engine():
        net = FCDenseNet(dparams, 11, layer=['pool4'], n_layers_per_block=nl, growth=16, mma='bf16c8')
        return _ii(net, StandardDAE(dp, 11, n_filters=16, padding=0, mma='bf16c8'), F32)
    Xs = [torch.from_numpy(S.make_images(3, 64, 96, seed=80 + i)).cuda() for i in (0, 1, 2, 1, 3)]
    outs = {}
    for graph in (False, None):
        ii = engine()
        res = []
        for X in Xs:
            o = ii.pred_fcn_fn(X)
            res.append([t.clone() for t in ii.refine(o[:-1], o[-1], 0.1, 6, early_stop=False, graph=graph)[:3]])
        outs[graph] = res
        if graph is None:
            assert len(ii._graphs) == 1
            ctx = next(iter(ii._graphs.values()))
            assert ctx['graph'] is not None and len(ii.dae._scratch_sessions) == 1
    for k, (a, b) in enumerate(zip(outs[False], outs[None])):
        for x, y in zip(a, b):
            assert torch.equal(x, y), k
    assert torch.equal(outs[None][1][0], outs[None][3][0])      # the repeated batch: same result both times
