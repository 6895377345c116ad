"""Fixed-tolerance end-to-end parity of the two FAST paths (fp32 MFMA, bf16-operand MFMA) on the
DAMPED synthetic workload (iterative_inference_segm_amd/synthetic.py, `DAMPED`).

BASELINE configs[1] as it stands -- real FCN-8 + 64-filter standard DAE (pool4), 224x224, 11
classes, 10 refinement steps of 0.1, early stop off -- with the four gains of the damped set, on
which the reference's own function (float64) is CONTRACTIVE: a perturbation of y0 decays instead of
growing 40x per step as it does with the default (chaotic) set.  So here the free-running refined map
of a float32 implementation can be held to the 1e-4 of north_star with fixed numbers, no fitted
criterion:

  (A) fp32, free-running in y with the float64 trajectory's DePool2D mask DECISIONS injected
      (layers/mylayers.py:111-114 is a discontinuity; everything else is arithmetic):
      max |Y_fp32 - Y_f64| <= 1e-4 after all 10 steps -- strict.
  (B) fp32, free-running, its own masks (the product path, `refine()`): >= 0.999 of the pixels
      within 1e-4, max error bounded; every mask bit that differs from the float64 trajectory sits in
      a pooling window whose float64 top-2 gap is tiny against the activations (a verified near-tie),
      and such bits are < 1e-5 of all bits at every step.
  (C) the float64 HIP path itself is pinned to the float64 CPU oracle on THIS set (one image, two
      steps, 1e-10), as it is on the default set by tests/test_gpu_f64.py.
  (D) bf16 operands (fp32 activations, mma='bf16', and bf16 C8 activations, mma='bf16c8'): refined
      argmax agreement >= 0.99 with float64; labels = argmax of the float64-refined map, so
      mIoU(float64) == 1 and |mIoU(bf16) - 1| <= 0.05 is a real statement.  The split-operand mode
      (mma='bf16x3' on the DAE, FCN-8 fp32: DESIGN 3.8) is held to the fp32 path's numbers here
      (agreement >= 0.9999, mIoU within 1e-3 of 1); its pixel-level 1e-4 check is tests/test_gpu_x3.py.
"""
import numpy as np
import pytest
import torch

from oracle import dae as odae
from oracle import fcn8 as ofcn8
from oracle import refine as orefine
from iterative_inference_segm_amd import synthetic as S
from _parity_helpers import TOL, host, to64

pytestmark = pytest.mark.gpu

F32, F64 = torch.float32, torch.float64
STEP, NSTEPS, LEVELS = 0.1, 10, 6


def engine(dtype, mma=None):
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp, dp, temp = S.make_damped_set()
    return IterativeInference(
        FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'], temperature=temp, dtype=dtype,
             mma=None if mma == 'bf16x3' else mma),
        StandardDAE(dp, 11, dtype=dtype, mma=mma), 11, [11], dtype=dtype)


def mask_bits(pre, pool):
    """DePool2D's mask (pre == repeat(pool)) as a bool tensor over the pooled extent."""
    h2, w2 = pool.shape[2] * 2, pool.shape[3] * 2
    return pre[:, :, :h2, :w2] == pool.repeat_interleave(2, 2).repeat_interleave(2, 3)


def top2_gap(pre, pool):
    """Per pooling window: (max - second largest, max) of the pre-pool map."""
    B, C, h, w = pool.shape
    win = pre[:, :, :2 * h, :2 * w].reshape(B, C, h, 2, w, 2).permute(0, 1, 2, 4, 3, 5)
    top = win.reshape(B, C, h, w, 4).topk(2, dim=-1).values
    return top[..., 0] - top[..., 1], top[..., 0]


def stepwise(ii, H, Y, n, inject=None, record=False):
    """The loop of iterative_inference.py:265-273 one DAE forward at a time (early stop off).
    inject: per step {level: (bool mask, pre-pool (H, W))} decisions to use instead of the path's
    own (a tensor pair whose equality reproduces them exactly goes in as `mask_override`); record: return
    the per-step {level: (pre, pool)} of the path (device tensors)."""
    from iterative_inference_segm_amd import ops
    y = Y.clone()
    st = ops.RefineState(y.shape[0], y.shape[2], y.shape[3], y.device)
    rec = []
    for k in range(n):
        override = None
        if inject is not None:
            override = {}
            for p, (m, pre_hw) in inject[k].items():
                full = torch.zeros((m.shape[0], m.shape[1]) + tuple(pre_hw), dtype=y.dtype,
                                   device=y.device)
                full[:, :, :m.shape[2], :m.shape[3]] = m.to(y.dtype)
                override[p] = (full, torch.ones((m.shape[0], m.shape[1], m.shape[2] // 2,
                                                 m.shape[3] // 2), dtype=y.dtype, device=y.device))
        if record:
            ii.dae.trace = {}
        score = ii.dae.scores(H, y, mask_override=override)
        if record:
            rec.append({p: (ii.dae.trace['pre%d' % p], ii.dae.trace['pool%d' % p])
                        for p in range(1, LEVELS + 1)})
            ii.dae.trace = None
        ops.refine_update(score, y, st, STEP, off=(0, 0))
        ops.refine_finalize(st, -1.0)
    return y, rec


def test_damped_set_is_contractive_in_float64(built_lib):
    """The property the set is built for, under test: the float64 path on y0 and on y0 + 1e-5 u --
    the mean deviation after 10 steps is BELOW the initial one (it grows 4 orders of magnitude on
    the default set: test_reference_function_is_chaotic_at_float32_resolution)."""
    ii = engine(F64)
    X = S.make_images(2, 224, 224, seed=1234)
    out = ii.pred_fcn_fn(X)
    H, Y = out[:-1], out[-1]
    assert float(Y.amax(1).mean()) >= 0.9                       # confident y0
    u = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, size=tuple(Y.shape))).cuda()
    Yp = (Y + 1e-5 * u).clamp(0, 1)
    d0 = float((Y - Yp).abs().mean())
    a = ii.refine(H, Y, STEP, NSTEPS, early_stop=False)[0]
    b = ii.refine(H, Yp, STEP, NSTEPS, early_stop=False)[0]
    e = (a - b).abs()
    frac = float((e.amax(1) <= TOL).double().mean())
    print('damped set, float64, y0 vs y0 + 1e-5 u: mean |dy| %.2e -> %.2e after %d steps, max %.2e, '
          'pixels within 1e-4 %.5f' % (d0, float(e.mean()), NSTEPS, float(e.max()), frac))
    assert float(e.mean()) <= d0 and frac >= 0.99
    assert float((a.argmax(1) == b.argmax(1)).double().mean()) >= 0.9999


def test_float64_path_is_pinned_to_the_oracle_on_the_damped_set(built_lib):
    fp, dp, temp = S.make_damped_set()
    ii = engine(F64)
    X = S.make_images(1, 224, 224, seed=1234)
    out = ii.pred_fcn_fn(X)
    h_ref, y_ref = ofcn8.fcn8_forward(to64(fp), X.astype(np.float64),
                                      layer=['pool4', 'probs_dimshuffle'], temperature=temp)
    assert np.abs(host(out[-1]) - y_ref).max() <= 1e-10
    dp64 = to64(dp)
    # ALL 10 steps of the loop, free-running (the reference of (B) is pinned end to end on the set it is
    # used on, as test_full_size_end_to_end_strict pins it on the default set)
    yii_ref, it_ref = orefine.refine_batch(lambda hh, yy: odae.dae_forward(dp64, hh, yy), [h_ref],
                                           y_ref, STEP, NSTEPS)
    Yii, iters, _ = ii.refine(out[:-1], out[-1], STEP, NSTEPS)
    err = np.abs(host(Yii) - yii_ref).max()
    print('damped set: float64 HIP vs oracle after %d steps: %.3e' % (NSTEPS, err))
    assert list(host(iters)) == list(it_ref) and err <= 1e-9


def test_fp32_free_running_fixed_tolerance(built_lib):
    """(A) and (B) of the module docstring, 2 images, all 10 steps."""
    ii32, ii64 = engine(F32), engine(F64)
    X = S.make_images(2, 224, 224, seed=1234)
    o32, o64 = ii32.pred_fcn_fn(X), ii64.pred_fcn_fn(X)
    H32, Y32, H64, Y64 = o32[:-1], o32[-1], o64[:-1], o64[-1]
    assert float((Y32.double() - Y64).abs().max()) <= TOL      # FCN-8 output (no masks): strict

    # the float64 trajectory, with its mask decisions and pre-pool maps per step
    y64, rec64 = stepwise(ii64, H64, Y64, NSTEPS, record=True)
    base = ii64.refine(H64, Y64, STEP, NSTEPS, early_stop=False)[0]
    assert torch.equal(y64, base)                               # stepwise() IS the refine() loop
    dec64 = [{p: mask_bits(*r[p]) for p in r} for r in rec64]
    inject = [{p: (d[p], r[p][0].shape[2:]) for p in d} for d, r in zip(dec64, rec64)]

    # (A) fp32 arithmetic, float64 decisions: strict 1e-4 on the refined map
    ya, _ = stepwise(ii32, H32, Y32, NSTEPS, inject=inject)
    err_a = float((ya.double() - y64).abs().max())
    print('(A) fp32 with the float64 mask decisions, %d steps free-running in y: max |err| %.3e'
          % (NSTEPS, err_a))
    assert err_a <= TOL

    # (B) the product path; its own masks step by step for the flip accounting
    got = ii32.refine(H32, Y32, STEP, NSTEPS, early_stop=False)[0]
    yb, rec32 = stepwise(ii32, H32, Y32, NSTEPS, record=True)
    assert torch.equal(yb, got)
    e = (got.double() - y64).abs()
    frac = float((e.amax(1) <= TOL).double().mean())
    agree = float((got.argmax(1) == y64.argmax(1)).double().mean())
    print('(B) fp32 free-running, own masks: pixels within 1e-4 %.5f, max %.3e, mean %.3e, argmax '
          'agreement %.6f' % (frac, float(e.max()), float(e.mean()), agree))
    # The CLAIM is the fraction (>= 0.999 of the pixels within 1e-4) and the mean; the pixels beyond 1e-4
    # are where a DePool2D near-tie went the other way (accounted for bit by bit below).  Their size is
    # bounded as a regression check only: 64 images measure max 1.26e-3 (profiles/r04_parity_damped_64.md,
    # scripts/parity_report.py) -- asserted at 2e-3, with at most 1e-6 of the pixels beyond 1e-3.
    assert frac >= 0.999 and float(e.mean()) <= 1e-5
    assert float(e.max()) <= 2e-3 and float((e.amax(1) > 1e-3).double().mean()) <= 1e-6
    assert agree >= 0.9999
    total_flips = 0
    for k in range(NSTEPS):
        bits = flips = 0
        worst = 0.0
        for p in range(1, LEVELS + 1):
            m32 = mask_bits(*rec32[k][p])
            diff = m32 != dec64[k][p]
            bits += diff.numel()
            n = int(diff.sum())
            flips += n
            if n:
                gap, top = top2_gap(*rec64[k][p])
                B, C, h, w = gap.shape
                wdiff = diff.reshape(B, C, h, 2, w, 2).any(dim=5).any(dim=3)
                rel = gap[wdiff] / (1.0 + top[wdiff].abs())
                worst = max(worst, float(rel.max()))
                if k == 0:
                    # same y0 up to the fp32 FCN-8's rounding: the gap is below twice the measured
                    # fp32 error of this very tensor
                    pre_err = float((rec32[0][p][0].double() - rec64[0][p][0]).abs().max())
                    assert pre_err <= 1e-4 * (1 + float(rec64[0][p][0].abs().max()))
                    assert float(gap[wdiff].max()) <= 2 * pre_err
        total_flips += flips
        print('    step %2d: %d of %d mask bits differ, largest float64 top-2 gap among them %.2e '
              '(relative to 1 + |max|)' % (k + 1, flips, bits, worst))
        assert flips <= 1e-5 * bits and worst <= 1e-3
    print('    %d differing bits over %d steps' % (total_flips, NSTEPS))


def test_bf16_agreement_and_miou_fixed_tolerance(built_lib):
    """(D): 16 images, 10 steps; labels are the argmax of the float64-refined map."""
    ii16, ii32, ii64, iic8 = engine(F32, 'bf16'), engine(F32), engine(F64), engine(F32, 'bf16c8')
    iix3 = engine(F32, 'bf16x3')
    cm = {k: np.zeros((11, 12)) for k in ('bf16', 'c8', 'x3', 'f32', 'f64')}
    agree, agree_fcn = {'bf16': [], 'c8': [], 'x3': [], 'f32': []}, []
    for i in range(2):
        X = S.make_images(8, 224, 224, seed=700 + i)
        res = {}
        engines = (('f64', ii64), ('f32', ii32), ('bf16', ii16), ('c8', iic8), ('x3', iix3))
        for k, ii in engines:
            out = ii.pred_fcn_fn(X)
            Yii = ii.refine(out[:-1], out[-1], STEP, NSTEPS, early_stop=False)[0]
            res[k] = (out[-1], Yii)
        T = S.labels_from_map(host(res['f64'][1]), seed=800 + i)
        for k, ii in engines:
            cm[k] += ii.val_device(res[k][1], T).cm.cpu().numpy().reshape(11, 12)
        ref = res['f64'][1].argmax(1)
        agree_fcn.append(float((res['bf16'][0].argmax(1) == res['f64'][0].argmax(1)).double().mean()))
        for k in agree:
            agree[k].append(float((res[k][1].argmax(1) == ref).double().mean()))
    miou = {}
    for k in cm:
        c = cm[k][:, :11]
        tp = np.diag(c)
        with np.errstate(invalid='ignore', divide='ignore'):
            miou[k] = float(np.nanmean(tp / (c.sum(1) + c.sum(0) - tp)))
    print('damped set, 16 images x %d steps: mIoU f64 %.5f fp32 %.5f bf16x3 %.5f bf16 %.5f bf16+C8 %.5f; '
          'refined argmax agreement with f64: fp32 %.6f bf16x3 %.6f bf16 %.5f bf16+C8 %.5f (bf16 FCN-8 '
          'output %.5f)'
          % (NSTEPS, miou['f64'], miou['f32'], miou['x3'], miou['bf16'], miou['c8'],
             np.mean(agree['f32']), np.mean(agree['x3']), np.mean(agree['bf16']), np.mean(agree['c8']),
             np.mean(agree_fcn)))
    assert miou['f64'] == 1.0
    assert np.mean(agree['f32']) >= 0.9999 and abs(miou['f32'] - 1.0) <= 1e-3
    assert np.mean(agree['x3']) >= 0.9999 and abs(miou['x3'] - 1.0) <= 1e-3     # fp32-class
    assert np.mean(agree['bf16']) >= 0.99 and abs(miou['bf16'] - 1.0) <= 0.05
    assert np.mean(agree['c8']) >= 0.99 and abs(miou['c8'] - 1.0) <= 0.05     # C8 activations
