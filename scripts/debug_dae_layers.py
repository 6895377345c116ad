"""Layer-by-layer HIP vs oracle comparison of one full-size DAE forward (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dae as odae, fcn8 as ofcn8, nn as onn
from iterative_inference_segm_amd import synthetic as S
from iterative_inference_segm_amd.dae import StandardDAE
from iterative_inference_segm_amd.fcn8 import FCN8

fp, dp = S.make_fcn8_params(), S.make_dae_params()
to64 = lambda p: {k: tuple(np.asarray(a, np.float64) for a in v) for k, v in p.items()}
X = S.make_images(1, 224, 224, seed=1234)
fcn = FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'])
dae = StandardDAE(dp, 11)
dae.dce = False   # full maps, so every level can be compared
h, y = fcn(torch.from_numpy(X).cuda())
dae.trace = {}
score = dae.scores([h], y)
torch.cuda.synchronize()
h64, y64 = h.cpu().numpy().astype(np.float64), y.cpu().numpy().astype(np.float64)
r, net = odae.dae_forward(to64(dp), [h64], y64, return_net=True)
for k in ['pool1', 'pool2', 'pool3', 'pool4', 'pool5', 'pool6', 'fused_up6', 'fused_up5', 'fused_up4',
          'fused_up3', 'fused_up2', 'fused_up1']:
    g = dae.trace[k].cpu().numpy(); o = net[k]
    e = np.abs(g - o)
    print('%-10s shape %-20s max|ref| %.3f  max err %.3e  mean err %.3e  frac>1e-3 %.3e' %
          (k, g.shape, np.abs(o).max(), e.max(), e.mean(), (e > 1e-3).mean()))
# mask agreement per level: GPU masks from GPU tensors vs oracle masks from oracle tensors
for p in range(6, 0, -1):
    pre_g = dae.trace['pre%d' % p].cpu().numpy(); pool_g = dae.trace['pool%d' % p].cpu().numpy()
    pre_o = net['pre%d' % p]; pool_o = net['pool%d' % p]
    Hh, Ww = pool_g.shape[2] * 2, pool_g.shape[3] * 2
    mg = pre_g[:, :, :Hh, :Ww] == np.repeat(np.repeat(pool_g, 2, 2), 2, 3)
    mo = pre_o[:, :, :Hh, :Ww] == np.repeat(np.repeat(pool_o, 2, 2), 2, 3)
    diff = np.argwhere(mg != mo)
    print('level %d: mask density gpu %.4f oracle %.4f, disagreements %d' % (p, mg.mean(), mo.mean(), len(diff)))
    for (b, c, yy, xx) in diff[:6]:
        y0, x0 = yy // 2 * 2, xx // 2 * 2
        print('   at c=%d y=%d x=%d  gpu window %s  oracle window %s' % (
            c, yy, xx, pre_g[b, c, y0:y0+2, x0:x0+2].ravel().tolist(), pre_o[b, c, y0:y0+2, x0:x0+2].ravel().tolist()))
