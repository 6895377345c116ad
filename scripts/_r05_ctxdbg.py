import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from iterative_inference_segm_amd import ops, synthetic as S
from iterative_inference_segm_amd.contextmod import ContextModDAE
H, W, B = int(sys.argv[1]), int(sys.argv[2]), 2
dae = ContextModDAE(S.make_contextmod_params(), 11)
X = torch.from_numpy(S.make_images(B, H, W, seed=7)).cuda()
g = torch.Generator(device='cuda').manual_seed(1)
y0 = torch.softmax(torch.randn(B, 11, H, W, device='cuda', generator=g), 1)
# unfused
ya = y0.clone(); sa = ops.RefineState(B, H, W, 'cuda')
sess = dae.new_session([X], ya)
score = dae.scores([X], ya, session=sess)
ops.refine_update(score, ya, sa, 0.5, off=(0, 0))
cat_a = sess['cat'].clone(); cat_a[:, 3:, 1:-1, 1:-1].copy_(ya)
# fused
yb = y0.clone(); sb = ops.RefineState(B, H, W, 'cuda')
sess = dae.new_session([X], yb)
nblk = dae.fused_step([X], yb, sb, 0.5, sess)
torch.cuda.synchronize()
d = (ya - yb).abs()
print('nblk', nblk, 'max diff y', float(d.max()))
idx = torch.nonzero(d > 0)
print('differing elements', idx.shape[0], idx[:10].tolist(), 'cols', sorted(set(idx[:, 3].tolist()))[:20], 'rows', sorted(set(idx[:, 2].tolist()))[:20])
dc = (cat_a - sess['cat']).abs()
idc = torch.nonzero(dc > 0)
print('cat diffs', idc.shape[0], idc[:10].tolist())
ops.refine_finalize(sa, 1e-3); ops.refine_finalize(sb, 1e-3, nblk=nblk)
print('norms', sa.last_norm.tolist(), sb.last_norm.tolist())
