"""Per-launch times of one contextmod-DAE step (configs[4] variant (i), batch 64, 224x224) by HIP events.
Usage: python scripts/ctx_step_profile.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iterative_inference_segm_amd import ops, synthetic as S
from iterative_inference_segm_amd.contextmod import ContextModDAE

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dae = ContextModDAE(S.make_contextmod_params(), 11)
X = torch.from_numpy(S.make_images(B, 224, 224, seed=7)).cuda()
y = torch.softmax(torch.randn(B, 11, 224, 224, device='cuda'), 1)
sess = dae.new_session([X], y)
for _ in range(3):
    dae.scores([X], y, session=sess)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    dae.scores([X], y, session=sess)
e1.record(); torch.cuda.synchronize()
print('scores() wall %.3f ms per step' % (e0.elapsed_time(e1) / 10))
torch.cuda._sleep(int(3e8))
ops.CONV_PROFILE = prof = []
for _ in range(3):
    dae.scores([X], y, session=sess)
torch.cuda.synchronize()
ops.CONV_PROFILE = None
n = len(prof) // 3
names = ['conv1'] + ['dil%d (d=%d)' % (i + 1, d) for i, d in enumerate([1, 2, 4, 8, 16, 1])] + ['1x1']
for i in range(n):
    ms = sorted(prof[i + r * n][2].elapsed_time(prof[i + r * n][3]) for r in range(3))[1]
    k, f = prof[i][0], prof[i][1]
    print('%-12s %-22s %.4f ms  %6.1f TF/s' % (names[i] if i < len(names) else '?', k, ms, f / ms / 1e9))

# the fused step (dilconv6 + dilconv7 + softmax + update as one launch, csrc/conv_small.hip ctx_tail_kernel)
st = ops.RefineState(B, 224, 224, 'cuda')
sess = dae.new_session([X], y)
for _ in range(3):
    dae.fused_step([X], y, st, 0.1, sess)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    dae.fused_step([X], y, st, 0.1, sess)
e1.record(); torch.cuda.synchronize()
print('fused_step() wall %.3f ms per step (scores + update in %d launches)' % (e0.elapsed_time(e1) / 10, 7))
torch.cuda._sleep(int(3e8))
ops.CONV_PROFILE = prof = []
for _ in range(3):
    dae.fused_step([X], y, st, 0.1, sess)
torch.cuda.synchronize()
ops.CONV_PROFILE = None
n = len(prof) // 3
names = ['conv1'] + ['dil%d (d=%d)' % (i + 1, d) for i, d in enumerate([1, 2, 4, 8, 16])] + ['dil6 + 1x1 + tail']
for i in range(n):
    ms = sorted(prof[i + r * n][2].elapsed_time(prof[i + r * n][3]) for r in range(3))[1]
    k, f = prof[i][0], prof[i][1]
    print('%-18s %-22s %.4f ms  %6.1f TF/s' % (names[i] if i < len(names) else '?', k, ms, f / ms / 1e9))
