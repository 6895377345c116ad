import torch, time
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = torch.empty(64, 128, 211, 211, device='cuda')
src = torch.rand(64, 128, 116, 116, device='cuda')
ms = t(lambda: x.fill_(1.0)); print('fill 1.46 GB contiguous: %.3f ms %.2f TB/s' % (ms, x.numel() * 4 / ms / 1e9))
w = x[:, :, 48:164, 48:164]
ms = t(lambda: w.copy_(src)); print('window copy 441 MB write + 441 MB read: %.3f ms  write %.2f TB/s' % (ms, src.numel() * 4 / ms / 1e9))
ms = t(lambda: w.fill_(2.0)); print('window fill 441 MB: %.3f ms %.2f TB/s' % (ms, src.numel() * 4 / ms / 1e9))
y = torch.empty_like(src)
ms = t(lambda: y.copy_(src)); print('dense copy 441 MB: %.3f ms (r+w %.2f TB/s)' % (ms, 2 * src.numel() * 4 / ms / 1e9))
ms = t(lambda: y.fill_(3.0)); print('dense fill 441 MB: %.3f ms %.2f TB/s' % (ms, src.numel() * 4 / ms / 1e9))
