import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
d = (a - b).abs()
bad = (d > 2e-3).nonzero()[:, 0]
print('bad points', bad.numel(), 'of', a.numel(), 'first', bad[:20].tolist())
