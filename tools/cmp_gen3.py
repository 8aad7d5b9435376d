"""compare two dumps written by cmp_gen.py"""
import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
d = (a - b).abs()
print(f'max |diff| {float(d.max()):.3e}  mean {float(d.mean()):.3e}  nan {int(torch.isnan(a).sum())}/{int(torch.isnan(b).sum())}  n {a.numel()}')
