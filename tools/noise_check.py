import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')]
from test_mlp_gpu import make_ac
DEV='cuda:0'
ac = make_ac(60, 2); ac.set_seed(3)
N = 1 << 20
obs = torch.zeros(N, 60, device=DEV)
with torch.no_grad():
    mean = ac.step(obs, deterministic=True)[0]
    es = []
    for k in range(16):
        a = ac.step(obs)[0]
        es.append((a - mean).double())
    E = torch.stack(es)  # [16, N, 2], log_std = 0
print('n', E.numel(), 'mean %.5f var %.5f skew %.5f kurt %.5f' % (E.mean(), E.var(), (E**3).mean(), (E**4).mean()))
print('corr dims', float((E[..., 0] * E[..., 1]).mean()), 'corr consecutive calls', float((E[:-1] * E[1:]).mean()),
      'corr consecutive rows', float((E[:, :-1] * E[:, 1:]).mean()), 'corr sq dims', float(((E[...,0]**2-1) * (E[...,1]**2-1)).mean()),
      'corr sq calls', float(((E[:-1]**2-1) * (E[1:]**2-1)).mean()))
q = torch.quantile(E.flatten()[:8_000_000].float(), torch.tensor([0.001, 0.01, 0.1, 0.5, 0.9, 0.99, 0.999], device=DEV))
print('quantiles', q.cpu().numpy().round(4), 'normal: [-3.0902 -2.3263 -1.2816 0 1.2816 2.3263 3.0902]')
print('P(|e|>3) %.6f (normal 0.002700)  P(|e|>4) %.3e (normal 6.334e-05)' % (float((E.abs() > 3).double().mean()), float((E.abs() > 4).double().mean())))
