"""GPU tool (round 6): where a chunk of the Fisher-vector product kernel (csrc/fvp_kernel.hip) spends its cycles.
Needs the clock build of the library:

    tools/build_variant_lib.sh fvpclocks fvp_kernel.hip -DOFV_CLOCKS
    OSA_LIB_PATH=omnisafe_amd/lib/libomnisafe_amd_fvpclocks.so python tools/fvp_phase_clocks.py [--out profiles/r6_fvp_phase_clocks.txt]

Thread 0 of workgroup 0 accumulates the shader cycles (s_memtime) between marks over its chunks; MFMA issue cycles per
phase are counted from the source (v_mfma_f32_16x16x4_f32 = 32 cycles each, one wave per SIMD)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnisafe_amd import _lib  # noqa: E402
from omnisafe_amd.trust_region import TrustRegionSolver  # noqa: E402

PH = ['loop / previous tail', 'barrier: tiles free', 'forward (3 layers, tanh)', 'tangent pass (J v) + dL/d(out)',
      'next rows requested', 'backward through W3, W2', 'S -> F tiles in LDS (80 stores) + barrier',
      'dW2 (A operands + 64 MFMAs)', 'dW3', 'bias sums', 'barrier + x tile + barrier', 'dW1']
# MFMAs per wave and chunk of each phase at 60 / 2, hidden 64 (KB 4, HT 4, OT 1, NSB 4)
MF = [0, 0, 64 + 64 + 16, 64 + 128 + 32, 0, 16 + 64, 0, 64, 16, 0, 0, 64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    ap.add_argument('--rows', type=int, default=65536)
    args = ap.parse_args()
    lib = _lib.load(require_gpu=True)
    fn = lib.osa_debug_fvp_clocks  # (only the -DOFV_CLOCKS build exports it)
    fn.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    torch.manual_seed(1)
    import types

    import numpy as np

    from omnisafe_amd.models import ConstraintActorCritic
    from omnisafe_amd.spaces import Box

    ns = types.SimpleNamespace
    mc = ns(actor=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4), critic=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
            weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)
    ac = ConstraintActorCritic(Box(-np.inf, np.inf, (60,)), Box(-1, 1, (2,)), mc, 4, device='cuda:0')
    obs = torch.randn(args.rows, 60, device='cuda:0')
    v = ac.actor.pad(torch.randn(ac.actor.num_params))
    s = TrustRegionSolver(ac, cg_iters=10, cg_damping=0.1)
    s.begin(obs)
    for _ in range(3):
        s.fvp(v)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 16)()
    fn(buf, 1)
    reps = 20
    for _ in range(reps):
        s.fvp(v)
    torch.cuda.synchronize()
    fn(buf, 0)
    chunks = reps * (((args.rows + 63) // 64 + 255) // 256)  # (workgroup 0 of 256 takes every 256th chunk)
    tot = sum(buf[:12])
    lines = [f'osa_fvp_kernel<1, 4>, {args.rows} rows, workgroup 0: shader cycles per 64-row chunk (mean of {chunks} chunks)',
             f'{"phase":48s} {"cycles":>8s} {"share":>7s} {"MFMA issue":>11s} {"not MFMA":>9s}']
    for k in range(12):
        c = buf[k] / chunks
        lines.append(f'{PH[k]:48s} {c:8.0f} {100 * buf[k] / tot:6.1f}% {32 * MF[k]:11d} {c - 32 * MF[k]:9.0f}')
    lines.append(f'{"chunk":48s} {tot / chunks:8.0f} {"":7s} {32 * sum(MF):11d} {tot / chunks - 32 * sum(MF):9.0f}')
    txt = '\n'.join(lines)
    print(txt)
    if args.out:
        open(args.out, 'w').write(txt + '\n')


if __name__ == '__main__':
    main()
