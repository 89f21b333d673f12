"""Builds libomnisafe_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m omnisafe_amd.build            # rebuild if any source is newer than the library

The library is git-ignored but travels to the GPU box with the working tree.  hipcc cross-compiles
without a GPU, so this also runs in the (GPU-less) build container.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB_DIR = os.path.join(PKG, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libomnisafe_amd.so')
ARCH = 'gfx950'
# per-source compiler options.  Pass kernel: let MFMA results land in architectural VGPRs where they are
# consumed by VALU code (tanh, norms, Adam) instead of AGPRs + v_accvgpr_read moves (-0.8 % step time)
PER_FILE_FLAGS = {'ppo_pass_kernel.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1'],
                  'part_grad_kernel.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1'],
                  'p2p_pass_kernel.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1']}


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (need ROCm >= 7.0)')


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


ABI_INFO = 'abi_info.hip'  # defines osa_abi_digest(); recompiled whenever any other source changes


def source_digest() -> str:
    """sha256 over the header and every kernel source (content, not mtimes: the library travels to the GPU
    box with the working tree and file times do not survive that).  Compiled into the library
    (osa_abi_digest) and compared by _lib.load(): a stale or foreign libomnisafe_amd.so is rebuilt or
    refused instead of being called through prototypes it does not have."""
    import hashlib

    h = hashlib.sha256()
    files = [f for f in sources() if os.path.basename(f) != ABI_INFO]
    files += sorted(glob.glob(os.path.join(CSRC, '*.h')))
    files += sorted(glob.glob(os.path.join(os.path.dirname(PKG), 'include', '*.h')))
    for f in files:
        h.update(os.path.basename(f).encode() + b'\0')
        h.update(open(f, 'rb').read())
    for k in sorted(PER_FILE_FLAGS):
        h.update((k + ' '.join(PER_FILE_FLAGS[k])).encode())
    return h.hexdigest()[:32]


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    stamp = os.path.join(LIB_DIR, 'abi_digest.txt')
    if not os.path.exists(stamp) or open(stamp).read().strip() != source_digest():
        return True
    lib_m = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(
        os.path.join(os.path.dirname(PKG), 'include', '*.h'))
    return any(os.path.getmtime(d) > lib_m for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    digest = source_digest()
    stamp = os.path.join(LIB_DIR, 'abi_digest.txt')
    digest_changed = not os.path.exists(stamp) or open(stamp).read().strip() != digest
    cmds = []
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src) + '.o')
        is_info = os.path.basename(src) == ABI_INFO
        if force or not os.path.exists(obj) or (is_info and digest_changed) or os.path.getmtime(obj) < max(
                os.path.getmtime(src),
                *(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, '*.h'))),
                *(os.path.getmtime(h) for h in glob.glob(os.path.join(os.path.dirname(PKG), 'include', '*.h')))):
            cmds.append([_hipcc(), f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', obj,
                         '-Wall', '-Wno-unused-function'] + PER_FILE_FLAGS.get(os.path.basename(src), []) +
                        ([f'-DOSA_ABI_DIGEST="{digest}"'] if is_info else []) +
                        os.environ.get('OSA_EXTRA_CFLAGS', '').split())
        objs.append(obj)
    if cmds:
        # the sources are independent translation units (the two big ones instantiate ~90 persistent-pass and
        # ~30 per-step kernels: 2-3 minutes each): compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)

        with ThreadPoolExecutor(max_workers=min(len(cmds), max(1, (os.cpu_count() or 2)))) as ex:
            list(ex.map(run, cmds))
    cmd = [_hipcc(), f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB_PATH, *objs]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, 'w') as f:
        f.write(digest + '\n')
    return LIB_PATH


if __name__ == '__main__':
    build_library(force='--force' in sys.argv)
    print(LIB_PATH)
