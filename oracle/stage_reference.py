"""TEST INFRASTRUCTURE ONLY -- stages the *unmodified* reference package so that it can travel to the GPU
box (which has no /root/reference) as a checker and as the CPU baseline of bench.py.

    python oracle/stage_reference.py        # /root/reference/omnisafe -> oracle/_ref/omnisafe_ref.zip

The archive is an OUTPUT of the build container (like a compiled `.so`): `oracle/_ref/` is git-ignored, so
no reference source enters the repository's history, but it is not gpurun-ignored, so it ships with the
working tree.  Nothing under `omnisafe_amd/` reads it.  Users: `oracle/ref_harness.py` (which unpacks it
into a scratch directory when /root/reference is absent), and through it `tests/test_reference_facade_gpu.py`
(the reference's own `omnisafe.Agent` driving the plugin) and `bench.py`'s `cpu_baseline` leg.

`__graft_entry__.build()` calls `stage()` whenever /root/reference is present.
"""
from __future__ import annotations

import hashlib
import os
import sys
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = os.environ.get('OMNISAFE_REFERENCE_ROOT', '/root/reference')
OUT_DIR = os.path.join(HERE, '_ref')
ARCHIVE = os.path.join(OUT_DIR, 'omnisafe_ref.zip')
_KEEP = ('.py', '.yaml', '.yml', '.json', '.txt')


def _files():
    pkg = os.path.join(REF_SRC, 'omnisafe')
    for root, dirs, files in os.walk(pkg):
        dirs[:] = sorted(d for d in dirs if d != '__pycache__')
        for f in sorted(files):
            if f.endswith(_KEEP):
                full = os.path.join(root, f)
                yield full, os.path.relpath(full, REF_SRC)
    # the reference's own test double of a user-registered single env (tests/simple_env.py: 'Test-v0'), driven
    # through the installed plugin by tests/test_host_env_gpu.py
    extra = os.path.join(REF_SRC, 'tests', 'simple_env.py')
    if os.path.exists(extra):
        yield extra, os.path.relpath(extra, REF_SRC)


def stage(verbose: bool = True) -> str | None:
    """Write the archive (deterministic: sorted members, fixed timestamps).  Returns its path, or None
    when the reference is not present (GPU box: the prebuilt archive is used as it is)."""
    if not os.path.isdir(os.path.join(REF_SRC, 'omnisafe')):
        return ARCHIVE if os.path.exists(ARCHIVE) else None
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = ARCHIVE + '.tmp'
    digest = hashlib.sha256()
    n = 0
    with zipfile.ZipFile(tmp, 'w', zipfile.ZIP_DEFLATED) as z:
        for full, rel in _files():
            data = open(full, 'rb').read()
            digest.update(rel.encode() + b'\0' + data)
            info = zipfile.ZipInfo(rel, date_time=(2020, 1, 1, 0, 0, 0))
            info.compress_type = zipfile.ZIP_DEFLATED
            info.external_attr = 0o644 << 16
            z.writestr(info, data)
            n += 1
        z.writestr(zipfile.ZipInfo('STAGED_FROM', date_time=(2020, 1, 1, 0, 0, 0)),
                   f'{REF_SRC}\nfiles {n}\nsha256 {digest.hexdigest()}\n')
    os.replace(tmp, ARCHIVE)
    if verbose:
        print(f'[stage_reference] {n} files -> {ARCHIVE} ({os.path.getsize(ARCHIVE)} bytes, '
              f'sha256 {digest.hexdigest()[:16]})')
    return ARCHIVE


if __name__ == '__main__':
    sys.exit(0 if stage() else 1)
