"""Static guard on the generated gfx950 code of the kernels that issue asynchronous loads from inline asm.

A register that is the target of an in-flight asm load must not be copied: hipcc knows nothing about the latency of
an asm statement, so a v_mov it inserts to reconcile two definitions of such a register at a control-flow join (or
to split a live range) reads the register before the data has landed.  That happened in the Winograd kernel (about
one workgroup in 10^5 summed stale operands; tools/race_kernel.py found it) and is invisible in the source.  The
test compiles the kernels to assembly (no GPU needed) and fails on any move out of a register that an asm load of the
MFMA region writes (moves of ordinary values, e.g. saved lane indices, are fine), and on any scratch access there.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def _asm(src, tmp_path):
    out = os.path.join(tmp_path, os.path.basename(src) + '.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC,
                    '-S', '--cuda-device-only', '-o', out, src], check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(text, pattern):
    for m in re.finditer(r'^(_Z\S*' + pattern + r'\S*):\s*;', text, re.M):
        end = text.index('.Lfunc_end', m.end())
        yield m.group(1), [ln.strip() for ln in text[m.end():end].split('\n') if ln.strip() and not ln.strip().startswith(';')]


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not available')
@pytest.mark.parametrize('src,pattern', [('conv_wino.hip', 'conv_wino_kernel')])
def test_no_copies_of_async_load_targets(tmp_path, src, pattern):
    text = _asm(os.path.join(CSRC, src), str(tmp_path))
    seen = 0
    for name, lines in _kernels(text, pattern):
        mf = [i for i, ln in enumerate(lines) if ln.startswith('v_mfma')]
        assert mf, name
        seen += 1
        first_load = next(i for i, ln in enumerate(lines) if ln.startswith('buffer_load_dwordx2'))
        region = lines[min(first_load, mf[0]):mf[-1] + 1]          # from the first asynchronous load to the last MFMA
        # Registers written by the asynchronous asm loads.  Inside the loop nest (from the first MFMA on) any move out of
        # such a register is suspect, whatever the listing order (the load may sit later in the loop body).  Before
        # the first MFMA only loads that precede the move count: a register may hold an ordinary value there (a saved
        # lane index) and become a load target later.
        load_re = re.compile(r'(buffer_load_dwordx2|ds_read_b64)\s+v\[(\d+):(\d+)\]')
        targets = set()
        for ln in region:
            m = load_re.match(ln)
            if m:
                targets.update(range(int(m.group(2)), int(m.group(3)) + 1))
        assert targets, name
        first_mfma = mf[0] - min(first_load, mf[0])
        copies, seen_targets = [], set()
        for i, ln in enumerate(region):
            m = load_re.match(ln)
            if m:
                seen_targets.update(range(int(m.group(2)), int(m.group(3)) + 1))
                continue
            m = re.match(r'v_mov_b(32|64)(?:_e32|_e64)?\s+\S+,\s*v(?:\[(\d+):(\d+)\]|(\d+))', ln)
            if m:
                src = range(int(m.group(2)), int(m.group(3)) + 1) if m.group(2) else [int(m.group(4))]
                pool = targets if i >= first_mfma else seen_targets
                if any(r in pool for r in src):
                    copies.append(ln)
            elif ln.startswith('scratch_') or ln.startswith('v_pk_mov'):
                copies.append(ln)
        assert not copies, (name, copies[:8])
    assert seen >= 5
