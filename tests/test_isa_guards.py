"""Static guard on the generated gfx950 code of the kernels that issue asynchronous loads from inline asm.

A register that is the target of an in-flight asm load must not be copied: hipcc knows nothing about the latency of
an asm statement, so a v_mov it inserts to reconcile two definitions of such a register at a control-flow join (or
to split a live range) reads the register before the data has landed.  That happened in the Winograd kernel (about
one workgroup in 10^5 summed stale operands; tools/race_kernel.py found it) and is invisible in the source.  The
test compiles the kernels to assembly (no GPU needed) and fails on any move out of a register that an asm load of the
MFMA region writes (moves and spill stores of ordinary values, e.g. saved lane indices, are fine), and on any other
scratch access there (reloads, spills of accumulators).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

LOAD_RE = re.compile(r'(?:buffer_load_dwordx4|buffer_load_dwordx2|buffer_load_dword|global_load_dwordx4|global_load_dword|ds_read_b128|ds_read_b64)\s+v(?:\[(\d+):(\d+)\]|(\d+)(?!\d))')
SPILL_RE = re.compile(r'scratch_store_dword(?:x[234])?\s+off,\s*v(?:\[(\d+):(\d+)\]|(\d+)(?!\d))')
DEST_RE = re.compile(r'v_\w+\s+v(?:\[(\d+):(\d+)\]|(\d+)(?!\d))')
MOVE_RE = re.compile(r'v_mov_b(?:32|64)(?:_e32|_e64)?\s+\S+,\s*v(?:\[(\d+):(\d+)\]|(\d+)(?!\d))')


def _regs(m):
    return range(int(m.group(1)), int(m.group(2)) + 1) if m.group(1) else [int(m.group(3))]


EXTRA = {'wgrad_wino.hip': ['-fno-slp-vectorize']}          # per-file flags of csrc/build.sh: the guard must see the shipped code


def _asm(src, tmp_path):
    out = os.path.join(tmp_path, os.path.basename(src) + '.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC] +
                   EXTRA.get(os.path.basename(src), []) + ['-S', '--cuda-device-only', '-o', out, src], check=True,
                   stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(text, pattern):
    for m in re.finditer(r'^(_Z\S*' + pattern + r'\S*):\s*;', text, re.M):
        end = text.index('.Lfunc_end', m.end())
        yield m.group(1), [ln.strip() for ln in text[m.end():end].split('\n') if ln.strip() and not ln.strip().startswith(';')]


def suspicious_copies(lines):
    """Moves out of registers that asynchronous asm loads write, between the first such load and the last MFMA.
    Inside the loop nest (from the first MFMA on) any such move is suspect, whatever the listing order (the load may
    sit later in the loop body).  Before the first MFMA only loads that precede the move count: a register may hold
    an ordinary value there (a saved lane index) and become a load target later.  (v_accvgpr_write out of such
    registers is the accumulator shuffle at a loop exit: registers whose loads completed long ago serve as
    temporaries there.)  A move that follows a full wait (vmcnt(0) / lgkmcnt(0)) in straight-line code is harmless."""
    mf = [i for i, ln in enumerate(lines) if ln.startswith('v_mfma')]
    loads = [i for i, ln in enumerate(lines) if LOAD_RE.match(ln)]
    assert mf and loads
    start = min(loads[0], mf[0])
    region = lines[start:mf[-1] + 1]
    targets = set()
    for ln in region:
        m = LOAD_RE.match(ln)
        if m:
            targets.update(_regs(m))
    first_mfma = mf[0] - start
    # hot[i]: line i lies inside a run of MFMAs (gaps of at most 150 lines: one chunk of the pipeline)
    hot = [False] * len(region)
    rm = [i - start for i in mf]
    for a, b in zip(rm, rm[1:]):
        if b - a <= 150:
            for j in range(a, b + 1):
                hot[j] = True
    copies, seen_vm, seen_lds, landed = [], set(), set(), set()
    for i, ln in enumerate(region):
        m = LOAD_RE.match(ln)
        if m:
            (seen_lds if ln.startswith('ds_') else seen_vm).update(_regs(m))
            landed.difference_update(_regs(m))
            continue
        if ln.startswith('s_waitcnt'):
            if 'vmcnt(0)' in ln:
                landed.update(seen_vm)                      # every vector-memory load issued so far has landed
            if 'lgkmcnt(0)' in ln:
                landed.update(seen_lds)                     # every LDS read issued so far has landed
            continue
        if ln.startswith('.LBB'):
            # a loop header (target of a later, backward branch): the state along the back edge is unknown.  Forward
            # joins (skipped blocks without loads) keep the state.
            label = ln.split(':')[0]
            back = [j for j in range(i + 1, len(region)) if region[j].startswith(('s_cbranch', 's_branch')) and region[j].split()[-1] == label]
            # (a loop whose body issues no asynchronous load - an epilogue's loop over destinations - cannot bring one in flight)
            if back and any(LOAD_RE.match(x) for x in region[i:back[-1] + 1]):
                landed.clear()
            continue
        m = MOVE_RE.match(ln)
        if m:
            pool = targets if i >= first_mfma else (seen_vm | seen_lds)
            if any(r in pool and r not in landed for r in _regs(m)):
                copies.append(ln)
        # a vector instruction that overwrites a former load target makes it an ordinary register again: its live range as a
        # load target ended with its last use, behind the wait (e.g. the zero that initialises the accumulators of the next
        # block lives in a register the loads of the chunk use)
        d = DEST_RE.match(ln)
        if d and not ln.startswith('v_mfma'):
            landed.update(_regs(d))
            if i < first_mfma:
                seen_vm.difference_update(_regs(d))
                seen_lds.difference_update(_regs(d))
        if m:
            continue
        if SPILL_RE.match(ln):
            # a spill store is a copy out of its source registers: harmless for an ordinary value (it can only make a
            # counted vmcnt wait stricter, never weaker), a stale-data bug for the target of a load still in flight
            pool = targets if i >= first_mfma else (seen_vm | seen_lds)
            if any(r in pool and r not in landed for r in _regs(SPILL_RE.match(ln))):
                copies.append(ln)
        elif (ln.startswith('scratch_') and hot[i]) or ln.startswith('v_pk_mov'):
            # scratch traffic between the MFMAs of a chunk (accumulator or operand spills); the reloads of loop invariants in
            # the epilogue of a persistent kernel, which also lies between the first and the last MFMA, are ordinary code
            copies.append(ln)
    return copies


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not available')
@pytest.mark.parametrize('src,pattern,nmin', [('conv_wino.hip', 'conv_winoh_kernel', 3), ('wgrad_wino.hip', 'wino_wgrad_kernel', 1),
                                              ('conv_wino44.hip', 'wino44_kernel', 2)])
def test_no_copies_of_async_load_targets(tmp_path, src, pattern, nmin):
    text = _asm(os.path.join(CSRC, src), str(tmp_path))
    seen = 0
    for name, lines in _kernels(text, pattern):
        seen += 1
        copies = suspicious_copies(lines)
        assert not copies, (name, copies[:8])
    assert seen >= nmin


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not available')
def test_direct_implicit_gemm_never_copies_a_load_target(tmp_path):
    """conv_igemm_kernel<..., DIRECT> (csrc/conv_igemm.hip) loads its MFMA operands straight into registers one K step ahead and waits for them
    with counted s_waitcnt vmcnt(N).  Until round 4 the K loop had two tails that hipcc folded into one block behind v_mov copies of the operand
    registers - copies of registers whose loads could still be in flight, in front of the wait: under cross-stream memory load the last K step of
    a tile multiplied stale operands (on the GPU: tests/test_parity_r04.py::test_cold_training_steps_repeat_bit_for_bit_beside_
    the_helper_stream).  Static check of every DIRECT instantiation: between the first operand load and the last MFMA no move reads a register that
    a buffer_load_dwordx4 of the kernel writes."""
    text = _asm(os.path.join(CSRC, 'conv_igemm.hip'), str(tmp_path))
    reg_re = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')
    seen = 0
    for name, lines in _kernels(text, 'conv_igemm_kernel'):
        if 'Lb1EEE' not in name:                       # the LDS-staged variant: its loads are consumed behind a full wait and a barrier
            continue
        seen += 1
        targets = set()
        for ln in lines:
            m = re.match(r'buffer_load_dwordx4\s+v\[(\d+):(\d+)\]', ln)
            if m:
                targets.update(range(int(m.group(1)), int(m.group(2)) + 1))
        first = min(i for i, ln in enumerate(lines) if ln.startswith('buffer_load_dwordx4'))
        last = max(i for i, ln in enumerate(lines) if ln.startswith('v_mfma'))
        bad = []
        for ln in lines[first:last + 1]:
            if ln.startswith(('v_mov_b32', 'v_mov_b64', 'v_pk_mov_b32', 'v_accvgpr_write_b32')) or ln.startswith('scratch_store'):
                ops_ = ln.split(None, 1)[1].split(',', 1)
                src = set()
                for a, b, c in reg_re.findall(ops_[1] if len(ops_) > 1 else ''):
                    src.update(range(int(a), int(b) + 1) if a else [int(c)])
                if src & targets:
                    bad.append(ln)
        assert not bad, (name, len(bad), bad[:4])
    assert seen >= 9


TR_RE = re.compile(r'ds_read_b64_tr_b16\s+v\[(\d+):(\d+)\]')


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not available')
def test_lds_dma_weight_gradient_kernel_keeps_its_pipeline(tmp_path):
    """wgrad_bf16_dma_kernel (csrc/wgrad_bf16.hip, round 3) hides its row loads only while (a) the rows arrive by LDS-DMA, (b) the only
    vector-memory waits are the hand-counted ones - hipcc drains every outstanding DMA (vmcnt(0)) in front of an LDS read it knows of,
    which is why the fragment reads are inline asm - and (c) no register that an asm fragment read writes is copied or spilled between
    the read and the MFMAs (the compiler does not know those reads are asynchronous).  Static check of the shipped code, no GPU."""
    text = _asm(os.path.join(CSRC, 'wgrad_bf16.hip'), str(tmp_path))
    kernels = list(_kernels(text, 'wgrad_bf16_dma_kernel'))
    assert len(kernels) == 1
    name, lines = kernels[0]
    assert sum(ln.startswith('global_load_lds_dwordx4') for ln in lines) >= 20
    assert not any(ln.startswith(('global_load_dword', 'buffer_load_dword')) and 'lds' not in ln for ln in lines), 'a register-staged load crept in'
    waits = [ln for ln in lines if ln.startswith('s_waitcnt') and 'vmcnt' in ln]
    counted = sorted(int(re.search(r'vmcnt\((\d+)\)', ln).group(1)) for ln in waits)
    # the drain at the top of a run of steps and the one before the workgroup retires, plus the two counted waits of the step loop
    assert counted == [0, 0, 8, 12], waits
    mf = [i for i, ln in enumerate(lines) if ln.startswith('v_mfma')]
    assert len(mf) == 40                                          # 36 per step + 4 for the bias gradient
    # (c) as a simulation of the LDS queue: a fragment read is in flight from its issue until an s_waitcnt lgkmcnt(N) leaves at most N
    # younger LDS operations outstanding (LDS returns in order); until then NO other instruction may name one of its target registers.
    # (Since round 4 the kernel builds the shifted fragments in registers - v_perm / v_mov on read targets - which is fine behind the wait.)
    first_tr = min(i for i, ln in enumerate(lines) if TR_RE.match(ln))
    reg_re = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')

    def named(ln):
        out = set()
        for a, b, c in reg_re.findall(ln):
            out.update(range(int(a), int(b) + 1) if a else [int(c)])
        return out

    flight, bad, nread = [], [], 0
    for ln in lines[first_tr:mf[-1] + 1]:
        m = TR_RE.match(ln)
        if m:
            flight.append(set(range(int(m.group(1)), int(m.group(2)) + 1)))
            nread += 1
            continue
        w = re.search(r'lgkmcnt\((\d+)\)', ln) if ln.startswith('s_waitcnt') else None
        if w:
            flight = flight[len(flight) - int(w.group(1)):] if int(w.group(1)) < len(flight) else flight
            if int(w.group(1)) == 0:
                flight = []
            continue
        if ln.startswith('ds_') or ln.startswith('s_load'):
            bad.append('another lgkmcnt operation inside the counted window: ' + ln)
        pending = set().union(*flight) if flight else set()
        if pending & named(ln):
            bad.append(ln)
        if ln.startswith('scratch_') or SPILL_RE.match(ln):
            bad.append(ln)
    assert nread >= 30 and not bad, bad[:8]


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not available')
def test_half_domain_weight_gradient_kernel_fits_two_workgroups_per_cu(tmp_path):
    """wino_wgrad_half_kernel (csrc/wgrad_wino.hip, round 3) is faster than the one-workgroup kernel only because TWO of its workgroups fit
    a CU - one wave of each per SIMD, one's loads and transform adds under the other's MFMAs: at most 256 registers per wave, at most
    80 KB of LDS per workgroup, no scratch, and all 128 accumulators of a half (64 MFMAs per group pass pair) in the loop."""
    text = _asm(os.path.join(CSRC, 'wgrad_wino.hip'), str(tmp_path))
    m = re.search(r'\.amdhsa_kernel (\S*wino_wgrad_half_kernel\S*)(.*?)\.end_amdhsa_kernel', text, re.S)
    assert m, 'kernel not found'
    meta = m.group(2)
    val = lambda k: int(re.search(k + r'\s+(\d+)', meta).group(1))                              # noqa: E731
    assert val('amdhsa_next_free_vgpr') <= 256, 'more than 256 registers: one workgroup per CU again'
    assert val('amdhsa_group_segment_fixed_size') <= 80 * 1024
    assert val('amdhsa_private_segment_fixed_size') == 0, 'scratch in the weight-gradient kernel'
    kernels = list(_kernels(text, 'wino_wgrad_half_kernel'))
    assert len(kernels) == 1
    lines = kernels[0][1]
    assert not any(ln.startswith('scratch_') for ln in lines)
    assert sum(ln.startswith('v_mfma_f32_32x32x2') for ln in lines) >= 2 * 4 * 16 * 2            # both halves x 4 groups x 16 MFMAs x (steady state + peeled last quad)


_REG_RE = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')

def _named(s):
    regs = set()
    for a, b, c in _REG_RE.findall(s):
        regs.update(range(int(a), int(b) + 1) if a else [int(c)])
    return regs

def _inflight_analysis(body, partial_wait=None):
    """Forward data flow over the kernel's basic blocks: the set of registers with an asm global_load in flight at every instruction (union at
    joins).  Every instruction outside the asm statements that names such a register is reported.  ``partial_wait``: (text of a counted asm wait,
    regex of the loads it covers) - e.g. the wait for a bias word that leaves younger requests in flight."""
    blocks, cur, label_of = [], [], {}
    in_asm = False
    for raw in body:
        ln = raw.strip()
        if ln.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if ln.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not ln or ln.startswith(';'):
            continue
        if re.match(r'^\.LBB\d+_\d+:', ln):
            if cur:
                blocks.append(cur)
            cur = []
            label_of[ln.split(':')[0]] = len(blocks)
            continue
        if ln.startswith('.'):
            continue
        cur.append((ln.split(';')[0].strip(), in_asm))
        if ln.startswith(('s_branch', 's_cbranch', 's_endpgm')):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for i, b in enumerate(blocks):
        last = b[-1][0] if b else ''
        out = []
        if last.startswith(('s_branch', 's_cbranch')):
            out.append(label_of[last.split()[-1]])
        if not last.startswith(('s_branch', 's_endpgm')) and i + 1 < len(blocks):
            out.append(i + 1)
        succ.append(out)

    narrow = set()          # targets of the loads a partial wait covers

    def transfer(b, state, report):
        state = set(state)
        for ins, asm in b:
            if asm:
                ld = re.match(r'global_load_dwordx4\s+v\[(\d+):(\d+)\]', ins)
                if ld:
                    state.update(range(int(ld.group(1)), int(ld.group(2)) + 1))
                    report['loads'] += 1
                elif ins.startswith('s_waitcnt') and 'vmcnt(0)' in ins:
                    state.clear()
                    report['waits'] += 1
                continue
            if ins.startswith('s_waitcnt') and 'vmcnt(0)' in ins:
                state.clear()
                continue
            hit = state & _named(ins.split(None, 1)[1] if ' ' in ins else '')
            if hit:
                report['bad'].append((ins, sorted(hit)[:4]))
        return state

    ins_state = [None] * len(blocks)
    ins_state[0] = frozenset()
    work = [0]
    while work:
        i = work.pop()
        out = frozenset(transfer(blocks[i], ins_state[i], {'loads': 0, 'waits': 0, 'bad': []}))
        for j in succ[i]:
            new = out if ins_state[j] is None else ins_state[j] | out
            if new != ins_state[j]:
                ins_state[j] = new
                work.append(j)
    report = {'loads': 0, 'waits': 0, 'bad': []}
    left = set()
    for i, b in enumerate(blocks):
        if ins_state[i] is not None:
            end = transfer(b, ins_state[i], report)
            if b and b[-1][0].startswith('s_endpgm'):
                left |= end
    return report, left


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not available')
def test_bf16_convolution_epilogue_prefetches_are_not_touched_before_their_wait(tmp_path):
    """conv_bf16d_kernel (csrc/conv_bf16.hip) requests the previous cell state (LSTM epilogue) and, since round 5, the eleven operands of a round's
    gate items (LSTM_BWD epilogue) with inline-asm global_load_dwordx4 a whole round ahead and waits for them with an asm `s_waitcnt vmcnt(0)` that
    names them.  hipcc does not know those registers are in flight: between an asm load and the asm wait behind it NO instruction may name one of
    its target registers - no v_mov that splits a live range, no spill store, no reuse as a temporary (the LSTM_BWD instantiations do spill a few
    registers at times: they must be other ones).  The check follows the control flow (a forward data-flow analysis over the basic blocks, union at joins):
    the run-time choice between the prefetching and the generic form of the LSTM_BWD rounds is made once, around all four rounds, so that on the path
    that waits for a prefetch no other path's code - where those registers are dead and free for reuse - lies between the load and the wait."""
    src = os.path.join(CSRC, 'conv_bf16.hip')
    out = os.path.join(str(tmp_path), 'conv_bf16.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-S', '--cuda-device-only',
                    '-o', out, src], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    checked = 0
    for m in re.finditer(r'^(_Z\S*conv_bf16d_kernelILi[23]E\S*):\s*;', text, re.M):        # the LSTM and LSTM_BWD instantiations
        body = text[m.end():text.index('.Lfunc_end', m.end())].split('\n')
        report, left = _inflight_analysis(body)
        if report['loads'] == 0:                              # (the 64-column LSTM_BWD instantiations: no prefetching form)
            continue
        assert report['loads'] >= 8 and report['waits'] >= 1, (m.group(1), report['loads'], report['waits'])
        assert not left, (m.group(1), 'asm loads without a wait behind them', sorted(left)[:8])
        assert not report['bad'], (m.group(1), len(report['bad']), report['bad'][:6])
        checked += 1
    assert checked >= 4


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not available')
def test_wino44_kernels_have_no_register_target_in_flight_outside_their_rings(tmp_path):
    """wino44_kernel (csrc/conv_wino44.hip) at first requested the bias word and the previous cell state into registers with inline-asm global
    loads, as conv_wino.hip does.  At 251-256 registers hipcc (a) spilled one of the targets right behind its request and reused the register for
    the next address - a memory fault on the GPU - and, with fewer requests, (b) copied the targets to other registers in FRONT of the asm wait that
    names them as read-write operands (a tied operand is satisfied by a copy).  Those requests now go to LDS by LDS-DMA: the only registers with a
    load in flight are the weight ring and the LDS operand ring of the main loop (checked by test_no_copies_of_async_load_targets).  Here: no
    global load with a register target in the kernel's asm statements, three LDS-DMA request sites for the epilogue per instantiation, no spill."""
    text = _asm(os.path.join(CSRC, 'conv_wino44.hip'), str(tmp_path))
    seen = 0
    for m in re.finditer(r'^(_Z\S*wino44_kernelILi([01])E\S*):\s*;', text, re.M):          # the LSTM (0) and the plain-store (1) instantiation
        body = text[m.end():text.index('.Lfunc_end', m.end())].split('\n')
        report, left = _inflight_analysis(body)
        assert report['loads'] == 0 and not left, (m.group(1), report)
        dma = [ln for ln in body if re.search(r'buffer_load_dword(x4)?\s+v\d+, s\[\d+:\d+\], \S+ offen lds', ln)]
        nepi = 3 if m.group(2) == '0' else 1
        assert len([ln for ln in dma if 'dwordx4' not in ln]) >= 1 and len(dma) >= 9 * 2 + 9 + nepi, (m.group(1), len(dma))
        assert not any('scratch_' in ln for ln in body), (m.group(1), 'the kernel spills')
        seen += 1
    assert seen == 2
