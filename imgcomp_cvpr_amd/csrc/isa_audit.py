#!/usr/bin/env python3
"""Build gate for the kernels whose MFMAs are inline asm (conv3x3_wino4.hip, conv3x3_wino_tn.hip, ...): csrc/Makefile runs this over
the gfx950 assembly of every such file and the build FAILS on a finding.

Why (round 5, root cause of round 4's "packed fp32" failure; experiment log profiles/r05_w4_rootcause.md): an `asm volatile("v_mfma...")`
statement is one opaque instruction to hipcc.  It keeps the statement's register operands in program order, but it does not know the
statement is an MFMA, so around it GCNHazardRecognizer inserts NONE of the matrix-pipe wait states the hardware leaves to software:
  (A) XDL write -> any other access: a non-MFMA instruction (VALU, v_accvgpr_*, LDS / buffer / scratch access) that reads or writes a
      register of an MFMA's destination tuple needs >= passes + 3 wait states after it (hipcc puts `s_nop 9` behind a builtin
      v_mfma_f32_16x16x4_f32; this gate asks for 12).  Too early, it sees the accumulator as it was BEFORE the MFMA (or partly
      written: the last columns of each 16-lane row land last).
  (B) VALU write -> MFMA source: a VALU instruction that writes a register the MFMA reads as SrcA / SrcB / SrcC needs 2 wait states
      before it (hipcc: `s_nop 1`).
The hand-written code keeps both by construction (operands come from LDS / buffer loads, accumulators are read only after an
`s_nop 15 x 2` pad), but the REGISTER ALLOCATOR may add instructions of its own between two asm statements: with the SLP vectoriser's
64-bit temporaries in the input transform it spilled / copied the four VGPR-resident accumulators (`scratch_store_dwordx4`,
`v_mov_b64`) directly behind and in front of the MFMAs that own them -- wrong values whose place changed from launch to launch.
Patching wait states for (A) and (B) into that assembly made it bit-exact again (0 of 150 launches against 150 of 150), (A) or (B)
alone did not.  So: whatever flags or compiler produced the object, this audit proves the shipped ISA has neither pattern.
  (S) Spills: the code-object metadata of every audited kernel (`.vgpr_spill_count`, `.private_segment_fixed_size`) must show NO scratch
      -- a spill is where the allocator's own instructions come from -- unless the kernel's name matches --allow-scratch (the
      h12 phase-shuffle instantiations spill 1-4 registers in their epilogue, far from any MFMA; rules (A) / (B) still cover them).
Control flow: the scan is linear, a backward branch is followed across its back edge and a FORWARD branch into its target (round 6:
the 8-wave form skips its transform slices behind scalar branches -- the instructions skipped must not be counted as wait states),
each with a copy of the state, for as many wait states as a hazard can span.

usage: isa_audit.py file.s [--max-scratch BYTES] [--allow-scratch NAME-REGEX]   (exit 1 on a finding; BYTES per kernel, default 0)
Wait states are counted the way LLVM does: every instruction 1, `s_nop N` N + 1."""
import re, sys

XDL_TO_ANY = 12      # (A)
VALU_TO_SRC = 2      # (B)
_REG = re.compile(r'(?<![\w.])([va])(?:\[(\d+):(\d+)\]|(\d+)\b)')


def regs_of(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(4) is not None: out.add((m.group(1), int(m.group(4))))
        else: out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernels(path):
    name, body = None, None
    for ln in open(path):
        m = re.match(r'^([A-Za-z_][\w$.]*):', ln)
        if m and not ln.startswith('.L'):
            name, body = m.group(1), []
            continue
        if ln.startswith('.Lfunc_end') and body is not None:
            yield name, body
            name, body = None, None
            continue
        if body is None: continue
        s = ln.split(';')[0].strip()
        if not s or (s.startswith('.') and not s.endswith(':')): continue
        body.append(s)              # instructions and labels (`.LBBn_m:`): the back edges of loops are followed below


def _step(s, age, recent, findings, tag=''):
    """one instruction through both rules; age: register -> wait states since an MFMA wrote it; recent: recent VALU writes.  -> is MFMA"""
    op, _, rest = s.partition(' ')
    args = [a.strip() for a in rest.split(',')] if rest else []
    states = int(args[0], 0) + 1 if op == 's_nop' else 1
    mfma = op.startswith('v_mfma') or op.startswith('v_smfmac')
    if mfma:
        dst, src = regs_of(args[0]), set().union(*(regs_of(a) for a in args[1:4]))
        for ws, wr, text in recent:
            if ws < VALU_TO_SRC and (wr & src):
                findings.append('(B){} VALU write {} wait state(s) before an MFMA that reads it:  {}  ->  {}'.format(tag, ws, text, s))
        # an MFMA reading another MFMA's destination as SrcC (the accumulate chain) is the hardware's business; as SrcA / SrcB it is not
        early = [r for r in set().union(*(regs_of(a) for a in args[1:3])) if r in age and age[r] < XDL_TO_ANY]
        if early: findings.append('(A){} MFMA result used as SrcA / SrcB after {} wait states:  {}'.format(tag, min(age[r] for r in early), s))
        for r in dst: age[r] = 0
    else:
        touched = regs_of(rest) if not op.startswith('s_') else set()
        hit = [r for r in touched if r in age and age[r] < XDL_TO_ANY]
        if hit:
            findings.append('(A){} {} wait state(s) after an MFMA wrote {}{}:  {}'.format(tag, min(age[r] for r in hit), hit[0][0], hit[0][1], s))
    new_recent = None
    if op.startswith('v_') and not op.startswith(('v_mfma', 'v_smfmac', 'v_cmp', 'v_nop')) and args:
        new_recent = [0, regs_of(args[0]), s]
    for r in list(age):
        if not (mfma and age[r] == 0): age[r] += states
        if age[r] > 4 * XDL_TO_ANY: del age[r]
    for e in recent: e[0] += states
    recent[:] = [e for e in recent if e[0] < VALU_TO_SRC + 1]
    if new_recent: recent.append(new_recent)
    return mfma


def audit_kernel(name, body):
    findings = []
    age, recent = {}, []
    n_mfma = 0
    labels = {s[:-1]: i for i, s in enumerate(body) if s.endswith(':')}
    for i, s in enumerate(body):
        if s.endswith(':'):
            continue
        n_mfma += _step(s, age, recent, findings)
        # a backward branch: the first instructions of the loop body run again right behind the last ones -- follow the back edge
        # for as many wait states as a hazard can span, with a COPY of the state at the branch
        op, _, rest = s.partition(' ')
        if op.startswith(('s_cbranch', 's_branch')) and rest.strip() in labels:
            back = labels[rest.strip()] < i
            # (a taken FORWARD branch: the target's first instructions run right behind the branch, without the skipped ones)
            tag = ' [across the back edge of the loop at {}]' if back else ' [behind the taken forward branch to {}]'
            age2, recent2 = dict(age), [list(e) for e in recent]
            states, j = 0, labels[rest.strip()]
            while j < (i if back else len(body)) and states < XDL_TO_ANY + 2:
                t = body[j]
                j += 1
                if t.endswith(':'):
                    continue
                if t.startswith(('s_cbranch', 's_branch', 's_endpgm')) and not back:
                    break                                   # (the next branch forks its own copy when the linear scan reaches it)
                _step(t, age2, recent2, findings, tag.format(rest.strip()))
                states += int(t.split()[1], 0) + 1 if t.startswith('s_nop') else 1
    return n_mfma, findings


def scratch_of(path):
    """kernel name -> (private_segment_fixed_size, vgpr_spill_count) from the amdhsa metadata at the end of the assembly"""
    out, name, priv = {}, None, 0
    for ln in open(path):
        m = re.match(r'\s+\.name:\s+(\S+)', ln)
        if m: name, priv = m.group(1), 0
        m = re.match(r'\s+\.private_segment_fixed_size:\s+(\d+)', ln)
        if m: priv = int(m.group(1))
        m = re.match(r'\s+\.vgpr_spill_count:\s+(\d+)', ln)
        if m and name: out[name] = (priv, int(m.group(1)))
    return out


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('path')
    ap.add_argument('--max-scratch', type=int, default=0)
    ap.add_argument('--allow-scratch', default=None)
    a = ap.parse_args()
    path = a.path
    bad = 0
    audited = 0
    scratch = scratch_of(path)
    for name, body in kernels(path):
        n_mfma, findings = audit_kernel(name, body)
        if not n_mfma: continue
        audited += 1
        priv, spilled = scratch.get(name, (0, 0))
        if (priv > a.max_scratch or spilled) and not (a.allow_scratch and re.search(a.allow_scratch, name)):
            findings.append('(S) {} bytes of scratch, {} register(s) spilled (limit {} bytes; --allow-scratch {})'.format(priv, spilled, a.max_scratch, a.allow_scratch))
        for rule in ('(A)', '(B)', '(S)'):
            of_rule = [f for f in findings if f.startswith(rule)]
            for f in of_rule[:6]: print('%s: %s: %s' % (path, name, f))
            if len(of_rule) > 6: print('%s: %s: ... %d more of %s' % (path, name, len(of_rule) - 6, rule))
        bad += len(findings)
    print('isa_audit: %s: %d kernel(s) with MFMAs audited, %d finding(s)' % (path, audited, bad))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
