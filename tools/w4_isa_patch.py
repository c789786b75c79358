#!/usr/bin/env python3
"""Instrument the gfx950 assembly of conv3x3_wino4.hip with wait states at chosen instruction pairs (root-causing the packed-fp32
failure of round 4: which pair needs them?).  usage: w4_isa_patch.py in.s out.s rule[,rule...]
rules:
  pk_post:N       s_nop N-1 after every v_pk_*                      (packed result -> any consumer)
  pk_pre:N        s_nop N-1 before every v_pk_*                     (any producer -> packed source)
  mfma_war:N      s_nop N-1 after an MFMA whose SrcA/SrcB register is written by a VALU instruction within the next 8 instructions
  mfma_post:N     s_nop N-1 after every MFMA that is followed by a non-MFMA instruction
  mfma_pre:N      s_nop N-1 before every MFMA that follows a VALU instruction
  ds_pre:N        s_nop N-1 before every ds_write_b128
  dpp_pre:N       s_nop N-1 before every *_dpp
  acc_guard:N     before any non-MFMA instruction that reads or writes a register an MFMA wrote fewer than N wait states ago
                  (every instruction counted as one state): s_nop up to N states -- the XDL-write -> VALU / VMEM access hazard
  acc_guard_valu:N / acc_guard_mem:N   the same for VALU instructions only / memory instructions only
Only the wino4_3x3_kernel functions are touched."""
import re, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from w4_isa_hazards import regs, writes

def opof(ln):
    s = ln.split(';')[0].strip()
    if not s or s.startswith('.') or s.endswith(':'): return None, []
    p = s.split(None, 1)
    return p[0], ([a.strip() for a in p[1].split(',')] if len(p) > 1 else [])

def is_valu(op): return op is not None and op.startswith('v_') and not op.startswith('v_mfma')

def main():
    src, dst, rules = sys.argv[1], sys.argv[2], dict((r.split(':') + ['1'])[:2] for r in sys.argv[3].split(','))
    rules = {k: int(v) for k, v in rules.items()}
    lines = open(src).read().split('\n')
    out, inside, counts = [], False, {k: 0 for k in rules}
    nop = lambda n: '\ts_nop %d' % (n - 1)
    age = {}            # register -> wait states since an MFMA wrote it
    from w4_isa_hazards import reads
    for idx, ln in enumerate(lines):
        m = re.match(r'^(_Z\S+):', ln)
        if m: inside = 'wino4_3x3_kernel' in m.group(1)
        if ln.startswith('.Lfunc_end'): inside = False
        op, args = opof(ln)
        if not inside or op is None:
            out.append(ln); continue
        # previous / next real instructions
        def nxt(k):
            r, j = [], idx + 1
            while j < len(lines) and len(r) < k:
                o, a = opof(lines[j])
                if lines[j].startswith('.Lfunc_end'): break
                if o is not None: r.append((o, a))
                j += 1
            return r
        def prv():
            j = idx - 1
            while j >= 0:
                o, a = opof(lines[j])
                if o is not None: return o, a
                j -= 1
            return None, []
        pre, post = [], []
        for gk in ('acc_guard', 'acc_guard_valu', 'acc_guard_mem'):
            if gk in rules and not op.startswith('v_mfma') and not op.startswith('s_'):
                if gk == 'acc_guard_valu' and not is_valu(op): continue
                if gk == 'acc_guard_mem' and is_valu(op): continue
                touched = (writes(op, args) | reads(op, args)) & set(age)
                if touched:
                    need = rules[gk] - min(age[r] for r in touched)
                    if need > 0:
                        while need > 0: pre.append(nop(min(need, 16))); need -= 16
                        counts[gk] += 1
                        for r in list(age): age[r] += rules[gk]
        if op.startswith('v_mfma'):
            for r in regs(args[0]): age[r] = 0
        m_n = re.match(r's_nop', op)
        step = (int(args[0]) + 1) if m_n else 1
        for r in list(age):
            if not (op.startswith('v_mfma') and r in regs(args[0])): age[r] += step
            if age[r] > 64: del age[r]
        if op.startswith('v_pk_'):
            if 'pk_pre' in rules: pre.append(nop(rules['pk_pre'])); counts['pk_pre'] += 1
            if 'pk_post' in rules: post.append(nop(rules['pk_post'])); counts['pk_post'] += 1
        if op.startswith('v_mfma'):
            following = nxt(8)
            if 'mfma_war' in rules:
                ab = regs(args[1]) | regs(args[2])
                if any(is_valu(o) and (writes(o, a) & ab) for o, a in following):
                    post.append(nop(rules['mfma_war'])); counts['mfma_war'] += 1
            if 'mfma_post' in rules and following and not following[0][0].startswith('v_mfma'):
                post.append(nop(rules['mfma_post'])); counts['mfma_post'] += 1
            if 'mfma_pre' in rules and is_valu(prv()[0]):
                pre.append(nop(rules['mfma_pre'])); counts['mfma_pre'] += 1
        if op == 'ds_write_b128' and 'ds_pre' in rules:
            pre.append(nop(rules['ds_pre'])); counts['ds_pre'] += 1
        if op.endswith('_dpp') and 'dpp_pre' in rules:
            pre.append(nop(rules['dpp_pre'])); counts['dpp_pre'] += 1
        out.extend(pre); out.append(ln); out.extend(post)
    open(dst, 'w').write('\n'.join(out))
    print(dst, counts)

if __name__ == '__main__':
    main()
