#!/usr/bin/env python3
"""Scan the gfx950 ISA of conv3x3_wino4.hip for register dependences the compiler's hazard recogniser cannot see.

The kernel's MFMAs are inline asm: the compiler knows which registers each one reads and writes (so program order is kept) but not
that the instruction is an MFMA, so none of the matrix-pipe wait states of GCNHazardRecognizer are inserted around them.
For every `v_mfma` this lists, within a window of following instructions,
  WAR-AB  a VALU / DS / VMEM instruction that WRITES a register the MFMA reads as SrcA / SrcB,
  RAW-AB  (looking backwards) a VALU instruction that wrote SrcA / SrcB just before it,
  ACC     any non-MFMA instruction touching the MFMA's accumulator tuple.
usage: w4_isa_hazards.py file.s [kernel-name-substring] [window]
"""
import re, sys, collections

def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.fullmatch(r'([va])\[(\d+):(\d+)\]', tok)
    if m: return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r'([va])(\d+)', tok)
    if m: return {(m.group(1), int(m.group(2)))}
    return set()

def parse(path, want):
    kern, out, name = None, collections.OrderedDict(), None
    for ln in open(path):
        m = re.match(r'^(_Z\S+):', ln)
        if m:
            name = m.group(1); kern = [] if (want in name) else None
            if kern is not None: out[name] = kern
            continue
        if kern is None: continue
        if ln.startswith('.Lfunc_end'): kern = None; continue
        s = ln.split(';')[0].strip()
        if not s or s.startswith('.') or s.endswith(':'): continue
        parts = s.split(None, 1)
        op = parts[0]; args = [a.strip() for a in parts[1].split(',')] if len(parts) > 1 else []
        kern.append((op, args, s))
    return out

def writes(op, args):
    if op.startswith(('buffer_store', 'ds_write', 'global_store', 's_', 'ds_store')): return set()
    if op.startswith('v_cmp'): return set()
    return regs(args[0]) if args else set()

def reads(op, args):
    r = set()
    start = 0 if op.startswith(('buffer_store', 'ds_write', 'global_store')) else 1
    for a in args[start:]:
        r |= regs(a.split()[0]) if a else set()
    if op.endswith('_dpp') or op.startswith(('v_fmac', 'v_pk_fmac')): r |= regs(args[0])
    return r

def main():
    path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else 'wino4_3x3_kernel'; win = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    for name, k in parse(path, want).items():
        war = collections.Counter(); raw = collections.Counter(); acc = collections.Counter(); examples = {}
        n_mfma = 0
        for i, (op, args, s) in enumerate(k):
            if not op.startswith('v_mfma'): continue
            n_mfma += 1
            ab = regs(args[1]) | regs(args[2]); c = regs(args[0])
            for d in range(1, win + 1):
                if i + d >= len(k): break
                op2, a2, s2 = k[i + d]
                if op2.startswith('v_mfma'): continue
                w = writes(op2, a2)
                if w & ab:
                    kind = 'pk' if op2.startswith('v_pk') else ('dpp' if op2.endswith('_dpp') else ('ds' if op2.startswith('ds_') else ('vmem' if op2.startswith('buffer') else 'valu')))
                    # instructions between that are MFMAs (each holds the pipe 8 passes)
                    between = sum(1 for q in range(i + 1, i + d) if k[q][0].startswith('v_mfma'))
                    war[(kind, d, between)] += 1
                    examples.setdefault(('WAR', kind, d, between), (s, s2))
                if (w | reads(op2, a2)) & c:
                    acc[(op2, d)] += 1
            for d in range(1, 4):
                if i - d < 0: break
                op2, a2, s2 = k[i - d]
                if op2.startswith(('v_mfma', 'buffer', 'ds_', 's_')): continue
                if writes(op2, a2) & ab:
                    raw[('pk' if op2.startswith('v_pk') else 'valu', d)] += 1
                    examples.setdefault(('RAW', d), (s2, s))
        print(name, 'mfma:', n_mfma)
        print('  WAR on SrcA/SrcB (kind, distance in instructions, MFMAs in between): count')
        for key in sorted(war): print('   ', key, war[key])
        print('  RAW into SrcA/SrcB from VALU (kind, distance):', dict(raw))
        print('  non-MFMA instructions touching an accumulator within the window:', dict(acc))
        for key, (a, b) in examples.items():
            if key[0] == 'WAR' and key[1] in ('valu', 'pk', 'dpp') and key[2] <= 6: print('   e.g.', key, '|', a, '->', b)

if __name__ == '__main__':
    main()
