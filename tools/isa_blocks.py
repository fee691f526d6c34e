#!/usr/bin/env python3
"""Per-basic-block instruction counts of one kernel's gfx950 assembly (tools/eval_isa.sh writes it): tools/isa_blocks.py FILE [first last]"""
import re
import sys


def cls(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith(('v_readlane', 'v_writelane')):
        return 'spill'
    if op.startswith('v_mov'):
        return 'mov'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'scratch_', 'buffer_', 'flat_')):
        return 'vmem'
    if op.startswith('s_nop'):
        return 'nop'
    if op.startswith('s_'):
        return 'salu'
    return None


def main():
    L = open(sys.argv[1]).read().split('\n')
    lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, len(L))
    blocks, cur = [], None
    for i, l in enumerate(L):
        m = re.match(r'^(\.LBB\d+_\d+):', l) or re.match(r'^; %bb\.(\d+):', l)
        if m:
            cur = {'name': m.group(1), 'line': i + 1, 'cnt': {}, 'loop': ''}
            mm = re.search(r'in Loop: Header=(\S+) Depth=(\d+)', l)
            if mm:
                cur['loop'] = mm.group(1) + ' d' + mm.group(2)
            blocks.append(cur)
            continue
        if cur is None:
            continue
        t = l.strip().split()
        if not t or t[0].startswith(';') or t[0].startswith('.'):
            mm = re.search(r'Loop Header: Depth=(\d+)', l)
            if mm:
                cur['loop'] = 'HEADER d' + mm.group(1)
            continue
        c = cls(t[0])
        if c:
            cur['cnt'][c] = cur['cnt'].get(c, 0) + 1
    for b in blocks:
        if lo < b['line'] < hi:
            print(b['line'], b['name'], b['loop'], b['cnt'])


main()
