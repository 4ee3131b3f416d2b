#!/usr/bin/env python3
"""Instruction-mix summary of one kernel in a hipcc -S listing.
usage: isa_stats.py file.s <substring of mangled kernel name> [--dump]"""
import collections, re, sys

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if l.endswith(':') is False and re.match(r'^_Z\w*%s\w*:' % re.escape(key), l):
            start = i; break
        if re.match(r'^(_Z\w*%s\w*):' % re.escape(key), l):
            start = i; break
    if start is None:
        sys.exit('kernel not found')
    ops = collections.Counter(); body = []
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith('.Lfunc_end'): break
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
        op = t.split()[0]
        ops[op] += 1; body.append(t)
    cls = collections.Counter()
    for k, v in ops.items():
        c = ('trans' if re.match(r'v_(rsq|sqrt|rcp|exp|log|sin|cos)', k) else 'valu' if k.startswith('v_') else 'salu' if k.startswith('s_')
             else 'lds' if k.startswith('ds_') else 'vmem' if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other')
        cls[c] += v
    print(lines[start]); print('total', sum(ops.values()), dict(cls))
    for k, v in ops.most_common(60): print('  %-28s %d' % (k, v))
    if '--dump' in sys.argv: print('\n'.join(body))

main()
