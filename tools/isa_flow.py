"""Compact instruction-class flow of one kernel from the ISA text tools/isa.sh leaves in /tmp: where the global loads (G), LDS writes (W) /
reads (R), MFMAs (M), barriers (|B|) and s_waitcnt ([..]) sit relative to each other, one line per basic block.  Round 6 found with it that
the float32 weight gradient's "software pipeline" did not exist in the binary (vmcnt(0) + the masking selects stood before the MFMA loop).

  bash tools/isa.sh conv_wgrad && python tools/isa_flow.py conv_wgrad 'conv_wgrad_kernelILi3ELi3ELi2' [--blocks]
"""
import re
import sys


def kind(l):
    m = re.match(r'\s*(\S+)', l)
    op = m.group(1) if m else ''
    if op.startswith(('global_load', 'buffer_load')):
        return 'Gl' if ' lds' in l else 'G'
    if op.startswith(('global_store', 'buffer_store', 'global_atomic')):
        return 'S'
    if op.startswith('ds_write') or op.startswith('ds_store'):
        return 'W'
    if op.startswith('ds_read') or op.startswith('ds_load'):
        return 'R'
    if 'mfma' in op:
        return 'M'
    if op == 's_barrier':
        return '|B|'
    if op == 's_waitcnt':
        return '[' + l.split('s_waitcnt')[1].strip().replace('vmcnt', 'vm').replace('lgkmcnt', 'lgkm') + ']'
    if op.startswith(('s_cbranch', 's_branch')):
        return '<' + l.split()[-1].replace('.LBB', '') + '>'
    if re.match(r'\.LBB\d+_\d+:', op):
        return '\n' + op.replace('.LBB', '')
    if op.startswith(('v_readlane', 'v_writelane')):
        return 'l'
    if op.startswith('scratch_'):
        return '!'
    if op.startswith('s_'):
        return 's'
    if op.startswith('v_'):
        return 'v'
    return ''


def squeeze(t):
    # runs of one letter -> letter x count
    return re.sub(r'([svlMRWGS!])\1{3,}', lambda m: f'{m.group(1)}x{len(m.group(0))} ', t)


def main():
    src, pat = sys.argv[1], sys.argv[2]
    s = open(f'/tmp/{src}-hip-amdgcn-amd-amdhsa-gfx950.s').read()
    for m in re.finditer(r'^(_Z\S*' + re.escape(pat) + r'\S*):', s, re.M):
        i = m.end()
        j = s.index('s_endpgm', i)
        body = s[i:j].split('\n')
        print('==', m.group(1), len(body), 'lines')
        print(squeeze(''.join(kind(l) for l in body)))


if __name__ == '__main__':
    main()
