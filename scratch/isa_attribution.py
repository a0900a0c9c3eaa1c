"""Per-source-line attribution of a kernel's ISA: compiles die_pic.hip for gfx950 with line tables
(-gline-tables-only: the code is the shipped -O3 code), finds the kernel by its mangled name, and counts the instructions behind
every `.loc` by class.  usage: python scratch/isa_attribution.py [mangled-name-prefix] [top-N] > profiles/rNN_k1_isa_by_source_line.txt
Static counts: an instruction inside the chunk loop runs once per chunk of 64 agents and wave, one in the prologue once per tile."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAME = sys.argv[1] if len(sys.argv) > 1 else '_Z18k_pic_forward_moveIfLi1ELb1ELb0ELb1ELb0ELb0ELb0EEv7FwdArgs7PicArgs'
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 60
out = '/tmp/isa_attr_die_pic.s'
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-std=c++17', '-ffp-contract=on', '-w', '--cuda-device-only',
                       '-gline-tables-only', '-S', os.path.join(ROOT, 'die_amd/csrc/die_pic.hip'), '-o', out])
src = open(out).read().split('\n')
files = {}
for l in src:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
start = [i for i, l in enumerate(src) if l.startswith(NAME + ':')][0]
end = [i for i, l in enumerate(src) if i > start and l.startswith('.Lfunc_end')][0]


def klass(op):
    if op in ('v_readlane_b32', 'v_writelane_b32', 'v_readfirstlane_b32'):
        return 'lane'
    if op.startswith('v_'):
        return 'valu64' if re.search(r'_(f64|u64|i64|b64)', op) and not op.startswith('v_cmp') and not op.startswith('v_cndmask') else 'valu'
    if op.startswith('s_load') or op.startswith('s_buffer'):
        return 'smem'
    if op in ('s_waitcnt', 's_nop', 's_barrier', 's_setprio', 's_sleep'):
        return 'wait'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')):
        return 'vmem'
    return 'other'


per = collections.defaultdict(collections.Counter)
tot = collections.Counter()
cur = None
seq = []                                    # (class, file, line) in program order
for l in src[start:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    t = l.strip().split()
    if not t or t[0].startswith(('.', ';')) or t[0].endswith(':'):
        continue
    k = klass(t[0])
    per[cur][k] += 1
    tot[k] += 1
    seq.append((k, cur))
text = {}
for f in ('die_pic.hip', 'die_forward.h', 'die_common.h', 'die_rng.h'):
    text[f] = open(os.path.join(ROOT, 'die_amd/csrc', f)).read().split('\n')
order = ('valu', 'valu64', 'lane', 'salu', 'smem', 'lds', 'vmem', 'wait')
print(f'kernel {NAME}')
print(f'static instruction counts by class: {dict(tot)}  (total {sum(tot.values())})')
byfile = collections.Counter()
for (f, ln), c in per.items():
    byfile[f] += sum(c.values())
print('by file:', dict(byfile.most_common()))
# regions of die_pic.hip by line ranges of the kernel (the chunk loop is what runs per 64 agents)
body = [i for i, l in enumerate(text['die_pic.hip']) if 'for (;;) {' in l and 'a wave' in l]
print()
print(f'{"file:line":26s} ' + ' '.join(f'{k:>6s}' for k in order) + '   source')
for (f, ln), c in sorted(per.items(), key=lambda kv: -(kv[1]['valu'] + kv[1]['valu64'] + kv[1]['lane']))[:TOP]:
    line = text.get(f, [''] * (ln + 1))[ln - 1].strip()[:110] if f in text and ln - 1 < len(text[f]) else ''
    print(f'{f + ":" + str(ln):26s} ' + ' '.join(f'{c[k]:6d}' for k in order) + '   ' + line)
# the chunk loop by POSITION: from the first instruction of its head to the last instruction of its last statement — what a wave
# executes once per chunk of 64 agents (everything inlined into it included)
pic = text['die_pic.hip']
lo = [i + 1 for i, l in enumerate(pic) if 'for (;;) {' in l and 'a wave' in l][0]
hi = [i + 1 for i, l in enumerate(pic) if i + 1 > lo and l.strip() == 'first = false;'][0]
idx = [k for k, (c, loc) in enumerate(seq) if loc and loc[0] == 'die_pic.hip' and lo <= loc[1] <= hi]
region = collections.Counter(c for c, _ in seq[idx[0]:idx[-1] + 1])
print()
print(f'chunk loop (die_pic.hip:{lo}-{hi}, by position): {dict(region)}  (total {sum(region.values())} of {len(seq)})')
# forward (die_forward.h) in total, the move / feeding / binning part of the chunk loop in total
fw = collections.Counter()
for (f, ln), c in per.items():
    if f == 'die_forward.h':
        fw.update(c)
print()
print('die_forward.h (the per-agent forward, inlined once):', dict(fw))
