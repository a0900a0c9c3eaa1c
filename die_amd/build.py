"""Builds die_amd/libdie_hip.so (gfx950 code objects only) with hipcc.  No GPU needed."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libdie_hip.so')
SOURCES = ['die_agents.hip', 'die_env.hip', 'die_init.hip', 'die_sort.hip', 'die_pack.hip', 'die_ghost.hip', 'die_render.hip', 'die_pic.hip', 'die_pic_refresh.hip', 'die_nca.hip']
HEADERS = ['die_common.h', 'die_rng.h', 'die_forward.h', os.path.join('..', '..', 'include', 'die_hip.h')]
# -ffp-contract=on: fuse a*b+c only inside one source expression.  hipcc's default (fast) fuses across
# statements, so the same inlined device function could round differently in two kernels (the fused and the
# stand-alone forward must give identical bits: tests/fuzz_cases.py fuzz_paths).
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-ffp-contract=on', '-Wall', '-Wno-unused-function']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc, *FLAGS, '-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(OUT, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', OUT])
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
