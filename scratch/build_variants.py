"""Build ablation variants of libdie_hip.so into scratch/libs/ (kernel experiments)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'die_amd', 'csrc')
OUT = os.path.join(ROOT, 'scratch', 'libs')
os.makedirs(OUT, exist_ok=True)
VARIANTS = dict(a.split('=', 1) for a in sys.argv[1:]) if len(sys.argv) > 1 else {}
for name, flags in VARIANTS.items():
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-ffp-contract=on', '-shared', *flags.split(),
           *[os.path.join(CSRC, f) for f in ('die_agents.hip', 'die_env.hip', 'die_init.hip', 'die_sort.hip', 'die_pack.hip', 'die_ghost.hip', 'die_render.hip')], '-o', os.path.join(OUT, f'lib_{name}.so')]
    r = subprocess.run(cmd, capture_output=True, text=True)
    print(name, 'ok' if r.returncode == 0 else r.stderr[-2000:])
