"""A/B variants of one csrc file for phase-skip experiments: applies exact-string substitutions to a copy of the source, compiles it and links it with the other
objects of the product build into tools/bin/var/lib<name>.so (DCV_LIB=<path> makes deepcv_amd._native load it).  Not a product path: variants may compute
wrong results on purpose (they skip work to show what it costs).  usage: make_variants.py <spec.py>  — the spec defines SRC and VARIANTS = {name: [(old, new), ...]}"""
import os, subprocess, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = runpy.run_path(sys.argv[1])
src = os.path.join(ROOT, 'deepcv_amd', 'csrc', spec['SRC'])
text = open(src).read()
objs = [os.path.join(ROOT, 'build', 'csrc', f) for f in os.listdir(os.path.join(ROOT, 'build', 'csrc')) if f.endswith('.o') and f != spec['SRC'].replace('.hip', '.o')]
out_dir = os.path.join(ROOT, 'tools', 'bin', 'var'); os.makedirs(out_dir, exist_ok=True)
for name, subs in spec['VARIANTS'].items():
    t = text
    for old, new in subs:
        assert t.count(old) >= 1, (name, old[:60])
        t = t.replace(old, new)
    vsrc = os.path.join(ROOT, 'deepcv_amd', 'csrc', '_var_' + name + '.hip')
    open(vsrc, 'w').write(t)
    try:
        obj = os.path.join(out_dir, name + '.o')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-gpu-rdc', '-Wno-unused-function', '-c', vsrc, '-o', obj])
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-shared', '-fPIC', '--offload-arch=gfx950', '-o', os.path.join(out_dir, 'lib' + name + '.so'), obj] + objs)
        os.remove(obj)
    finally:
        os.remove(vsrc)
    print('built', name)
