import sys, os, re, json, importlib
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, '3dgan-inversion_amd'))
pkg = os.path.join(ROOT, '3dgan-inversion_amd', 'inv3d_amd')
out = {}
pat_mod = re.compile(r"^([A-Z][A-Z0-9_]*) = .*os\.environ\.get\('(EG3D_[A-Z0-9_]+)'", re.M)
pat_any = re.compile(r"os\.environ\.get\('(EG3D_[A-Z0-9_]+)',\s*('[^']*'|str\([^)]*\)|None)")
for dirpath, _, files in os.walk(pkg):
    for f in sorted(files):
        if not f.endswith('.py'):
            continue
        path = os.path.join(dirpath, f)
        src = open(path).read()
        rel = os.path.relpath(path, pkg)[:-3].replace(os.sep, '.')
        modname = 'inv3d_amd' if rel == '__init__' else 'inv3d_amd.' + rel.replace('.__init__', '')
        names = pat_mod.findall(src)
        if names:
            m = importlib.import_module(modname)
            for attr, env in names:
                out[f'{modname}.{attr}'] = {'env': env, 'value': getattr(m, attr)}
        for env, default in pat_any.findall(src):
            out.setdefault('inline_defaults', {}).setdefault(env, [])
            if default not in out['inline_defaults'][env]:
                out['inline_defaults'][env].append(default)
print(json.dumps(out, indent=1, sort_keys=True, default=str))
