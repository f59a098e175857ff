"""GPU-only tests, bench.py and the scripts never run in the CPU suite: a name that is not defined anywhere (a block pasted into the wrong function) would only show on the GPU box.
This walks every Python file of the repo and reports names that are loaded in a function without being bound in it, at module level or in builtins."""
import ast
import builtins
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bound_in(node):
    names = set()
    for t in ast.walk(node):
        if isinstance(t, ast.Name) and isinstance(t.ctx, (ast.Store, ast.Del)):
            names.add(t.id)
        elif isinstance(t, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(t.name)
        elif isinstance(t, (ast.Import, ast.ImportFrom)):
            names.update((a.asname or a.name).split(".")[0] for a in t.names)
        elif isinstance(t, ast.ExceptHandler) and t.name:
            names.add(t.name)
        elif isinstance(t, ast.arg):
            names.add(t.arg)
        elif isinstance(t, (ast.Global, ast.Nonlocal)):
            names.update(t.names)
    return names


def undefined_names(path):
    tree = ast.parse(open(path).read(), path)
    module = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    for n in tree.body:      # module level: everything bound outside function bodies, plus the functions / classes themselves
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            module.add(n.name)
        else:
            module |= _bound_in(n)
    for t in ast.walk(tree):
        if isinstance(t, ast.Global):
            module.update(t.names)
    out = []
    funcs = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef))]
    for c in tree.body:
        if isinstance(c, ast.ClassDef):
            module |= {m.name for m in c.body if isinstance(m, ast.FunctionDef)}
            funcs += [m for m in c.body if isinstance(m, (ast.FunctionDef, ast.AsyncFunctionDef))]
    for fn in funcs:         # a top-level function with everything nested in it: a name bound anywhere inside counts as bound (coarse, no false alarms)
        bound = module | _bound_in(fn)
        out += ["%s:%d: %r in %s" % (os.path.relpath(path, ROOT), t.lineno, t.id, fn.name) for t in ast.walk(fn)
                if isinstance(t, ast.Name) and isinstance(t.ctx, ast.Load) and t.id not in bound]
    return out


def test_no_undefined_names_in_any_python_file():
    files = [p for pat in ("tests/*.py", "tests/golden/*.py", "*.py", "scripts/*.py", "ground-fusion_amd/*.py", "oracle/*.py") for p in glob.glob(os.path.join(ROOT, pat))]
    assert len(files) > 40
    bad = [line for p in sorted(files) for line in undefined_names(p)]
    assert not bad, "\n".join(bad)
