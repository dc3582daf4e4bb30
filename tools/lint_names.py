#!/usr/bin/env python
"""Poor man's pyflakes (none in the image): names that are loaded somewhere in a module but bound nowhere in it."""
import ast, builtins, sys

def check(path):
    tree = ast.parse(open(path).read(), path)
    bound, loaded = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}, []
    for n in ast.walk(tree):
        if isinstance(n, ast.Name):
            (bound.add(n.id) if isinstance(n.ctx, (ast.Store, ast.Del)) else loaded.append((n.id, n.lineno)))
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            bound.add(n.name)
        elif isinstance(n, ast.arg):
            bound.add(n.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                bound.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            bound.update(n.names)
    bad = sorted({(name, ln) for name, ln in loaded if name not in bound})
    for name, ln in bad:
        print(f"{path}:{ln}: undefined name {name!r}")
    return len(bad)

if __name__ == "__main__":
    sys.exit(1 if sum(check(p) for p in sys.argv[1:]) else 0)
