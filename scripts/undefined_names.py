"""Poor man's pyflakes (no linter in the image): names a module loads but never binds anywhere (import, def, class, assignment,
argument, loop / with / comprehension target).  Usage: python scripts/undefined_names.py FILE..."""
import ast
import builtins
import sys


def check(path):
    tree = ast.parse(open(path).read(), path)
    bound = set(dir(builtins)) | {"__file__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                bound.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            bound.add(n.name)
            if not isinstance(n, ast.ClassDef):
                for a in n.args.args + n.args.kwonlyargs + n.args.posonlyargs:
                    bound.add(a.arg)
                if n.args.vararg:
                    bound.add(n.args.vararg.arg)
                if n.args.kwarg:
                    bound.add(n.args.kwarg.arg)
        elif isinstance(n, ast.Lambda):
            for a in n.args.args + n.args.kwonlyargs:
                bound.add(a.arg)
            if n.args.vararg:
                bound.add(n.args.vararg.arg)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            bound.add(n.id)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
    bad = sorted({(n.lineno, n.id) for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound})
    for ln, name in bad:
        print(f"{path}:{ln}: undefined name {name}")
    return len(bad)


if __name__ == "__main__":
    sys.exit(1 if sum(check(p) for p in sys.argv[1:]) else 0)
