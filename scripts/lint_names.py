#!/usr/bin/env python3
"""Tiny undefined-name check (no pyflakes in this image): every Name that is loaded must be a builtin, a module-level
binding, or bound somewhere in an enclosing function.  Catches the NameErrors that would otherwise cost a GPU call."""
import ast
import builtins
import sys


def names_bound(node):
    out = set()
    for n in ast.walk(node):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(n.name)
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            a = n.args
            for x in a.args + a.kwonlyargs + a.posonlyargs:
                out.add(x.arg)
            if a.vararg:
                out.add(a.vararg.arg)
            if a.kwarg:
                out.add(a.kwarg.arg)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for al in n.names:
                out.add((al.asname or al.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            out.update(n.names)
    return out


def check(path):
    tree = ast.parse(open(path).read(), path)
    bound = names_bound(tree) | set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    bad = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound:
            bad.append((n.lineno, n.id))
    return bad


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        for ln, name in check(p):
            print("%s:%d: undefined name %s" % (p, ln, name))
            rc = 1
    sys.exit(rc)
