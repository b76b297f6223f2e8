"""CPU tier: every name a function loads in bench.py, tools/*.py and tkmk/*.py is bound somewhere it can see (its own scope, an
enclosing function, the module, builtins).  No linter ships in this image; a stray block pasted into the wrong function (an
undefined `dist` in a bench leg that only runs on the GPU box) would otherwise surface only at round end."""
import ast
import builtins
import glob
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ([os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
         + sorted(glob.glob(os.path.join(ROOT, "tokamak-zk-evm_amd", "tkmk", "*.py"))))


def _bound_in(node):
    """names bound directly in this scope (not in nested function scopes): assignments, imports, defs, args, loop / with / except
    targets, comprehension variables"""
    out = set()
    if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
        a = node.args
        for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
            out.add(arg.arg)
    stack = list(ast.iter_child_nodes(node))
    while stack:
        n = stack.pop()
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(n.name)
            continue                                   # its body is another scope
        if isinstance(n, ast.Lambda):
            continue
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for al in n.names:
                out.add((al.asname or al.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            out.update(n.names)
        stack.extend(ast.iter_child_nodes(n))
    return out


def _undefined(tree):
    problems = []
    module_names = _bound_in(tree) | set(dir(builtins)) | {"__file__", "__name__", "__doc__"}

    def visit(node, visible):
        scope = visible | _bound_in(node)
        stack = list(ast.iter_child_nodes(node))
        while stack:
            n = stack.pop()
            if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
                for d in getattr(n, "decorator_list", []):
                    stack.append(d)
                for d in n.args.defaults + [k for k in n.args.kw_defaults if k is not None]:
                    stack.append(d)
                visit(n, scope)
                continue
            if isinstance(n, ast.ClassDef):
                visit(n, scope)
                continue
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in scope:
                problems.append((n.lineno, n.id))
            stack.extend(ast.iter_child_nodes(n))

    visit(tree, module_names)
    return problems


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.relpath(p, ROOT))
def test_no_undefined_names(path):
    tree = ast.parse(open(path).read(), path)
    assert _undefined(tree) == []


def test_the_checker_catches_a_misplaced_block():
    src = "import time\n\ndef main():\n    dist = None\n    return leg()\n\ndef leg():\n    if dist is not None:\n        return time.time()\n"
    assert _undefined(ast.parse(src)) == [(8, "dist")]
