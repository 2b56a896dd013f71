"""Poor man's pyflakes (not installed here): names that are loaded somewhere in a module but neither bound in an
enclosing scope, nor at module level, nor a builtin.  Usage: python tools/diag/undefined_names.py file.py ..."""
import ast
import builtins
import sys


def bound_in(node):
    names = set()
    for n in ast.walk(node):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            names.add(n.id)
        elif isinstance(n, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            names.add(n.name)
        elif isinstance(n, ast.arg):
            names.add(n.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                names.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            names.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            names.update(n.names)
    return names


def check(path):
    tree = ast.parse(open(path).read())
    module = set()
    for node in tree.body:  # module level bindings (incl. those made inside if / try / with blocks)
        module |= bound_in(node) if not isinstance(node, (ast.FunctionDef, ast.ClassDef)) else {node.name}
    bad = []

    def visit(node, scopes):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda, ast.ClassDef)):
            scopes = scopes + [bound_in(node)]
        for child in ast.iter_child_nodes(node):
            visit(child, scopes)
        if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Load):
            if node.id not in module and not hasattr(builtins, node.id) and node.id != "__file__" and not any(node.id in s for s in scopes):
                bad.append((node.lineno, node.id))

    visit(tree, [])
    return sorted(set(bad))


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        for ln, name in check(p):
            print("%s:%d: undefined name %r" % (p, ln, name))
            rc = 1
    sys.exit(rc)
