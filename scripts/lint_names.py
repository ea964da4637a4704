"""Undefined-name lint for the test tier (pyflakes is not installed): every Name that is LOADED in a module, function
or lambda must be bound somewhere in an enclosing scope, the module or builtins.  Also walks the subprocess scripts of
tests/plugin_scripts.py::CASES (source held in strings) and fails on statements that are unreachable leftovers of a
paste: a `def` body followed by more-indented code can not parse, a test body that names another script's variables
does not resolve.  Run by scripts/cpu_gate.sh; exit code 1 on any finding (VERDICT r4 "Next round" 1c)."""
import ast
import builtins
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILTINS = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__builtins__", "__spec__", "__path__", "__package__"}


class Scope:
    def __init__(self, node, parent, kind):
        self.node, self.parent, self.kind = node, parent, kind
        self.bound, self.loads, self.globals_ = set(), [], set()


def _bind_target(scope, t):
    for n in ast.walk(t):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            scope.bound.add(n.id)


class Walker(ast.NodeVisitor):
    def __init__(self, tree, predefined=()):
        self.scope = Scope(tree, None, "module")
        self.scope.bound |= set(predefined)
        self.all = [self.scope]
        self.star = False
        self.visit(tree)

    def _push(self, node, kind):
        s = Scope(node, self.scope, kind)
        self.all.append(s)
        self.scope = s
        return s

    def _pop(self):
        self.scope = self.scope.parent

    def _args(self, a):
        for x in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
            self.scope.bound.add(x.arg)

    def _func(self, node):
        self.scope.bound.add(node.name)
        for d in node.decorator_list:
            self.visit(d)
        for d in node.args.defaults + [k for k in node.args.kw_defaults if k is not None]:
            self.visit(d)
        for x in node.args.posonlyargs + node.args.args + node.args.kwonlyargs:
            if x.annotation is not None:
                self.visit(x.annotation)
        if node.returns is not None:
            self.visit(node.returns)
        self._push(node, "function")
        self._args(node.args)
        for st in node.body:
            self.visit(st)
        self._pop()

    visit_FunctionDef = visit_AsyncFunctionDef = _func

    def visit_Lambda(self, node):
        for d in node.args.defaults + [k for k in node.args.kw_defaults if k is not None]:
            self.visit(d)
        self._push(node, "function")
        self._args(node.args)
        self.visit(node.body)
        self._pop()

    def visit_ClassDef(self, node):
        self.scope.bound.add(node.name)
        for d in node.decorator_list + node.bases + [k.value for k in node.keywords]:
            self.visit(d)
        self._push(node, "class")
        for st in node.body:
            self.visit(st)
        self._pop()

    def _comp(self, node):
        self._push(node, "function")
        for g in node.generators:
            _bind_target(self.scope, g.target)
        for g in node.generators:
            self.visit(g.iter)
            for c in g.ifs:
                self.visit(c)
        if isinstance(node, ast.DictComp):
            self.visit(node.key)
            self.visit(node.value)
        else:
            self.visit(node.elt)
        self._pop()

    visit_ListComp = visit_SetComp = visit_GeneratorExp = visit_DictComp = _comp

    def visit_Name(self, node):
        if isinstance(node.ctx, ast.Load):
            self.scope.loads.append(node)
        else:
            self.scope.bound.add(node.id)

    def visit_NamedExpr(self, node):
        s = self.scope
        while s.kind == "function" and isinstance(s.node, (ast.ListComp, ast.SetComp, ast.GeneratorExp, ast.DictComp)):
            s = s.parent
        s.bound.add(node.target.id)
        self.visit(node.value)

    def visit_Import(self, node):
        for a in node.names:
            self.scope.bound.add((a.asname or a.name).split(".")[0])

    def visit_ImportFrom(self, node):
        for a in node.names:
            if a.name == "*":
                self.star = True
            else:
                self.scope.bound.add(a.asname or a.name)

    def visit_Global(self, node):
        self.scope.globals_ |= set(node.names)
        self.all[0].bound |= set(node.names)

    def visit_Nonlocal(self, node):
        self.scope.bound |= set(node.names)

    def visit_ExceptHandler(self, node):
        if node.name:
            self.scope.bound.add(node.name)
        self.generic_visit(node)

    def visit_MatchAs(self, node):
        if node.name:
            self.scope.bound.add(node.name)
        self.generic_visit(node)


def undefined_names(source, filename, predefined=()):
    tree = ast.parse(source, filename)
    w = Walker(tree, predefined)
    if w.star:
        return []
    out = []
    for s in w.all:
        for n in s.loads:
            t, found = s, False
            while t is not None:
                # (class scopes are skipped for lookups that start in a nested function, as Python does)
                if n.id in t.bound and (t.kind != "class" or t is s):
                    found = True
                    break
                t = t.parent
            if not found and n.id not in BUILTINS:
                out.append((filename, n.lineno, n.id))
    return out


def main():
    files = sorted(set(glob.glob(os.path.join(ROOT, "tests", "**", "*.py"), recursive=True) + glob.glob(os.path.join(ROOT, "*.py")) +
                       glob.glob(os.path.join(ROOT, "arrow_amd", "*.py")) + glob.glob(os.path.join(ROOT, "oracle", "*.py")) +
                       glob.glob(os.path.join(ROOT, "scripts", "*.py"))))
    bad = []
    for f in files:
        try:
            bad += undefined_names(open(f).read(), os.path.relpath(f, ROOT))
        except SyntaxError as e:
            bad.append((os.path.relpath(f, ROOT), e.lineno or 0, "SyntaxError: %s" % e.msg))
    sys.path.insert(0, ROOT)
    from tests import plugin_scripts as S

    ids = [c[0] for c in S.CASES]
    assert len(set(ids)) == len(ids), "duplicate case ids in tests/plugin_scripts.py::CASES"
    for cid, script, marker, _scale, _doc in S.CASES:
        try:
            bad += undefined_names(script, "plugin_scripts.py::CASES[%s]" % cid, predefined=("ROOT",))
        except SyntaxError as e:
            bad.append(("plugin_scripts.py::CASES[%s]" % cid, e.lineno or 0, "SyntaxError: %s" % e.msg))
        if ('print("%s' % marker) not in script:
            bad.append(("plugin_scripts.py::CASES[%s]" % cid, 0, "the script never prints its marker %s" % marker))
    for name in dir(S):
        if name.endswith("SCRIPT") and not any(getattr(S, name) is c[1] for c in S.CASES):
            bad.append(("tests/plugin_scripts.py", 0, "%s is in no row of CASES" % name))
    for f, line, what in bad:
        print("%s:%d: undefined name / defect: %s" % (f, line, what))
    print("lint_names: %d files + %d scripts, %d findings" % (len(files), len(S.CASES), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
