"""Static guard against the failure that made round 4's GPU suite red: a name loaded inside a function that is neither local, enclosing,
module-level, imported nor a builtin (a NameError that only fires when that line runs -- on the GPU box, under `-x`).

`scan(path)` walks the symbol tables Python itself builds (`symtable`) and returns [(file, scope, name)]; `python tools/undefined_names.py`
prints them for the whole tree and exits 1 when there are any.  Used by tests/test_host_logic.py (CPU, ~1 s)."""
import builtins
import os
import symtable
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TARGETS = ("tests", "demovlp_amd", "tools", "oracle", "bench.py", "__graft_entry__.py")
MODULE_DUNDERS = {"__file__", "__name__", "__doc__", "__package__", "__spec__", "__loader__", "__builtins__", "__path__", "__class__",
                  "__annotations__", "__dict__", "__module__", "__qualname__"}


def _module_names(top):
    """Names bound at module level, plus names functions bind there through `global x` declarations."""
    names = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}

    def walk(t):
        for s in t.get_symbols():
            if t is not top and s.is_declared_global() and s.is_assigned():
                names.add(s.get_name())
        for c in t.get_children():
            walk(c)
    walk(top)
    return names


def scan_source(src, filename="<src>"):
    top = symtable.symtable(src, filename, "exec")
    bound = _module_names(top)
    if "import *" in src:                                    # a star import can bind anything: nothing to say about such a file
        return []
    ok = bound | set(dir(builtins)) | MODULE_DUNDERS
    bad = []

    def walk(t, trail):
        for s in t.get_symbols():
            if not s.is_referenced():
                continue
            name = s.get_name()
            if t is top:
                unresolved = not (s.is_assigned() or s.is_imported() or s.is_namespace()) and name not in ok
            else:
                # inside a function / class / comprehension: locals, parameters and free (enclosing) names resolve there; what is left is
                # looked up in the module and then in builtins at run time
                unresolved = s.is_global() and name not in ok
            if unresolved:
                bad.append((filename, ".".join(trail) or "<module>", name))
        for c in t.get_children():
            walk(c, trail + [c.get_name()])
    walk(top, [])
    return bad


def scan(path):
    with open(path, encoding="utf-8") as f:
        return scan_source(f.read(), os.path.relpath(path, ROOT))


def tree_files(root=ROOT):
    for t in TARGETS:
        p = os.path.join(root, t)
        if os.path.isfile(p):
            yield p
        for d, dirs, files in os.walk(p):
            dirs[:] = [x for x in dirs if x not in ("__pycache__", ".pytest_cache", "lib", "csrc")]
            for f in sorted(files):
                if f.endswith(".py"):
                    yield os.path.join(d, f)


def main():
    bad = [b for p in tree_files() for b in scan(p)]
    for f, scope, name in bad:
        print("%s: %s: undefined name %r" % (f, scope, name))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
