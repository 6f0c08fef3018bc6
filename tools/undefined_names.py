"""Names a module's functions read from module scope that the imported module does not define (and builtins do not):
the NameErrors a code move leaves behind on paths no test runs.  usage: undefined_names.py package.module ..."""
import builtins
import importlib
import symtable
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def walk(table, mod, out, path=""):
    for child in table.get_children():
        name = path + "." + child.get_name() if path else child.get_name()
        if child.get_type() == "function":
            for sym in child.get_symbols():
                if sym.is_global() and sym.is_referenced() and not sym.is_assigned():
                    n = sym.get_name()
                    if not hasattr(mod, n) and not hasattr(builtins, n):
                        out.append((name, n))
        walk(child, mod, out, name)


bad = 0
for modname in sys.argv[1:]:
    mod = importlib.import_module(modname)
    src = open(mod.__file__).read()
    out = []
    walk(symtable.symtable(src, mod.__file__, "exec"), mod, out)
    for where, n in out:
        print("%s: %s reads undefined name %r" % (modname, where, n))
    bad += len(out)
sys.exit(1 if bad else 0)
