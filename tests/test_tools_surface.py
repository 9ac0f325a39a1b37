"""tools/ are probes and profile scripts written against this library over five rounds ("a reader cannot tell
which probes still run against the current ABI, and no test exercises them").  Here, on the CPU: every tools/*.py compiles, and
every name it takes from the package -- `ops.<name>`, `lib.carca_<name>` / `_lib.load().carca_<name>`, `M.<name>` of modules,
`from carca_replication_amd.<module> import <name>` -- exists today.  A tool that falls out of step with the library fails here, not on a GPU
box a round later.  (What the tools PRINT is evidence, quoted in TUNING.md; it is not checked.)"""
import ast
import glob
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))


def _names_used(tree, aliases):
    """(alias, attribute) pairs of every `alias.attribute` expression whose alias is one of `aliases`."""
    out = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in aliases:
            out.add((node.value.id, node.attr))
    return out


@pytest.mark.parametrize("path", TOOLS, ids=[os.path.basename(p) for p in TOOLS])
def test_tool_compiles_and_its_library_names_exist(path):
    from carca_replication_amd import _lib

    src = open(path).read()
    tree = ast.parse(src, filename=path)  # (a syntax error fails here)
    compile(src, path, "exec")
    exported = set(_lib.declared_symbols())
    # module aliases: `from carca_replication_amd import ops, _lib, modules as M` and `import carca_replication_amd.x as y`
    alias_of = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("carca_replication_amd"):
            if node.module == "carca_replication_amd":
                for a in node.names:
                    alias_of[a.asname or a.name] = "carca_replication_amd." + a.name
            else:
                mod = importlib.import_module(node.module)
                for a in node.names:
                    assert hasattr(mod, a.name), f"{os.path.basename(path)}: {node.module} has no {a.name}"
        elif isinstance(node, ast.Import):
            for a in node.names:
                if a.name.startswith("carca_replication_amd."):
                    alias_of[a.asname or a.name.split(".")[0]] = a.name
    mods = {}
    for alias, name in alias_of.items():
        try:
            mods[alias] = importlib.import_module(name)
        except ImportError:
            pass  # (a name imported from the package that is no module: checked below as an attribute of the package)
    for alias, attr in sorted(_names_used(tree, set(mods))):
        assert hasattr(mods[alias], attr), f"{os.path.basename(path)}: {mods[alias].__name__} has no attribute {attr}"
    # C entry points called through a loaded library handle: lib.carca_xxx( ... )
    for name in sorted(set(re.findall(r"\b(?:lib|_lib\.load\(\))\.(carca_[a-z0-9_]+)\s*\(", src))):
        assert name in exported, f"{os.path.basename(path)}: {name} is not declared in include/carca_hip.h"


def test_shell_tools_name_files_that_exist():
    """tools/*.sh drive rocprofv3 over tools/*.py and bench.py: every python file they name is there."""
    for sh in sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh"))):
        text = open(sh).read()
        for rel in set(re.findall(r"(tools/[A-Za-z0-9_]+\.py|bench\.py)", text)):
            assert os.path.exists(os.path.join(ROOT, rel)), f"{os.path.basename(sh)} names {rel}, which does not exist"
