#!/usr/bin/env python
"""Register / LDS / scratch / code-size table of every kernel in the built libgm_hip.so.

    python tools/kernel_resources.py                  # print the table
    python tools/kernel_resources.py --write          # refresh profiles/kernel_resources.json (the committed baseline)
    python tools/kernel_resources.py --diff           # what changed against the baseline

The library's .hip_fatbin section holds one offload bundle per translation unit; each is unbundled with
clang-offload-bundler and read with llvm-readelf (kernel metadata notes + symbol sizes).  No GPU needed.
tests/test_host_cpu.py compares the built library with the committed baseline: a shipped kernel whose register count,
LDS size, scratch use or code size moves must move the baseline in the same commit (round 3's dominant forward kernel
drifted 8.00 -> 8.46 us across a round in which nothing in it was meant to change)."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "generative_models_amd", "libgm_hip.so")
BASELINE = os.path.join(ROOT, "profiles", "kernel_resources.json")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
          "group_segment_fixed_size", "private_segment_fixed_size")


def _run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def tools_available():
    return all(os.path.isfile(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))


def kernel_table(lib=LIB):
    """{demangled kernel name: {vgpr, agpr, sgpr, spills, lds, scratch, code_bytes}} of every kernel in `lib`."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        _run(os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
        for i, a in enumerate(starts):
            b = starts[i + 1] if i + 1 < len(starts) else len(data)
            piece, co = os.path.join(td, "b%d.bin" % i), os.path.join(td, "b%d.co" % i)
            open(piece, "wb").write(data[a:b])
            _run(os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                 "--input=" + piece, "--output=" + co, "--unbundle")
            if os.path.getsize(co) == 0:
                continue
            notes = _run(os.path.join(LLVM, "llvm-readelf"), "--notes", co)
            syms = _run(os.path.join(LLVM, "llvm-readelf"), "-s", "-W", co)
            size = {}
            for ln in syms.splitlines():
                f = ln.split()
                if len(f) >= 8 and f[3] == "FUNC":
                    size[f[7]] = int(f[2])
            for blk in notes.split("- .agpr_count:")[1:]:
                blk = ".agpr_count:" + blk
                nm = re.search(r"\.name:\s+(\S+)", blk).group(1)
                row = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in FIELDS}
                row["code_bytes"] = size.get(nm, 0)
                out[nm] = row
    names = list(out)
    dem = _run("c++filt", *names).strip().split("\n") if names else []
    table = {}
    for nm, d in zip(names, dem):
        d = d.replace("(anonymous namespace)::", "")
        d = re.sub(r"\(.*\)$", "", d)                      # argument list: the template arguments identify the kernel
        d = re.sub(r"^void ", "", d)
        r = out[nm]
        table[d] = {"vgpr": r["vgpr_count"], "agpr": r["agpr_count"], "sgpr": r["sgpr_count"],
                    "spills": r["vgpr_spill_count"] + r["sgpr_spill_count"], "lds": r["group_segment_fixed_size"],
                    "scratch": r["private_segment_fixed_size"], "code_bytes": r["code_bytes"]}
    return table


def diff(table, base):
    lines = []
    for k in sorted(set(table) | set(base)):
        if k not in base:
            lines.append("+ %s %s" % (k, table[k]))
        elif k not in table:
            lines.append("- %s %s" % (k, base[k]))
        elif table[k] != base[k]:
            ch = {f: (base[k][f], table[k][f]) for f in table[k] if base[k].get(f) != table[k][f]}
            lines.append("~ %s %s" % (k, ch))
    return lines


if __name__ == "__main__":
    t = kernel_table()
    if "--write" in sys.argv:
        json.dump(t, open(BASELINE, "w"), indent=0, sort_keys=True)
        print("wrote %s (%d kernels)" % (BASELINE, len(t)))
    elif "--diff" in sys.argv:
        d = diff(t, json.load(open(BASELINE)))
        print("\n".join(d) if d else "no change against %s" % BASELINE)
    else:
        for k in sorted(t):
            print("%-110s %s" % (k[:110], t[k]))
