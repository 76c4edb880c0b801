"""The ABI structs of include/ivp_hip.h, member by member: offsets and sizes as the C compiler lays them out, against the
table kept beside the (uncompiled) Rust binding -- rust/ivp-hip-sys/abi_layout.json -- and against the `#[repr(C)]`
structs of rust/ivp-hip-sys/src/lib.rs (same members, same order; every function the header declares is declared there).

The reference is Rust (/root/reference/src/solve/solve_ivp.rs:99-108 is the call the boundary replaces,
src/solve/options.rs:75-123 the Options it mirrors) and this image has no Rust toolchain: this is what stands in for
`cargo build` of the binding.  Regenerate the table after an ABI change with
    python tests/test_abi_layout.py --write
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "ivp_hip.h")
TABLE = os.path.join(ROOT, "rust", "ivp-hip-sys", "abi_layout.json")
RUST = os.path.join(ROOT, "rust", "ivp-hip-sys", "src", "lib.rs")
STRUCTS = ("ivp_problem_t", "ivp_options_t", "ivp_batch_result_t", "ivp_run_stats_t", "ivp_step_log_t", "ivp_shard_t")


def header_structs():
    """struct name -> member names in declaration order, parsed from the header (comments stripped)."""
    src = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        body, name = m.group(1), m.group(2)
        if name not in STRUCTS:
            continue
        members = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):   # `int32_t a, b` declares two members
                mm = re.search(r"(\w+)\s*(\[\s*\d+\s*\])?\s*$", part.strip())
                members.append(mm.group(1))
        out[name] = members
    return out


def c_layout(tmp):
    """Compile a C program that prints offsetof / sizeof of every member with the system C compiler."""
    structs = header_structs()
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "ivp_hip.h"', 'int main(void) {', '  printf("{\\n");']
    for si, (s, members) in enumerate(structs.items()):
        lines.append(f'  printf("  \\"{s}\\": {{\\"sizeof\\": %zu, \\"members\\": [", sizeof({s}));')
        for i, mbr in enumerate(members):
            sep = ", " if i else ""
            lines.append(f'  printf("{sep}[\\"{mbr}\\", %zu, %zu]", offsetof({s}, {mbr}), sizeof((({s} *)0)->{mbr}));')
        lines.append('  printf("]}%s\\n", ' + ('","' if si + 1 < len(structs) else '""') + ');')
    lines += ['  printf("}\\n");', '  return 0;', '}']
    src = os.path.join(tmp, "abi_layout.c")
    open(src, "w").write("\n".join(lines) + "\n")
    exe = os.path.join(tmp, "abi_layout")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
    return json.loads(subprocess.check_output([exe]).decode())


def rust_structs():
    src = open(RUST).read()
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub struct (\w+)\s*\{(.*?)\n\}", src, flags=re.S):
        out[m.group(1)] = re.findall(r"pub (\w+)\s*:", m.group(2))
    return out, set(re.findall(r"pub fn (ivp_\w+)\s*\(", src))


def test_c_layout_matches_the_table_kept_with_the_rust_binding(tmp_path):
    got = c_layout(str(tmp_path))
    want = json.load(open(TABLE))
    assert set(got) == set(STRUCTS) == set(want["structs"])
    for s in STRUCTS:
        assert got[s] == want["structs"][s], s
    assert want["abi_version"] == int(re.search(r"#define\s+IVP_HIP_ABI_VERSION\s+(\d+)", open(HDR).read()).group(1))


def test_rust_structs_list_the_header_members_in_order_and_every_function_is_declared():
    structs, fns = rust_structs()
    hdr = header_structs()
    for s in STRUCTS:
        assert structs.get(s) == hdr[s], (s, structs.get(s), hdr[s])
    declared = set(re.findall(r"^\s*(?:int|void|const char \*)\s*\*?\s*(ivp_\w+)\s*\(", open(HDR).read(), flags=re.M))
    assert declared == fns, declared ^ fns
    src = open(RUST).read()
    assert re.search(r"IVP_HIP_ABI_VERSION: c_int = (\d+)", src).group(1) == re.search(r"#define\s+IVP_HIP_ABI_VERSION\s+(\d+)", open(HDR).read()).group(1)


if __name__ == "__main__" and "--write" in sys.argv:
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        lay = c_layout(d)
    ver = int(re.search(r"#define\s+IVP_HIP_ABI_VERSION\s+(\d+)", open(HDR).read()).group(1))
    json.dump({"abi_version": ver, "what": "offsetof / sizeof of every member of every struct of include/ivp_hip.h on x86-64 Linux (LP64), "
               "[member, offset, size]; the #[repr(C)] structs of src/lib.rs must produce the same layout", "structs": lay},
              open(TABLE, "w"), indent=1)
    print("wrote", TABLE)


def test_the_ctypes_binding_has_the_same_layout():
    """ivp_amd/_lib.py is the binding that is actually exercised: its ctypes structures against the same table."""
    sys.path.insert(0, ROOT)
    from ivp_amd import _lib
    import ctypes as C
    want = json.load(open(TABLE))["structs"]
    pairs = {"ivp_problem_t": _lib.ProblemT, "ivp_options_t": _lib.OptionsT, "ivp_batch_result_t": _lib.BatchResultT,
             "ivp_run_stats_t": _lib.RunStatsT, "ivp_step_log_t": _lib.StepLogT, "ivp_shard_t": _lib.ShardT}
    for name, st in pairs.items():
        assert C.sizeof(st) == want[name]["sizeof"], name
        assert [f[0] for f in st._fields_] == [m[0] for m in want[name]["members"]], name
        for mbr, off, size in want[name]["members"]:
            d = getattr(st, mbr)
            assert (d.offset, d.size) == (off, size), (name, mbr)
