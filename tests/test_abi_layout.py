"""The three statements of the ABI structs agree field for field: include/spock_amd.h (what the library is compiled against),
spock.jl_amd/_lib.py (ctypes, what the tests and the benchmark call through) and spock.jl_amd/julia/SpockAMD.jl (the `ccall`
binding a maintainer of the reference adds, src/model.jl:273-297 -- unexecutable here, so this test is what pins it).

Round 2 shipped a Julia `SpockProblemC` that lacked the trailing `n_rms` of `spock_problem_t`: spock_create then read four bytes
past the Julia struct.  Only symbol names were tested.  This file parses all three sources and compares names, order, types,
and the ctypes sizes against what the compiled library itself reports (spock_abi_struct_sizes)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "spock_amd.h")
JULIA = os.path.join(ROOT, "spock.jl_amd", "julia", "SpockAMD.jl")

# C struct name -> (Julia struct name, ctypes mirror name)
STRUCTS = {"spock_problem_t": ("SpockProblemC", "Problem"), "spock_options_t": ("SpockOptionsC", "Options"),
           "spock_result_t": ("SpockResultC", "Result"), "spock_sizes_t": ("SpockSizesC", "Sizes")}


def parse_header(path=HEADER):
    """{struct: [(field, kind)]} with kind in {"i32", "f64", "ptr"}."""
    txt = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", txt, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(const\s+)?(int32_t|double)\s+(.*)", decl)
            assert m, decl
            base = "i32" if m.group(2) == "int32_t" else "f64"
            for item in m.group(3).split(","):
                item = item.strip()
                ptr = item.startswith("*")
                fields.append((item.lstrip("* "), "ptr" if ptr else base))
        out[name] = fields
    return out


def parse_julia(path=JULIA):
    txt = open(path).read()
    out = {}
    for name, body in re.findall(r"^struct\s+(\w+)[^\n]*\n(.*?)^end", txt, flags=re.S | re.M):
        fields = []
        for line in body.splitlines():
            line = line.split("#")[0]
            for item in line.split(";"):
                item = item.strip()
                if not item:
                    continue
                f, t = item.split("::")
                kind = {"Cint": "i32", "Cdouble": "f64"}.get(t.strip(), "ptr" if t.strip().startswith("Ptr{") else t.strip())
                fields.append((f.strip(), kind))
        out[name] = fields
    return out


def ctypes_fields(cls):
    kind = {C.c_int32: "i32", C.c_double: "f64", C.c_void_p: "ptr"}
    return [(n, kind[t]) for n, t in cls._fields_]


def test_header_parser_sees_all_four_structs():
    h = parse_header()
    assert set(STRUCTS) <= set(h)
    assert [f for f, _ in h["spock_problem_t"]][-1] == "n_rms" and len(h["spock_problem_t"]) == 26
    assert ("anc", "ptr") in h["spock_problem_t"] and ("x_min", "f64") in h["spock_problem_t"]


@pytest.mark.parametrize("cname", sorted(STRUCTS))
def test_julia_struct_matches_header(cname):
    h, j = parse_header()[cname], parse_julia()[STRUCTS[cname][0]]
    assert [f for f, _ in j] == [f for f, _ in h], f"{cname}: field names / order differ"
    assert j == h, f"{cname}: field types differ"


@pytest.mark.parametrize("cname", sorted(STRUCTS))
def test_ctypes_struct_matches_header(cname):
    from spock_jl_amd import _lib
    assert ctypes_fields(getattr(_lib, STRUCTS[cname][1])) == parse_header()[cname]


def test_round2_julia_struct_would_have_failed():
    """The defect itself: a Julia problem struct without the trailing n_rms is caught by the comparison."""
    h = parse_header()["spock_problem_t"]
    stale = [f for f in parse_julia()["SpockProblemC"] if f[0] != "n_rms"]
    assert stale != h and len(stale) == len(h) - 1


def test_sizes_reported_by_the_library(S):
    """ctypes.sizeof of the mirrors == sizeof in the compiled library (both precisions); the Julia binding asserts the same at
    load (gpu_check_layout)."""
    import __graft_entry__ as g
    g.build()
    from spock_jl_amd import _lib
    for path in (None, _lib.F32_LIB):
        lib = S.load_library(path)
        sz = (C.c_int32 * 4)()
        assert lib.spock_abi_struct_sizes(sz, 4) == 0
        assert list(sz) == [C.sizeof(_lib.Problem), C.sizeof(_lib.Options), C.sizeof(_lib.Result), C.sizeof(_lib.Sizes)]
    # natural alignment of the header's field list gives the same numbers (what Julia's isbits struct layout does too)
    def natural(fields):
        off = 0
        for _, k in fields:
            a = 4 if k == "i32" else 8
            off = (off + a - 1) // a * a + a
        return (off + 7) // 8 * 8 if any(k != "i32" for _, k in fields) else off
    h = parse_header()
    assert [natural(h[c]) for c in ("spock_problem_t", "spock_options_t", "spock_result_t", "spock_sizes_t")] == list(sz)
    assert "gpu_check_layout()" in open(JULIA).read() and "spock_abi_struct_sizes" in open(JULIA).read()


def test_julia_binding_passes_per_node_risk_measures():
    """src/risk_measures.jl:127-136: one risk measure per non-leaf node -> n_rms = length(rms) with E / F / b stacked."""
    txt = open(JULIA).read()
    assert "Cint(n_rms))" in txt and "n_rms = uniform_rms ? 1 : length(rms)" in txt
    assert "non-uniform risk measures: not supported" not in txt
