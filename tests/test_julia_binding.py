"""Static verification of the Julia boundary -- no Julia needed, none is in this image.

julia/MI355X.jl and the Julia code blocks of INTEGRATION.md are the `ccall` binding a MetaFEM.jl maintainer adds; one wrong `Cint` / `Int64` in
an argument tuple, or one struct mirror whose field order drifts from the header, is a silent stack / memory corruption on first use.  This test
parses, textually,
  * every `ccall((:name, lib), Ret, (ArgTypes...), args...)`: the symbol must be declared in include/metafem_mi355x.h (or _debug.h), the tuple must
    have the declaration's arity, every Julia type must be one the C parameter type admits, the call must pass as many values as the tuple has
    types, and a `Ref{Struct}` / `Ptr{Struct}` must name the mirror of the C struct the parameter points to;
  * every `struct` mirror (`# == mfem_xxx` names its C struct): field names, order and types against the C struct, and -- compiled with gcc from the
    header itself -- `offsetof` of every field and `sizeof` against the layout Julia gives an isbits struct (natural alignment);
  * the ABI_VERSION constant against MFEM_ABI_VERSION.
Interfaces mirrored: solver/01_Types.jl:164-166 (the three Function fields), misc/04_GPU_Utils.jl:1-38 (array backend)."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", "metafem_mi355x.h"), os.path.join(ROOT, "include", "metafem_mi355x_debug.h")]

# C parameter type (normalised) -> Julia types an argument tuple may use for it
SCALARS = {
    "int": {"Cint", "Int32"}, "int32_t": {"Int32", "Cint"}, "int64_t": {"Int64", "Clonglong"}, "uint32_t": {"UInt32", "Cuint"},
    "uint64_t": {"UInt64", "Culonglong"}, "double": {"Float64", "Cdouble"}, "long long": {"Int64", "Clonglong"},
    "unsigned long long": {"UInt64", "Culonglong"}, "size_t": {"Csize_t", "UInt64"},
}
HANDLES = {"mfem_context", "mfem_csr", "mfem_brick", "mfem_mesh", "mfem_comm"}  # typedef'd struct pointers
JL_ELEM = {"double": "Float64", "int32_t": "Int32", "int64_t": "Int64", "uint16_t": "UInt16", "uint8_t": "UInt8", "int": "Cint", "uint32_t": "UInt32",
           "uint64_t": "UInt64"}
JL_SIZE = {"Int32": 4, "Cint": 4, "UInt32": 4, "Int64": 8, "UInt64": 8, "Float64": 8, "Cdouble": 8, "UInt16": 2, "UInt8": 1}


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def parse_header():
    """-> (functions: name -> (ret, [param types]), structs: name -> [(field, ctype, array_len)])"""
    funcs, structs = {}, {}
    for path in HEADERS:
        text = _strip_c_comments(open(path).read())
        text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
        for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
            fields = []
            for decl in m.group(1).split(";"):
                decl = " ".join(decl.split())
                if not decl:
                    continue
                fp = re.match(r"(\w+)\s*\(\*\s*(\w+)\)\s*\(", decl)  # function pointer member
                if fp:
                    fields.append((fp.group(2), "fnptr", 0))
                    continue
                mm = re.match(r"(.*?)(\**)\s*(\w+(?:\s*,\s*\w+)*)(?:\s*\[(\d+)\])?$", decl)
                base, stars, names, arr = mm.group(1).strip(), mm.group(2), mm.group(3), mm.group(4)
                for nm in [x.strip() for x in names.split(",")]:
                    fields.append((nm, (base + stars).replace("const ", "").strip(), int(arr) if arr else 0))
            structs[m.group(2)] = fields
        body = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
        body = re.sub(r"typedef\s+enum\s*\{.*?\}\s*\w+\s*;", " ", body, flags=re.S)
        body = re.sub(r"typedef\s+struct\s+\w+\s*\*\s*\w+\s*;", " ", body)
        body = body.replace('extern "C" {', " ")
        for m in re.finditer(r"([\w\s\*]+?)\b(mfem_\w+)\s*\(([^;{}]*?)\)\s*;", body, flags=re.S):
            ret = " ".join(m.group(1).split())
            params = " ".join(m.group(3).split())
            plist = []
            if params and params != "void":
                for prm in params.split(","):
                    prm = prm.strip()
                    mm = re.match(r"(.*?)(\w+)$", prm)  # drop the parameter name
                    typ = mm.group(1).strip() if mm and mm.group(1).strip() else prm
                    plist.append(" ".join(typ.replace(" *", "*").split()))
            funcs[m.group(2)] = (ret, plist)
    return funcs, structs


def _balanced(text, start):
    """index just after the parenthesis group opening at text[start] == '('"""
    depth = 0
    for i in range(start, len(text)):
        if text[i] == "(":
            depth += 1
        elif text[i] == ")":
            depth -= 1
            if depth == 0:
                return i + 1
    raise AssertionError("unbalanced parentheses")


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def julia_sources():
    src = {"julia/MI355X.jl": open(os.path.join(ROOT, "julia", "MI355X.jl")).read()}
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```julia\n(.*?)```", md, flags=re.S)
    src["INTEGRATION.md"] = "\n".join(blocks)
    ex = os.path.join(ROOT, "julia", "examples")
    if os.path.isdir(ex):
        for f in sorted(os.listdir(ex)):
            if f.endswith(".jl"):
                src["julia/examples/" + f] = open(os.path.join(ex, f)).read()
    return src


def _strip_jl_comments(text):
    return "\n".join(re.sub(r"#(?![^\"]*\"[^\"]*$).*$", "", ln) if '"' not in ln else ln.split(" #")[0] for ln in text.splitlines())


def parse_ccalls(text):
    text = _strip_jl_comments(text)
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*lib\)", text):
        end = _balanced(text, m.start() + len("ccall"))
        inner = text[m.start() + len("ccall("):end - 1]
        parts = _split_top(inner)
        ret, tup, args = parts[1], parts[2], parts[3:]
        assert tup.startswith("(") and tup.endswith(")"), (m.group(1), tup)
        types = _split_top(tup[1:-1])
        calls.append((m.group(1), ret, types, args))
    return calls


def parse_jl_structs(text):
    out = {}
    for m in re.finditer(r"^(?:Base\.@kwdef\s+)?(?:mutable\s+)?struct\s+(\w+)(?![^\n]*;\s*end)[^\n]*?#\s*==\s*(mfem_\w+)[^\n]*\n(.*?)^end", text, flags=re.S | re.M):
        fields = []
        for line in m.group(3).splitlines():
            for ln in line.split("#")[0].split(";"):  # (several fields per line, separated by semicolons)
                ln = ln.strip()
                if not ln:
                    continue
                mm = re.match(r"(\w+)::([\w\{\}, ]+?)(?:\s*=.*)?$", ln)
                assert mm, (m.group(1), ln)
                fields.append((mm.group(1), mm.group(2).strip()))
        out[m.group(1)] = (m.group(2), fields)
    # one-line form used in INTEGRATION.md: struct Name; a::T; b::T; end   # == mfem_xxx
    for m in re.finditer(r"^struct\s+(\w+);(.*?);\s*end\s*#\s*==\s*(mfem_\w+)", text, flags=re.M):
        fields = [tuple(x.strip().split("::")) for x in m.group(2).split(";") if x.strip()]
        out[m.group(1)] = (m.group(3), fields)
    return out


def _jl_allowed(ctype, jl, mirrors):
    """Is Julia type `jl` acceptable for the C parameter type `ctype`?"""
    c = ctype.replace("const ", "").strip()
    if c in SCALARS:
        return jl in SCALARS[c]
    if c in HANDLES:
        return jl == "Ptr{Cvoid}"
    if c.endswith("*"):
        base = c[:-1].strip()
        if base in HANDLES:                       # out-handle
            return jl in ("Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}")
        if base == "void":
            return jl in ("Ptr{Cvoid}", "Ptr{UInt8}")  # (the 128-byte RCCL id is a byte buffer)
        if base == "char":
            return jl == "Cstring"
        if base.startswith("mfem_"):              # pointer to a struct of the header: must name ITS mirror
            mm = re.match(r"(?:Ref|Ptr)\{(\w+)\}$", jl)
            return bool(mm) and mm.group(1) in mirrors and mirrors[mm.group(1)][0] == base
        if base in JL_ELEM:                       # typed array / out-scalar: untyped device pointer, or a pointer / reference of the right element type
            return jl == "Ptr{Cvoid}" or jl in (f"Ptr{{{JL_ELEM[base]}}}", f"Ref{{{JL_ELEM[base]}}}") or \
                (base == "int" and jl in ("Ptr{Int32}", "Ref{Int32}"))
    return False


def test_every_ccall_matches_the_header():
    funcs, cstructs = parse_header()
    assert len(funcs) > 80 and "mfem_solve" in funcs and "mfem_brick_assemble_thermal" in funcs
    total = 0
    for fname, text in julia_sources().items():
        mirrors = parse_jl_structs(text)
        if fname != "julia/MI355X.jl":  # code blocks and examples use the module's mirrors as well
            mirrors = {**parse_jl_structs(open(os.path.join(ROOT, "julia", "MI355X.jl")).read()), **mirrors}
        for name, ret, types, args in parse_ccalls(text):
            where = f"{fname}: ccall :{name}"
            assert name in funcs, f"{where}: not declared in include/*.h"
            cret, cparams = funcs[name]
            assert len(types) == len(cparams), f"{where}: {len(types)} argument types, the header declares {len(cparams)} parameters {cparams}"
            assert len(args) == len(types), f"{where}: {len(args)} values passed for {len(types)} argument types"
            if cret == "const char*":
                assert ret == "Cstring", f"{where}: returns {cret}, bound as {ret}"
            else:
                assert ret in SCALARS.get(cret, set()) or (cret.endswith("*") and ret.startswith("Ptr{")), f"{where}: returns {cret}, bound as {ret}"
            for i, (ct, jt) in enumerate(zip(cparams, types)):
                assert _jl_allowed(ct, jt, mirrors), f"{where}: parameter {i + 1} is `{ct}` in the header, `{jt}` in the binding"
            total += 1
    assert total >= 40, total  # (31 in the module at the end of round 4, the rest in INTEGRATION.md)


def _c_layout(cstructs):
    """offsetof / sizeof of every header struct, from gcc on the header itself."""
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "metafem_mi355x.h"', "int main(void) {"]
    for sname, fields in cstructs.items():
        lines.append(f'  printf("S {sname} %zu\\n", sizeof({sname}));')
        for fn, _, _ in fields:
            lines.append(f'  printf("F {sname} {fn} %zu\\n", offsetof({sname}, {fn}));')
    lines += ["  return 0;", "}"]
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "layout.c"), os.path.join(td, "layout")
        open(src, "w").write("\n".join(lines))
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = subprocess.check_output([exe], text=True)
    size, off = {}, {}
    for ln in out.splitlines():
        p = ln.split()
        if p[0] == "S":
            size[p[1]] = int(p[2])
        else:
            off[(p[1], p[2])] = int(p[3])
    return size, off


def _jl_field(jt):
    """(size, alignment) of an isbits Julia field type"""
    if jt in JL_SIZE:
        return JL_SIZE[jt], JL_SIZE[jt]
    if jt.startswith("Ptr{") or jt == "Ptr":
        return 8, 8
    m = re.match(r"NTuple\{(\d+),\s*(\w+)\}$", jt)
    if m:
        return int(m.group(1)) * JL_SIZE[m.group(2)], JL_SIZE[m.group(2)]
    raise AssertionError(f"unknown Julia field type {jt}")


def test_struct_mirrors_have_the_c_layout():
    funcs, cstructs = parse_header()
    hdr_only = {k: v for k, v in cstructs.items() if k in open(HEADERS[0]).read()}
    size, off = _c_layout(hdr_only)
    seen = set()
    for fname, text in julia_sources().items():
        for jname, (cname, jfields) in parse_jl_structs(text).items():
            where = f"{fname}: struct {jname} (== {cname})"
            assert cname in cstructs, f"{where}: no such struct in the header"
            cfields = cstructs[cname]
            assert [f for f, _ in jfields] == [f for f, _, _ in cfields], f"{where}: fields {[f for f, _ in jfields]} != header's {[f for f, _, _ in cfields]}"
            o, maxal = 0, 1
            for (fn, jt), (_, ct, arr) in zip(jfields, cfields):
                sz, al = _jl_field(jt)
                o = (o + al - 1) // al * al
                assert o == off[(cname, fn)], f"{where}: field {fn} at offset {o} in Julia, {off[(cname, fn)]} in C"
                # element type
                if ct.endswith("*") or ct == "fnptr":
                    assert jt.startswith("Ptr"), f"{where}: field {fn} is a pointer in C, `{jt}` in Julia"
                elif arr:
                    assert jt == f"NTuple{{{arr}, {JL_ELEM[ct]}}}", f"{where}: field {fn} is {ct}[{arr}], `{jt}` in Julia"
                else:
                    assert jt in SCALARS[ct], f"{where}: field {fn} is {ct} in C, `{jt}` in Julia"
                o += sz
                maxal = max(maxal, al)
            o = (o + maxal - 1) // maxal * maxal
            assert o == size[cname], f"{where}: sizeof {o} in Julia, {size[cname]} in C"
            seen.add(cname)
    # every struct that crosses the boundary by pointer in a bound call has a mirror
    for need in ("mfem_solve_options", "mfem_solve_stats", "mfem_thermal_params", "mfem_elasticity_params", "mfem_op_layout", "mfem_kval_term",
                 "mfem_res_term", "mfem_var_term", "mfem_const_term"):
        assert need in seen, f"no Julia mirror of {need}"


def test_abi_version_constant():
    hdr = open(HEADERS[0]).read()
    ver = int(re.search(r"#define MFEM_ABI_VERSION (\d+)", hdr).group(1))
    jl = open(os.path.join(ROOT, "julia", "MI355X.jl")).read()
    assert int(re.search(r"const ABI_VERSION = (\d+)", jl).group(1)) == ver


def test_the_checker_catches_a_wrong_binding():
    """The parser is not vacuous: a swapped Cint / Int64 and a reordered struct are reported."""
    funcs, cstructs = parse_header()
    mirrors = parse_jl_structs(open(os.path.join(ROOT, "julia", "MI355X.jl")).read())
    assert _jl_allowed("int64_t", "Int64", mirrors) and not _jl_allowed("int64_t", "Cint", mirrors)
    assert _jl_allowed("int32_t", "Int32", mirrors) and not _jl_allowed("int32_t", "Int64", mirrors)
    assert _jl_allowed("const mfem_solve_options*", "Ref{SolveOptions}", mirrors) and not _jl_allowed("const mfem_solve_options*", "Ref{SolveStats}", mirrors)
    assert not _jl_allowed("double", "Ptr{Cvoid}", mirrors) and not _jl_allowed("const double*", "Float64", mirrors)
    bad = "x = ccall((:mfem_dot, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), c, n, a, b, out)"
    (name, ret, types, args), = parse_ccalls(bad)
    assert name == "mfem_dot" and not _jl_allowed(funcs[name][1][1], types[1], mirrors)
