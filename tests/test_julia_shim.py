"""The Julia shim (nmf.jl_amd/julia/NMFX.jl) cannot be parsed by a Julia here (the image has none), so what CAN be checked without one
is checked here, against include/nmfx.h:

* `COpts` / `CResult` are field for field (order, width, name) the C structs `nmfx_opts` / `nmfx_result`, and every positional
  constructor call passes exactly that many values;
* every `ccall((:sym, libnmfx), Ret, (ArgTypes...), args...)` names a function the header declares, with the header's return type, the
  header's argument count, an argument-type tuple whose entries are ABI-compatible with the C parameter types, and as many actual
  arguments as the tuple has entries;
* the file's block structure closes (every `module / struct / function / if / for / try / do / let / begin` has its `end`);
* the zero-edit drop-in is there: `DeviceMatrix{T} <: AbstractMatrix{T}` and one `NMF.solve!` method per iterative algorithm whose X
  argument is typed on it (one per type -- a Union would be ambiguous with the reference's methods, src/multupd.jl:45 etc.);
* the test file a maintainer with Julia runs (julia/test/runtests.jl) exists and goes through the wrapper.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "nmf.jl_amd", "julia", "NMFX.jl")
HDR = os.path.join(ROOT, "include", "nmfx.h")


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def _strip_jl_comments(s):
    out = []
    for line in s.splitlines():
        # (no '#' inside string literals of this file except in docstrings / comments; keep it simple and check that assumption)
        i, in_str = 0, False
        while i < len(line):
            c = line[i]
            if c == '"' and (i == 0 or line[i - 1] != "\\"):
                in_str = not in_str
            if c == "#" and not in_str:
                break
            i += 1
        out.append(line[:i])
    return "\n".join(out)


def c_struct_fields(hdr, name):
    end = re.search(r"\}\s*" + name + r"\s*;", hdr)
    assert end, name
    beg = hdr.rfind("typedef struct", 0, end.start())
    body = hdr[hdr.index("{", beg) + 1:end.start()]
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        mm = re.match(r"(int32_t|int64_t|uint64_t|double|float|int)\s+(.*)$", decl, flags=re.S)
        assert mm, decl
        for nm in mm.group(2).split(","):
            fields.append((mm.group(1), nm.strip()))
    return fields


def jl_struct_fields(jl, name):
    m = re.search(r"^struct " + name + r"\n(.*?)^end", jl, flags=re.S | re.M)
    assert m, name
    fields = []
    for line in m.group(1).splitlines():
        line = line.strip()
        if not line:
            continue
        mm = re.match(r"(\w+)::(\w+)$", line)
        assert mm, line
        fields.append((mm.group(2), mm.group(1)))
    return fields


JL2C = {"Int32": "int32_t", "Int64": "int64_t", "UInt64": "uint64_t", "Float64": "double", "Float32": "float", "Cint": "int"}


def split_top(s, sep=","):
    """split at top-level separators (outside (), [], {} and string literals)"""
    parts, depth, cur, in_str = [], 0, [], False
    for i, c in enumerate(s):
        if c == '"' and (i == 0 or s[i - 1] != "\\"):
            in_str = not in_str
        if not in_str:
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            elif c == sep and depth == 0:
                parts.append("".join(cur).strip())
                cur = []
                continue
        cur.append(c)
    tail = "".join(cur).strip()
    if tail:
        parts.append(tail)
    return parts


def balanced(s, start):
    """s[start] == '(' -> index just past its matching ')'"""
    depth, in_str = 0, False
    for i in range(start, len(s)):
        c = s[i]
        if c == '"' and s[i - 1] != "\\":
            in_str = not in_str
        if in_str:
            continue
        if c == "(":
            depth += 1
        elif c == ")":
            depth -= 1
            if depth == 0:
                return i + 1
    raise AssertionError("unbalanced parentheses")


def c_prototypes(hdr):
    protos = {}
    for m in re.finditer(r"^(int|void|const char \*)\s*(nmfx_\w+)\s*\(", hdr, flags=re.M):
        end = balanced(hdr, m.end() - 1)
        args = hdr[m.end():end - 1].strip()
        params = [] if args in ("", "void") else [a.strip() for a in split_top(args)]
        protos[m.group(2)] = (m.group(1).strip(), params)
    return protos


def test_structs_match_the_header():
    hdr = _strip_c_comments(open(HDR).read())
    jl = _strip_jl_comments(open(JL).read())
    for cname, jname in (("nmfx_opts", "COpts"), ("nmfx_result", "CResult")):
        cf, jf = c_struct_fields(hdr, cname), jl_struct_fields(jl, jname)
        assert len(cf) == len(jf), (cname, len(cf), len(jf))
        for (ct, cn), (jt, jn) in zip(cf, jf):
            assert JL2C[jt] == ct, (cname, cn, ct, jn, jt)
            assert cn == jn, (cname, cn, jn)
        # every positional constructor call passes one value per field
        for m in re.finditer(r"(?<![\w{.])" + jname + r"\(", jl):
            call = jl[m.end() - 1:balanced(jl, m.end() - 1)]
            assert len(split_top(call[1:-1])) == len(cf), (jname, call[:80])


def _arg_compatible(jl_t, c_t):
    c_t = c_t.strip()
    is_ptr = "*" in c_t
    if jl_t.startswith(("Ptr{", "Ref{")) or jl_t == "Cstring":
        return is_ptr
    if is_ptr:
        return False
    base = re.sub(r"\b(const)\b", "", c_t).split()
    ctype = " ".join(base[:-1]) if len(base) > 1 else base[0]
    want = {"Cint": ("int", "int32_t"), "Int32": ("int", "int32_t"), "Int64": ("int64_t", "long long"), "UInt64": ("uint64_t",),
            "Float64": ("double",)}.get(jl_t)
    return want is not None and ctype in want


def test_every_ccall_matches_its_prototype():
    hdr = _strip_c_comments(open(HDR).read())
    jl = _strip_jl_comments(open(JL).read())
    protos = c_prototypes(hdr)
    assert len(protos) >= 30
    seen = set()
    for m in re.finditer(r"ccall\(", jl):
        call = jl[m.end() - 1:balanced(jl, m.end() - 1)]
        parts = split_top(call[1:-1])
        sym = re.match(r"\(:(\w+),\s*libnmfx\)$", parts[0])
        assert sym, parts[0]
        name = sym.group(1)
        assert name in protos, f"{name} is not declared in include/nmfx.h"
        ret_c, params = protos[name]
        ret_j = parts[1]
        assert {"int": "Cint", "void": "Cvoid", "const char *": "Cstring"}[ret_c] == ret_j, (name, ret_c, ret_j)
        assert parts[2].startswith("(") and parts[2].endswith(")"), (name, parts[2])
        argt = split_top(parts[2][1:-1])
        assert len(argt) == len(params), (name, argt, params)
        for jt, ct in zip(argt, params):
            assert _arg_compatible(jt, ct), (name, jt, ct)
        assert len(parts) - 3 == len(argt), (name, "actual arguments", len(parts) - 3, "tuple", len(argt))
        seen.add(name)
    # the shim reaches every entry point of the solve path and of the front end
    for need in ("nmfx_create", "nmfx_destroy", "nmfx_set_X", "nmfx_solve", "nmfx_get_factors", "nmfx_last_error", "nmfx_get_iter_trace",
                 "nmfx_alspgrad_subsolve", "nmfx_solve_replicates", "nmfx_check_nonneg", "nmfx_randinit", "nmfx_nndsvd", "nmfx_rsvd_begin",
                 "nmfx_rsvd_finish", "nmfx_spa_init", "nmfx_pdsolve", "nmfx_pdrsolve", "nmfx_comm_init_local", "nmfx_comm_p2p_export"):
        assert need in seen, need


def test_blocks_close():
    jl = _strip_jl_comments(open(JL).read())
    jl = re.sub(r'""".*?"""', "", jl, flags=re.S)
    jl = re.sub(r'"(?:\\.|[^"\\])*"', '""', jl)
    depth = 0
    for ln, line in enumerate(jl.splitlines(), 1):
        s = line.strip()
        if not s:
            continue
        toks = re.findall(r"[A-Za-z_]\w*|\[|\]", s)
        opens, bracket = 0, 0
        for i, t in enumerate(toks):
            if t == "[":
                bracket += 1
            elif t == "]":
                bracket -= 1
            elif bracket == 0:
                if t in ("module", "function", "for", "while", "try", "let", "begin", "do", "macro", "quote"):
                    opens += 1
                elif t == "struct":      # `mutable struct` counts once
                    opens += 1
                elif t == "if" and (i == 0 or toks[i - 1] not in ("elseif",)):
                    opens += 1
                elif t == "end":
                    opens -= 1
        depth += opens
        assert depth >= 0, f"NMFX.jl:{ln}: more `end` than blocks"
    assert depth == 0, f"NMFX.jl: {depth} block(s) left open"


def test_zero_edit_drop_in_is_there():
    jl = _strip_jl_comments(open(JL).read())
    assert re.search(r"mutable struct DeviceMatrix\{T<:Union\{Float32,Float64\}\} <: AbstractMatrix\{T\}", jl)
    for need in ("Base.size(A::DeviceMatrix)", "Base.getindex(A::DeviceMatrix, i::Int)", "Base.IndexStyle(::Type{<:DeviceMatrix})"):
        assert need in jl, need
    for alg in ("MultUpdate", "ProjectedALS", "ALSPGrad", "CoordinateDescent", "GreedyCD"):
        assert re.search(r"^NMF\.solve!\(alg::NMF\." + alg + r"\{T\}, X::DeviceMatrix\{T\}, W::Matrix\{T\}, H::Matrix\{T\}\) where T", jl, flags=re.M), alg
    assert not re.search(r"NMF\.solve!\(alg::Union", jl)          # ambiguous with the reference's own methods
    for fn in ("NMF.nndsvd(X::DeviceMatrix{T}", "NMF.spa(X::DeviceMatrix{T}", "NMF.alspgrad_updateh!(X::DeviceMatrix{T}", "NMF.alspgrad_updatew!(X::DeviceMatrix{T}"):
        assert fn in jl, fn
    rt = open(os.path.join(ROOT, "nmf.jl_amd", "julia", "test", "runtests.jl")).read()
    assert "NMFX.DeviceMatrix" in rt and "NMF.nnmf(Xd, k" in rt and "NMF.solve!(NMF.MultUpdate{T}" in rt and "alspgrad_updateh!(Xd" in rt
