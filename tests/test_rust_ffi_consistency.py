"""The Rust crate cannot be compiled in this image (no rustc), so its raw bindings are checked as text:
every `extern "C"` declaration, `#[repr(C)]` struct and `RGR_*` constant of rust/rmqtt-gpu-router/src/ffi.rs
must agree with include/rmqtt_gpu_router.h — same function names, arity, parameter and return types
(through the obvious C -> Rust type map), same struct fields in the same order, same constant values.
A drift here is exactly the bug class a missing compiler would otherwise hide."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rmqtt_gpu_router.h")
FFI = os.path.join(ROOT, "rust", "rmqtt-gpu-router", "src", "ffi.rs")

SCALAR = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "uint16_t": "u16", "uint8_t": "u8", "char": "c_char", "void": "c_void",
          "double": "f64", "float": "f32", "int": "i32", "size_t": "usize"}


def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def c_type_to_rust(t):
    """'const uint8_t*' -> '*const u8', 'rgr_handle**' -> '*mut *mut rgr_handle', 'uint32_t' -> 'u32'"""
    t = t.strip()
    stars = t.count("*")
    base = t.replace("*", " ").split()
    const = "const" in base
    base = [w for w in base if w not in ("const", "struct")]
    assert len(base) == 1, t
    r = SCALAR.get(base[0], base[0])
    for k in range(stars):
        # in the header only the innermost pointee is ever const-qualified (const T* / const T* const* do not occur beyond one level)
        r = ("*const " if const and k == 0 else "*mut ") + r
    return r


def split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [p.strip() for p in out]


def parse_header():
    raw = open(HEADER).read()
    src = strip_comments(raw)
    src = re.sub(r"^\s*#[^\n]*(?:\\\n[^\n]*)*", " ", src, flags=re.M)            # preprocessor lines
    funcs, structs, consts, fnptr = {}, {}, {}, {}
    for m in re.finditer(r"typedef\s+[\w\s\*]*?\(\s*\*\s*(\w+)\s*\)\s*\([^;]*\)\s*;", src):
        fnptr[m.group(1)] = True
    for m in re.finditer(r"typedef\s+struct\s*\w*\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            arr = re.match(r"(.*?)(\w+)\s*\[\s*(\w+)\s*\]$", decl)
            if arr:
                fields.append((arr.group(2), f"[{c_type_to_rust(arr.group(1))}; {arr.group(3)}]"))
                continue
            first, *rest = [d.strip() for d in decl.split(",")]          # "uint32_t a, b" declares several fields of one type
            name = re.search(r"(\w+)$", first).group(1)
            base = first[: -len(name)]
            fields.append((name, c_type_to_rust(base)))
            base_type = base.replace("*", " ")
            for d in rest:
                fields.append((d.replace("*", "").strip(), c_type_to_rust(base_type + "*" * d.count("*"))))
        structs[m.group(2)] = fields
    body = re.sub(r"typedef\s+struct[^;{]*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    body = re.sub(r"typedef[^;]*;", " ", body)
    body = re.sub(r'extern\s+"C"\s*\{', " ", body)
    for m in re.finditer(r"([\w\s\*]+?)\b(rgr_\w+)\s*\(([^;{]*?)\)\s*;", body, flags=re.S):
        ret, name, params = " ".join(m.group(1).split()), m.group(2), m.group(3).strip()
        ret = ret.split("}")[-1].strip()
        ptypes = []
        if params and params != "void":
            for p in split_params(params):
                p = " ".join(p.split())
                pm = re.match(r"(.*?)(\w+)$", p)
                ptypes.append(c_type_to_rust(pm.group(1)) if pm.group(1).strip() not in ("", "const") else c_type_to_rust(p))
        funcs[name] = (None if ret == "void" else c_type_to_rust(ret), ptypes)
    for m in re.finditer(r"^\s*#\s*define\s+(RGR_\w+)\s+\(?\s*(-?(?:0x[0-9A-Fa-f]+|\d+))[uU]?[lL]*\s*(?:<<\s*(\d+))?\s*\)?\s*(?:/\*.*)?$", raw, flags=re.M):
        v = int(m.group(2), 0)
        consts[m.group(1)] = v << int(m.group(3)) if m.group(3) else v
    for m in re.finditer(r"\b(RGR_\w+)\s*=\s*(-?(?:0x[0-9A-Fa-f]+|\d+))[uU]?\s*(?:<<\s*(\d+))?", src):     # enumerators
        v = int(m.group(2), 0)
        consts.setdefault(m.group(1), v << int(m.group(3)) if m.group(3) else v)
    return funcs, structs, consts, fnptr


def parse_rust():
    src = strip_comments(open(FFI).read())
    funcs, structs, consts = {}, {}, {}
    for m in re.finditer(r"pub fn (rgr_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", src, flags=re.S):
        params = [p.split(":", 1)[1].strip() for p in split_params(m.group(2)) if p.strip()]
        funcs[m.group(1)] = (m.group(3).strip() if m.group(3) else None, [re.sub(r"\s+", " ", p) for p in params])
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[derive\([^\)]*\)\])?\s*pub struct (\w+)\s*\{(.*?)\}", src, flags=re.S):
        fields = []
        for f in split_params(m.group(2)):
            if not f.strip():
                continue
            name, ty = f.split(":", 1)
            fields.append((name.replace("pub", "").strip(), re.sub(r"\s+", " ", ty.strip())))
        structs[m.group(1)] = fields
    for m in re.finditer(r"pub const (RGR_\w+)\s*:\s*\w+\s*=\s*([^;]+);", src):
        consts[m.group(1)] = int(eval(m.group(2).replace("_", ""), {"__builtins__": {}}))       # "1 << 2", "0xFFFF_FFFF", "-2"
    return funcs, structs, consts


def test_every_rust_extern_matches_the_header():
    hf, _, _, fnptr = parse_header()
    rf, _, _ = parse_rust()
    assert len(rf) >= 40
    problems = []
    for name, (ret, params) in rf.items():
        if name not in hf:
            problems.append(f"{name}: declared in ffi.rs, absent from the header")
            continue
        hret, hparams = hf[name]
        if (ret or None) != hret:
            problems.append(f"{name}: returns {ret!r} in Rust, {hret!r} in C")
        if len(params) != len(hparams):
            problems.append(f"{name}: {len(params)} parameters in Rust, {len(hparams)} in C")
            continue
        for i, (a, b) in enumerate(zip(params, hparams)):
            if b in fnptr:                       # callbacks are spelled out as Option<unsafe extern "C" fn(..)> in Rust
                assert "fn(" in a, (name, i, a)
                continue
            if a != b:
                problems.append(f"{name}: parameter {i} is {a!r} in Rust, {b!r} in C")
    assert not problems, "\n".join(problems)


def test_repr_c_structs_match_the_header():
    _, hs, _, _ = parse_header()
    _, rs, _ = parse_rust()
    problems = []
    checked = 0
    for name, fields in rs.items():
        if fields == [("_p", "[u8; 0]")]:        # opaque handles
            continue
        assert name in hs, f"struct {name} of ffi.rs is not in the header"
        checked += 1
        hfields = hs[name]
        if [f[0].lstrip("_") for f in fields] != [f[0].lstrip("_") for f in hfields]:
            problems.append(f"{name}: fields {[f[0] for f in fields]} in Rust, {[f[0] for f in hfields]} in C")
            continue
        for (fn, a), (_, b) in zip(fields, hfields):
            if a.replace("*const", "*mut") != b.replace("*const", "*mut"):      # pointer constness of a field is not ABI
                problems.append(f"{name}.{fn}: {a!r} in Rust, {b!r} in C")
    assert checked >= 6
    assert not problems, "\n".join(problems)


def test_constants_match_the_header():
    _, _, hc, _ = parse_header()
    _, _, rc = parse_rust()
    assert len(rc) >= 15
    for name, v in rc.items():
        assert name in hc, f"{name} of ffi.rs is not in the header"
        assert (hc[name] & 0xFFFFFFFF) == (v & 0xFFFFFFFF), (name, hc[name], v)


def _call_args(src, start):
    """Arguments of the call whose '(' is at src[start]: top-level comma split with (), [], {} and <> balanced."""
    depth, cur, args, i = 0, "", [], start
    while True:
        ch = src[i]
        if ch in "([{":
            depth += 1
            if depth > 1:
                cur += ch
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                break
            cur += ch
        elif ch == "," and depth == 1:
            args.append(cur)
            cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip():
        args.append(cur)
    return args


def test_every_ffi_call_site_passes_the_declared_number_of_arguments():
    """rustc would refuse a call with the wrong arity; here the call sites of the crate are counted against ffi.rs."""
    rf, _, _ = parse_rust()
    src_dir = os.path.dirname(FFI)
    calls = 0
    problems = []
    for fn in sorted(os.listdir(src_dir)):
        if not fn.endswith(".rs") or fn == "ffi.rs":
            continue
        src = strip_comments(open(os.path.join(src_dir, fn)).read())
        for m in re.finditer(r"\b(rgr_\w+)\s*\(", src):
            name = m.group(1)
            if name not in rf:
                if re.search(r"\b(struct|type)\s+" + name + r"\b", src) or name in ("rgr_config", "rgr_result", "rgr_retain_result", "rgr_publish_attr", "rgr_tuple"):
                    continue
                problems.append(f"{fn}: calls {name}, which ffi.rs does not declare")
                continue
            args = _call_args(src, m.end() - 1)
            calls += 1
            if len(args) != len(rf[name][1]):
                problems.append(f"{fn}: {name} called with {len(args)} arguments, declared with {len(rf[name][1])}")
    assert calls >= 20
    assert not problems, "\n".join(problems)


def test_rust_sources_are_bracket_balanced():
    """The cheapest syntax check available without rustc: (), [] and {} balance in every source file of the crate
    (comments, string and char literals removed)."""
    src_dir = os.path.dirname(FFI)
    pairs = {")": "(", "]": "[", "}": "{"}
    for fn in sorted(os.listdir(src_dir)):
        if not fn.endswith(".rs"):
            continue
        src = strip_comments(open(os.path.join(src_dir, fn)).read())
        src = re.sub(r'r#"(?:.|\n)*?"#', '""', src)                      # raw strings
        src = re.sub(r'"(?:\\.|[^"\\])*"', '""', src)                    # string literals
        src = re.sub(r"'(?:\\.|[^'\\])'", "' '", src)                    # char literals (lifetimes have no closing quote)
        stack = []
        for ln, line in enumerate(src.split("\n"), 1):
            for ch in line:
                if ch in "([{":
                    stack.append((ch, ln))
                elif ch in ")]}":
                    assert stack and stack[-1][0] == pairs[ch], f"{fn}:{ln}: unmatched {ch!r}"
                    stack.pop()
        assert not stack, f"{fn}: {stack[-1][0]!r} opened at line {stack[-1][1]} is never closed"
