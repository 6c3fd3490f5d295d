"""include/starphase_hip.h -> include/starphase_hip.rs: the complete `extern "C"` block a Rust host binds (INTEGRATION.md shows the call sites; this file is
the mechanical part: every exported function, structs as opaque or `#[repr(C)]` names).  usage: rust_externs.py [--check]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = os.path.join(ROOT, "include", "starphase_hip.h")
OUT = os.path.join(ROOT, "include", "starphase_hip.rs")
SCALAR = {"int8_t": "i8", "uint8_t": "u8", "int16_t": "i16", "uint16_t": "u16", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64",
          "double": "f64", "float": "f32", "char": "c_char", "void": "c_void", "int": "c_int", "size_t": "usize"}


def rust_type(c):
    c = " ".join(c.replace("*", " * ").split())
    toks = c.split()
    # strip a leading const of the pointee, remember it
    out, ptrs = None, []
    base, i = [], 0
    while i < len(toks) and toks[i] != "*":
        base.append(toks[i]); i += 1
    const_base = "const" in base
    base = [t for t in base if t not in ("const", "struct", "unsigned")]
    name = base[-1] if base else "c_void"
    ty = SCALAR.get(name, name)
    rest = toks[i:]
    # every '*' (optionally followed by const) adds a pointer level
    levels = []
    j = 0
    while j < len(rest):
        if rest[j] == "*":
            is_const = j + 1 < len(rest) and rest[j + 1] == "const"
            levels.append(is_const); j += 2 if is_const else 1
        else:
            j += 1
    for n, lvl_const in enumerate(levels):
        pointee_const = const_base if n == 0 else levels[n - 1]
        ty = ("*const " if pointee_const else "*mut ") + ty
    return ty


def declarations(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    for m in re.finditer(r"^\s*([A-Za-z_][A-Za-z0-9_ \*]*?)\s*\b(sp_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef") or "(" in ret:
            continue
        yield ret, name, args


def constants(text):
    out = []
    for name, val in re.findall(r"#define\s+(SP_[A-Z0-9_]+)\s+(\S+)", text):
        v = val.rstrip("uU")
        if re.fullmatch(r"-?\d+", v):
            out.append((name, int(v)))
        elif val == "INT32_MIN":
            out.append((name, -2147483648))
    return out


def struct_fields(body):
    """`int32_t a, b[4]; const char* const* c;` -> [(name, rust type)]"""
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        first = decl.split(",")[0]
        m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)((?:\[[^\]]+\])*)$", first.strip())
        ctype = m.group(1).strip()
        for k, part in enumerate(decl.split(",")):
            part = part.strip()
            if k == 0:
                nm, dims = m.group(2), m.group(3)
            else:
                mm = re.match(r"^(\**)\s*([A-Za-z_][A-Za-z0-9_]*)((?:\[[^\]]+\])*)$", part)
                nm, dims = mm.group(2), mm.group(3)
            ty = rust_type(ctype)
            for d in reversed(re.findall(r"\[([^\]]+)\]", dims)):
                ty = f"[{ty}; {d} as usize]" if not d.isdigit() else f"[{ty}; {d}]"
            if nm in ("type", "ref", "in", "match", "mod", "box", "move", "self", "fn", "use", "where", "loop", "impl"):
                nm += "_"
            fields.append((nm, ty))
    return fields


def main():
    text = open(H).read()
    clean = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    clean = re.sub(r"//[^\n]*", " ", clean)
    lines = ["// Generated from include/starphase_hip.h by profiles/scripts/rust_externs.py -- do not edit (tests/test_abi.py checks it is up to date).",
             "// Handles are opaque; the structs a host fills or reads are `#[repr(C)]` mirrors of the header's, field for field.",
             "#![allow(non_camel_case_types, non_snake_case, dead_code)]", "use std::os::raw::{c_char, c_int, c_void};", ""]
    for name, val in constants(text):
        lines.append(f"pub const {name}: i64 = {val};")
    lines.append("")
    defined = set()
    for tag, body, name in re.findall(r"typedef struct (\w+)?\s*\{(.*?)\}\s*(\w+);", clean, flags=re.S):
        lines.append("#[repr(C)]")
        lines.append(f"pub struct {name} {{")
        for fn, ty in struct_fields(body):
            lines.append(f"    pub {fn}: {ty},")
        lines.append("}")
        defined.add(name)
    opaque = sorted({b for a, b in re.findall(r"typedef struct (sp_[a-z0-9_]+)\s+(sp_[a-z0-9_]+);", clean)} - defined)
    lines.append("")
    for st in opaque:
        lines.append(f"#[repr(C)] pub struct {st} {{ _private: [u8; 0] }}")
    lines += ["", '#[link(name = "starphase_hip")]', 'extern "C" {']
    names = []
    for ret, name, args in declarations(text):
        params = []
        if args and args != "void":
            for k, a in enumerate(args.split(",")):
                a = a.strip()
                m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
                ctype, pname = (m.group(1).strip(), m.group(2)) if m and m.group(1).strip() else (a, f"arg{k}")
                if pname in ("type", "ref", "in", "match", "mod", "box", "move", "self", "fn", "use", "where", "loop", "impl"):
                    pname += "_"
                params.append(f"{pname}: {rust_type(ctype)}")
        r = "" if ret == "void" else f" -> {rust_type(ret)}"
        lines.append(f"    pub fn {name}({', '.join(params)}){r};")
        names.append(name)
    lines.append("}")
    out = "\n".join(lines) + "\n"
    known = set(re.findall(r"pub struct (sp_\w+)", out))
    used = set(re.findall(r"(?:\*const |\*mut |: |\[|-> )(sp_[a-z0-9_]+)", out))
    assert used <= known, f"types without a definition: {sorted(used - known)}"
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == out
        print("up to date" if ok else "STALE: run profiles/scripts/rust_externs.py")
        sys.exit(0 if ok else 1)
    open(OUT, "w").write(out)
    print(f"{len(names)} functions, {len(defined)} structs with fields, {len(opaque)} opaque handles -> {OUT}")


if __name__ == "__main__":
    main()
