"""CPU: the Rust side of the boundary (rust/ndarray-interp-hip/src/hip_ffi.rs) against include/ndinterp.h,
mechanically -- every `#[repr(C)]` struct field by field (order, name, C type), every `extern "C"` function
argument by argument (order, name, C type) and its return type, every enumerator value, and the consumer callback
type.  There is no rustc in the build image, so this (not a compiler) is what keeps the two files in step: swapping
two fields or arguments in either file fails here.  Also: the ctypes binding's structs against the same header."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "ndinterp.h")
RUST = os.path.join(ROOT, "rust", "ndarray-interp-hip", "src", "hip_ffi.rs")

C_BASE = {
    "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "float": "f32",
    "size_t": "usize", "void": "c_void", "char": "c_char", "int": "i32",
    # enums travel as plain 32-bit integers
    "ndi_status": "i32", "ndi_dtype": "i32", "ndi_memspace": "i32", "ndi_strategy1d": "i32", "ndi_bc_kind": "i32",
    "ndi_monotonic": "i32", "ndi_path": "i32",
}


def c_to_rust(ctype: str) -> str:
    """Canonical Rust spelling of a C type (pointers with their constness, fixed-width integers, opaque structs)."""
    tokens = re.findall(r"[A-Za-z_][A-Za-z0-9_]*|\*", ctype)
    tokens = [t for t in tokens if t != "struct"]
    i, base_const = 0, False
    if tokens[i] == "const":
        base_const, i = True, i + 1
    base = tokens[i]
    i += 1
    if i < len(tokens) and tokens[i] == "const":
        base_const, i = True, i + 1
    ptrs = []
    while i < len(tokens):
        assert tokens[i] == "*", ctype
        i += 1
        own_const = i < len(tokens) and tokens[i] == "const"
        if own_const:
            i += 1
        ptrs.append(own_const)
    cur = C_BASE.get(base, base)
    pointee_const = base_const
    for own_const in ptrs:
        cur = ("*const " if pointee_const else "*mut ") + cur
        pointee_const = own_const
    return cur


def _strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _split_decl(decl):
    """'const void* x' -> ('const void*', 'x')"""
    decl = decl.strip()
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", decl, flags=re.S)
    return m.group(1).strip(), m.group(2)


def parse_header():
    text = _strip_comments(open(HEADER).read())
    structs, opaque, funcs, enums = {}, set(), {}, {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for f in m.group(2).split(";"):
            if f.strip():
                first, *more = f.split(",")   # `uint64_t nx, ny;` declares two fields of one type
                ctype, name = _split_decl(first)
                fields.append((name, c_to_rust(ctype)))
                for extra in more:
                    assert "*" not in extra, f
                    fields.append((extra.strip(), c_to_rust(ctype)))
        structs[m.group(3)] = fields
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text):
        opaque.add(m.group(2))
    for m in re.finditer(r"typedef\s+enum\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        nxt = 0
        for e in m.group(1).split(","):
            e = e.strip()
            if not e:
                continue
            if "=" in e:
                name, val = [t.strip() for t in e.split("=")]
                nxt = int(val, 0)
            else:
                name = e
            enums[name] = nxt
            nxt += 1
    body = re.sub(r"typedef[^;{]*\{.*?\}[^;]*;", "", text, flags=re.S)
    body = re.sub(r"typedef[^;]*;", "", body)
    body = re.sub(r"^\s*#.*$", "", body, flags=re.M)          # preprocessor lines
    body = body.replace('extern "C" {', "")
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(ndi_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", body, flags=re.S):
        ret = m.group(1).strip()
        args = []
        arg_text = m.group(3).strip()
        if arg_text and arg_text != "void":
            for a in arg_text.split(","):
                ctype, name = _split_decl(a)
                args.append((name, c_to_rust(ctype)))
        funcs[m.group(2)] = (args, None if ret == "void" else c_to_rust(ret))
    cb = re.search(r"typedef\s+([\w\s\*]+?)\(\s*\*\s*ndi_ring_consumer\s*\)\s*\(([^)]*)\)\s*;", text, flags=re.S)
    cb_args = [(n, c_to_rust(t)) for t, n in (_split_decl(a) for a in cb.group(2).split(","))]
    consumer = (cb_args, c_to_rust(cb.group(1)))
    return structs, opaque, funcs, enums, consumer


def _norm(ty):
    return re.sub(r"\s+", " ", ty.strip())


def parse_rust():
    text = open(RUST).read()
    text = re.sub(r"//[^\n]*", "", text)
    structs, opaque, funcs, consts = {}, set(), {}, {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*pub struct (\w+)\s*\{([^}]*)\}", text):
        fields = []
        for f in m.group(2).split(","):
            f = f.strip()
            if not f:
                continue
            fm = re.match(r"(pub\s+)?(\w+)\s*:\s*(.+)$", f, flags=re.S)
            fields.append((fm.group(2), _norm(fm.group(3)), bool(fm.group(1))))
        if len(fields) == 1 and fields[0][0] == "_private":
            opaque.add(m.group(1))
        else:
            assert all(pub for _, _, pub in fields), f"{m.group(1)}: every field of an ABI struct is pub"
            structs[m.group(1)] = [(n, t) for n, t, _ in fields]
    ext = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    for m in re.finditer(r"pub fn (\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", ext, flags=re.S):
        args = []
        for a in m.group(2).split(","):
            a = a.strip()
            if a:
                name, ty = a.split(":", 1)
                args.append((name.strip(), _norm(ty)))
        funcs[m.group(1)] = (args, _norm(m.group(3)) if m.group(3) else None)
    for m in re.finditer(r"pub const (NDI_\w+): i32 = (-?\d+);", text):
        consts[m.group(1)] = int(m.group(2))
    cb = re.search(r"pub type ndi_ring_consumer\s*=\s*Option<unsafe extern \"C\" fn\(([^)]*)\)\s*->\s*([^>]+)>;", text,
                   flags=re.S)
    cb_args = []
    for a in cb.group(1).split(","):
        name, ty = a.split(":", 1)
        cb_args.append((name.strip(), _norm(ty)))
    consumer = (cb_args, _norm(cb.group(2)))
    return structs, opaque, funcs, consts, consumer


def test_c_type_normaliser():
    assert c_to_rust("const void*") == "*const c_void"
    assert c_to_rust("void*") == "*mut c_void"
    assert c_to_rust("ndi_interp1d**") == "*mut *mut ndi_interp1d"
    assert c_to_rust("const ndi_interp1d* const*") == "*const *const ndi_interp1d"
    assert c_to_rust("void* const*") == "*const *mut c_void"
    assert c_to_rust("const int32_t*") == "*const i32"
    assert c_to_rust("uint64_t") == "u64"


def test_structs_match_field_by_field():
    hs, hopaque, _, _, _ = parse_header()
    rs, ropaque, _, _, _ = parse_rust()
    assert len(hs) >= 9, sorted(hs)   # boundary, 2 descs, oob_info, eval_opts, ring_chunk, ring_desc, shard_io, profile
    assert sorted(hs) == sorted(rs), (sorted(hs), sorted(rs))
    for name in hs:
        assert hs[name] == rs[name], f"{name}:\n  header {hs[name]}\n  rust   {rs[name]}"
    assert hopaque == ropaque == {"ndi_interp1d", "ndi_interp2d", "ndi_locator"}


def test_functions_match_argument_by_argument():
    _, _, hf, _, hcb = parse_header()
    _, _, rf, _, rcb = parse_rust()
    assert len(hf) >= 30
    assert sorted(hf) == sorted(rf), (sorted(set(hf) - set(rf)), sorted(set(rf) - set(hf)))
    for name in hf:
        assert hf[name] == rf[name], f"{name}:\n  header {hf[name]}\n  rust   {rf[name]}"
    assert hcb == rcb, (hcb, rcb)


def test_enumerators_match():
    _, _, _, he, _ = parse_header()
    _, _, _, _, _ = parse_rust()
    rc = parse_rust()[3]
    assert len(he) >= 29
    assert he == rc, (sorted(set(he.items()) ^ set(rc.items())))


def test_the_check_catches_a_swap(tmp_path, monkeypatch):
    """Self-test of the checker: swapping two fields / two arguments in the Rust file must be seen."""
    src = open(RUST).read()
    swapped = src.replace("    pub q_memspace: i32,\n    pub out_memspace: i32,", "    pub out_memspace: i32,\n    pub q_memspace: i32,")
    assert swapped != src
    p = tmp_path / "hip_ffi.rs"
    p.write_text(swapped)
    monkeypatch.setattr(__import__(__name__), "RUST", str(p))
    with pytest.raises(AssertionError):
        test_structs_match_field_by_field()
    swapped = src.replace("        lo: *mut u64,\n        hi: *mut u64", "X").replace(
        "pub fn ndi_shard_bounds(nq: u64, shard: u32, n_shards: u32, lo: *mut u64, hi: *mut u64);",
        "pub fn ndi_shard_bounds(nq: u64, n_shards: u32, shard: u32, lo: *mut u64, hi: *mut u64);")
    assert swapped != src
    p.write_text(swapped)
    with pytest.raises(AssertionError):
        test_functions_match_argument_by_argument()
    # a widened type is caught too
    p.write_text(src.replace("pub slot: u32,", "pub slot: u64,"))
    with pytest.raises(AssertionError):
        test_structs_match_field_by_field()


CT = {"i32": C.c_int32, "u32": C.c_uint32, "i64": C.c_int64, "u64": C.c_uint64, "f64": C.c_double}


def test_ctypes_structs_match_the_header(pkg):
    """The Python binding's Structures against the header: field names, order and sizes."""
    hs = parse_header()[0]
    cap = pkg._capi
    pairs = {"ndi_boundary": cap.Boundary, "ndi_interp1d_desc": cap.Interp1DDesc, "ndi_interp2d_desc": cap.Interp2DDesc,
             "ndi_oob_info": cap.OobInfo, "ndi_eval_opts": cap.EvalOpts, "ndi_profile": cap.Profile}
    for cname, cls in pairs.items():
        got = [(n, t) for n, t in cls._fields_]
        assert [n for n, _ in got] == [n for n, _ in hs[cname]], cname
        for (n, ct), (_, rt) in zip(got, hs[cname]):
            if rt in CT:
                assert C.sizeof(ct) == C.sizeof(CT[rt]), (cname, n)
            elif rt.startswith("*"):
                assert C.sizeof(ct) == C.sizeof(C.c_void_p), (cname, n)
            else:   # nested struct by value
                assert C.sizeof(ct) == C.sizeof(pairs[rt]), (cname, n)


PATCH = os.path.join(ROOT, "rust", "patches", "ndarray-interp-0.6.0-batched-hook.patch")
STRATEGIES = os.path.join(ROOT, "rust", "ndarray-interp-hip", "src", "strategies.rs")


def test_patch_adds_both_hooks_and_the_strategies_override_them():
    """Round 6: `interp_array` reaches the strategy through `interp_array_into_owned` (so the device strategies can pass
    NDI_EVAL_FRESH_OUTPUT), `interp_array_into` through `interp_array_into`; both are defaulted trait methods on both
    traits, and every device strategy overrides both."""
    patch = open(PATCH).read()
    for f in ("src/interp1d/strategies/mod.rs", "src/interp2d/strategies/mod.rs"):
        part = patch.split(f"+++ b/{f}")[1].split("\ndiff ")[0]
        assert re.search(r"^\+    fn interp_array_into<", part, flags=re.M), f
        assert re.search(r"^\+    fn interp_array_into_owned<", part, flags=re.M), f
    for f, call in (("src/interp1d/mod.rs", "self.interp_array_into_owned(xs, ys.view_mut())"),
                    ("src/interp2d/mod.rs", "self.interp_array_into_owned(xs, ys, zs.view_mut())")):
        part = patch.split(f"+++ b/{f}")[1].split("\ndiff ")[0]
        assert "+        " + call in part, f
        assert "+            return self.strategy.interp_array_into_owned(" in part or \
            re.search(r"\+\s+\.interp_array_into_owned\(self, xs_1d, ys_1d, buffer_d\)", part), f
    src = open(STRATEGIES).read()
    assert len(re.findall(r"fn interp_array_into<", src)) == 3 and len(re.findall(r"fn interp_array_into_owned<", src)) == 3
    assert src.count("ffi::NDI_EVAL_FRESH_OUTPUT)") == 3       # one per owned hook
    ffi = open(RUST).read()
    assert "pub const NDI_EVAL_FRESH_OUTPUT: i32 = 1;" in ffi and "pub const NDI_EVAL_ROWS_AFTER_ERROR_UNSPECIFIED: i32 = 2;" in ffi


def test_patch_applies_to_the_reference_sources(tmp_path):
    """Where the reference checkout is present (this container; not the GPU box): `patch --dry-run` of the hook patch."""
    import shutil
    import subprocess
    ref = "/root/reference/src"
    if not os.path.isdir(ref) or shutil.which("patch") is None:
        pytest.skip("no reference checkout / no patch(1) here")
    r = subprocess.run(["patch", "-p1", "--dry-run", "-d", "/root/reference", "-i", PATCH], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
