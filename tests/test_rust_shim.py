"""CPU: the reference-side Rust binding (rust-shim/) agrees with the C ABI it binds.

No Rust toolchain exists in this image, so the shim cannot be compiled; what can be checked mechanically is the part a compiler would
NOT check anyway -- that the `extern "C"` declarations in rust-shim/zkhip_ffi.rs describe the functions include/zkhip.h declares (name,
arity, pointer vs integer, width, constness of every parameter, return type) and that libzkhip.so exports them -- plus the structural
promises of the patch: the upstream generic signatures are kept (halo2-axiom `best_multiexp<C: CurveAffine>` / `best_fft<Scalar: Field,
G: FftGroup<Scalar>>`, reached from /root/reference/aggregator/src/wrapper.rs:129), dispatch is by TypeId with a CPU fall-through, and
`ParamsKZG` unregisters its bases in `Drop`.  The call sequence the shim issues runs on the GPU in tests/cpp/shim_sequence.c."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "rust-shim")

# canonical parameter classes: (kind, bits, const) -- what has to agree across the FFI for the call to be sound on x86-64 / SysV
C_TYPES = {
    "int": ("int", 32), "uint32_t": ("uint", 32), "int32_t": ("int", 32), "size_t": ("uint", 64), "uint64_t": ("uint", 64),
    "uint8_t": ("uint", 8), "char": ("int", 8), "void": ("void", 0),
}
RUST_TYPES = {
    "c_int": ("int", 32), "i32": ("int", 32), "u32": ("uint", 32), "usize": ("uint", 64), "u64": ("uint", 64), "u8": ("uint", 8),
    "c_char": ("int", 8), "c_void": ("void", 0),
}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def c_param(p):
    """'const uint64_t *scalars' / 'uint64_t out_xyz[12]' / 'size_t n' -> ('ptr', pointee_class, const) or ('val', class)"""
    p = p.strip()
    is_array = "[" in p
    p = re.sub(r"\[[^\]]*\]", "", p)
    const = bool(re.search(r"\bconst\b", p))
    stars = p.count("*")
    toks = [t for t in re.sub(r"[*]", " ", p).replace("const", " ").split()]
    base = toks[0]
    assert base in C_TYPES, f"unknown C type in {p!r}"
    if stars or is_array:
        assert stars + int(is_array) == 1, f"pointer depth of {p!r}"
        return ("ptr", C_TYPES[base], const)
    return ("val", C_TYPES[base])


def c_prototypes():
    text = strip_c_comments(open(os.path.join(ROOT, "include", "zkhip.h")).read())
    protos = {}
    for m in re.finditer(r"\b(int|void|const char \*|size_t)\s*(zkhip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if "(*" in params:                       # array-of-array parameters (zkhip_profile_read): not bound by the shim
            protos[name] = (ret, None)
            continue
        try:
            plist = [] if params in ("void", "") else [c_param(p) for p in params.split(",")]
        except AssertionError:                   # pointer-to-pointer / struct parameters (device-pipeline entry points): not bound by the shim
            plist = None
        protos[name] = (ret, plist)
    return protos


def rust_param(p):
    name, ty = [x.strip() for x in p.split(":", 1)]
    m = re.fullmatch(r"\*(const|mut)\s+(\w+)", ty)
    if m:
        assert m.group(2) in RUST_TYPES, f"unknown Rust pointee in {p!r}"
        return ("ptr", RUST_TYPES[m.group(2)], m.group(1) == "const")
    assert ty in RUST_TYPES, f"unknown Rust type in {p!r}"
    return ("val", RUST_TYPES[ty])


def rust_externs():
    text = open(os.path.join(SHIM, "zkhip_ffi.rs")).read()
    text = re.sub(r"//[^\n]*", "", text)
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S)
    assert block, 'no extern "C" block in rust-shim/zkhip_ffi.rs'
    out = {}
    for m in re.finditer(r"fn\s+(zkhip_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block.group(1), flags=re.S):
        params = " ".join(m.group(2).split())
        out[m.group(1)] = ((m.group(3) or "()").strip(), [rust_param(p) for p in params.split(",") if p.strip()])
    return out, text


def test_extern_block_matches_the_header():
    protos = c_prototypes()
    externs, _ = rust_externs()
    assert {"zkhip_msm_g1", "zkhip_ntt_fr", "zkhip_register_bases", "zkhip_unregister_bases", "zkhip_last_error", "zkhip_init"} <= set(externs)
    for name, (rret, rparams) in externs.items():
        assert name in protos, f"{name} is declared in the Rust shim but not in include/zkhip.h"
        cret, cparams = protos[name]
        assert cparams is not None, f"{name}: the header's parameter list is outside what this test models"
        want_ret = {"int": "c_int", "void": "()", "const char *": "*const c_char", "size_t": "usize"}[cret]
        assert rret == want_ret, f"{name}: returns {rret} in Rust, {cret} in C"
        assert len(rparams) == len(cparams), f"{name}: {len(rparams)} parameters in Rust, {len(cparams)} in C"
        for i, (r, c) in enumerate(zip(rparams, cparams)):
            assert r[0] == c[0], f"{name} parameter {i}: {r} vs {c} (pointer / value)"
            assert r[1] == c[1], f"{name} parameter {i}: {r} vs {c} (class / width)"
            if r[0] == "ptr":
                assert r[2] == c[2], f"{name} parameter {i}: *const / *mut does not match the header's const"


def test_library_exports_what_the_shim_binds(lib):
    externs, _ = rust_externs()
    for name in externs:
        assert hasattr(lib, name), f"libzkhip.so does not export {name}"


def test_every_ffi_call_in_the_shim_is_declared():
    externs, ffi_text = rust_externs()
    used = set(re.findall(r"\b(zkhip_[a-z0-9_]+)\s*\(", ffi_text))
    assert used <= set(externs), f"called but not declared: {sorted(used - set(externs))}"
    # the patch files only go through zkhip_ffi::*, never through the C ABI directly
    for f in ("arithmetic_patch.rs", "commitment_patch.rs", "domain_patch.rs"):
        body = re.sub(r"//[^\n]*", "", open(os.path.join(SHIM, f)).read())
        for call in re.findall(r"zkhip_ffi::(\w+)", body):
            assert re.search(r"pub\(crate\)\s+fn\s+" + call + r"\b", ffi_text), f"{f} calls zkhip_ffi::{call}, which zkhip_ffi.rs does not define"
        assert not re.search(r'extern\s+"C"', body), f'{f} must not declare its own extern "C" items'


def test_patch_keeps_the_generic_signatures_and_the_cpu_fall_through():
    a = open(os.path.join(SHIM, "arithmetic_patch.rs")).read()
    assert re.search(r"pub fn best_multiexp<C: CurveAffine>\(coeffs: &\[C::Scalar\], bases: &\[C\]\) -> C::Curve", a)
    assert re.search(r"pub fn best_fft<Scalar: Field, G: FftGroup<Scalar>>\(a: &mut \[G\], omega: Scalar, log_n: u32\)", a)
    assert "best_multiexp_cpu(coeffs, bases)" in a and "best_fft_cpu(a, omega, log_n)" in a          # every other instantiation / any error
    assert "assert_eq!(coeffs.len(), bases.len())" in a                                                 # the reference's panic, unchanged
    ffi = open(os.path.join(SHIM, "zkhip_ffi.rs")).read()
    assert "TypeId::of::<T>() == TypeId::of::<U>()" in ffi
    for fn in ("try_msm_g1", "try_ntt_fr", "pin", "unpin"):
        body = re.search(r"pub\(crate\) fn " + fn + r"\b.*?\n\}", ffi, flags=re.S).group(0)
        assert "is::<" in body, f"{fn} does not dispatch on the type"
    assert "panic!" not in re.sub(r"//[^\n]*", "", ffi), "a non-zero status falls through to the CPU body, it does not panic"


def test_params_kzg_unregisters_in_drop_and_pins_in_every_constructor():
    c = open(os.path.join(SHIM, "commitment_patch.rs")).read()
    drop = re.search(r"impl<E: Engine> Drop for ParamsKZG<E> \{.*?\n\}", c, flags=re.S)
    assert drop and "self.zkhip_unpin()" in drop.group(0)
    unpin = re.search(r"fn zkhip_unpin\(&self\) \{.*?\}", c, flags=re.S).group(0)
    assert "unpin::<E::G1Affine>(&self.g)" in unpin and "unpin::<E::G1Affine>(&self.g_lagrange)" in unpin
    for ctor in ("setup", "from_parts", "read_custom", "downsize"):
        assert re.search(r"`(?:Params::)?" + ctor + r"`", c), f"no instruction for {ctor}"
    assert re.search(r"impl<E: Engine> Clone for ParamsKZG<E>.*?\.zkhip_pinned\(\)", c, flags=re.S)


@pytest.mark.parametrize("name", ["zkhip_ffi.rs", "arithmetic_patch.rs", "commitment_patch.rs", "domain_patch.rs", "build.rs"])
def test_rust_files_are_lexically_balanced(name):
    """the cheapest stand-in for a parse: brackets balance outside comments, strings and char / lifetime tokens"""
    text = open(os.path.join(SHIM, name)).read()
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text)
    text = re.sub(r"'(?:\\.|[^'\\])'", "' '", text)
    stack, pairs = [], {")": "(", "]": "[", "}": "{"}
    for ch in text:
        if ch in "([{":
            stack.append(ch)
        elif ch in ")]}":
            assert stack and stack.pop() == pairs[ch], f"{name}: unbalanced {ch}"
    assert not stack, f"{name}: unclosed {stack}"


def test_integration_md_reproduces_the_shim_files_verbatim():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in ("Cargo.patch.toml", "build.rs", "zkhip_ffi.rs", "arithmetic_patch.rs", "commitment_patch.rs", "domain_patch.rs"):
        body = open(os.path.join(SHIM, name)).read().rstrip()
        assert body in doc, f"INTEGRATION.md section 2 is out of date with rust-shim/{name}"
