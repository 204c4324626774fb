#!/usr/bin/env python3
"""Rewrites the code blocks of INTEGRATION.md section 2 from the files under rust-shim/ (tests/test_rust_shim.py asserts they agree)."""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
doc_path = os.path.join(ROOT, "INTEGRATION.md")
doc = open(doc_path).read()
for name in ("Cargo.patch.toml", "build.rs", "zkhip_ffi.rs", "arithmetic_patch.rs", "commitment_patch.rs", "domain_patch.rs", "prover_patch.rs"):
    body = open(os.path.join(ROOT, "rust-shim", name)).read().rstrip()
    pat = re.compile(r"(`rust-shim/" + re.escape(name) + r"` — [^\n]*:\n\n```\w+\n)(.*?)(\n```\n)", re.S)
    assert pat.search(doc), name
    doc = pat.sub(lambda m: m.group(1) + body + m.group(3), doc, count=1)
open(doc_path, "w").write(doc)
print("INTEGRATION.md section 2 synchronised")
