"""libdg16.so loads (no GPU needed) and exports every symbol include/dg16.h declares; creating a
context without a GPU fails loudly instead of falling back."""

import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dg16.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dg16_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    import dg16_amd
    from dg16_amd.lib import load, EXPORTED
    L = load()
    decl = declared_symbols()
    assert len(decl) >= 15
    for s in decl:
        assert hasattr(L, s), "include/dg16.h declares %s but libdg16.so does not export it" % s
    assert sorted(EXPORTED) == decl, "python binding list and header disagree"


RUST_SYS = os.path.join(ROOT, "bindings", "dg16-sys", "src", "lib.rs")


def test_rust_extern_block_lists_the_header_symbols():
    """bindings/dg16-sys/src/lib.rs (the crate a maintainer builds where cargo exists) declares exactly the functions of
    include/dg16.h, once each, and the shim only calls functions the sys crate declares."""
    rust = open(RUST_SYS).read()
    fns = re.findall(r"pub fn (dg16_[a-z0-9_]+)\s*\(", rust)
    assert sorted(fns) == declared_symbols(), "dg16-sys and include/dg16.h disagree"
    shim = os.path.join(ROOT, "bindings", "dg16-shim", "src")
    used = set()
    for f in os.listdir(shim):
        used |= set(re.findall(r"sys::(dg16_[a-z0-9_]+)\s*\(", open(os.path.join(shim, f)).read()))
    assert used and used <= set(fns), sorted(used - set(fns))
    # every patch names a file of the reference by the path the survey cites
    for f in os.listdir(os.path.join(ROOT, "bindings", "patches")):
        txt = open(os.path.join(ROOT, "bindings", "patches", f)).read()
        assert re.search(r"^--- a/(dist-primitives|groth16|ark-circom)/", txt, flags=re.M), f


def test_rust_struct_mirrors_have_the_fields_of_the_header():
    """#[repr(C)] mirrors: same field names in the same order as the C structs (types are checked by a maintainer's
    `cargo build --features static-check`; the order is what a silent ABI break would change)."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dg16.h")).read(), flags=re.S)
    rust = open(RUST_SYS).read()
    pairs = {"dg16_pk_info": "Dg16PkInfo", "dg16_r1cs_header": "Dg16R1csHeader", "dg16_zkey_header": "Dg16ZkeyHeader",
             "dg16_csr": "Dg16Csr", "dg16_arkkey_layout_t": "Dg16ArkKeyLayout", "dg16_comm": "Dg16Comm", "dg16_net": "Dg16Net"}
    for cname, rname in pairs.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, flags=re.S).group(1)
        cfields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.search(r"\(\*(\w+)\)", decl)            # function pointer member
            if m:
                cfields.append(m.group(1))
                continue
            for nm in decl.split(","):
                cfields.append(re.findall(r"\w+", nm)[-1])          # the declarator's last identifier is the member name
        rbody = re.search(r"pub struct %s \{(.*?)\n\}" % rname, rust, flags=re.S).group(1)
        rbody = re.sub(r"//.*", "", rbody)
        rfields = [f.rstrip("_") for f in re.findall(r"pub (\w+):", rbody)]
        assert [c for c in cfields if c] == rfields, (cname, cfields, rfields)


def test_flag_constants_agree_with_the_header():
    """enum dg16_flags of include/dg16.h == the F_* constants the Python binding passes (and the Rust block of
    dg16-sys crate)."""
    from dg16_amd import lib
    txt = open(os.path.join(ROOT, "include", "dg16.h")).read()
    body = re.search(r"enum dg16_flags \{(.*?)\};", txt, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    hdr = {k: int(v) for k, v in re.findall(r"DG16_(F_[A-Z_]+)\s*=\s*(\d+)u", body)}
    assert len(hdr) >= 6 and len(set(hdr.values())) == len(hdr)
    for v in hdr.values():
        assert v & (v - 1) == 0, "flags are single bits"
    for name, value in hdr.items():
        assert getattr(lib, name) == value, name
    rust = open(RUST_SYS).read()
    rflags = {n: int(v) for n, v in re.findall(r"pub const DG16_(F_[A-Z_]+): c_uint = (\d+);", rust)}
    assert rflags == hdr, "dg16-sys flag constants"


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import dg16_amd
    with pytest.raises(dg16_amd.Dg16Error):
        dg16_amd.Context(0)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under distributed-groth16_amd/ may import, link or
    dlopen it."""
    pkg = os.path.join(ROOT, "distributed-groth16_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "liboracle", "oracle/c", "corc"):
                    assert needle not in txt, "%s references the oracle (%s)" % (f, needle)


def test_header_is_valid_c_and_cpp():
    import subprocess
    hdr = os.path.join(ROOT, "include", "dg16.h")
    subprocess.check_call(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-Werror", hdr])
    subprocess.check_call(["g++", "-fsyntax-only", "-x", "c++", "-std=c++11", "-Wall", "-Werror", hdr])


def test_plain_c_consumer_parses_a_zkey_and_verifies_a_proof(tmp_path):
    """tests/abi_c/consumer.c, built with gcc against include/dg16.h and libdg16.so: zkey -> vk, proof.bin bytes ->
    proof, dg16_groth16_verify -- the reference's `zk-cli verify` path from C."""
    import random
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import dg16_amd  # noqa: F401
    from dg16_amd import serialize as S
    from oracle.pyref import groth16 as G
    from oracle.pyref.fields import FR
    from test_zkey_reader import small_key
    from zkey_writer import write_zkey
    lib_dir = os.path.join(ROOT, "distributed-groth16_amd")
    exe = str(tmp_path / "consumer")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_c", "consumer.c"), "-o", exe, "-L", lib_dir, "-ldg16",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    F = FR["bn254"]
    r1cs, w, pk, m = small_key(seed=4, nc=17, ni=2, nw=9)
    rng = random.Random(2)
    proof = G.create_proof("bn254", pk, rng.randrange(1, F.p), rng.randrange(1, F.p), r1cs, w)
    (tmp_path / "k.zkey").write_bytes(write_zkey(pk, r1cs, m))
    (tmp_path / "proof.bin").write_bytes(S.proof_to_bytes(*proof))
    for name, vals, expect in (("good", w[1:2], "accepted=1"), ("bad", [(w[1] + 1) % F.p], "accepted=0")):
        (tmp_path / (name + ".bin")).write_bytes(b"".join(v.to_bytes(32, "little") for v in vals))
        out = subprocess.run([exe, str(tmp_path / "k.zkey"), str(tmp_path / "proof.bin"), str(tmp_path / (name + ".bin"))],
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert expect in out.stdout and "n_public=1" in out.stdout
        assert "arkkey_layout=3 (key file truncated)" in out.stdout
        import torch
        if not torch.cuda.is_available():
            assert "ctx_create=0" not in out.stdout          # no GPU: a status, not a context
