"""CPU: the `.zkey` reader (distributed-groth16_amd/zkey.py, mirror of ark-circom/src/zkey.rs:53-388) on a key
written in the snarkjs layout by tests/zkey_writer.py from the big-int oracle's setup.  (The reference's own
test.zkey is not in its tree.)  The Montgomery constant snarkjs writes for G1.F.one is the reference's in-tree
known answer zkey.rs:417-427 and is checked in tests/test_oracle_kats.py."""

import random
import struct

import numpy as np
import pytest

import dg16_amd  # noqa: F401  (package shim)
from dg16_amd.zkey import ZKey, ZKeyError
from oracle.pyref import groth16 as G
from oracle.pyref.fields import FQ, FR
from zkey_writer import write_zkey


def small_key(seed=3, nc=21, ni=3, nw=17):
    F = FR["bn254"]
    r1cs, w = G.synthetic_r1cs(F, num_constraints=nc, num_instance=ni, num_witness=nw, seed=seed)
    rng = random.Random(seed)
    td = tuple(rng.randrange(1, F.p) for _ in range(5))
    pk, sc = G.setup("bn254", r1cs, td)
    return r1cs, w, pk, sc["m"]


def dec_g1(row):
    Fq = FQ["bn254"]
    v = [Fq.from_mont(int.from_bytes(row[4 * i:4 * i + 4].tobytes(), "little")) for i in range(2)]
    return None if v == [0, 0] else tuple(v)


def test_zkey_round_trip():
    r1cs, w, pk, m = small_key()
    z = ZKey(write_zkey(pk, r1cs, m))
    ni, nw, nc = r1cs["num_instance"], r1cs["num_witness"], r1cs["num_constraints"]
    assert (z.n_vars, z.n_public, z.domain_size) == (ni + nw, ni - 1, m)
    assert (z.num_constraints, z.num_instance_variables, z.num_witness_variables) == (nc, ni, nw + 1)
    assert [dec_g1(r) for r in z.a_query] == pk["a_query"]
    assert [dec_g1(r) for r in z.b_g1_query] == pk["b_g1_query"]
    assert [dec_g1(r) for r in z.l_query] == pk["l_query"]
    assert [dec_g1(r) for r in z.h_query] == pk["h_query"] and len(z.h_query) == m
    assert [dec_g1(r) for r in z.ic] == pk["gamma_abc_g1"]
    assert dec_g1(z.alpha_g1) == pk["alpha_g1"] and dec_g1(z.delta_g1) == pk["delta_g1"]
    assert z.b_g2_query.shape == (ni + nw, 16)
    # coefficients: stored v R^2, the public-input rows are dropped (zkey.rs:176-180)
    F = FR["bn254"]
    R2inv = F.inv(F.R * F.R % F.p)
    for k, name in enumerate(("a", "b")):
        ptr, col, val = z._csr_r2[k]
        assert len(ptr) == nc + 1
        for row in range(nc):
            got = [(int.from_bytes(val[j].tobytes(), "little") * R2inv % F.p, int(col[j]))
                   for j in range(ptr[row], ptr[row + 1])]
            assert got == [(cf % F.p, idx) for cf, idx in r1cs[name][row]]


def test_zkey_rejects_malformed_files():
    r1cs, w, pk, m = small_key()
    good = write_zkey(pk, r1cs, m)
    with pytest.raises(ZKeyError):
        ZKey(b"r1cs" + good[4:])
    with pytest.raises(ZKeyError):
        ZKey(good[:len(good) // 2])
    # wrong base field prime
    bad = bytearray(good)
    p = good.index(FQ["bn254"].p.to_bytes(32, "little"))
    bad[p] ^= 1
    with pytest.raises(ZKeyError):
        ZKey(bytes(bad))
    # drop the last section (H)
    n = struct.unpack_from("<I", good, 8)[0]
    off, cut = 12, None
    for _ in range(n):
        sid, size = struct.unpack_from("<IQ", good, off)
        if sid == 9:
            cut = off
        off += 12 + size
    trunc = good[:8] + struct.pack("<I", n - 1) + good[12:cut]
    with pytest.raises(ZKeyError):
        ZKey(trunc)
