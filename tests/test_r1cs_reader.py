"""The `.r1cs` reader against the one real circom R1CS present in the reference tree
(ark-circom/test-vectors/complex-circuit/complex-circuit-10000-10000.r1cs, committed xz-compressed under
tests/golden) and the error behaviour of r1cs_reader.rs:58-72,161-188."""

import lzma
import os
import struct

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "complex_circuit_10000.r1cs.xz")


def witness_for_complex_circuit(r, p):
    """complex-circuit-10000-10000.circom:9-17: b[0] = a*a, b[i] = b[i-1]^2, c = b[last]; wire order
    1, c, a, b[0..]: filled by solving the constraints in order with a = 3."""
    w = [0] * r.n_wires
    w[0] = 1
    w[2] = 3
    A, B, C = r.rows(0), r.rows(1), r.rows(2)
    for ra, rb, rc in zip(A, B, C):
        (ca, ia), (cb, ib), (cc, ic) = ra[0], rb[0], rc[0]
        assert len(ra) == len(rb) == len(rc) == 1
        w[ic] = ca * w[ia] * cb * w[ib] * pow(cc, p - 2, p) % p
    return w


def test_header_and_shape():
    import dg16_amd  # noqa: F401
    from dg16_amd.r1cs import R1CS
    r = R1CS.from_file(FIX)
    # 10 002 wires, 10 000 constraints, 1 public output, 1 private input
    assert (r.n_wires, r.n_constraints, r.n_pub_out, r.n_pub_in, r.n_prv_in) == (10002, 10000, 1, 0, 1)
    assert r.num_inputs == 2 and r.num_aux == 10000
    assert r.wire_mapping[0] == 0 and len(r.wire_mapping) == 10002
    for k in range(3):
        row_ptr, col, coeff = r.csr[k]
        assert len(row_ptr) == 10001 and row_ptr[-1] == len(col) == coeff.shape[0] == 10000


def test_witness_satisfies_constraints():
    import dg16_amd  # noqa: F401
    from dg16_amd.r1cs import R1CS
    from oracle.pyref.fields import FR
    from oracle.pyref import groth16 as G
    p = FR["bn254"].p
    r = R1CS.from_file(FIX)
    w = witness_for_complex_circuit(r, p)
    assert w[3] == 9 and w[4] == 81
    r1cs = dict(num_instance=r.num_inputs, num_witness=r.num_aux, num_constraints=r.n_constraints,
                a=r.rows(0), b=r.rows(1), c=r.rows(2))
    assert G.is_satisfied(r1cs, w, p)


def test_reader_errors():
    import dg16_amd  # noqa: F401
    from dg16_amd.r1cs import R1CS, R1CSError
    raw = bytearray(lzma.decompress(open(FIX, "rb").read()))
    with pytest.raises(R1CSError, match="magic"):
        R1CS(b"xxxx" + bytes(raw[4:]))
    bad = bytearray(raw)
    struct.pack_into("<I", bad, 4, 2)
    with pytest.raises(R1CSError, match="version"):
        R1CS(bytes(bad))
    # corrupt the prime in the header section (first section in this file is the header)
    bad = bytearray(raw)
    off = 12
    while True:
        typ, size = struct.unpack_from("<IQ", bad, off)
        off += 12
        if typ == 1:
            bad[off + 4] ^= 1
            break
        off += size
    with pytest.raises(R1CSError, match="bn256"):
        R1CS(bytes(bad))
