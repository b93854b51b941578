"""CPU: the known answers the reference's OWN tests hold for the file formats, reproduced through the native readers
of libdg16 (host code) -- so that the `.r1cs` / `.zkey` rows are pinned by reference-held vectors, not only by files
this repo's test writer produced:

  * ark-circom/src/circom/r1cs_reader.rs:257-339  `sample()`: a complete 816-byte .r1cs image and the header fields,
    constraint entries and wire map the reference asserts after parsing it            -> dg16_r1cs_parse
  * ark-circom/src/zkey.rs:417-491  fq_buf / g1_buf / g2_buf: snarkjs' byte images of G1.F.one, G1.one and G2.one and
    the coordinates `can_deser_g1` / `can_deser_g2` expect (this pins the Fq2 component order c0 || c1 and the x || y
    order of the 128-byte G2 layout that dg16_pk_create / dg16_msm consume as they lie in the file)
                                                                                      -> dg16_zkey_parse / _points
The vectors are data copied from the reference (tests/golden/reference_kats.json); the GPU half -- the same bytes
used as MSM bases -- is tests/test_gpu_zkey.py::test_reference_g2_bytes_are_the_generator_on_the_gpu."""

import json
import os
import struct

import numpy as np

import dg16_amd  # noqa: F401
from oracle.pyref.curves import CURVES
from oracle.pyref.fields import FQ, FR

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


def test_r1cs_sample_vector_of_the_reference():
    from dg16_amd.r1cs import R1CS
    raw = bytes.fromhex(KAT["r1cs_sample_hex"].replace(" ", ""))
    want = KAT["r1cs_sample_asserts"]
    assert len(raw) == 816 and raw[12 + 12 + 4:12 + 12 + 4 + 32].hex() == want["prime_hex"]
    assert int.from_bytes(bytes.fromhex(want["prime_hex"]), "little") == FR["bn254"].p
    r = R1CS(raw)
    for k in ("n_wires", "n_pub_out", "n_pub_in", "n_prv_in", "n_labels", "n_constraints"):
        assert getattr(r, k) == want[k], k
    A, B, C = r.rows(0), r.rows(1), r.rows(2)
    assert len(A) == len(B) == len(C) == 3
    # r1cs_reader.rs:323-329: constraints[0].0 has two terms, the first is (wire 5, coefficient 3);
    # constraints[2].1[0] = (wire 0, 6); constraints[1].2 is empty   [(coefficient, wire) pairs here]
    assert len(A[0]) == want["c0_a_len"] and A[0][0] == (want["c0_a0"][1], want["c0_a0"][0])
    assert B[2][0] == (want["c2_b0"][1], want["c2_b0"][0])
    assert len(C[1]) == want["c1_c_len"]
    # every entry of the image, in file order
    assert A == [[(3, 5), (8, 6)], [(4, 1), (8, 4), (3, 5)], [(4, 6)]]
    assert B == [[(2, 0), (20, 2), (12, 3)], [(44, 3), (6, 6)], [(6, 0), (11, 2), (5, 3)]]
    assert C == [[(5, 0), (7, 2)], [], [(600, 6)]]
    assert r.wire_mapping is not None and len(r.wire_mapping) == want["wire_mapping_len"]
    assert int(r.wire_mapping[1]) == want["wire_mapping_1"]
    assert [int(x) for x in r.wire_mapping] == [0, 3, 10, 11, 12, 15, 324]


def _mont_le(b, Fq):
    return Fq.from_mont(int.from_bytes(bytes(b), "little"))


def test_zkey_point_images_of_the_reference_decode_to_the_generators():
    Fq = FQ["bn254"]
    fq_buf, g1_buf, g2_buf = (bytes(KAT[k]) for k in ("zkey_fq_buf", "zkey_g1_buf", "zkey_g2_buf"))
    assert _mont_le(fq_buf, Fq) == 1                                                   # can_deser_fq
    assert [_mont_le(g1_buf[i:i + 32], Fq) for i in (0, 32)] == [int(v) for v in KAT["zkey_g1_one"]]
    g2 = [_mont_le(g2_buf[i:i + 32], Fq) for i in (0, 32, 64, 96)]                     # can_deser_g2
    assert g2[:2] == [int(v) for v in KAT["zkey_g2_one"]["x"]] and g2[2:] == [int(v) for v in KAT["zkey_g2_one"]["y"]]
    # ... and they are THE generators of the oracle's curve description (ark-bn254's G1 / G2 generators)
    assert tuple(int(v) for v in KAT["zkey_g1_one"]) == CURVES["bn254", "g1"].gen
    c2 = CURVES["bn254", "g2"]
    assert ((g2[0], g2[1]), (g2[2], g2[3])) == c2.gen and c2.on_curve(c2.gen)


def test_native_zkey_reader_hands_out_the_reference_images_as_they_lie_in_the_file():
    """A zkey whose every G1 slot is g1_buf and every G2 slot is g2_buf (the shape of the reference's
    can_deser_g1_vec / can_deser_g2_vec, zkey.rs:503-560): dg16_zkey_points must return exactly those bytes --
    they go to dg16_pk_create without conversion."""
    from dg16_amd.zkey import ZKey
    Fq, Fr = FQ["bn254"], FR["bn254"]
    g1, g2 = bytes(KAT["zkey_g1_buf"]), bytes(KAT["zkey_g2_buf"])
    n_vars, n_public, domain = 5, 1, 8
    sec = {1: struct.pack("<I", 1),
           2: (struct.pack("<I", 32) + Fq.p.to_bytes(32, "little") + struct.pack("<I", 32) + Fr.p.to_bytes(32, "little")
               + struct.pack("<III", n_vars, n_public, domain) + g1 + g1 + g2 + g2 + g1 + g2),
           3: g1 * (n_public + 1), 4: struct.pack("<I", 0), 5: g1 * n_vars, 6: g1 * n_vars, 7: g2 * n_vars,
           8: g1 * (n_vars - n_public - 1), 9: g1 * domain}
    raw = b"zkey" + struct.pack("<II", 1, len(sec))
    for sid in sorted(sec):
        raw += struct.pack("<IQ", sid, len(sec[sid])) + sec[sid]
    z = ZKey(raw)
    assert (z.n_vars, z.n_public, z.domain_size) == (n_vars, n_public, domain)
    g1w, g2w = np.frombuffer(g1, dtype=np.uint64), np.frombuffer(g2, dtype=np.uint64)
    for name in ("alpha_g1", "beta_g1", "delta_g1"):
        assert np.array_equal(getattr(z, name).reshape(-1), g1w), name
    for name in ("beta_g2", "gamma_g2", "delta_g2"):
        assert np.array_equal(getattr(z, name).reshape(-1), g2w), name
    for name, cnt in (("a_query", n_vars), ("b_g1_query", n_vars), ("l_query", n_vars - n_public - 1), ("h_query", domain),
                      ("ic", n_public + 1)):
        q = getattr(z, name)
        assert q.shape == (cnt, 8) and all(np.array_equal(row, g1w) for row in q), name
    assert z.b_g2_query.shape == (n_vars, 16) and all(np.array_equal(row, g2w) for row in z.b_g2_query)
