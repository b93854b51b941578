"""TEST INFRASTRUCTURE: serialise a Groth16 key of the big-int oracle (oracle/pyref/groth16.setup) in the snarkjs
`.zkey` layout the reference reads (ark-circom/src/zkey.rs:53-388): section table, protocol id, Groth16 header,
IC, coefficients (values times R^2, public-input rows appended like snarkjs does), A, B1, B2, C(=L), H."""

import struct

from oracle.pyref.fields import FQ, FR


def _fq(v):
    F = FQ["bn254"]
    return F.to_bytes(v, mont=True)


def _g1(P):
    return bytes(64) if P is None else _fq(P[0]) + _fq(P[1])


def _g2(P):
    return bytes(128) if P is None else _fq(P[0][0]) + _fq(P[0][1]) + _fq(P[1][0]) + _fq(P[1][1])


def write_zkey(pk, r1cs, domain_size):
    Fr, Fq = FR["bn254"], FQ["bn254"]
    ni, nw, nc = r1cs["num_instance"], r1cs["num_witness"], r1cs["num_constraints"]
    n_vars, n_public = ni + nw, ni - 1
    sec = {}
    sec[1] = struct.pack("<I", 1)
    sec[2] = (struct.pack("<I", 32) + Fq.p.to_bytes(32, "little") + struct.pack("<I", 32) + Fr.p.to_bytes(32, "little")
              + struct.pack("<III", n_vars, n_public, domain_size)
              + _g1(pk["alpha_g1"]) + _g1(pk["beta_g1"]) + _g2(pk["beta_g2"]) + _g2(pk["gamma_g2"])
              + _g1(pk["delta_g1"]) + _g2(pk["delta_g2"]))
    sec[3] = b"".join(_g1(P) for P in pk["gamma_abc_g1"])
    coeffs = []
    R2 = Fr.R * Fr.R % Fr.p
    for k, name in enumerate(("a", "b")):
        for row, lc in enumerate(r1cs[name]):
            for cf, idx in lc:
                coeffs.append(struct.pack("<III", k, row, idx) + (cf * R2 % Fr.p).to_bytes(32, "little"))
    for i in range(n_public + 1):          # snarkjs: one extra row per public signal, A[nc + i][i] = 1
        coeffs.append(struct.pack("<III", 0, nc + i, i) + R2.to_bytes(32, "little"))
    sec[4] = struct.pack("<I", len(coeffs)) + b"".join(coeffs)
    sec[5] = b"".join(_g1(P) for P in pk["a_query"])
    sec[6] = b"".join(_g1(P) for P in pk["b_g1_query"])
    sec[7] = b"".join(_g2(P) for P in pk["b_g2_query"])
    sec[8] = b"".join(_g1(P) for P in pk["l_query"])
    sec[9] = b"".join(_g1(P) for P in pk["h_query"])
    out = b"zkey" + struct.pack("<II", 1, len(sec))
    for sid in sorted(sec):
        out += struct.pack("<IQ", sid, len(sec[sid])) + sec[sid]
    return out
