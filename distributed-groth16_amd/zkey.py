"""snarkjs `.zkey` reader (Groth16, BN254) -- host-side mirror of the reference's `read_zkey`
(ark-circom/src/zkey.rs:53-388): section table, Groth16 header, the five point sections that feed the MSMs
and the coefficient section that gives the A / B constraint matrices.

Points in a zkey are x || y little-endian limbs ALREADY in Montgomery form with the identity encoded as
(0, 0) (zkey.rs:342-377) -- exactly libdg16's base layout -- so every query is handed to `dg16_pk_create` as a
zero-copy view of the file.  Matrix coefficients are stored multiplied by R^2 (zkey.rs:332-337): one
Montgomery reduction (`dg16_field_op` from_mont) turns them into the Montgomery-form values `dg16_qap` takes.

Like `BinFile::matrices` (zkey.rs:149-198) the rows above `num_constraints` (the public-input rows snarkjs
appends) are dropped, `num_instance_variables = n_public + 1`, and `num_constraints` is the largest row index
minus `n_public`."""

import struct

import numpy as np

from .r1cs import BN254_R_LE

BN254_Q_LE = bytes.fromhex("47fd7cd8168c203c8dca7168916a81975d588181b64550b829a031e1724e6430")


class ZKeyError(ValueError):
    pass


class ZKey:
    G1 = 64     # bytes per affine point
    G2 = 128

    def __init__(self, data):
        data = memoryview(data)
        if bytes(data[:4]) != b"zkey":
            raise ZKeyError("Invalid magic number")
        self.version, n_sections = struct.unpack_from("<II", data, 4)
        off = 12
        sections = {}
        for _ in range(n_sections):
            if off + 12 > len(data):
                raise ZKeyError("truncated section table")
            sid, size = struct.unpack_from("<IQ", data, off)
            off += 12
            sections.setdefault(sid, (off, size))       # first section of an id wins (zkey.rs:145-147)
            off += size
        if off > len(data):
            raise ZKeyError("truncated file")
        for sid in (1, 2, 3, 4, 5, 6, 7, 8, 9):
            if sid not in sections:
                raise ZKeyError("missing section %d" % sid)
        (protocol,) = struct.unpack_from("<I", data, sections[1][0])
        if protocol != 1:
            raise ZKeyError("not a Groth16 key (protocol %d)" % protocol)

        # ---- section 2: header (zkey.rs:296-330) ----
        p = sections[2][0]
        (n8q,) = struct.unpack_from("<I", data, p)
        if n8q != 32 or bytes(data[p + 4:p + 36]) != BN254_Q_LE:
            raise ZKeyError("base field is not BN254's")
        p += 4 + n8q
        (n8r,) = struct.unpack_from("<I", data, p)
        if n8r != 32 or bytes(data[p + 4:p + 36]) != BN254_R_LE:
            raise ZKeyError("scalar field is not BN254's")
        p += 4 + n8r
        self.n_vars, self.n_public, self.domain_size = struct.unpack_from("<III", data, p)
        p += 12
        if self.domain_size & (self.domain_size - 1):
            raise ZKeyError("domain size is not a power of two")

        def pts(nbytes):
            nonlocal p
            v = np.frombuffer(data, dtype=np.uint64, count=nbytes // 8, offset=p)
            p += nbytes
            return v
        self.alpha_g1, self.beta_g1 = pts(self.G1), pts(self.G1)
        self.beta_g2, self.gamma_g2 = pts(self.G2), pts(self.G2)
        self.delta_g1, self.delta_g2 = pts(self.G1), pts(self.G2)

        def section(sid, count, nbytes):
            o, size = sections[sid]
            if size < count * nbytes:
                raise ZKeyError("section %d too short" % sid)
            return np.frombuffer(data, dtype=np.uint64, count=count * nbytes // 8, offset=o).reshape(count, nbytes // 8)
        nv, npub = self.n_vars, self.n_public
        self.ic = section(3, npub + 1, self.G1)                 # gamma_abc_g1
        self.a_query = section(5, nv, self.G1)
        self.b_g1_query = section(6, nv, self.G1)
        self.b_g2_query = section(7, nv, self.G2)
        self.l_query = section(8, nv - npub - 1, self.G1)
        self.h_query = section(9, self.domain_size, self.G1)

        # ---- section 4: coefficients (zkey.rs:149-198) ----
        o, size = sections[4]
        (n_coeffs,) = struct.unpack_from("<I", data, o)
        if size < 4 + n_coeffs * 44:
            raise ZKeyError("coefficient section too short")
        rec = np.dtype([("matrix", "<u4"), ("row", "<u4"), ("col", "<u4"), ("val", "<u8", 4)])
        co = np.frombuffer(data, dtype=rec, count=n_coeffs, offset=o + 4)
        if n_coeffs and (co["matrix"].max() > 1 or co["row"].max() >= self.domain_size):
            raise ZKeyError("coefficient out of range")
        max_row = int(co["row"].max()) if n_coeffs else 0
        self.num_constraints = max_row - npub
        self.num_instance_variables = npub + 1
        self.num_witness_variables = nv - npub                   # as the reference reports it (zkey.rs:186)
        self._csr_r2 = []
        for k in (0, 1):
            sel = co[(co["matrix"] == k) & (co["row"] < self.num_constraints)]
            order = np.argsort(sel["row"], kind="stable")        # file order inside a row, like the push() loop
            sel = sel[order]
            ptr = np.zeros(self.num_constraints + 1, dtype=np.uint32)
            np.add.at(ptr, sel["row"].astype(np.int64) + 1, 1)
            self._csr_r2.append((np.cumsum(ptr, dtype=np.uint32), np.ascontiguousarray(sel["col"]),
                                 np.ascontiguousarray(sel["val"])))

    @classmethod
    def from_file(cls, path):
        with open(path, "rb") as f:
            return cls(f.read())

    def matrices(self, ctx):
        """CSR (row_ptr, col, coeff) of A and B with Montgomery-form coefficients, ready for `Context.qap`."""
        out = []
        for ptr, col, val in self._csr_r2:
            mont = ctx.field_op("bn254", "fr", "from_mont", val) if len(val) else val    # v R^2 -> v R
            out.append((ptr, col, mont))
        return out

    def proving_key(self, ctx, shard=0, n_shards=1):
        """Resident GPU key (`dg16_pk_create`): the queries go over as they lie in the file."""
        fixed = np.concatenate([self.alpha_g1, self.beta_g1, self.delta_g1, self.beta_g2, self.delta_g2])
        return ctx.pk_create("bn254", self.n_vars, self.n_public + 1, self.domain_size, self.a_query,
                             self.b_g1_query, self.b_g2_query, self.h_query, self.l_query, fixed,
                             shard=shard, n_shards=n_shards)
