"""`.r1cs` reader (iden3 binary format v1) -- host-side mirror of the reference's
`R1CSFile::new` / `R1CS::from` (ark-circom/src/circom/r1cs_reader.rs:54-249, r1cs.rs): header checks,
constraint section -> CSR arrays the GPU kernel `dg16_qap` consumes, wire map.

Same error behaviour as the reference: bad magic, version != 1, field size != 32 bytes and a prime other
than the BN254 scalar modulus are rejected (r1cs_reader.rs:58-72,161-188)."""

import struct

import numpy as np

BN254_R_LE = bytes.fromhex("010000f093f5e1439170b97948e833285d588181b64550b829a031e1724e6430")


class R1CSError(ValueError):
    pass


class R1CS:
    """num_inputs counts the constant 1 (arkworks' instance variable 0), like `R1CS::from`."""

    def __init__(self, data: bytes):
        if data[:4] != b"r1cs":
            raise R1CSError("Invalid magic number")
        version, n_sections = struct.unpack_from("<II", data, 4)
        if version != 1:
            raise R1CSError("Unsupported version")
        off = 12
        sections = {}
        for _ in range(n_sections):
            typ, size = struct.unpack_from("<IQ", data, off)
            off += 12
            sections[typ] = (off, size)
            off += size
        if 1 not in sections:
            raise R1CSError("No section offset for header type found")
        hoff, hsize = sections[1]
        (field_size,) = struct.unpack_from("<I", data, hoff)
        if field_size != 32:
            raise R1CSError("This parser only supports 32-byte fields")
        if hsize != 32 + field_size:
            raise R1CSError("Invalid header section size")
        if data[hoff + 4: hoff + 36] != BN254_R_LE:
            raise R1CSError("This parser only supports bn256")
        (self.n_wires, self.n_pub_out, self.n_pub_in, self.n_prv_in, self.n_labels,
         self.n_constraints) = struct.unpack_from("<IIIIQI", data, hoff + 36)
        self.num_inputs = 1 + self.n_pub_in + self.n_pub_out
        self.num_variables = self.n_wires
        self.num_aux = self.n_wires - self.num_inputs
        if 2 not in sections:
            raise R1CSError("No section offset for constraint type found")
        coff, _ = sections[2]
        mats = [([0], [], []) for _ in range(3)]        # (row_ptr, col, coeff bytes) for A, B, C
        p = coff
        for _ in range(self.n_constraints):
            for k in range(3):
                (n_vec,) = struct.unpack_from("<I", data, p)
                p += 4
                row_ptr, col, coeff = mats[k]
                for _ in range(n_vec):
                    (idx,) = struct.unpack_from("<I", data, p)
                    col.append(idx)
                    coeff.append(data[p + 4: p + 36])
                    p += 36
                row_ptr.append(len(col))
        self.csr = []
        for row_ptr, col, coeff in mats:
            c = np.frombuffer(b"".join(coeff), dtype=np.uint64).reshape(-1, 4) if coeff else \
                np.zeros((0, 4), dtype=np.uint64)
            self.csr.append((np.asarray(row_ptr, dtype=np.uint32), np.asarray(col, dtype=np.uint32), c.copy()))
        if 3 in sections:
            moff, msize = sections[3]
            if msize != self.n_wires * 8:
                raise R1CSError("Invalid map section size")
            self.wire_mapping = np.frombuffer(data, dtype=np.uint64, count=self.n_wires, offset=moff).copy()
            if self.wire_mapping[0] != 0:
                raise R1CSError("Wire 0 should always be mapped to 0")
        else:
            self.wire_mapping = None

    @classmethod
    def from_file(cls, path):
        import lzma
        raw = open(path, "rb").read()
        if path.endswith(".xz"):
            raw = lzma.decompress(raw)
        return cls(raw)

    def rows(self, k):
        """Matrix k (0 = A, 1 = B, 2 = C) as a list of [(coeff_int, index)] rows (canonical integers)."""
        row_ptr, col, coeff = self.csr[k]
        ints = [int.from_bytes(coeff[i].tobytes(), "little") for i in range(coeff.shape[0])]
        return [[(ints[j], int(col[j])) for j in range(row_ptr[i], row_ptr[i + 1])]
                for i in range(self.n_constraints)]
