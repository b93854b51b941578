"""`.r1cs` reader (iden3 binary format v1) -- thin wrapper over the native reader of libdg16
(`dg16_r1cs_parse`, csrc/formats.hip), the counterpart of the reference's `R1CSFile::new` / `R1CS::from`
(ark-circom/src/circom/r1cs_reader.rs:54-249, r1cs.rs): header checks, constraint section -> CSR arrays the GPU
kernel `dg16_qap` consumes, wire map.

Same error behaviour as the reference: bad magic, version != 1, field size != 32 bytes and a prime other
than the BN254 scalar modulus are rejected (r1cs_reader.rs:58-72,161-188)."""

import ctypes

import numpy as np

from . import lib as _lib

BN254_R_LE = bytes.fromhex("010000f093f5e1439170b97948e833285d588181b64550b829a031e1724e6430")


class R1CSError(ValueError):
    pass


class R1CS:
    """num_inputs counts the constant 1 (arkworks' instance variable 0), like `R1CS::from`."""

    def __init__(self, data: bytes):
        L = _lib.load()
        data = bytes(data)
        h = ctypes.c_void_p()
        if L.dg16_r1cs_parse(data, len(data), ctypes.byref(h)) != 0:
            raise R1CSError(L.dg16_io_error().decode())
        try:
            hd = _lib.R1csHeader()
            L.dg16_r1cs_header_get(h, ctypes.byref(hd))
            self.n_wires, self.n_pub_out, self.n_pub_in = hd.n_wires, hd.n_pub_out, hd.n_pub_in
            self.n_prv_in, self.n_labels, self.n_constraints = hd.n_prv_in, hd.n_labels, hd.n_constraints
            self.num_inputs = 1 + self.n_pub_in + self.n_pub_out
            self.num_variables = self.n_wires
            self.num_aux = self.n_wires - self.num_inputs
            self.csr = []
            for k in range(3):
                m = _lib.Csr()
                L.dg16_r1cs_matrix(h, k, ctypes.byref(m))
                self.csr.append(m.arrays())
            self.wire_mapping = None
            if hd.has_wire_map:
                p = ctypes.POINTER(ctypes.c_uint64)()
                L.dg16_r1cs_wire_map(h, ctypes.byref(p))
                self.wire_mapping = np.ctypeslib.as_array(p, shape=(self.n_wires,)).copy()
        finally:
            L.dg16_r1cs_free(h)

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
