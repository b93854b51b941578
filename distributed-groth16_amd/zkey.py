"""snarkjs `.zkey` reader (Groth16, BN254) -- thin wrapper over the native reader of libdg16 (`dg16_zkey_parse`,
csrc/formats.hip), the counterpart of the reference's `read_zkey` (ark-circom/src/zkey.rs:53-388): section table,
Groth16 header, the five point sections that feed the MSMs and the coefficient section that gives the A / B
constraint matrices.

Points in a zkey are x || y little-endian limbs ALREADY in Montgomery form with the identity encoded as
(0, 0) (zkey.rs:342-377) -- exactly libdg16's base layout -- so every query is handed to `dg16_pk_create` as a
zero-copy view of the file.  Matrix coefficients are stored multiplied by R^2 (zkey.rs:332-337): one
Montgomery reduction (`dg16_field_op` from_mont) turns them into the Montgomery-form values `dg16_qap` takes.

Like `BinFile::matrices` (zkey.rs:149-198) the rows above `num_constraints` (the public-input rows snarkjs
appends) are dropped, `num_instance_variables = n_public + 1`, and `num_constraints` is the largest row index
minus `n_public`."""

import ctypes

import numpy as np

from . import lib as _lib

SECTIONS = {"alpha_g1": 0, "beta_g1": 1, "beta_g2": 2, "gamma_g2": 3, "delta_g1": 4, "delta_g2": 5, "ic": 6,
            "a_query": 7, "b_g1_query": 8, "b_g2_query": 9, "l_query": 10, "h_query": 11}     # DG16_ZKEY_*
G2_SECTIONS = ("beta_g2", "gamma_g2", "delta_g2", "b_g2_query")


class ZKeyError(ValueError):
    pass


class ZKey:
    def __init__(self, data):
        L = _lib.load()
        self._data = bytes(data)                 # the native handle points into this buffer
        self._buf = np.frombuffer(self._data, dtype=np.uint8)
        base = self._buf.ctypes.data
        h = ctypes.c_void_p()
        if L.dg16_zkey_parse(ctypes.c_void_p(base), len(self._data), ctypes.byref(h)) != 0:
            raise ZKeyError(L.dg16_io_error().decode())
        try:
            hd = _lib.ZkeyHeader()
            L.dg16_zkey_header_get(h, ctypes.byref(hd))
            self.n_vars, self.n_public, self.domain_size = hd.n_vars, hd.n_public, hd.domain_size
            self.num_constraints = hd.num_constraints
            self.num_instance_variables = self.n_public + 1
            self.num_witness_variables = self.n_vars - self.n_public     # as the reference reports it (zkey.rs:186)
            for name, which in SECTIONS.items():
                ptr, cnt = ctypes.c_void_p(), ctypes.c_size_t()
                L.dg16_zkey_points(h, which, ctypes.byref(ptr), ctypes.byref(cnt))
                words = 16 if name in G2_SECTIONS else 8
                off = (ptr.value or base) - base
                view = self._buf[off:off + cnt.value * words * 8].view(np.uint64)    # zero-copy view of the file
                setattr(self, name, view.reshape(cnt.value, words) if which >= 6 else view)
            self._csr_r2 = []
            for k in (0, 1):
                m = _lib.Csr()
                L.dg16_zkey_matrix(h, k, ctypes.byref(m))
                self._csr_r2.append(m.arrays())
        finally:
            L.dg16_zkey_free(h)

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
