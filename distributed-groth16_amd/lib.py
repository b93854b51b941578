"""ctypes binding of libdg16.so (C ABI: include/dg16.h).  Mirrors the reference-side seams:
`Context.msm` <- G::msm (dist-primitives/src/dmsm/mod.rs:82), `Context.ntt` <- Radix2 fft/ifft
(ark-circom/src/circom/qap.rs:64-85), `Context.h_poly` <- witness_map_from_matrices (qap.rs:64-91).

Arrays cross this boundary as numpy uint64 arrays (host) or raw device pointers (ints, e.g.
`torch.Tensor.data_ptr()`), in the byte layout documented in include/dg16.h.
"""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CURVES = {"bn254": 0, "bls12_381": 1, "bls12_377": 2}
FQ_LIMBS64 = {"bn254": 4, "bls12_381": 6, "bls12_377": 6}

F_SCALARS_MONT = 1
F_DEVICE_PTRS = 2
F_OUT_AFFINE = 4
F_H_CYCLIC = 8
F_SERIAL_CHANNELS = 16
F_OVERLAP_TAIL = 32
F_BASES_IN_SUBGROUP = 64

STATUS = {1: "LENGTH_MISMATCH", 2: "BAD_CURVE", 3: "BAD_ARG", 4: "OOM", 5: "HIP", 6: "NET",
          7: "UNSUPPORTED"}


class Dg16Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dg16 status %d (%s): %s" % (code, STATUS.get(code, "?"), msg))
        self.code = code


class R1csHeader(ctypes.Structure):       # dg16_r1cs_header
    _fields_ = [(n, ctypes.c_uint32) for n in ("n_wires", "n_pub_out", "n_pub_in", "n_prv_in", "n_constraints",
                                                "has_wire_map")] + [("n_labels", ctypes.c_uint64)]


class ZkeyHeader(ctypes.Structure):       # dg16_zkey_header
    _fields_ = [(n, ctypes.c_uint32) for n in ("n_vars", "n_public", "domain_size", "num_constraints")]


class PkInfo(ctypes.Structure):           # dg16_pk_info
    _fields_ = [("n_ab", ctypes.c_uint64), ("n_l", ctypes.c_uint64), ("n_h", ctypes.c_uint64),
                ("c_ab", ctypes.c_uint32), ("c_l", ctypes.c_uint32), ("c_h", ctypes.c_uint32),
                ("shard", ctypes.c_uint32), ("n_shards", ctypes.c_uint32), ("table_bytes", ctypes.c_uint64),
                ("table_stride", ctypes.c_uint32)]


class Csr(ctypes.Structure):              # dg16_csr
    _fields_ = [("n_rows", ctypes.c_uint64), ("nnz", ctypes.c_uint64), ("row_ptr", ctypes.POINTER(ctypes.c_uint32)),
                ("col", ctypes.POINTER(ctypes.c_uint32)), ("coeff", ctypes.c_void_p)]

    def arrays(self):
        """Copies out (row_ptr uint32[n_rows + 1], col uint32[nnz], coeff uint64[nnz][4])."""
        n, nnz = int(self.n_rows), int(self.nnz)
        ptr = np.ctypeslib.as_array(self.row_ptr, shape=(n + 1,)).copy() if n or nnz else np.zeros(1, dtype=np.uint32)
        if nnz == 0:
            return ptr, np.zeros(0, dtype=np.uint32), np.zeros((0, 4), dtype=np.uint64)
        col = np.ctypeslib.as_array(self.col, shape=(nnz,)).copy()
        raw = (ctypes.c_uint64 * (4 * nnz)).from_address(self.coeff)
        return ptr, col, np.frombuffer(raw, dtype=np.uint64).reshape(nnz, 4).copy()


_COMM_N = ctypes.CFUNCTYPE(ctypes.c_uint, ctypes.c_void_p)
_COMM_GATHER = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                ctypes.c_void_p)
_COMM_A2A = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                             ctypes.c_void_p)


class ArkKeyLayout(ctypes.Structure):     # dg16_arkkey_layout_t
    _fields_ = [(n, ctypes.c_uint64) for n in
                ("n_ic", "n_a", "n_b1", "n_b2", "n_h", "n_l", "off_alpha_g1", "off_beta_g2", "off_gamma_g2", "off_delta_g2",
                 "off_ic", "off_beta_g1", "off_delta_g1", "off_a", "off_b1", "off_b2", "off_h", "off_l", "bytes")]


class CommStruct(ctypes.Structure):       # dg16_comm
    _fields_ = [("self", ctypes.c_void_p), ("n_ranks", _COMM_N), ("rank", _COMM_N), ("all_gather", _COMM_GATHER),
                ("all_to_all", _COMM_A2A)]


_NET_COLL = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                             ctypes.c_void_p, ctypes.c_void_p)
_NET_P2P = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p,
                            ctypes.c_size_t, ctypes.c_void_p)


class NetStruct(ctypes.Structure):        # dg16_net (MpcNet: mpc-net/src/lib.rs:36-140)
    _fields_ = [("self", ctypes.c_void_p), ("n_parties", _COMM_N), ("party_id", _COMM_N),
                ("gather_to_king", _NET_COLL), ("scatter_from_king", _NET_COLL),
                ("is_init", ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)), ("send_to", _NET_P2P),
                ("recv_from", _NET_P2P)]


def lib_path():
    # DG16_LIB: an explicitly named build of the same library (A/B timing of two builds on one box)
    return os.environ.get("DG16_LIB") or os.path.join(_HERE, "libdg16.so")


_lib = None


def load():
    """Loads libdg16.so; raises (never falls back) if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: torch ships its own libamdhip64; if libdg16 (linked against /opt/rocm)
    # initialises first, torch.cuda later reports "No HIP GPUs are available".  Let torch load first when
    # it is importable (it is only plumbing for device memory / streams / torch.distributed).
    try:
        import torch
        torch.cuda.is_available()
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "libdg16.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C distributed-groth16_amd/csrc`.  There is no CPU fallback." % path)
    L = ctypes.CDLL(path)
    vp, sz, i, u, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint, ctypes.c_uint64
    L.dg16_ctx_create.argtypes = [i, ctypes.POINTER(vp)]
    L.dg16_ctx_destroy.argtypes = [vp]
    L.dg16_ctx_destroy.restype = None
    L.dg16_last_error.argtypes = [vp]
    L.dg16_last_error.restype = ctypes.c_char_p
    L.dg16_set_stream.argtypes = [vp, i, vp]
    L.dg16_sync.argtypes = [vp, i]
    L.dg16_device_info.argtypes = [vp, ctypes.c_char_p, sz, ctypes.POINTER(i)]
    L.dg16_field_op.argtypes = [vp, i, i, vp, vp, vp, sz, u, i]
    L.dg16_ntt.argtypes = [vp, i, vp, u, i, vp, u, i]
    L.dg16_h_poly.argtypes = [vp, i, vp, vp, vp, u, vp, u, i]
    L.dg16_msm.argtypes = [vp, i, i, vp, vp, sz, sz, u, i, vp]
    L.dg16_gen_bases.argtypes = [vp, i, i, u64, sz, vp, u, i]
    if hasattr(L, "dg16_ctx_set_table_budget"):      # (absent from older builds named by DG16_LIB for A/B timing)
        L.dg16_ctx_set_table_budget.argtypes = [vp, u64]
    L.dg16_bases_upload.argtypes = [vp, i, i, vp, sz, u, ctypes.POINTER(vp)]
    L.dg16_bases_free.argtypes = [vp]
    L.dg16_bases_free.restype = None
    L.dg16_bases_info.argtypes = [vp, ctypes.POINTER(sz), ctypes.POINTER(u), ctypes.POINTER(u64)]
    L.dg16_msm_resident.argtypes = [vp, vp, vp, sz, u, i, vp]
    L.dg16_d_msm_resident.argtypes = [vp, vp, vp, vp, vp, sz, u, i, vp]
    L.dg16_to_affine.argtypes = [vp, i, i, vp, vp, sz, u, i]
    L.dg16_last_kernel_ms.argtypes = [vp, i, i, ctypes.POINTER(ctypes.c_float)]
    L.dg16_pk_create.argtypes = [vp, i, sz, sz, sz, vp, vp, vp, vp, vp, vp, u, ctypes.POINTER(vp)]
    L.dg16_pk_destroy.argtypes = [vp]
    L.dg16_pk_destroy.restype = None
    L.dg16_pk_info_get.argtypes = [vp, ctypes.POINTER(PkInfo)]
    L.dg16_groth16_prove.argtypes = [vp, vp, vp, vp, vp, vp, vp, u, vp]
    L.dg16_pk_create_shard.argtypes = [vp, i, sz, sz, sz, vp, vp, vp, vp, vp, vp, u, u, u, ctypes.POINTER(vp)]
    L.dg16_groth16_results_bytes.argtypes = [i]
    L.dg16_groth16_results_bytes.restype = sz
    L.dg16_groth16_msms.argtypes = [vp, vp, vp, vp, vp, vp, vp, u, vp]
    L.dg16_groth16_assemble.argtypes = [vp, vp, vp, sz, vp, u, vp]
    L.dg16_qap.argtypes = [vp, i, sz, sz, sz, u, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u, i]
    L.dg16_qap_rows.argtypes = [vp, i, sz, sz, sz, u, vp, vp, vp, vp, vp, vp, vp, sz, sz, vp, vp, vp, u, i]
    L.dg16_h_poly_dist.argtypes = [vp, i, vp, vp, vp, vp, u, vp, u, i]
    L.dg16_ntt_dist.argtypes = [vp, i, vp, vp, vp, u, i, u, i]
    L.dg16_ntt_dist_stage.argtypes = [vp, i, u, u, u, i, i, vp, vp, u, i]
    L.dg16_h_poly_dist_stage.argtypes = [vp, i, u, u, u, i, ctypes.POINTER(vp), vp, u, i]
    L.dg16_groth16_msms_h.argtypes = [vp, vp, vp, vp, vp, u, vp]
    L.dg16_groth16_prove_dist.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, u, vp]
    L.dg16_codec_error.argtypes = []
    L.dg16_codec_error.restype = ctypes.c_char_p
    L.dg16_arkkey_layout.argtypes = [vp, sz, i, ctypes.POINTER(ArkKeyLayout)]
    L.dg16_points_compress.argtypes = [vp, i, i, vp, sz, vp, u, i]
    L.dg16_points_decompress.argtypes = [vp, i, i, vp, sz, i, vp, u, i]
    L.dg16_wire_fr_bytes.argtypes = [sz]
    L.dg16_wire_fr_bytes.restype = sz
    L.dg16_wire_fr_encode.argtypes = [vp, i, vp, sz, vp, u, i]
    L.dg16_wire_fr_decode.argtypes = [vp, i, vp, sz, vp, ctypes.POINTER(sz), u, i]
    L.dg16_rccl_unique_id.argtypes = [vp]
    L.dg16_rccl_create.argtypes = [vp, vp, u, u, ctypes.POINTER(vp)]
    L.dg16_rccl_ranks.argtypes = [vp, ctypes.POINTER(u), ctypes.POINTER(u)]
    if hasattr(L, "dg16_rccl_channels_split"):          # (absent from A/B builds made before round 4: DG16_LIB)
        L.dg16_rccl_channels_split.argtypes = [vp]
    L.dg16_rccl_comm.argtypes = [vp]
    L.dg16_rccl_comm.restype = vp
    L.dg16_rccl_net.argtypes = [vp]
    L.dg16_rccl_net.restype = vp
    L.dg16_rccl_destroy.argtypes = [vp]
    L.dg16_rccl_destroy.restype = None
    L.dg16_rccl_error.argtypes = []
    L.dg16_rccl_error.restype = ctypes.c_char_p
    L.dg16_localnet_create.argtypes = [u, ctypes.POINTER(vp)]
    L.dg16_localnet_party.argtypes = [vp, u]
    L.dg16_localnet_party.restype = vp
    L.dg16_localnet_destroy.argtypes = [vp]
    L.dg16_localnet_destroy.restype = None
    L.dg16_pss_create.argtypes = [vp, i, u, ctypes.POINTER(vp)]
    L.dg16_pss_destroy.argtypes = [vp]
    L.dg16_pss_destroy.restype = None
    L.dg16_pss_apply.argtypes = [vp, vp, i, vp, sz, vp, u, i]
    L.dg16_pss_apply_exp.argtypes = [vp, vp, i, i, vp, sz, vp, u, i]
    L.dg16_d_fft.argtypes = [vp, vp, vp, vp, sz, u, i, u, i, i, vp, u, i]
    L.dg16_localnet_abort.argtypes = [vp]
    L.dg16_localnet_abort.restype = None
    L.dg16_localnet_reset.argtypes = [vp, u]
    L.dg16_localnet_reset.restype = None
    L.dg16_d_msm.argtypes = [vp, vp, vp, i, vp, vp, sz, sz, u, i, vp]
    L.dg16_deg_red.argtypes = [vp, vp, vp, vp, sz, vp, u, i]
    L.dg16_d_pp.argtypes = [vp, vp, vp, vp, vp, sz, vp, u, i]
    L.dg16_ext_wit_h.argtypes = [vp, vp, vp, vp, vp, vp, u, vp, u]
    L.dg16_prove_a.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz, sz, u, i, vp]
    L.dg16_prove_b.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz, sz, u, i, vp]
    L.dg16_prove_c.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, sz, vp, vp, sz, sz, vp, vp, sz, sz, u, vp]
    # file-format readers (host side of the library)
    L.dg16_io_error.argtypes = []
    L.dg16_io_error.restype = ctypes.c_char_p
    L.dg16_r1cs_parse.argtypes = [vp, sz, ctypes.POINTER(vp)]
    L.dg16_r1cs_header_get.argtypes = [vp, ctypes.POINTER(R1csHeader)]
    L.dg16_r1cs_matrix.argtypes = [vp, i, ctypes.POINTER(Csr)]
    L.dg16_r1cs_wire_map.argtypes = [vp, ctypes.POINTER(ctypes.POINTER(u64))]
    L.dg16_r1cs_free.argtypes = [vp]
    L.dg16_r1cs_free.restype = None
    L.dg16_zkey_parse.argtypes = [vp, sz, ctypes.POINTER(vp)]
    L.dg16_zkey_header_get.argtypes = [vp, ctypes.POINTER(ZkeyHeader)]
    L.dg16_zkey_points.argtypes = [vp, i, ctypes.POINTER(vp), ctypes.POINTER(sz)]
    L.dg16_zkey_matrix.argtypes = [vp, i, ctypes.POINTER(Csr)]
    L.dg16_zkey_free.argtypes = [vp]
    L.dg16_zkey_free.restype = None
    L.dg16_serialize_error.argtypes = []
    L.dg16_serialize_error.restype = ctypes.c_char_p
    L.dg16_proof_compress.argtypes = [i, vp, vp]
    L.dg16_proof_decompress.argtypes = [i, vp, i, vp]
    L.dg16_verify_error.argtypes = []
    L.dg16_verify_error.restype = ctypes.c_char_p
    L.dg16_groth16_verify.argtypes = [i, vp, vp, vp, vp, vp, sz, vp, sz, vp, u, ctypes.POINTER(i)]
    _lib = L
    return L


EXPORTED = ["dg16_ctx_create", "dg16_ctx_destroy", "dg16_last_error", "dg16_set_stream", "dg16_sync",
            "dg16_device_info", "dg16_field_op", "dg16_ntt", "dg16_h_poly", "dg16_msm", "dg16_pk_info_get",
            "dg16_gen_bases", "dg16_to_affine", "dg16_last_kernel_ms", "dg16_pk_create", "dg16_pk_destroy",
            "dg16_groth16_prove", "dg16_pk_create_shard", "dg16_groth16_results_bytes", "dg16_groth16_msms",
            "dg16_groth16_assemble", "dg16_localnet_create", "dg16_localnet_party", "dg16_localnet_destroy", "dg16_localnet_abort",
            "dg16_localnet_reset",
            "dg16_pss_create", "dg16_pss_destroy", "dg16_pss_apply", "dg16_pss_apply_exp", "dg16_d_fft",
            "dg16_d_msm", "dg16_deg_red", "dg16_d_pp", "dg16_ext_wit_h", "dg16_qap", "dg16_qap_rows",
            "dg16_h_poly_dist", "dg16_h_poly_dist_stage", "dg16_ntt_dist", "dg16_ntt_dist_stage", "dg16_groth16_msms_h", "dg16_groth16_prove_dist",
            "dg16_rccl_unique_id", "dg16_rccl_create", "dg16_rccl_comm", "dg16_rccl_net", "dg16_rccl_destroy", "dg16_rccl_ranks",
            "dg16_rccl_channels_split",
            "dg16_rccl_error", "dg16_bases_upload", "dg16_bases_free", "dg16_bases_info", "dg16_msm_resident",
            "dg16_d_msm_resident", "dg16_codec_error", "dg16_arkkey_layout", "dg16_points_compress",
            "dg16_points_decompress", "dg16_wire_fr_bytes", "dg16_wire_fr_encode", "dg16_wire_fr_decode",
            "dg16_io_error", "dg16_r1cs_parse", "dg16_r1cs_header_get", "dg16_r1cs_matrix", "dg16_r1cs_wire_map",
            "dg16_r1cs_free", "dg16_zkey_parse", "dg16_zkey_header_get", "dg16_zkey_points", "dg16_zkey_matrix",
            "dg16_zkey_free", "dg16_serialize_error", "dg16_proof_compress", "dg16_proof_decompress",
            "dg16_verify_error", "dg16_groth16_verify", "dg16_prove_a", "dg16_prove_b", "dg16_prove_c",
            "dg16_ctx_set_table_budget"]


def _ptr(x):
    """numpy array -> host pointer; int -> raw (device) pointer; None -> NULL."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data_as(ctypes.c_void_p)
    return ctypes.c_void_p(int(x))


class ResidentBases:
    """dg16_bases: a base vector resident in HBM as its table of window multiples."""

    def __init__(self, ctx, handle, curve, group, n):
        self.ctx, self.h, self.curve, self.group, self.n = ctx, handle, curve, group, n

    def info(self):
        n, c, b = ctypes.c_size_t(), ctypes.c_uint(), ctypes.c_uint64()
        self.ctx._chk(self.ctx.L.dg16_bases_info(self.h, ctypes.byref(n), ctypes.byref(c), ctypes.byref(b)))
        return {"n": n.value, "window_bits": c.value, "table_bytes": b.value}

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.L.dg16_bases_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ProvingKey:
    """Resident Groth16 proving key (dg16_pk)."""

    def __init__(self, ctx, handle, curve, num_vars, num_inputs, domain_size):
        self.ctx, self.h, self.curve = ctx, handle, curve
        self.num_vars, self.num_inputs, self.domain_size = num_vars, num_inputs, domain_size

    def info(self):
        """dg16_pk_info as a dict (window bits, points per launch, table bytes)."""
        i = PkInfo()
        self.ctx._chk(self.ctx.L.dg16_pk_info_get(self.h, ctypes.byref(i)))
        return {n: getattr(i, n) for n, _ in PkInfo._fields_}

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.L.dg16_pk_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One per process and GPU (dg16_ctx).  Host-array methods are synchronous; `*_dev` methods take
    device pointers and are stream-ordered on the channel's stream."""

    def __init__(self, device=0):
        self.L = load()
        h = ctypes.c_void_p()
        rc = self.L.dg16_ctx_create(device, ctypes.byref(h))
        if rc != 0:
            raise Dg16Error(rc, "dg16_ctx_create failed (no GPU visible? libdg16 has no CPU path)")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.dg16_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise Dg16Error(rc, self.L.dg16_last_error(self.h).decode())

    # ---- plumbing ------------------------------------------------------------------------------
    def set_stream(self, channel, stream_ptr):
        self._chk(self.L.dg16_set_stream(self.h, channel, ctypes.c_void_p(stream_ptr or 0)))

    def set_table_budget(self, nbytes):
        """HBM budget of the window tables of one resident key / base set built from now on (0 = unlimited)."""
        self._chk(self.L.dg16_ctx_set_table_budget(self.h, int(nbytes)))
        self.table_budget = int(nbytes)

    def sync(self, channel=0):
        self._chk(self.L.dg16_sync(self.h, channel))

    def device_info(self):
        buf = ctypes.create_string_buffer(64)
        cu = ctypes.c_int()
        self._chk(self.L.dg16_device_info(self.h, buf, 64, ctypes.byref(cu)))
        return buf.value.decode(), cu.value

    def last_kernel_mhz(self, channel=0):
        """Shader clock (MHz) under the bucket accumulation last_kernel_ms(channel, 1) timed, measured by the kernel itself."""
        return self.last_kernel_ms(channel, 2)

    def last_kernel_ms(self, channel=0, which=1):
        ms = ctypes.c_float()
        self._chk(self.L.dg16_last_kernel_ms(self.h, channel, which, ctypes.byref(ms)))
        return ms.value

    # ---- host-array API ---------------------------------------------------------------------------
    def field_op(self, curve, kind, op, a, b=None, channel=0):
        from_ops = {"add": 0, "sub": 1, "mul": 2, "sqr": 3, "inv": 4, "to_mont": 5, "from_mont": 6, "neg": 7}
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = None if b is None else np.ascontiguousarray(b, dtype=np.uint64)
        out = np.empty_like(a)
        fid = CURVES[curve] + (16 if kind == "fr" else 0)
        self._chk(self.L.dg16_field_op(self.h, fid, from_ops[op], _ptr(a), _ptr(b), _ptr(out), a.shape[0], 0,
                                       channel))
        return out

    def ntt(self, curve, data, inverse=False, coset=None, channel=0):
        data = np.array(data, dtype=np.uint64, copy=True)
        n = data.shape[0]
        log_n = n.bit_length() - 1
        if n == 0 or (1 << log_n) != n:
            raise ValueError("NTT size must be a power of two")
        coset = None if coset is None else np.ascontiguousarray(coset, dtype=np.uint64)
        self._chk(self.L.dg16_ntt(self.h, CURVES[curve], _ptr(data), log_n, int(inverse), _ptr(coset), 0, channel))
        return data

    def h_poly(self, curve, a, b, c, channel=0):
        a, b, c = (np.ascontiguousarray(v, dtype=np.uint64) for v in (a, b, c))
        m = a.shape[0]
        log_m = m.bit_length() - 1
        if (1 << log_m) != m or b.shape != a.shape or c.shape != a.shape:
            raise ValueError("a, b, c must have equal power-of-two length")
        out = np.empty_like(a)
        self._chk(self.L.dg16_h_poly(self.h, CURVES[curve], _ptr(a), _ptr(b), _ptr(c), log_m, _ptr(out), 0,
                                     channel))
        return out

    def msm(self, curve, group, bases, scalars, scalars_mont=False, affine=False, channel=0, in_subgroup=False):
        """in_subgroup: DG16_F_BASES_IN_SUBGROUP -- the caller's word that the bases are elements of the order-r group
        (an arkworks G1Affine / G2Affine that came through Validate::Yes or CRS generation is), so the library may split
        the scalars with the curve's endomorphism (half the Horner tail: 15-20 % of a plain 2^20 MSM).  The default is the
        C ABI's: flag NOT set, plain Pippenger for every group of cofactor != 1, the group element VariableBaseMSM::msm
        gives for ANY points of the curve (decoded with validate = 0, say).  BN254 G1 (cofactor one) splits either way."""
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        nl = FQ_LIMBS64[curve] * (2 if group == 2 else 1)
        out = np.zeros((1, nl * (2 if affine else 3)), dtype=np.uint64)
        flags = ((F_SCALARS_MONT if scalars_mont else 0) | (F_OUT_AFFINE if affine else 0) |
                 (F_BASES_IN_SUBGROUP if in_subgroup else 0))
        self._chk(self.L.dg16_msm(self.h, CURVES[curve], group, _ptr(bases), _ptr(scalars), bases.shape[0],
                                  scalars.shape[0], flags, channel, _ptr(out)))
        return out

    def bases_upload(self, curve, group, bases, n=None, device_ptrs=False):
        """bases: numpy array of affine points (host) or a raw device pointer with n given."""
        h = ctypes.c_void_p()
        if not device_ptrs:
            bases = np.ascontiguousarray(bases, dtype=np.uint64)
            n = bases.shape[0]
        self._chk(self.L.dg16_bases_upload(self.h, CURVES[curve], group, _ptr(bases), n,
                                           F_DEVICE_PTRS if device_ptrs else 0, ctypes.byref(h)))
        return ResidentBases(self, h, curve, group, n)

    def msm_resident(self, hb, scalars, scalars_mont=False, affine=False, channel=0):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        nl = FQ_LIMBS64[hb.curve] * (2 if hb.group == 2 else 1)
        out = np.zeros((1, nl * (2 if affine else 3)), dtype=np.uint64)
        flags = (F_SCALARS_MONT if scalars_mont else 0) | (F_OUT_AFFINE if affine else 0)
        self._chk(self.L.dg16_msm_resident(self.h, hb.h, _ptr(scalars), scalars.shape[0], flags, channel, _ptr(out)))
        return out

    def msm_resident_dev(self, hb, scalars_ptr, n, out_ptr, scalars_mont=False, affine=False, channel=0):
        flags = F_DEVICE_PTRS | (F_SCALARS_MONT if scalars_mont else 0) | (F_OUT_AFFINE if affine else 0)
        self._chk(self.L.dg16_msm_resident(self.h, hb.h, _ptr(scalars_ptr), n, flags, channel, _ptr(out_ptr)))

    @staticmethod
    def _coord_bytes(curve):
        return 32 if curve == "bn254" else 48

    def points_compress(self, curve, group, affine, channel=0):
        """affine: uint64 array [n][2 * limbs * group] (x || y Montgomery limbs; limbs = 4 for BN254, 6 for BLS12-377)
        -> bytes, one coordinate (32 / 48 bytes, x2 for G2) per point (arkworks Compress::Yes)."""
        fb = self._coord_bytes(curve)
        affine = np.ascontiguousarray(affine, dtype=np.uint64).reshape(-1, fb // 4 * group)
        out = np.zeros(affine.shape[0] * fb * group, dtype=np.uint8)
        self._chk(self.L.dg16_points_compress(self.h, CURVES[curve], group, _ptr(affine), affine.shape[0], _ptr(out), 0,
                                              channel))
        return out.tobytes()

    def points_decompress(self, curve, group, raw, validate=False, channel=0):
        fb = self._coord_bytes(curve)
        raw = np.frombuffer(bytes(raw), dtype=np.uint8)
        n = raw.size // (fb * group)
        out = np.zeros((n, fb // 4 * group), dtype=np.uint64)
        self._chk(self.L.dg16_points_decompress(self.h, CURVES[curve], group, _ptr(raw), n, int(validate), _ptr(out), 0,
                                                channel))
        return out

    def points_decompress_dev(self, curve, group, in_ptr, n, out_ptr, validate=False, channel=0):
        self._chk(self.L.dg16_points_decompress(self.h, CURVES[curve], group, _ptr(in_ptr), n, int(validate),
                                                _ptr(out_ptr), F_DEVICE_PTRS, channel))

    def points_compress_dev(self, curve, group, in_ptr, n, out_ptr, channel=0):
        self._chk(self.L.dg16_points_compress(self.h, CURVES[curve], group, _ptr(in_ptr), n, _ptr(out_ptr),
                                              F_DEVICE_PTRS, channel))

    def wire_fr_encode(self, curve, mont, channel=0):
        """Montgomery Fr elements -> ark-serialize compressed Vec<F> bytes (u64 length || canonical elements)."""
        mont = np.ascontiguousarray(mont, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(self.L.dg16_wire_fr_bytes(mont.shape[0]), dtype=np.uint8)
        self._chk(self.L.dg16_wire_fr_encode(self.h, CURVES[curve], _ptr(mont), mont.shape[0], _ptr(out), 0, channel))
        return out.tobytes()

    def wire_fr_decode(self, curve, raw, channel=0):
        raw = np.frombuffer(bytes(raw), dtype=np.uint8)
        out = np.zeros((max(raw.size - 8, 0) // 32, 4), dtype=np.uint64)
        n = ctypes.c_size_t()
        self._chk(self.L.dg16_wire_fr_decode(self.h, CURVES[curve], _ptr(raw), raw.size, _ptr(out), ctypes.byref(n), 0,
                                             channel))
        return out[:n.value]

    def gen_bases(self, curve, group, seed, n, channel=0):
        nl = FQ_LIMBS64[curve] * 2 * (2 if group == 2 else 1)
        out = np.zeros((n, nl), dtype=np.uint64)
        self._chk(self.L.dg16_gen_bases(self.h, CURVES[curve], group, seed, n, _ptr(out), 0, channel))
        return out

    def to_affine(self, curve, group, jac, channel=0):
        jac = np.ascontiguousarray(jac, dtype=np.uint64)
        nl = FQ_LIMBS64[curve] * (2 if group == 2 else 1)
        jac = jac.reshape(-1, 3 * nl)
        out = np.zeros((jac.shape[0], 2 * nl), dtype=np.uint64)
        self._chk(self.L.dg16_to_affine(self.h, CURVES[curve], group, _ptr(jac), _ptr(out), jac.shape[0], 0,
                                        channel))
        return out

    # ---- QAP (R1CS x witness) ---------------------------------------------------------------------------
    def qap(self, curve, num_constraints, num_inputs, csr_a, csr_b, full_assignment, scalars_mont=True, channel=0):
        """csr_* = (row_ptr uint32, col uint32, coeff uint64[nnz][4] in Montgomery form).  Returns a, b, c."""
        w = np.ascontiguousarray(full_assignment, dtype=np.uint64).reshape(-1, 4)
        need = num_constraints + num_inputs
        log_m = max(need - 1, 0).bit_length()
        m = 1 << log_m
        out = [np.zeros((m, 4), dtype=np.uint64) for _ in range(3)]
        ap, ac, av = (np.ascontiguousarray(x) for x in csr_a)
        bp, bc, bv = (np.ascontiguousarray(x) for x in csr_b)
        self._chk(self.L.dg16_qap(self.h, CURVES[curve], num_constraints, num_inputs, w.shape[0], log_m, _ptr(ap),
                                  _ptr(ac), _ptr(av), _ptr(bp), _ptr(bc), _ptr(bv), _ptr(w), _ptr(out[0]),
                                  _ptr(out[1]), _ptr(out[2]), F_SCALARS_MONT if scalars_mont else 0, channel))
        return out

    def qap_dev(self, curve, num_constraints, num_inputs, num_vars, log_m, a_ptr_p, a_col_p, a_val_p, b_ptr_p,
                b_col_p, b_val_p, w_p, a_out, b_out, c_out, scalars_mont=True, channel=0):
        """Device pointers throughout; stream-ordered on the channel (index errors surface at the next sync)."""
        self._chk(self.L.dg16_qap(self.h, CURVES[curve], num_constraints, num_inputs, num_vars, log_m, _ptr(a_ptr_p),
                                  _ptr(a_col_p), _ptr(a_val_p), _ptr(b_ptr_p), _ptr(b_col_p), _ptr(b_val_p),
                                  _ptr(w_p), _ptr(a_out), _ptr(b_out), _ptr(c_out),
                                  F_DEVICE_PTRS | (F_SCALARS_MONT if scalars_mont else 0), channel))

    def qap_rows_dev(self, curve, num_constraints, num_inputs, num_vars, log_m, a_ptr_p, a_col_p, a_val_p, b_ptr_p,
                     b_col_p, b_val_p, w_p, row_start, row_stride, a_out, b_out, c_out, scalars_mont=True, channel=0):
        """qap_dev for the rows row_start + row_stride * j only (this rank's cyclic rows), written densely."""
        self._chk(self.L.dg16_qap_rows(self.h, CURVES[curve], num_constraints, num_inputs, num_vars, log_m,
                                       _ptr(a_ptr_p), _ptr(a_col_p), _ptr(a_val_p), _ptr(b_ptr_p), _ptr(b_col_p),
                                       _ptr(b_val_p), _ptr(w_p), row_start, row_stride, _ptr(a_out), _ptr(b_out),
                                       _ptr(c_out), F_DEVICE_PTRS | (F_SCALARS_MONT if scalars_mont else 0), channel))

    # ---- one process per GPU: sharded h-polynomial, distributed prove -------------------------------------------
    def h_poly_dist_dev(self, curve, comm, a_ptr, b_ptr, c_ptr, log_m, out_ptr, channel=0):
        """comm: an object with `.comm_ptr` (RcclComm, TorchComm) or None."""
        self._chk(self.L.dg16_h_poly_dist(self.h, CURVES[curve], comm.comm_ptr if comm is not None else None,
                                          _ptr(a_ptr), _ptr(b_ptr), _ptr(c_ptr), log_m, _ptr(out_ptr), F_DEVICE_PTRS,
                                          channel))

    def ntt_dist_dev(self, curve, comm, in_ptr, out_ptr, log_n, inverse=False, channel=0):
        self._chk(self.L.dg16_ntt_dist(self.h, CURVES[curve], comm.comm_ptr if comm is not None else None, _ptr(in_ptr),
                                       _ptr(out_ptr), log_n, int(inverse), F_DEVICE_PTRS, channel))

    def ntt_dist_stage_dev(self, curve, log_n, rank, n_ranks, inverse, stage, in_ptr, out_ptr, channel=0):
        self._chk(self.L.dg16_ntt_dist_stage(self.h, CURVES[curve], log_n, rank, n_ranks, int(inverse), stage, _ptr(in_ptr),
                                             _ptr(out_ptr), F_DEVICE_PTRS, channel))

    def h_poly_dist_stage_dev(self, curve, log_m, rank, n_ranks, stage, in_ptrs, out_ptr, channel=0):
        arr = (ctypes.c_void_p * 3)(*[int(x) for x in list(in_ptrs) + [0] * (3 - len(in_ptrs))])
        self._chk(self.L.dg16_h_poly_dist_stage(self.h, CURVES[curve], log_m, rank, n_ranks, stage, arr, _ptr(out_ptr),
                                                F_DEVICE_PTRS, channel))

    def groth16_msms_h_dev(self, pk, h_ptr, w_ptr, rs_host, results_ptr, scalars_mont=True):
        rs_host = np.ascontiguousarray(rs_host, dtype=np.uint64)
        self._chk(self.L.dg16_groth16_msms_h(self.h, pk.h, _ptr(h_ptr), _ptr(w_ptr), _ptr(rs_host),
                                             F_DEVICE_PTRS | (F_SCALARS_MONT if scalars_mont else 0), _ptr(results_ptr)))

    def prove_dist_dev(self, pk, comm, a_ptr, b_ptr, c_ptr, w_ptr, rs_host, out_ptr, scalars_mont=True, overlap_tail=False):
        """overlap_tail (DG16_F_OVERLAP_TAIL): H's bucket reduction, the all-gather of the records and the assembly run on
        channel 2's stream; the proof is complete after sync(2)."""
        rs_host = np.ascontiguousarray(rs_host, dtype=np.uint64)
        flags = F_DEVICE_PTRS | (F_SCALARS_MONT if scalars_mont else 0) | (F_OVERLAP_TAIL if overlap_tail else 0)
        self._chk(self.L.dg16_groth16_prove_dist(self.h, pk.h, comm.comm_ptr if comm is not None else None,
                                                 _ptr(a_ptr), _ptr(b_ptr), _ptr(c_ptr), _ptr(w_ptr), _ptr(rs_host),
                                                 flags, _ptr(out_ptr)))

    # ---- Groth16 prover ---------------------------------------------------------------------------------
    def pk_create(self, curve, num_vars, num_inputs, domain_size, a_query, b_g1_query, b_g2_query, h_query,
                  l_query, fixed_points, device_ptrs=False, shard=0, n_shards=1, h_cyclic=False):
        """Makes an arkworks-shaped ProvingKey resident (see include/dg16.h).  Arguments are numpy
        arrays (host) or raw device pointers (device_ptrs=True).  With n_shards > 1 only slice
        `shard` of every MSM range is kept (one process per GPU); h_cyclic: the h bases of the shard are
        h_query[shard + n_shards * j], the output layout of the sharded h-polynomial."""
        h = ctypes.c_void_p()
        args = [a_query, b_g1_query, b_g2_query, h_query, l_query, fixed_points]
        if not device_ptrs:
            args = [np.ascontiguousarray(x, dtype=np.uint64) for x in args]
        self._chk(self.L.dg16_pk_create_shard(self.h, CURVES[curve], num_vars, num_inputs, domain_size,
                                              *[_ptr(x) for x in args], shard, n_shards,
                                              (F_DEVICE_PTRS if device_ptrs else 0) | (F_H_CYCLIC if h_cyclic else 0),
                                              ctypes.byref(h)))
        return ProvingKey(self, h, curve, num_vars, num_inputs, domain_size)

    def results_bytes(self, curve):
        return self.L.dg16_groth16_results_bytes(CURVES[curve])

    def groth16_msms_dev(self, pk, a_ptr, b_ptr, c_ptr, w_ptr, rs_host, results_ptr, scalars_mont=True):
        rs_host = np.ascontiguousarray(rs_host, dtype=np.uint64)
        self._chk(self.L.dg16_groth16_msms(self.h, pk.h, _ptr(a_ptr), _ptr(b_ptr), _ptr(c_ptr), _ptr(w_ptr),
                                           _ptr(rs_host), F_DEVICE_PTRS | (F_SCALARS_MONT if scalars_mont else 0),
                                           _ptr(results_ptr)))

    def groth16_assemble_dev(self, pk, gathered_ptr, n_shards, rs_host, proof_ptr, scalars_mont=True):
        rs_host = np.ascontiguousarray(rs_host, dtype=np.uint64)
        self._chk(self.L.dg16_groth16_assemble(self.h, pk.h, _ptr(gathered_ptr), n_shards, _ptr(rs_host),
                                               F_DEVICE_PTRS | (F_SCALARS_MONT if scalars_mont else 0),
                                               _ptr(proof_ptr)))

    def groth16_msms(self, pk, a, b, c, full_assignment, r, s, scalars_mont=True):
        """Host arrays in, this shard's results record (uint8 array) out."""
        a, b, c, w = (np.ascontiguousarray(v, dtype=np.uint64) for v in (a, b, c, full_assignment))
        rs = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.uint64).reshape(1, 4),
                                                  np.asarray(s, dtype=np.uint64).reshape(1, 4)]))
        out = np.zeros(self.results_bytes(pk.curve), dtype=np.uint8)
        self._chk(self.L.dg16_groth16_msms(self.h, pk.h, _ptr(a), _ptr(b), _ptr(c), _ptr(w), _ptr(rs),
                                           F_SCALARS_MONT if scalars_mont else 0, _ptr(out)))
        return out

    def groth16_assemble(self, pk, gathered, r, s, scalars_mont=True):
        gathered = np.ascontiguousarray(gathered, dtype=np.uint8)
        n_shards = gathered.size // self.results_bytes(pk.curve)
        rs = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.uint64).reshape(1, 4),
                                                  np.asarray(s, dtype=np.uint64).reshape(1, 4)]))
        nl = FQ_LIMBS64[pk.curve]
        out = np.zeros(12 * nl, dtype=np.uint64)
        self._chk(self.L.dg16_groth16_assemble(self.h, pk.h, _ptr(gathered), n_shards, _ptr(rs),
                                               F_SCALARS_MONT if scalars_mont else 0, _ptr(out)))
        return out[:3 * nl].reshape(1, -1), out[3 * nl:9 * nl].reshape(1, -1), out[9 * nl:].reshape(1, -1)

    def prove(self, pk, a, b, c, full_assignment, r, s, scalars_mont=True):
        """Host arrays in, (A, B, C) Jacobian arrays out."""
        a, b, c, w = (np.ascontiguousarray(v, dtype=np.uint64) for v in (a, b, c, full_assignment))
        rs = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.uint64).reshape(1, 4),
                                                  np.asarray(s, dtype=np.uint64).reshape(1, 4)]))
        nl = FQ_LIMBS64[pk.curve]
        out = np.zeros(3 * nl + 6 * nl + 3 * nl, dtype=np.uint64)
        self._chk(self.L.dg16_groth16_prove(self.h, pk.h, _ptr(a), _ptr(b), _ptr(c), _ptr(w), _ptr(rs),
                                            F_SCALARS_MONT if scalars_mont else 0, _ptr(out)))
        return out[:3 * nl].reshape(1, -1), out[3 * nl:9 * nl].reshape(1, -1), out[9 * nl:].reshape(1, -1)

    def prove_dev(self, pk, a_ptr, b_ptr, c_ptr, w_ptr, rs_host, out_ptr, scalars_mont=True, overlap_tail=False):
        """overlap_tail (DG16_F_OVERLAP_TAIL): the proof is complete on channel 2's stream (sync(2)), and the work
        enqueued next on channel 0 -- the next proof of a queue -- starts under this one's last bucket reduction."""
        rs_host = np.ascontiguousarray(rs_host, dtype=np.uint64)
        flags = F_DEVICE_PTRS | (F_SCALARS_MONT if scalars_mont else 0) | (F_OVERLAP_TAIL if overlap_tail else 0)
        self._chk(self.L.dg16_groth16_prove(self.h, pk.h, _ptr(a_ptr), _ptr(b_ptr), _ptr(c_ptr), _ptr(w_ptr),
                                            _ptr(rs_host), flags, _ptr(out_ptr)))

    # ---- device-pointer API (stream-ordered) ----------------------------------------------------------
    def msm_dev(self, curve, group, bases_ptr, scalars_ptr, n, out_ptr, scalars_mont=False, affine=False,
                channel=0, n_scalars=None, in_subgroup=False):
        flags = (F_DEVICE_PTRS | (F_SCALARS_MONT if scalars_mont else 0) | (F_OUT_AFFINE if affine else 0) |
                 (F_BASES_IN_SUBGROUP if in_subgroup else 0))
        self._chk(self.L.dg16_msm(self.h, CURVES[curve], group, _ptr(bases_ptr), _ptr(scalars_ptr), n,
                                  n if n_scalars is None else n_scalars, flags, channel, _ptr(out_ptr)))

    def ntt_dev(self, curve, data_ptr, log_n, inverse=False, coset=None, channel=0):
        coset = None if coset is None else np.ascontiguousarray(coset, dtype=np.uint64)
        self._chk(self.L.dg16_ntt(self.h, CURVES[curve], _ptr(data_ptr), log_n, int(inverse), _ptr(coset),
                                  F_DEVICE_PTRS, channel))

    def h_poly_dev(self, curve, a_ptr, b_ptr, c_ptr, log_m, out_ptr, channel=0):
        self._chk(self.L.dg16_h_poly(self.h, CURVES[curve], _ptr(a_ptr), _ptr(b_ptr), _ptr(c_ptr), log_m,
                                     _ptr(out_ptr), F_DEVICE_PTRS, channel))

    def gen_bases_dev(self, curve, group, seed, n, out_ptr, channel=0):
        self._chk(self.L.dg16_gen_bases(self.h, CURVES[curve], group, seed, n, _ptr(out_ptr), F_DEVICE_PTRS,
                                        channel))

    def field_op_dev(self, curve, kind, op, a_ptr, b_ptr, out_ptr, n, channel=0):
        fid = CURVES[curve] + (16 if kind == "fr" else 0)
        self._chk(self.L.dg16_field_op(self.h, fid, op, _ptr(a_ptr), _ptr(b_ptr), _ptr(out_ptr), n,
                                       F_DEVICE_PTRS, channel))


# ---- transports of the one-process-per-GPU prover (dg16_comm) ---------------------------------------------------
def rccl_unique_id():
    """128 bytes made on rank 0 and handed to the other ranks out of band (e.g. torch.distributed's store)."""
    L = load()
    buf = ctypes.create_string_buffer(128)
    rc = L.dg16_rccl_unique_id(buf)
    if rc != 0:
        raise Dg16Error(rc, L.dg16_rccl_error().decode())
    return buf.raw


class RcclComm:
    """Native RCCL communicator of libdg16 (csrc/rccl_net.hip): `.comm_ptr` for prove_dist / h_poly_dist,
    `.net_ptr` = the same communicator behind the MpcNet vtable (d_fft, d_msm, ... one party per GPU)."""

    def __init__(self, ctx, unique_id, n_ranks, rank):
        self.ctx, self.L = ctx, ctx.L
        h = ctypes.c_void_p()
        rc = self.L.dg16_rccl_create(ctx.h, unique_id, n_ranks, rank, ctypes.byref(h))
        if rc != 0:
            raise Dg16Error(rc, self.L.dg16_rccl_error().decode())
        self.h = h
        self.n_ranks, self.rank = n_ranks, rank
        self.comm_ptr = ctypes.c_void_p(self.L.dg16_rccl_comm(h))
        self.net_ptr = ctypes.c_void_p(self.L.dg16_rccl_net(h))

    def describe(self):
        return "native RCCL (grouped ncclSend/ncclRecv + ncclAllGather, stream-ordered)"

    def ranks(self):
        """(n_ranks, rank) as the communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        n, me = ctypes.c_uint(0), ctypes.c_uint(0)
        rc = self.L.dg16_rccl_ranks(self.h, ctypes.byref(n), ctypes.byref(me))
        if rc != 0:
            raise Dg16Error(rc, self.L.dg16_rccl_error().decode())
        return n.value, me.value

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.L.dg16_rccl_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TorchComm:
    """dg16_comm implemented by the caller with torch.distributed: the `nccl` backend stays stream-ordered (the
    collective is issued under the library's stream), `gloo` stages through host memory -- that one exists so that
    the distributed entry points can be driven by several processes that share ONE GPU (tests), where RCCL refuses
    to form a communicator."""

    def __init__(self, dist, device, n_ranks, rank):
        import torch
        self.torch, self.dist, self.device = torch, dist, device
        self.n_ranks, self.rank = n_ranks, rank
        self.backend = dist.get_backend()
        self.errors = []
        self._cb = (_COMM_N(lambda _s: n_ranks), _COMM_N(lambda _s: rank), _COMM_GATHER(self._all_gather),
                    _COMM_A2A(self._all_to_all))
        self.struct = CommStruct(None, *self._cb)
        self.comm_ptr = ctypes.cast(ctypes.pointer(self.struct), ctypes.c_void_p)

    def describe(self):
        return "torch.distributed (%s)" % self.backend

    def close(self):
        pass

    def _tensor(self, ptr, nbytes):
        class Buf:
            pass
        b = Buf()
        b.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        return self.torch.as_tensor(b, device=self.device)

    def _run(self, stream, fn):
        torch = self.torch
        try:
            ext = torch.cuda.ExternalStream(int(stream), device=self.device)
            if self.backend == "nccl":
                with torch.cuda.stream(ext):
                    fn(False)
            else:
                ext.synchronize()
                fn(True)
                torch.cuda.synchronize(self.device)
            return 0
        except Exception as e:                       # never let an exception cross the C ABI
            self.errors.append(repr(e))
            return 6

    def _all_gather(self, _s, send, nbytes, recv, stream):
        def fn(host):
            src, dst = self._tensor(send, nbytes), self._tensor(recv, nbytes * self.n_ranks)
            if host:
                out = self.torch.empty(nbytes * self.n_ranks, dtype=self.torch.uint8)
                self.dist.all_gather_into_tensor(out, src.cpu())
                dst.copy_(out)
            else:
                self.dist.all_gather_into_tensor(dst, src)
        return self._run(stream, fn)

    def _all_to_all(self, _s, send, recv, per_peer, stream):
        def fn(host):
            n = self.n_ranks
            src, dst = self._tensor(send, per_peer * n), self._tensor(recv, per_peer * n)
            if host:
                # gloo has no all-to-all: gather everything, keep the column addressed to this rank
                allbuf = self.torch.empty(per_peer * n * n, dtype=self.torch.uint8)
                self.dist.all_gather_into_tensor(allbuf, src.cpu())
                dst.copy_(allbuf.view(n, n, per_peer)[:, self.rank, :].reshape(-1))
            else:
                self.dist.all_to_all_single(dst, src)
        return self._run(stream, fn)


def probe_channels(net_struct, n_parties, party_id, dist, soft_s=2.0, hard_s=120.0, rounds=2, device=None):
    """Watchdog for the JOINED form of prove::C (groth16/src/prove.rs:113-125: three d_msm in flight on channels 0 / 1
    / 2, which dg16_prove_c drives from three host threads): before the first proof every party runs a tiny
    gather-to-king + scatter-from-king on each channel of its dg16_net AT ONCE, its three threads started in an order
    that differs from party to party -- the access pattern of the real call.  A transport whose channels are
    independent finishes in a round trip; one whose channels share an ordered pipe (or whose concurrent use deadlocks)
    does not.  The decision is COLLECTIVE (one all-reduce over `dist`'s default group, the pattern of
    parallel.make_prover): if ANY party misses the soft deadline, EVERY party gets "serial" and passes
    DG16_F_SERIAL_CHANNELS from then on (the three d_msm one after another, in the order 0, 1, 2 on every party: same
    proof).  A probe that has not finished by the hard deadline is a dead transport: DG16_ERR_NET on every party.
    -> "joined" | "serial".   net_struct: a NetStruct (TorchNet.struct, or the struct behind dg16_rccl_net).

    device: where the payloads of the net live -- the dg16_net contract is DEVICE pointers plus the stream they were
    produced on for a GPU net (a TorchNet made with a cuda device, the library's RCCL net: ncclSend from pageable host
    memory faults), host pointers and a NULL stream only for a CPU-device TorchNet (the gloo tests).  None / "cpu": host
    buffers; a cuda device: the probe's payloads are torch tensors on it, each channel's thread issues on a stream of its
    own, and the decision tensor goes to the device when the default group's backend is nccl."""
    import threading
    import time
    import torch
    nbytes = 8
    done = [None] * 3
    errs = []
    selfp = getattr(net_struct, "self")          # the vtable's `self` member (NULL for a Python-implemented net)
    on_gpu = device is not None and torch.device(device).type == "cuda"

    def one(c):
        t0 = time.monotonic()
        try:
            if on_gpu:
                torch.cuda.set_device(device)
                stream = torch.cuda.Stream(device=device)
                sptr = ctypes.c_void_p(stream.cuda_stream)
            for it in range(rounds):
                send_h = np.full(nbytes, (16 * c + party_id + it) & 0xFF, dtype=np.uint8)
                back_h = np.repeat(np.arange(n_parties, dtype=np.uint8) + 3 * c + it, nbytes) if party_id == 0 else None
                if on_gpu:
                    with torch.cuda.stream(stream):
                        send = torch.from_numpy(send_h).to(device)
                        gathered = torch.zeros(nbytes * n_parties, dtype=torch.uint8, device=device)
                        back = torch.from_numpy(back_h).to(device) if party_id == 0 else None
                        got = torch.zeros(nbytes, dtype=torch.uint8, device=device)
                    stream.synchronize()
                    rc = net_struct.gather_to_king(selfp, c, send.data_ptr(), nbytes,
                                                   gathered.data_ptr() if party_id == 0 else None, sptr)
                    rc2 = net_struct.scatter_from_king(selfp, c, back.data_ptr() if party_id == 0 else None, nbytes,
                                                       got.data_ptr(), sptr)
                    stream.synchronize()
                    got_h = got.cpu().numpy()
                else:
                    gathered = np.zeros(nbytes * n_parties, dtype=np.uint8)
                    rc = net_struct.gather_to_king(selfp, c, send_h.ctypes.data, nbytes,
                                                   gathered.ctypes.data if party_id == 0 else None, None)
                    got_h = np.zeros(nbytes, dtype=np.uint8)
                    rc2 = net_struct.scatter_from_king(selfp, c, back_h.ctypes.data if party_id == 0 else None, nbytes,
                                                       got_h.ctypes.data, None)
                if rc or rc2 or not bool((got_h == (party_id + 3 * c + it) & 0xFF).all()):
                    errs.append("channel %d: rc %d / %d or a payload of another channel" % (c, rc, rc2))
                    return
            done[c] = time.monotonic() - t0
        except Exception as e:        # noqa: BLE001 -- reported through the collective decision
            errs.append("channel %d: %r" % (c, e))

    order = [(party_id + i) % 3 for i in range(3)]
    ths = [threading.Thread(target=one, args=(c,), daemon=True) for c in order]
    t_start = time.monotonic()
    for t in ths:
        t.start()
    for t in ths:
        t.join(max(0.0, soft_s - (time.monotonic() - t_start)))
    in_time = all(d is not None and d <= soft_s for d in done) and not errs
    for t in ths:                     # a stalled channel: wait it out (its messages must not meet the real protocol's)
        t.join(max(0.0, hard_s - (time.monotonic() - t_start)))
    alive = any(t.is_alive() for t in ths)
    flag_dev = device if (on_gpu and dist.get_backend() == "nccl") else "cpu"    # an nccl group reduces device tensors only
    flag = torch.tensor([0 if (alive or errs) else (2 if in_time else 1)], dtype=torch.int32, device=flag_dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    v = int(flag.item())
    if v == 0:
        raise Dg16Error(6, "channel probe: a channel of the transport never completed (%s)" % (errs or "hard deadline"))
    return "joined" if v == 2 else "serial"


class TorchNet:
    """dg16_net (the MpcNet vtable: mpc-net/src/lib.rs:36-140) implemented by the caller with torch.distributed, one
    process per party, king = rank 0.  ONE PROCESS GROUP PER CHANNEL: the three MultiplexedStreamIDs
    (mpc-net/src/lib.rs:29-33) are independent streams and dg16_prove_c drives them from three host threads at once
    (groth16/src/prove.rs:113-125), in an order that differs from party to party -- a collective of channel c is only
    ever matched against the peers' collectives of channel c.  `nccl` groups stay stream-ordered, `gloo` groups stage
    through host memory (several parties can then share ONE GPU: the multi-process tests).  `before` (optional):
    called as before(channel, op) ahead of every exchange -- the tests inject per-channel delays there."""

    def __init__(self, dist, device, n_parties, party_id, before=None):
        import torch
        self.torch, self.dist, self.device = torch, dist, device
        self.n, self.me, self.before = n_parties, party_id, before
        self.on_host = torch.device(device).type == "cpu"
        self.backend = dist.get_backend()
        ranks = list(range(n_parties))
        self.groups = [dist.new_group(ranks=ranks, backend=self.backend) for _ in range(3)]   # collective: all ranks
        self.errors = []
        self._cb = (_COMM_N(lambda _s: n_parties), _COMM_N(lambda _s: party_id), _NET_COLL(self._gather),
                    _NET_COLL(self._scatter), ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)(lambda _s: 1),
                    _NET_P2P(self._send_to), _NET_P2P(self._recv_from))
        self.struct = NetStruct(None, *self._cb)
        self.net_ptr = ctypes.cast(ctypes.pointer(self.struct), ctypes.c_void_p)

    def describe(self):
        return "torch.distributed (%s), one process group per channel" % self.backend

    def _tensor(self, ptr, nbytes):
        if self.on_host:      # host buffers (the transport's own CPU tests: no GPU involved)
            return self.torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(int(ptr)), dtype=self.torch.uint8)
        return TorchComm._tensor(self, ptr, nbytes)

    def _run(self, channel, op, stream, fn):
        torch = self.torch
        try:
            if not 0 <= channel < 3:
                raise ValueError("channel must be 0, 1 or 2")
            if self.before is not None:
                self.before(channel, op)
            if self.on_host:
                fn(True, self.groups[channel])
                return 0
            ext = torch.cuda.ExternalStream(int(stream), device=self.device) if stream else None
            if self.backend == "nccl" and ext is not None:
                with torch.cuda.stream(ext):
                    fn(False, self.groups[channel])
            else:
                (ext.synchronize() if ext is not None else torch.cuda.synchronize(self.device))
                fn(True, self.groups[channel])
                torch.cuda.synchronize(self.device)
            return 0
        except Exception as e:                       # never let an exception cross the C ABI
            self.errors.append("channel %d %s: %r" % (channel, op, e))
            return 6

    def _gather(self, _s, channel, send, nbytes, recv, stream):          # client_send_or_king_receive, lib.rs:61-99
        def fn(host, group):
            src = self._tensor(send, nbytes)
            src = src.cpu() if host else src
            if self.me == 0:
                parts = [self.torch.empty_like(src) for _ in range(self.n)]
                self.dist.gather(src, gather_list=parts, dst=0, group=group)
                self._tensor(recv, nbytes * self.n).copy_(self.torch.cat(parts))
            else:
                self.dist.gather(src, dst=0, group=group)
        return self._run(channel, "gather", stream, fn)

    def _scatter(self, _s, channel, send, nbytes, recv, stream):         # client_receive_or_king_send, lib.rs:102-140
        def fn(host, group):
            dst = self._tensor(recv, nbytes)
            out = self.torch.empty(nbytes, dtype=self.torch.uint8) if host else dst
            if self.me == 0:
                allbuf = self._tensor(send, nbytes * self.n)
                allbuf = allbuf.cpu() if host else allbuf
                self.dist.scatter(out, scatter_list=list(allbuf.view(self.n, nbytes).unbind(0)), src=0, group=group)
            else:
                self.dist.scatter(out, src=0, group=group)
            if host:
                dst.copy_(out)
        return self._run(channel, "scatter", stream, fn)

    def _send_to(self, _s, peer, channel, send, nbytes, stream):
        def fn(host, group):
            src = self._tensor(send, nbytes)
            self.dist.send(src.cpu() if host else src, dst=peer, group=group)
        return self._run(channel, "send_to", stream, fn)

    def _recv_from(self, _s, peer, channel, recv, nbytes, stream):
        def fn(host, group):
            dst = self._tensor(recv, nbytes)
            out = self.torch.empty(nbytes, dtype=self.torch.uint8) if host else dst
            self.dist.recv(out, src=peer, group=group)
            if host:
                dst.copy_(out)
        return self._run(channel, "recv_from", stream, fn)
