"""arkworks compressed `ProvingKey<Bn254>` / `VerifyingKey<Bn254>` files -- what the reference's service writes at
setup and reloads on every proof request (`pk.serialize_with_mode(.., Compress::Yes)`,
`ProvingKey::deserialize_with_mode(.., Compress::Yes, Validate::No)`: mpc-api/src/main.rs:154-171, :459-512).

Thin composition of two native pieces of libdg16: `dg16_arkkey_layout` (host code: walks the container) and the batched
GPU point codec `dg16_points_compress` / `dg16_points_decompress` (csrc/ark_codec.hip: one lane per point, the same
routines as the proof.bin codec).  A Rust shim does the same in ~30 lines.  The arrays that come out
are the ones `Context.pk_create` takes: affine x || y Montgomery limbs, identity = zeros."""

import ctypes
import struct

import numpy as np

from . import lib as _lib

G1_FIELDS = ("alpha_g1", "beta_g1", "delta_g1")
G2_FIELDS = ("beta_g2", "gamma_g2", "delta_g2")
G1_VECS = {"gamma_abc_g1": ("off_ic", "n_ic"), "a_query": ("off_a", "n_a"), "b_g1_query": ("off_b1", "n_b1"),
           "h_query": ("off_h", "n_h"), "l_query": ("off_l", "n_l")}


class ArkKeyError(ValueError):
    pass


def layout(data, verifying_key_only=False):
    L = _lib.load()
    data = bytes(data)
    out = _lib.ArkKeyLayout()
    if L.dg16_arkkey_layout(data, len(data), int(verifying_key_only), ctypes.byref(out)) != 0:
        raise ArkKeyError(L.dg16_codec_error().decode())
    return {n: getattr(out, n) for n, _ in _lib.ArkKeyLayout._fields_}


def _read(ctx, data, vk_only, validate):
    lay = layout(data, vk_only)
    data = bytes(data)
    key = {}

    def pts(group, off, n):
        return ctx.points_decompress("bn254", group, data[off:off + n * 32 * group], validate=validate)

    key["alpha_g1"] = pts(1, lay["off_alpha_g1"], 1)
    for name in G2_FIELDS:
        key[name] = pts(2, lay["off_" + name], 1)
    key["gamma_abc_g1"] = pts(1, lay["off_ic"], lay["n_ic"])
    if not vk_only:
        key["beta_g1"] = pts(1, lay["off_beta_g1"], 1)
        key["delta_g1"] = pts(1, lay["off_delta_g1"], 1)
        for name in ("a_query", "b_g1_query", "h_query", "l_query"):
            off, cnt = G1_VECS[name]
            key[name] = pts(1, lay[off], lay[cnt])
        key["b_g2_query"] = pts(2, lay["off_b2"], lay["n_b2"])
    return key


def read_proving_key(ctx, data, validate=False):
    """-> dict of uint64 arrays (affine Montgomery limbs): alpha_g1, beta_g1, delta_g1, beta_g2, gamma_g2, delta_g2,
    gamma_abc_g1, a_query, b_g1_query, b_g2_query, h_query, l_query.  validate=False is the reference's Validate::No
    (the square roots are still taken: an x off the curve is an error); True adds the G2 subgroup checks."""
    return _read(ctx, data, False, validate)


def read_verifying_key(ctx, data, validate=True):
    """The reference deserialises a VerifyingKey with Validate::Yes (mpc-api/src/main.rs:207-210, :505; only the
    ProvingKey takes Validate::No), hence the default.  Stricter than arkworks 0.4 in one corner: an infinity-flagged
    encoding whose x bits are not zero is rejected here (ark-serialize accepts it as the identity)."""
    return _read(ctx, data, True, validate)


def _vec(ctx, group, arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 8 * group)
    return struct.pack("<Q", arr.shape[0]) + ctx.points_compress("bn254", group, arr)


def write_verifying_key(ctx, key):
    c = lambda g, name: ctx.points_compress("bn254", g, key[name])      # noqa: E731
    return c(1, "alpha_g1") + c(2, "beta_g2") + c(2, "gamma_g2") + c(2, "delta_g2") + _vec(ctx, 1, key["gamma_abc_g1"])


def write_proving_key(ctx, key):
    c = lambda g, name: ctx.points_compress("bn254", g, key[name])      # noqa: E731
    return (write_verifying_key(ctx, key) + c(1, "beta_g1") + c(1, "delta_g1") + _vec(ctx, 1, key["a_query"])
            + _vec(ctx, 1, key["b_g1_query"]) + _vec(ctx, 2, key["b_g2_query"]) + _vec(ctx, 1, key["h_query"])
            + _vec(ctx, 1, key["l_query"]))


def resident_key(ctx, key, num_inputs, domain_size, shard=0, n_shards=1):
    """`Context.pk_create` from a key read by `read_proving_key` (the mapping of groth16/src/proving_key.rs:48-65)."""
    fixed = np.concatenate([key["alpha_g1"].reshape(-1), key["beta_g1"].reshape(-1), key["delta_g1"].reshape(-1),
                            key["beta_g2"].reshape(-1), key["delta_g2"].reshape(-1)])
    return ctx.pk_create("bn254", key["a_query"].shape[0], num_inputs, domain_size, key["a_query"], key["b_g1_query"],
                         key["b_g2_query"], key["h_query"], key["l_query"], fixed, shard=shard, n_shards=n_shards)
