"""TEST INFRASTRUCTURE ONLY -- pure-Python big-int restatement of the arithmetic the
reference obtains from arkworks (ark-ff / ark-ec / ark-poly 0.4, NOT vendored under
/root/reference).  Nothing in the product path may import this.

Parity status: the reference holds no golden MSM/NTT vectors; the
arithmetic here is pinned by (i) the curve-family parametrisations q(u), r(u), (ii) the
in-tree known answers listed in SURVEY.md section 0 (tests/test_oracle_kats.py),
(iii) definitional identities (generator order, on-curve, DFT by definition) and (iv) the
reference's snarkjs-generated Groth16 proof triple (fixtures/million), which the pairing
verifier built on this arithmetic accepts (oracle/pyref/pairing.py, tests/test_oracle_pairing.py).

Field elements are plain Python ints in [0, p).  Montgomery form (the arkworks in-memory
representation, R = 2^(64*limbs), pinned by /root/reference/ark-circom/src/zkey.rs:417-427)
is only a (de)serialisation detail here.
"""

from dataclasses import dataclass


@dataclass(frozen=True)
class PrimeField:
    name: str
    p: int
    limbs64: int          # number of 64-bit limbs in the arkworks BigInt
    generator: int = 0    # arkworks `GENERATOR` (multiplicative generator), Fr only

    @property
    def R(self):
        return (1 << (64 * self.limbs64)) % self.p

    @property
    def nbytes(self):
        return 8 * self.limbs64

    @property
    def two_adicity(self):
        s, t = 0, self.p - 1
        while t % 2 == 0:
            t //= 2
            s += 1
        return s

    @property
    def two_adic_root(self):
        """arkworks TWO_ADIC_ROOT_OF_UNITY = GENERATOR^((p-1)/2^s)."""
        s = self.two_adicity
        return pow(self.generator, (self.p - 1) >> s, self.p)

    def root_of_unity(self, n):
        """arkworks FftField::get_root_of_unity(n), n a power of two."""
        assert n & (n - 1) == 0 and n > 0
        k = n.bit_length() - 1
        s = self.two_adicity
        assert k <= s
        return pow(self.two_adic_root, 1 << (s - k), self.p)

    def inv(self, a):
        return pow(a, self.p - 2, self.p)

    def to_mont(self, a):
        return a * self.R % self.p

    def from_mont(self, a):
        return a * self.inv(self.R) % self.p

    # -- little-endian byte (de)serialisation, limbs are u64 LE == plain LE bytes
    def to_bytes(self, a, mont=False):
        if mont:
            a = self.to_mont(a)
        return int(a).to_bytes(self.nbytes, "little")

    def from_bytes(self, b, mont=False):
        a = int.from_bytes(b, "little")
        return self.from_mont(a) if mont else a


# --- curve families (q, r verified from the BN / BLS12 parametrisations in
# tests/test_oracle_kats.py) ---------------------------------------------------------------

BN254_Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BLS12_381_Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
BLS12_381_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
BLS12_377_Q = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
BLS12_377_R = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001

FQ = {
    "bn254": PrimeField("bn254_fq", BN254_Q, 4),
    "bls12_381": PrimeField("bls12_381_fq", BLS12_381_Q, 6),
    "bls12_377": PrimeField("bls12_377_fq", BLS12_377_Q, 6),
}
# multiplicative generators: 5 / 7 / 22 (checked to be generators of the
# 2-Sylow part -- i.e. quadratic non-residues -- in tests/test_oracle_kats.py)
FR = {
    "bn254": PrimeField("bn254_fr", BN254_R, 4, generator=5),
    "bls12_381": PrimeField("bls12_381_fr", BLS12_381_R, 4, generator=7),
    "bls12_377": PrimeField("bls12_377_fr", BLS12_377_R, 4, generator=22),
}
CURVE_IDS = {"bn254": 0, "bls12_381": 1, "bls12_377": 2}
