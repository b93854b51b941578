"""TEST INFRASTRUCTURE ONLY -- PackedSharingParams, restating
/root/reference/secret-sharing/src/pss.rs:13-148 line by line (n = 4l, t = l-1; share domain of
size n, secret domain = size-(l+t+1) coset with offset F::GENERATOR, secret2 = size-2(l+t+1)
coset, pss.rs:34-62)."""

from .fields import PrimeField
from .poly import Domain


class PackedSharingParams:
    def __init__(self, field: PrimeField, l: int):
        self.F = field
        self.l = l
        self.n = 4 * l                      # pss.rs:36
        self.t = l - 1                      # pss.rs:37
        assert self.n == 2 * (self.t + l + 1)
        self.share = Domain(field, self.n)                                           # :39
        self.secret = Domain(field, l + self.t + 1).get_coset(field.generator)        # :40-43
        self.secret2 = Domain(field, 2 * (l + self.t + 1)).get_coset(field.generator) # :44-47
        assert self.share.size == self.n
        assert self.secret.size == l + self.t + 1
        assert self.secret2.size == 2 * (l + self.t + 1)

    # pss.rs:86-92
    def pack_from_public(self, secrets):
        assert len(secrets) == self.l, "Secrets length mismatch"
        coeffs = self.secret.ifft(secrets)       # interpolate on the secrets domain
        return self.share.fft(coeffs)            # evaluate on the share domain

    # pss.rs:110-127
    def unpack(self, shares):
        coeffs = self.share.ifft(shares)
        evals = self.secret.fft(coeffs)          # fft_in_place truncates to |secret| coeffs
        return evals[: self.l]

    # pss.rs:131-148
    def unpack2(self, shares):
        coeffs = self.share.ifft(shares)
        evals = self.secret2.fft(coeffs)
        return evals[0: 2 * self.l: 2]

    # ---- "in the exponent" (dist-primitives/src/dmsm/mod.rs:7-68) ------------------------
    def unpackexp(self, curve, shares, degree2):
        coeffs = self.share.ifft_group(curve, shares)          # dmsm/mod.rs:14
        if degree2:
            ev = self.secret2.fft_group(curve, coeffs)          # :38
            return ev[0: 2 * self.l: 2]                         # :39-43
        ev = self.secret.fft_group(curve, coeffs)               # :44
        return ev[: self.l]                                     # :45

    def packexp_from_public(self, curve, secrets):
        coeffs = self.secret.ifft_group(curve, secrets)         # :56
        return self.share.fft_group(curve, coeffs)              # :59
