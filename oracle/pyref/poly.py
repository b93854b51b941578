"""TEST INFRASTRUCTURE ONLY -- restatement of ark-poly 0.4 `Radix2EvaluationDomain` as used by the
reference (call sites: /root/reference/secret-sharing/src/pss.rs:39-47,88-91,112,123,133,144;
dist-primitives/src/dfft/mod.rs:40,49,78,81; ark-circom/src/circom/qap.rs:34,64-85,105-108).

Conventions restated (ark-poly is not vendored):
  * size = next power of two >= requested, group_gen = get_root_of_unity(size),
    element(i) = offset * group_gen^i;
  * fft:  evals[j] = sum_i coeffs[i] * element(j)^i      (natural order in and out);
  * ifft: exact inverse of fft (includes the size_inv scale and offset^-i);
  * an input shorter than the domain is zero-extended; an input longer than the domain is
    truncated to `size` (Vec::resize in fft_in_place).
Both the O(n^2) definition (`dft_def`) and an O(n log n) iterative form (`ntt`) are provided and
tested against each other.
"""

from .fields import PrimeField


def bitrev_permute(a):
    n = len(a)
    lg = n.bit_length() - 1
    out = list(a)
    for i in range(n):
        j = int(format(i, "0%db" % lg)[::-1], 2) if lg else 0
        if j > i:
            out[i], out[j] = out[j], out[i]
    return out


def ntt(vals, omega, p):
    """In-order radix-2 DIT NTT: out[j] = sum_i vals[i] * omega^(i*j) mod p."""
    n = len(vals)
    assert n & (n - 1) == 0
    a = bitrev_permute([v % p for v in vals])
    length = 2
    while length <= n:
        w_len = pow(omega, n // length, p)
        half = length // 2
        tw = [1] * half
        for k in range(1, half):
            tw[k] = tw[k - 1] * w_len % p
        for start in range(0, n, length):
            for k in range(half):
                x = a[start + k]
                y = a[start + k + half] * tw[k] % p
                a[start + k] = (x + y) % p
                a[start + k + half] = (x - y) % p
        length *= 2
    return a


class Domain:
    """Radix2EvaluationDomain<F> (optionally a coset: `get_coset(offset)`)."""

    def __init__(self, field: PrimeField, n: int, offset: int = 1):
        size = 1
        while size < n:
            size *= 2
        self.F = field
        self.p = field.p
        self.size = size
        self.log_size = size.bit_length() - 1
        self.group_gen = field.root_of_unity(size)
        self.group_gen_inv = field.inv(self.group_gen)
        self.size_inv = field.inv(size % field.p)
        self.offset = offset % field.p
        self.offset_inv = field.inv(self.offset)

    def get_coset(self, offset):
        return Domain(self.F, self.size, offset)

    def element(self, i):
        return self.offset * pow(self.group_gen, i, self.p) % self.p

    def _fit(self, v, zero):
        v = list(v)[: self.size]
        return v + [zero] * (self.size - len(v))

    # ---- field-element transforms --------------------------------------------------------
    def fft(self, coeffs):
        p = self.p
        c = self._fit(coeffs, 0)
        if self.offset != 1:
            o = 1
            for i in range(self.size):
                c[i] = c[i] * o % p
                o = o * self.offset % p
        return ntt(c, self.group_gen, p)

    def ifft(self, evals):
        p = self.p
        e = self._fit(evals, 0)
        c = ntt(e, self.group_gen_inv, p)
        o = self.size_inv
        for i in range(self.size):
            c[i] = c[i] * o % p
            o = o * self.offset_inv % p
        return c

    def fft_def(self, coeffs):
        """O(n^2) definition, for cross-checking `fft`."""
        p = self.p
        c = self._fit(coeffs, 0)
        return [sum(ci * pow(self.element(j), i, p) for i, ci in enumerate(c)) % p
                for j in range(self.size)]

    # ---- group-element transforms (`DomainCoeff` for G: used by unpackexp / packexp,
    #      dist-primitives/src/dmsm/mod.rs:14,38,44,56,59) -- O(n^2) by definition ---------
    def fft_group(self, curve, pts):
        r = self.p
        c = self._fit(pts, None)
        out = []
        for j in range(self.size):
            x = self.element(j)
            acc = curve.to_jac(None)
            for i, P in enumerate(c):
                if P is not None:
                    acc = curve.jadd(acc, curve.jmul(curve.to_jac(P), pow(x, i, r)))
            out.append(curve.to_affine(acc))
        return out

    def ifft_group(self, curve, pts):
        r = self.p
        e = self._fit(pts, None)
        out = []
        for i in range(self.size):
            acc = curve.to_jac(None)
            for j, P in enumerate(e):
                if P is not None:
                    k = pow(self.group_gen_inv, i * j, r) * self.size_inv % r
                    k = k * pow(self.offset_inv, i, r) % r
                    acc = curve.jadd(acc, curve.jmul(curve.to_jac(P), k))
            out.append(curve.to_affine(acc))
        return out
