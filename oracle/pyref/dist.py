"""TEST INFRASTRUCTURE ONLY -- n-party simulations of the reference's dist-primitives, restated
line by line.  A "network round" is modelled the way `LocalTestNet::simulate_network_round`
(/root/reference/mpc-net/src/multi.rs:289-316) presents it: every function takes the list of
per-party inputs (index = party id, party 0 = king) and returns the list of per-party outputs.

  d_msm                     dist-primitives/src/dmsm/mod.rs:70-98
  d_fft / d_ifft            dist-primitives/src/dfft/mod.rs:17-54 / 56-95
  fft1_in_place             dfft/mod.rs:98-140
  fft2_in_place             dfft/mod.rs:142-182
  fft2_with_rearrange_pad   dfft/mod.rs:185-256
  fft_in_place_rearrange    dfft/mod.rs:258-271
  pack_vec / transpose      utils/pack.rs:4-33
  deg_red                   utils/deg_red.rs:10-28
  d_pp                      dpp/mod.rs:17-88
"""

from .pss import PackedSharingParams
from .poly import Domain


def log2(x):
    """ark_std::log2 = ceil(log2(x))."""
    return 0 if x <= 1 else (x - 1).bit_length()


def transpose(m):                                   # utils/pack.rs:18-33
    assert len(m) > 0
    return [list(col) for col in zip(*m)]


def pack_vec(secrets, pp):                          # utils/pack.rs:4-16
    assert len(secrets) % pp.l == 0
    return [pp.pack_from_public(list(secrets[i:i + pp.l])) for i in range(0, len(secrets), pp.l)]


def fft_in_place_rearrange(data):                   # dfft/mod.rs:258-271 (bit reversal)
    data = list(data)
    target = 0
    n = len(data)
    for pos in range(n):
        if target > pos:
            data[target], data[pos] = data[pos], data[target]
        mask = n >> 1
        while target & mask:
            target &= ~mask
            mask >>= 1
        target |= mask
    return data


def fft1_in_place(px, dom: Domain, pp, gen):        # dfft/mod.rs:98-140
    p = dom.p
    px = list(px)
    for i in range(log2(dom.size), log2(pp.l), -1):                 # (log2 l + 1 ..= log2 m).rev()
        poly_size = dom.size // (1 << i)
        factor_stride = pow(gen, 1 << (i - 1), p)
        factor = factor_stride
        for k in range(poly_size):
            for j in range((1 << (i - 1)) // pp.l):
                x = px[(2 * j) * poly_size + k]
                y = px[(2 * j + 1) * poly_size + k] * factor % p
                px[j * (2 * poly_size) + k] = (x + y) % p
                px[j * (2 * poly_size) + k + poly_size] = (x - y) % p
            factor = factor * factor_stride % p
    return px


def fft2_in_place(s1, dom: Domain, pp, gen):        # dfft/mod.rs:142-182
    p = dom.p
    s1 = list(s1)
    s2 = [0] * len(s1)
    for i in range(log2(pp.l), 0, -1):                              # (1..=log2 l).rev()
        poly_size = dom.size // (1 << i)
        factor_stride = pow(gen, 1 << (i - 1), p)
        factor = factor_stride
        for k in range(poly_size):
            for j in range(1 << (i - 1)):
                x = s1[k * (1 << i) + 2 * j]
                y = s1[k * (1 << i) + 2 * j + 1] * factor % p
                s2[k * (1 << (i - 1)) + j] = (x + y) % p
                s2[(k + poly_size) * (1 << (i - 1)) + j] = (x - y) % p
            factor = factor * factor_stride % p
        s1, s2 = s2, s1
    return s1[-1:] + s1[:-1]                                        # rotate_right(1)


def fft2_with_rearrange_pad(px_all, rearrange, pad, degree2, dom, pp, gen):   # dfft/mod.rs:185-256
    mbyl = len(px_all[0])
    all_shares = transpose(px_all)                                  # gather + transpose :204-207
    s1 = [0] * (mbyl * pp.l)
    for i, share in enumerate(all_shares):
        tmp = pp.unpack2(share) if degree2 else pp.unpack(share)    # :211-215
        for j in range(pp.l):
            s1[i * pp.l + j] = tmp[j]
    s1 = fft2_in_place(s1, dom, pp, gen)                            # :222
    if pad > 1:
        s1 = s1 + [0] * ((pad - 1) * len(s1))                       # :225-227
    if rearrange:
        s1 = fft_in_place_rearrange(s1)                             # :231
        stride = len(s1) // pp.l
        out_shares = [pp.pack_from_public(s1[i::stride]) for i in range(stride)]   # :233-243
        return transpose(out_shares)
    return transpose(pack_vec(s1, pp))                              # :247


def d_fft(pcoeff_shares, rearrange, pad, degree2, dom: Domain, pp):           # dfft/mod.rs:17-54
    assert len(pcoeff_shares[0]) * pp.l == dom.size
    local = [fft1_in_place(s, dom, pp, dom.group_gen) for s in pcoeff_shares]
    return fft2_with_rearrange_pad(local, rearrange, pad, degree2, dom, pp, dom.group_gen)


def d_ifft(peval_shares, rearrange, pad, degree2, dom: Domain, pp):           # dfft/mod.rs:56-95
    assert len(peval_shares[0]) * pp.l == dom.size
    p = dom.p
    local = []
    for s in peval_shares:
        s = [x * dom.size_inv % p for x in s]                                 # :78
        local.append(fft1_in_place(s, dom, pp, dom.group_gen_inv))            # :81
    return fft2_with_rearrange_pad(local, rearrange, pad, degree2, dom, pp, dom.group_gen_inv)


def share_for_dfft(x, pp):
    """How the reference's tests/QAP::pss prepare d_fft input (dfft/mod.rs:305-317,
    groth16/src/qap.rs:152-165): bit-reverse, stride-(m/l) l-tuples, pack; returns the
    per-party share vectors (n lists of m/l elements)."""
    x = fft_in_place_rearrange(x)
    m = len(x)
    stride = m // pp.l
    packed = [pp.pack_from_public(x[i::stride]) for i in range(stride)]
    return transpose(packed)


def d_msm(curve, bases_per_party, scalars_per_party, pp):                     # dmsm/mod.rs:70-98
    c_shares = [curve.msm(b, s) for b, s in zip(bases_per_party, scalars_per_party)]   # :82
    output = None
    for P in pp.unpackexp(curve, c_shares, True):                              # :93
        output = curve.add(output, P)
    return [output] * pp.n                                                     # :94


def deg_red(px_all, pp):                                                      # utils/deg_red.rs:10-28
    px_shares = transpose(px_all)
    out = [pp.pack_from_public(pp.unpack2(s)) for s in px_shares]
    return transpose(out)


def d_pp(num_all, den_all, pp):                                               # dpp/mod.rs:17-88
    p = pp.F.p
    s = 1
    sinv = 1
    numden_all = [[x * s % p for x in num] + [x * s % p for x in den]
                  for num, den in zip(num_all, den_all)]                       # :27-34
    shares = transpose(numden_all)                                             # :48
    numden = [v for sh in shares for v in pp.unpack2(sh)]                      # :53-56
    half = len(numden) // 2
    for i in range(half):                                                      # :58-61
        numden[i] = numden[i] * pp.F.inv(numden[i + half]) % p
    numden = numden[:half]                                                     # :63
    for i in range(1, len(numden)):                                            # :66-69
        numden[i] = numden[i] * numden[i - 1] % p
    out = transpose(pack_vec(numden, pp))                                      # :74-79
    out = [[x * sinv % p for x in party] for party in out]                     # :86
    return deg_red(out, pp)                                                    # :87
