"""TEST INFRASTRUCTURE ONLY -- Groth16 (circom/snarkjs flavour) restated in Python big ints.

  witness_map_from_matrices   /root/reference/ark-circom/src/circom/qap.rs:27-92
  h_query_scalars             ark-circom/src/circom/qap.rs:94-110
  qap / QAP::pss              groth16/src/qap.rs:44-91, 143-187
  ext_wit::h (n-party)        groth16/src/ext_wit.rs:16-101
  A/B/C::compute              groth16/src/prove.rs:21-136
  pack_from_arkworks_pk       groth16/src/proving_key.rs:35-110
  example driver              groth16/examples/sha256.rs:26-95, 97-121, 173-212

`create_proof_with_reduction_and_matrices` and `generate_parameters` live in the reference's
third-party fork of ark-groth16 (branch distributed-groth16, not vendored: SURVEY.md 8(c)); they
are restated from the published Groth16 algorithm and anchored on the call sites
groth16/examples/sha256.rs:133-168 and the base-vector mapping at proving_key.rs:48-65.  The
end-to-end check that does not need arkworks: setup with a KNOWN trapdoor, so A, B, C are known
multiples of the generators and the pairing equation can be checked in the exponent
(`proof_scalars_from_trapdoor`, `verify_in_exponent`).

An R1CS is a dict: num_instance (incl. the constant 1), num_witness, num_constraints and
a, b, c = list of rows, each row a list of (coeff, variable_index).
"""

import random

from .fields import FR
from .curves import CURVES
from .poly import Domain
from .pss import PackedSharingParams
from . import dist


def evaluate_constraint(row, assignment, p):
    return sum(c * assignment[i] for c, i in row) % p


def qap(r1cs, full_assignment, F):
    """groth16/src/qap.rs:44-91 == ark-circom/src/circom/qap.rs:34-62."""
    p = F.p
    ni, nc = r1cs["num_instance"], r1cs["num_constraints"]
    dom = Domain(F, nc + ni)
    m = dom.size
    a = [0] * m
    b = [0] * m
    for i in range(nc):
        a[i] = evaluate_constraint(r1cs["a"][i], full_assignment, p)
        b[i] = evaluate_constraint(r1cs["b"][i], full_assignment, p)
    a[nc:nc + ni] = [x % p for x in full_assignment[:ni]]
    c = [0] * m
    for i in range(nc):
        c[i] = a[i] * b[i] % p
    return a, b, c, dom


def distribute_powers(v, g, p):
    out, o = [], 1
    for x in v:
        out.append(x * o % p)
        o = o * g % p
    return out


def witness_map_from_abc(a, b, c, dom):
    """ark-circom/src/circom/qap.rs:64-91."""
    F, p = dom.F, dom.p
    root = Domain(F, 2 * dom.size).element(1)                     # :67-73
    a = dom.fft(distribute_powers(dom.ifft(a), root, p))          # :64,72,76
    b = dom.fft(distribute_powers(dom.ifft(b), root, p))
    c = dom.fft(distribute_powers(dom.ifft(c), root, p))          # :83-85
    return [(x * y - z) % p for x, y, z in zip(a, b, c)]          # :79,87-89


def witness_map_from_matrices(r1cs, full_assignment, F):
    a, b, c, dom = qap(r1cs, full_assignment, F)
    return witness_map_from_abc(a, b, c, dom)


def h_query_scalars(max_power, t, delta_inverse, F):
    """ark-circom/src/circom/qap.rs:94-110."""
    p = F.p
    scalars = [delta_inverse * pow(t, i, p) % p for i in range(2 * max_power + 1)]
    dom = Domain(F, len(scalars))
    return dom.ifft(scalars)[1::2]


# ---------------------------------------------------------------------------------------------
# setup with known trapdoor
# ---------------------------------------------------------------------------------------------

def setup_scalars(r1cs, F, trapdoor):
    """Per-variable QAP evaluations at tau (LibsnarkReduction::instance_map_with_evaluation,
    reached through ark-circom/src/circom/qap.rs:20-25) and the query scalars of the key."""
    p = F.p
    alpha, beta, gamma, delta, tau = trapdoor
    ni, nw, nc = r1cs["num_instance"], r1cs["num_witness"], r1cs["num_constraints"]
    nv = ni + nw
    dom = Domain(F, nc + ni)
    m = dom.size
    # Lagrange coefficients L_i(tau) on the size-m domain
    zt = (pow(tau, m, p) - 1) % p
    u = []
    for i in range(m):
        wi = dom.element(i)
        u.append(zt * wi % p * F.inv(m * (tau - wi) % p) % p)
    a = [0] * nv
    b = [0] * nv
    c = [0] * nv
    for i in range(ni):
        a[i] = u[nc + i]
    for i in range(nc):
        for coeff, idx in r1cs["a"][i]:
            a[idx] = (a[idx] + u[i] * coeff) % p
        for coeff, idx in r1cs["b"][i]:
            b[idx] = (b[idx] + u[i] * coeff) % p
        for coeff, idx in r1cs["c"][i]:
            c[idx] = (c[idx] + u[i] * coeff) % p
    dinv = F.inv(delta)
    ginv = F.inv(gamma)
    abc = [(beta * a[i] + alpha * b[i] + c[i]) % p for i in range(nv)]
    return dict(
        m=m, zt=zt, a=a, b=b, c=c,
        gamma_abc=[x * ginv % p for x in abc[:ni]],
        l=[x * dinv % p for x in abc[ni:]],
        h=h_query_scalars(m - 1, tau, dinv, F),
        abc=abc,
    )


def setup(curve_name, r1cs, trapdoor):
    """Proving key as affine points (ark_groth16::ProvingKey field names)."""
    F = FR[curve_name]
    g1, g2 = CURVES[curve_name, "g1"], CURVES[curve_name, "g2"]
    alpha, beta, gamma, delta, tau = trapdoor
    s = setup_scalars(r1cs, F, trapdoor)
    G1 = lambda k: g1.mul(g1.gen, k)
    G2 = lambda k: g2.mul(g2.gen, k)
    pk = dict(
        alpha_g1=G1(alpha), beta_g1=G1(beta), beta_g2=G2(beta), delta_g1=G1(delta),
        delta_g2=G2(delta), gamma_g2=G2(gamma),
        a_query=[G1(x) for x in s["a"]],
        b_g1_query=[G1(x) for x in s["b"]],
        b_g2_query=[G2(x) for x in s["b"]],
        h_query=[G1(x) for x in s["h"]],
        l_query=[G1(x) for x in s["l"]],
        gamma_abc_g1=[G1(x) for x in s["gamma_abc"]],
    )
    return pk, s


# ---------------------------------------------------------------------------------------------
# single prover (the value a 1-GPU prove must equal)
# ---------------------------------------------------------------------------------------------

def create_proof(curve_name, pk, r, s, r1cs, full_assignment):
    """Groth16::create_proof_with_reduction_and_matrices (call site
    groth16/examples/sha256.rs:159)."""
    F = FR[curve_name]
    g1, g2 = CURVES[curve_name, "g1"], CURVES[curve_name, "g2"]
    ni = r1cs["num_instance"]
    h = witness_map_from_matrices(r1cs, full_assignment, F)
    w = [x % F.p for x in full_assignment]
    h_acc = g1.msm(pk["h_query"], h)
    l_aux_acc = g1.msm(pk["l_query"], w[ni:])
    g_a = g1.add(g1.add(g1.mul(pk["delta_g1"], r), pk["alpha_g1"]),
                 g1.add(pk["a_query"][0], g1.msm(pk["a_query"][1:], w[1:])))
    if r != 0:
        g1_b = g1.add(g1.add(g1.mul(pk["delta_g1"], s), pk["beta_g1"]),
                      g1.add(pk["b_g1_query"][0], g1.msm(pk["b_g1_query"][1:], w[1:])))
    else:
        g1_b = None
    g2_b = g2.add(g2.add(g2.mul(pk["delta_g2"], s), pk["beta_g2"]),
                  g2.add(pk["b_g2_query"][0], g2.msm(pk["b_g2_query"][1:], w[1:])))
    rs_delta = g1.mul(pk["delta_g1"], r * s % F.p)
    g_c = g1.add(g1.mul(g_a, s), g1.mul(g1_b, r))
    g_c = g1.add(g_c, g1.neg(rs_delta))
    g_c = g1.add(g_c, g1.add(l_aux_acc, h_acc))
    return g_a, g2_b, g_c


def proof_scalars_from_trapdoor(r1cs, F, trapdoor, sc, r, s, full_assignment):
    """Discrete logs of (A, B, C) computed directly in Fr -- independent of every MSM/NTT."""
    p = F.p
    alpha, beta, gamma, delta, tau = trapdoor
    ni = r1cs["num_instance"]
    w = [x % p for x in full_assignment]
    At = sum(x * y for x, y in zip(sc["a"], w)) % p
    Bt = sum(x * y for x, y in zip(sc["b"], w)) % p
    Ct = sum(x * y for x, y in zip(sc["c"], w)) % p
    a = (alpha + At + r * delta) % p
    b = (beta + Bt + s * delta) % p
    hz = (At * Bt - Ct) % p
    aux = sum(x * y for x, y in zip(sc["abc"][ni:], w[ni:])) % p
    c = ((aux + hz) * F.inv(delta) + s * a + r * b - r * s % p * delta) % p
    return a, b, c


def verify_in_exponent(r1cs, F, trapdoor, sc, abc_scalars, full_assignment):
    """e(A,B) = e(alpha,beta) e(sum_pub, gamma) e(C, delta), checked on discrete logs."""
    p = F.p
    alpha, beta, gamma, delta, tau = trapdoor
    ni = r1cs["num_instance"]
    a, b, c = abc_scalars
    pub = sum(x * y for x, y in zip(sc["abc"][:ni], full_assignment[:ni])) % p
    return a * b % p == (alpha * beta + pub + c * delta) % p


# ---------------------------------------------------------------------------------------------
# the reference's n-party path
# ---------------------------------------------------------------------------------------------

def qap_pss(a, b, c, pp):
    """QAP::pss (groth16/src/qap.rs:143-187): per-party (a, b, c) share vectors."""
    sa, sb, sc_ = (dist.share_for_dfft(v, pp) for v in (a, b, c))
    return [(sa[i], sb[i], sc_[i]) for i in range(pp.n)]


def ext_wit_h(qap_shares, dom, pp):
    """ext_wit::h (groth16/src/ext_wit.rs:16-101): per-party packed shares of h."""
    F, p = dom.F, dom.p
    m = dom.size
    dom2 = Domain(F, 2 * m)                                                     # :31-32
    outs = []
    for k in range(3):
        shares = [q[k] for q in qap_shares]
        coeff = dist.d_ifft(shares, True, 2, False, dom, pp)                    # :34-39
        outs.append(dist.d_fft(coeff, False, 1, False, dom2, pp))               # :44-49

    def unpack_shares(v):                                                       # :67-79
        s1 = [x for sh in dist.transpose(v) for x in pp.unpack(sh)]
        for i in range(m):
            j = i * pp.l + pp.t
            s1[i], s1[j] = s1[j], s1[i]
        return s1[:m]                                                           # :83-85

    pe, qe, we = (unpack_shares(v) for v in outs)
    h = [(x * y - z) % p for x, y, z in zip(pe, qe, we)]                        # :88-92
    return dist.transpose(dist.pack_vec(h, pp))                                 # :94


def pack_from_witness(pp, assignment):
    """groth16/examples/sha256.rs:97-121."""
    packed = []
    for i in range(0, len(assignment), pp.l):
        chunk = list(assignment[i:i + pp.l])
        chunk += [0] * (pp.l - len(chunk))
        packed.append(pp.pack_from_public(chunk))
    return dist.transpose(packed)


def pack_proving_key(curve_name, pk, pp):
    """PackedProvingKeyShare::pack_from_arkworks_proving_key (proving_key.rs:35-110):
    s <- a_query[1..], u <- h_query, w <- l_query, h <- b_g1_query[1..], v <- b_g2_query[1..]."""
    g1, g2 = CURVES[curve_name, "g1"], CURVES[curve_name, "g2"]

    def pack(curve, pts):
        packed = [pp.packexp_from_public(curve, pts[i:i + pp.l]) for i in range(0, len(pts), pp.l)]
        return dist.transpose(packed)

    s = pack(g1, pk["a_query"][1:])
    u = pack(g1, pk["h_query"])
    w = pack(g1, pk["l_query"])
    h = pack(g1, pk["b_g1_query"][1:])
    v = pack(g2, pk["b_g2_query"][1:])
    return [dict(s=s[i], u=u[i], w=w[i], h=h[i], v=v[i]) for i in range(pp.n)]


def prove_A(g1, L, N, r, S, a, pp):
    """prove::A::compute, groth16/src/prove.rs:21-46, line by line (S, a: lists of per-party share vectors)."""
    v0 = g1.mul(N, r)                        # :36  Calculate (N)^r
    v1 = g1.add(L, v0)                       # :38  L.(N)^r
    prod = dist.d_msm(g1, S, a, pp)[0]       # :41
    return g1.add(v1, prod)                  # :43


def prove_B(g2, Z, K, s, V, a, pp):
    """prove::B::compute, prove.rs:62-85."""
    v0 = g2.mul(K, s)                        # :76
    v1 = g2.add(Z, v0)                       # :78
    prod = dist.d_msm(g2, V, a, pp)[0]       # :80
    return g2.add(v1, prod)                  # :82


def prove_C(g1, A, M, s, r, W, U, H, a, ax, h, pp):
    """prove::C::compute, prove.rs:106-136 (the three d_msm are joined on channels 0 / 1 / 2 there)."""
    w = dist.d_msm(g1, W, ax, pp)[0]         # :119
    u = dist.d_msm(g1, U, h, pp)[0]          # :121
    hh = dist.d_msm(g1, H, a, pp)[0]         # :123
    v0 = g1.mul(A, s)                        # :128  A^s
    v1 = g1.mul(M, r)                        # :130  M^r
    v2 = g1.mul(hh, r)                       # :132
    return g1.add(g1.add(g1.add(g1.add(w, u), v0), v1), v2)   # :134


def mpc_prove(curve_name, pk, r1cs, full_assignment, l=2, r=0, s=0, L=None, N=None, Z=None, K=None, M=None):
    """groth16/examples/sha256.rs:26-95,173-212.  The example runs r = s = 0 with Default (identity) L, N, Z, K, M
    and completes the proof afterwards (:208-212); with L = alpha_g1 + a_query[0], N = delta_g1,
    Z = beta_g2 + b_g2_query[0], K = delta_g2, M = beta_g1 + b_g1_query[0] the SAME three compute calls give the
    blinded single-prover proof for any r, s (C's r (M + d_msm(H, a)) is r B1 - r s delta_g1)."""
    F = FR[curve_name]
    g1, g2 = CURVES[curve_name, "g1"], CURVES[curve_name, "g2"]
    pp = PackedSharingParams(F, l)
    ni = r1cs["num_instance"]
    a, b, c, dom = qap(r1cs, full_assignment, F)
    qap_shares = qap_pss(a, b, c, pp)
    crs = pack_proving_key(curve_name, pk, pp)
    w = [x % F.p for x in full_assignment]
    ax_shares = pack_from_witness(pp, w[ni:])
    a_shares = pack_from_witness(pp, w[1:])
    h_shares = ext_wit_h(qap_shares, dom, pp)
    col = lambda k: [x[k] for x in crs]      # noqa: E731
    pi_a = prove_A(g1, L, N, r, col("s"), a_shares, pp)                                   # sha256.rs:45-57
    pi_b = prove_B(g2, Z, K, s, col("v"), a_shares, pp)                                   # sha256.rs:59-71
    pi_c = prove_C(g1, pi_a, M, s, r, col("w"), col("u"), col("h"), a_shares, ax_shares, h_shares, pp)   # :73-92
    if r == 0 and s == 0 and L is None and Z is None:
        pi_a = g1.add(pi_a, g1.add(pk["a_query"][0], pk["alpha_g1"]))                # sha256.rs:211
        pi_b = g2.add(pi_b, g2.add(pk["b_g2_query"][0], pk["beta_g2"]))              # sha256.rs:212
    return pi_a, pi_b, pi_c


# ---------------------------------------------------------------------------------------------
# synthetic satisfied R1CS (our own generator)
# ---------------------------------------------------------------------------------------------

def synthetic_r1cs(F, num_constraints, num_instance, num_witness, seed, nnz=3):
    """Random sparse satisfied R1CS: each constraint i is
    (sum a_k w_k) * (sum b_k w_k) = w_out, with w_out a fresh witness variable while fresh
    variables remain, afterwards c = <random row> chosen so that the constraint holds."""
    rng = random.Random(seed)
    p = F.p
    nv = num_instance + num_witness
    w = [1] + [rng.randrange(p) for _ in range(nv - 1)]
    A, B, C = [], [], []
    next_free = num_instance + max(0, num_witness - num_constraints)
    for i in range(num_constraints):
        limit = next_free if next_free < nv else nv
        limit = max(limit, 1)
        ra = [(rng.randrange(1, p), rng.randrange(limit)) for _ in range(nnz)]
        rb = [(rng.randrange(1, p), rng.randrange(limit)) for _ in range(nnz)]
        va = evaluate_constraint(ra, w, p)
        vb = evaluate_constraint(rb, w, p)
        if next_free < nv:
            w[next_free] = va * vb % p
            rc = [(1, next_free)]
            next_free += 1
        else:
            # c-row = k * w_j with k chosen to satisfy the constraint
            j = rng.randrange(1, nv)
            while w[j] == 0:
                j = rng.randrange(1, nv)
            rc = [(va * vb % p * F.inv(w[j]) % p, j)]
        A.append(ra); B.append(rb); C.append(rc)
    r1cs = dict(num_instance=num_instance, num_witness=num_witness,
                num_constraints=num_constraints, a=A, b=B, c=C)
    return r1cs, w


def is_satisfied(r1cs, w, p):
    return all(evaluate_constraint(ra, w, p) * evaluate_constraint(rb, w, p) % p
               == evaluate_constraint(rc, w, p)
               for ra, rb, rc in zip(r1cs["a"], r1cs["b"], r1cs["c"]))
