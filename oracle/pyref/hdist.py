"""TEST INFRASTRUCTURE ONLY -- the sharded h-polynomial of the one-process-per-GPU prover, stage by stage, on
Python integers.  It restates what csrc/ntt.hip (h_poly_dist_launch) does on N ranks, so that (a) the index algebra
is checked on CPU against the single-prover witness_map (oracle/pyref/groth16.py, ark-circom/src/circom/qap.rs:64-91)
and (b) tests/test_parallel_gloo.py can run the exchange logic of distributed-groth16_amd/parallel.py with the gloo
backend and no GPU.

The reference's own distributed transform is d_fft / d_ifft (dist-primitives/src/dfft/mod.rs:17-95): local butterfly
levels on every party (fft1_in_place :98-140), gather to the king, the remaining levels there (fft2_in_place
:142-182), scatter.  On one MI355X node the king disappears: the m-point transform is split as m = N x M (N ranks),
every rank does M-point transforms locally and the N-point cross-rank part after ONE all-to-all per transform
(xGMI is point to point: an all-to-all uses all 7 links of a GPU at once, a gather would serialise on the king's).

Layouts (m = N M, S = M / N, rank rho):
  rows_in    a[N j + rho], j < M            (cyclic rows of the QAP evaluation vectors: each rank computes its own)
  stage 0    Y = iNTT_M(rows_in)             ->  piece sigma (S elements) goes to rank sigma          [all-to-all 1]
  stage 1    Z[i1][j] (from rank i1), k2 = rho S + j:
               t[i1]  = Z[i1][j] * w_m^(-i1 k2)
               c[k1]  = (1 / N) sum_i1 t[i1] w_N^(-i1 k1)        coefficient M k1 + k2 of the polynomial
               c[k1] *= g^(M k1 + k2),  g = w_2m                  (the shift of qap.rs:67-76)
               u[q]   = sum_k1 c[k1] w_N^(k1 q)
               U[q][j] = u[q] * w_m^(q k2)        ->  piece q goes to rank q                         [all-to-all 2]
  stage 2    W[k2] (k2 < M, natural),  X = NTT_M(W):  X[j] = evaluation  rho + N j  of the shifted polynomial
             h[j] = A[j] B[j] - C[j]  = the single prover's h[rho + N j]
"""

from .poly import Domain, ntt


def rows_of(v, rank, n_ranks):
    return v[rank::n_ranks]


def stage0(rows, F, m, n_ranks):
    M = m // n_ranks
    dom = Domain(F, M)
    return dom.ifft(rows)                      # includes 1 / M


def stage1(Z, F, m, n_ranks, rank):
    """Z[i1] = the S elements received from rank i1.  Returns U[q] = the S elements for rank q."""
    p = F.p
    N, M = n_ranks, m // n_ranks
    S = M // N
    w = Domain(F, m).group_gen
    w_inv = F.inv(w)
    g = Domain(F, 2 * m).group_gen
    n_inv = F.inv(N)
    wN, wN_inv = pow(w, M, p), pow(w_inv, M, p)
    U = [[0] * S for _ in range(N)]
    for j in range(S):
        k2 = rank * S + j
        t = [Z[i1][j] * pow(w_inv, i1 * k2, p) % p for i1 in range(N)]
        c = [sum(t[i1] * pow(wN_inv, i1 * k1, p) for i1 in range(N)) * n_inv % p for k1 in range(N)]
        c = [c[k1] * pow(g, M * k1 + k2, p) % p for k1 in range(N)]
        for q in range(N):
            u = sum(c[k1] * pow(wN, k1 * q, p) for k1 in range(N)) % p
            U[q][j] = u * pow(w, q * k2, p) % p
    return U


def stage2(Wa, Wb, Wc, F, m, n_ranks):
    p = F.p
    M = m // n_ranks
    dom = Domain(F, M)
    A, B, C = dom.fft(Wa), dom.fft(Wb), dom.fft(Wc)
    return [(x * y - z) % p for x, y, z in zip(A, B, C)]


def all_to_all(pieces):
    """pieces[src][dst] -> received[dst][src] (what RCCL's grouped send/recv does)."""
    n = len(pieces)
    return [[pieces[src][dst] for src in range(n)] for dst in range(n)]


def h_poly_sharded(a, b, c, F, n_ranks):
    """All ranks in one process: returns h_shards[rank][j] = h[rank + N j]."""
    m = len(a)
    N = n_ranks
    M = m // N
    S = M // N
    assert N * N <= m and m % (N * N) == 0
    W = []
    for v in (a, b, c):
        Y = [stage0(rows_of(v, r, N), F, m, N) for r in range(N)]
        Z = all_to_all([[Y[r][s * S:(s + 1) * S] for s in range(N)] for r in range(N)])
        U = [stage1(Z[r], F, m, N, r) for r in range(N)]
        R = all_to_all(U)
        W.append([[x for piece in R[r] for x in piece] for r in range(N)])
    return [stage2(W[0][r], W[1][r], W[2][r], F, m, N) for r in range(N)]


# ---- one transform with one all-to-all (dg16_ntt_dist) ---------------------------------------------------------------
def ntt_sharded(x, F, n_ranks, inverse=False):
    """All ranks in one process.  Rank rho starts with x[N j + rho]; returns out[rho][k1 S + j] = X[M k1 + rho S + j]
    (X = the m-point transform of x; inverse: inverse root and 1 / m), the layout csrc/ntt.hip: ntt_dist_stage leaves."""
    p, m, N = F.p, len(x), n_ranks
    M, S = m // N, m // N // N
    dom_m, dom_M = Domain(F, m), Domain(F, M)
    w = dom_m.group_gen_inv if inverse else dom_m.group_gen
    Y = [(dom_M.ifft(x[r::N]) if inverse else dom_M.fft(x[r::N])) for r in range(N)]
    Z = all_to_all([[Y[r][s * S:(s + 1) * S] for s in range(N)] for r in range(N)])
    n_inv = F.inv(N) if inverse else 1
    wN = pow(w, M, p)
    out = []
    for rho in range(N):
        o = [0] * M
        for j in range(S):
            k2 = rho * S + j
            t = [Z[rho][i1][j] * pow(w, i1 * k2, p) % p for i1 in range(N)]
            for k1 in range(N):
                o[k1 * S + j] = sum(t[i1] * pow(wN, i1 * k1, p) for i1 in range(N)) * n_inv % p
        out.append(o)
    return out
