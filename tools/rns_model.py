"""NumPy model of the residue-number-system (RNS) contraction used by the fp64-emulation prototype
("emulate_fp64": int8 MFMA with exact int32 accumulation; Ozaki scheme II, Ozaki / Uchino / Imamura 2025, adapted so
that the accumulator STAYS in residue form across all panel updates and is reconstructed once per column).

Pins the constants the HIP kernels use (moduli, CRT weights) and checks the float64 reconstruction recipe against exact
Python integers.  Test tooling: nothing here is imported by the package.

Scheme: values are held as fixed-point integers v = rint(x * 2^B), B = 52 (exact in float64).  The 14 moduli (P ~ 2^109.9)
are enough because every contraction of the path multiplies two ROWS of Euclidean norm <= 1/2 in these units (rows of the
Cholesky factor: sum_k L_ck^2 = Ky_cc; rows of L^-1 k*: the posterior variance is >= 0), so |X| <= 2^102 by Cauchy-Schwarz
whatever the contraction length; the self-test draws its operands accordingly and checks |X| < P/2.  For pairwise
coprime moduli p_l <= 256 the residues v mod p_l (symmetric, int8) of two operands are multiplied and summed exactly in
int32 (|r| <= 128, K <= 1023 per launch: < 2^24), reduced mod p_l and added to an int8 residue accumulator.  After all
updates the exact integer X = sum_k a_k b_k (|X| < P/2) follows from its residues by the CRT in "fraction" form:
    X / P  =  centred_frac( sum_l w_l / p_l ),   w_l = (r_l * q_l) mod p_l,   q_l = (P / p_l)^-1 mod p_l
evaluated with two float64 accumulators: H collects each term rounded to a multiple of 2^-44 (exact sums), L the
remainders (|.| < 2^-44, double-double reciprocal of p_l), so the centred fraction is good to ~2^-95 absolute.
"""
import numpy as np

MODULI = [253, 251, 249, 247, 245, 241, 239, 233, 229, 227, 223, 211, 199, 197]
B = 52


def crt_constants(moduli=MODULI):
    P = 1
    for p in moduli:
        P *= p
    q = [pow(P // p, -1, p) for p in moduli]
    return P, q


def to_fixed(x):
    v = np.rint(np.asarray(x, dtype=np.float64) * 2.0 ** B)
    assert np.all(np.abs(v) < 2.0 ** 53)
    return v


def residues(v, moduli=MODULI):
    """symmetric residues as int8 planes [t, ...] computed the way the kernel does (float64 fma-free form)."""
    out = np.empty((len(moduli),) + v.shape, dtype=np.int8)
    for l, p in enumerate(moduli):
        q = np.rint(v / p)
        r = v - q * p            # exact: |r| <= p/2 up to the rounding of v/p, fixed below
        assert np.abs(r).max() <= 127      # moduli <= 253: no fix-up step needed
        out[l] = r.astype(np.int8)
    return out


def split_recip(p):
    ih = 1.0 / p
    # il = 1/p - ih in float64 via exact rational arithmetic
    from fractions import Fraction
    il = float(Fraction(1, p) - Fraction(ih))
    return ih, il


def reconstruct(R, moduli=MODULI):
    """R: int residue planes [t, ...] of X -> float64 X / P (centred fraction), two-accumulator recipe."""
    P, q = crt_constants(moduli)
    H = np.zeros(R.shape[1:])
    L = np.zeros(R.shape[1:])
    MAGIC = 1.5 * 2.0 ** (52 - 44)
    for l, p in enumerate(moduli):
        w = (R[l].astype(np.float32) * np.float32(q[l]))             # |r q| < 2^15: exact in f32
        k = np.rint(w * np.float32(1.0 / p))
        w = (w - k * np.float32(p)).astype(np.float64)               # representative of (r q) mod p, |w| <= p
        ih, il = split_recip(p)
        t1 = w * ih
        hi = (t1 + MAGIC) - MAGIC
        lo = (w * ih - hi)                                           # numpy has no fma: emulate with exact rationals below
        from fractions import Fraction
        # exact fma(w, ih, -hi) + w * il for the model (the kernel uses two hardware fma)
        lo = np.vectorize(lambda ww, hh: float(Fraction(ww) * Fraction(ih) - Fraction(hh) + Fraction(ww) * Fraction(il)))(w, hi)
        H += hi
        L += lo
    f = (H - np.rint(H)) + L
    f = f - np.rint(f)
    return f, P


def selftest(seed=0, M=6, N=5, K=4096):
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, (M, K)) * rng.choice([1.0, 1e-3, 1e-9], (M, K))
    Bm = rng.uniform(-1, 1, (N, K))
    # rows of norm exactly 1/2 (the extreme case the bound allows); one pair of parallel rows reaches |X| = 2^102
    A *= 0.5 / np.linalg.norm(A, axis=1, keepdims=True)
    Bm *= 0.5 / np.linalg.norm(Bm, axis=1, keepdims=True)
    Bm[0] = A[0]
    va, vb = to_fixed(A), to_fixed(Bm)
    ra, rb = residues(va), residues(vb)
    # check residues against exact integers
    for l, p in enumerate(MODULI):
        ok = all((int(va[i, k]) - int(ra[l, i, k])) % p == 0 for i in range(M) for k in range(0, K, 97))
        assert ok, p
        assert np.abs(ra[l].astype(int)).max() <= 127
    # residue accumulation, 1023 columns per "launch" (int32-exact), accumulator kept as symmetric int8
    t = len(MODULI)
    R = np.zeros((t, M, N), dtype=np.int64)
    for k0 in range(0, K, 1023):
        for l, p in enumerate(MODULI):
            acc = ra[l, :, k0:k0 + 1023].astype(np.int64) @ rb[l, :, k0:k0 + 1023].astype(np.int64).T
            assert np.abs(acc).max() < 2 ** 24
            s = (R[l] + acc) % p
            R[l] = np.where(s > (p - 1) // 2, s - p, s)
            assert np.abs(R[l]).max() <= 127
    f, P = reconstruct(R)
    X = f * float(P) * 2.0 ** (-2 * B)
    exact = np.array([[sum(int(va[i, k]) * int(vb[j, k]) for k in range(K)) for j in range(N)] for i in range(M)], dtype=object)
    assert max(abs(int(x)) for x in exact.ravel()) < P // 2          # Cauchy-Schwarz: <= (2^51 + sqrt(K)/2)^2
    exact_f = np.array([[float(exact[i, j]) * 2.0 ** (-2 * B) for j in range(N)] for i in range(M)])
    ref = A @ Bm.T
    err_rns = np.max(np.abs(X - exact_f) / np.maximum(np.abs(exact_f), 1e-300))
    err_f64 = np.max(np.abs(ref - exact_f))
    print("K=%d: RNS reconstruct vs exact fixed-point product: rel %.2e; float64 dot vs the same: abs %.2e; |X|max %.3g"
          % (K, err_rns, err_f64, np.max(np.abs(exact_f))))
    assert err_rns < 1e-15
    return err_rns


if __name__ == "__main__":
    P, q = crt_constants()
    import math
    print("t = %d moduli, log2 P = %.2f, q =" % (len(MODULI), math.log2(P)), q)
    for K in (768, 4096, 16384):
        selftest(K=K)
