"""CPU prototype (numpy, f64) of the two-stage tridiagonalisation exactly as the HIP kernels of
eigh_twostage.hip implement it: dense -> band (CholeskyQR2 panels + Householder reconstruction, two-sided
block updates), band -> tridiagonal (bulge chasing, first-column elimination), back-transformation
Z = Q1 Q2 Y.  Developer tool: index conventions are validated here before they go on the GPU."""
import numpy as np


def chol_upper(g):
    return np.linalg.cholesky(g).T


def panel_qr_hr(P):
    """P (m x b) -> V (m x b unit lower trapezoidal), T (b x b upper), Rt (b x b upper) with
    (I - V T V^T)^T P = [Rt; 0].  CholeskyQR2 then Householder reconstruction (LU of Q - [D; 0])."""
    m, b = P.shape
    R1 = chol_upper(P.T @ P)
    Q1 = P @ np.linalg.inv(R1)
    R2 = chol_upper(Q1.T @ Q1)
    R2i = np.linalg.inv(R2)
    Q = Q1 @ R2i
    R = R2 @ R1
    # LU without pivoting of (Q - [D; 0]), D_ii = -sign of the pivot candidate
    Wk = Q[:b].copy()
    D = np.zeros(b)
    L = np.eye(b)
    U = np.zeros((b, b))
    for i in range(b):
        D[i] = -1.0 if Wk[i, i] >= 0 else 1.0
        Wk[i, i] -= D[i]
        U[i, i:] = Wk[i, i:]
        L[i + 1:, i] = Wk[i + 1:, i] / Wk[i, i]
        Wk[i + 1:, i + 1:] -= np.outer(L[i + 1:, i], Wk[i, i + 1:])
    Ui = np.linalg.inv(U)
    V = np.empty_like(P)
    V[:b] = L
    V[b:] = Q[b:] @ Ui            # = Q1[b:] @ (R2i @ Ui)
    # Q D - [I; 0] = V (U D)  and  = -V T V1^T  ->  T = -(U D) V1^-T
    T = -(U * D[None, :]) @ np.linalg.inv(L).T
    Rt = D[:, None] * R
    return V, T, Rt


def dense_to_band(A, b):
    """Returns (B dense with bandwidth b, list of (row0, V, T))."""
    A = A.copy()
    n = A.shape[0]
    refl = []
    j0 = 0
    while n - j0 - b >= 2:
        r0 = j0 + b
        m = n - r0
        bb = min(b, m)      # a short last panel (m < b) still gives an upper-trapezoidal R
        P = A[r0:, j0:j0 + b]
        if m >= b:
            V, T, Rt = panel_qr_hr(P)
            A[r0:r0 + b, j0:j0 + b] = Rt
            A[r0 + b:, j0:j0 + b] = 0.0
        else:
            raise NotImplementedError("n must be a multiple of b")
        A[j0:j0 + b, r0:] = A[r0:, j0:j0 + b].T
        A22 = A[r0:, r0:]
        W0 = A22 @ V
        Y1 = W0 @ T
        Z = V.T @ Y1
        X = Y1 - 0.5 * V @ (T.T @ Z)
        A22 -= V @ X.T + X @ V.T
        refl.append((r0, V, T))
        j0 += b
    return A, refl


def house(x):
    """v (v[0] = 1), tau, beta with (I - tau v v^T) x = beta e1 (LAPACK dlarfg convention)."""
    alpha = x[0]
    xn = np.linalg.norm(x[1:])
    if xn == 0.0:
        return np.concatenate([[1.0], np.zeros(len(x) - 1)]), 0.0, alpha
    beta = -np.copysign(np.hypot(alpha, xn), alpha)
    tau = (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1.0
    return v, tau, beta


def band_to_tridiag(B, b):
    """Bulge chasing on a dense copy (only band + bulge entries are touched).  Returns d, e, V2 (n x n, column s =
    stacked reflectors of sweep s), tau2 (n x npos)."""
    A = B.copy()
    n = A.shape[0]
    npos = (n + b - 1) // b + 1
    V2 = np.zeros((n, n))
    tau2 = np.zeros((n, npos))
    for s in range(n - 2):
        r = s + 1
        ln = min(b, n - r)
        if ln < 2:
            continue
        v, tau, beta = house(A[r:r + ln, s].copy())
        A[r, s] = beta; A[r + 1:r + ln, s] = 0.0
        A[s, r] = beta; A[s, r + 1:r + ln] = 0.0
        V2[r:r + ln, s] = v; tau2[s, 0] = tau
        D = A[r:r + ln, r:r + ln]
        w = tau * (D @ v)
        w -= 0.5 * tau * (w @ v) * v
        D -= np.outer(v, w) + np.outer(w, v)
        p = 1
        while True:
            c0, lc = r, ln                  # previous block's rows = this block's columns
            r = c0 + lc
            ln = min(b, n - r)
            if ln <= 0:
                break
            Bk = A[r:r + ln, c0:c0 + lc]
            Bk -= np.outer(Bk @ v, tau * v)            # right-apply the previous reflector
            if ln >= 2:
                vn, taun, beta = house(Bk[:, 0].copy())
                Bk[0, 0] = beta; Bk[1:, 0] = 0.0
                Bk[:, 1:] -= np.outer(taun * vn, vn @ Bk[:, 1:])   # left-apply to the other columns
            else:
                vn, taun = np.ones(1), 0.0
            A[c0:c0 + lc, r:r + ln] = Bk.T
            v, tau = vn, taun
            V2[r:r + ln, s] = v; tau2[s, p] = tau
            D = A[r:r + ln, r:r + ln]
            w = tau * (D @ v)
            w -= 0.5 * tau * (w @ v) * v
            D -= np.outer(v, w) + np.outer(w, v)
            p += 1
    return np.diag(A).copy(), np.diag(A, -1).copy(), V2, tau2, A


def apply_q2(V2, tau2, b, Y):
    """Y <- Q2 Y, Q2 = product of the bulge-chasing reflectors in generation order."""
    Y = Y.copy()
    n = V2.shape[0]
    for s in range(n - 3, -1, -1):
        r, p = s + 1, 0
        while r < n:
            ln = min(b, n - r)
            v = V2[r:r + ln, s]
            Y[r:r + ln] -= np.outer(tau2[s, p] * v, v @ Y[r:r + ln])
            r += ln; p += 1
    return Y


def apply_q1(refl, Y):
    Y = Y.copy()
    for r0, V, T in reversed(refl):
        Y[r0:] -= V @ (T @ (V.T @ Y[r0:]))
    return Y


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n, b in [(128, 32), (256, 32), (192, 16)]:
        y = rng.standard_normal((2 * n, n)) * np.logspace(0, -2, n)
        A = y.T @ y / (2 * n)
        A += np.eye(n) * 0.01 * np.trace(A) / n
        Bd, refl = dense_to_band(A, b)
        assert np.abs(np.tril(Bd, -b - 1)).max() < 1e-14, "band"
        assert np.abs(np.linalg.eigvalsh(Bd) - np.linalg.eigvalsh(A)).max() < 1e-13
        d, e, V2, tau2, At = band_to_tridiag(Bd, b)
        assert np.abs(np.tril(At, -2)).max() < 1e-14
        Tm = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        w, Yt = np.linalg.eigh(Tm)
        assert np.abs(w - np.linalg.eigvalsh(A)).max() < 1e-13
        k = n // 4
        Z = apply_q1(refl, apply_q2(V2, tau2, b, Yt[:, n - k:]))
        res = np.abs(A @ Z - Z * w[n - k:]).max()
        orth = np.abs(Z.T @ Z - np.eye(k)).max()
        print(n, b, "residual", res, "orth", orth)
        assert res < 1e-13 and orth < 1e-13
    print("ok")
