#!/usr/bin/env python3
"""Lane-level numpy emulation of the index math in csrc/gemm.hip and csrc/potrf.hip (no GPU in the build
container).  Each helper mirrors a device function line by line: 64-lane vectors stand for a wavefront,
`mfma_f64_16x16x4` follows the gfx950 operand/accumulator lane maps (cdna_hip_programming.md section 3).
Run:  python tools/emulate_kernels.py   (asserts; prints 'emulation ok').

Written before the first GPU run to validate lane maps and LDS layouts.  The GEMM part still describes the kernel's
operand images and accumulator layout (they did not change); the diagonal-factor part models the FIRST version of
potrf_diag128 (dense [128][130] image, substitution TRSM) -- the current kernel (packed 16x16 blocks, in-register
factor, incremental inverse) is validated on the GPU against LAPACK instead (tests/test_gpu_kernels.py)."""
import numpy as np

LANES = np.arange(64)


def mfma(a, b, acc):
    """a[64], b[64] doubles; acc[64,4].  A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D col=l&15,row=(l>>4)+4r."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    A[LANES & 15, LANES >> 4] = a
    B[LANES >> 4, LANES & 15] = b
    D = A @ B
    out = acc.copy()
    for r in range(4):
        out[:, r] += D[(LANES >> 4) + 4 * r, LANES & 15]
    return out


# ---------------------------------------------------------------------------------------------- gemm.hip
BM = BN = 128
BK = 16
MNLD = 144
OPBUF = 2304
KMAJOR, MNMAJOR = 0, 1


def gload(op, P, r0, R, k0, K, tid):
    v = np.zeros((4, 2))
    for i in range(4):
        idx = tid + 256 * i
        if op == KMAJOR:
            row, ch = idx >> 3, idx & 7
            gr, gk = r0 + row, k0 + ch * 2
            if gr < R:
                for e in range(2):
                    if gk + e < K:
                        v[i, e] = P[gr, gk + e]
        else:
            kk, c2 = idx >> 6, idx & 63
            gk, gr = k0 + kk, r0 + c2 * 2
            if gk < K:
                for e in range(2):
                    if gr + e < R:
                        v[i, e] = P[gk, gr + e]
    return v


def sstore(op, s, base, tid, v):
    for i in range(4):
        idx = tid + 256 * i
        if op == KMAJOR:
            row, ch = idx >> 3, idx & 7
            o = row * 16 + ((ch ^ ((row >> 1) & 7)) << 1)
        else:
            kk, c2 = idx >> 6, idx & 63
            o = kk * MNLD + c2 * 2
        s[base + o: base + o + 2] = v[i]


def frag_offsets(op, lane, wbase):
    off = []
    for kq in range(4):
        if op == KMAJOR:
            row = wbase + (lane & 15)
            c0 = (lane >> 5) ^ ((lane & 15) >> 1)
            off.append(row * 16 + ((((kq << 1) ^ c0)) << 1) + ((lane >> 4) & 1))
        else:
            off.append((kq * 4 + (lane >> 4)) * MNLD + wbase + (lane & 15))
    return off


def bank_conflicts(addrs_doubles):
    """ds_read_b64: 64 banks of 4 B, two 32-lane groups; returns max distinct-address multiplicity per bank."""
    worst = 1
    for half in (slice(0, 32), slice(32, 64)):
        a = np.asarray(addrs_doubles[half]) * 8
        banks = {}
        for x in a:
            for w in (0, 4):
                banks.setdefault(((x + w) // 4) % 64, set()).add(x)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def emulate_gemm_tile(opa, opb, m, n, k, rng):
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((k, n))
    Ag = A if opa == KMAJOR else A.T.copy()       # stored [m][k] or [k][m]
    Bg = B.T.copy() if opb == KMAJOR else B       # stored [n][k] or [k][n]
    C = np.zeros((BM, BN))
    smem = np.zeros(4 * OPBUF)
    nk = (k + BK - 1) // BK
    acc = np.zeros((4, 64, 4, 4, 4))  # wave, lane, i, j, r
    for kt in range(nk):
        base = (kt & 1) * 2 * OPBUF
        for tid in range(256):
            sstore(opa, smem, base, tid, gload(opa, Ag, 0, m, kt * BK, k, tid))
            sstore(opb, smem, base + OPBUF, tid, gload(opb, Bg, 0, n, kt * BK, k, tid))
        for wave in range(4):
            wm, wn = wave >> 1, wave & 1
            offA = frag_offsets(opa, LANES, wm * 64)
            offB = frag_offsets(opb, LANES, wn * 64)
            subA = 256 if opa == KMAJOR else 16
            subB = 256 if opb == KMAJOR else 16
            for kq in range(4):
                if kt == 0 and wave == 0:
                    assert bank_conflicts(offA[kq]) == 1, ("A conflict", opa, kq)
                    assert bank_conflicts(offB[kq]) == 1, ("B conflict", opb, kq)
                for i in range(4):
                    a = smem[base + offA[kq] + i * subA]
                    for j in range(4):
                        b = smem[base + OPBUF + offB[kq] + j * subB]
                        acc[wave, :, i, j, :] = mfma(a, b, acc[wave, :, i, j, :])
    for wave in range(4):
        wm, wn = wave >> 1, wave & 1
        for i in range(4):
            for r in range(4):
                rows = wm * 64 + i * 16 + (LANES >> 4) + 4 * r
                for j in range(4):
                    cols = wn * 64 + j * 16 + (LANES & 15)
                    C[rows, cols] = acc[wave, :, i, j, r]
    ref = np.zeros((BM, BN))
    ref[:m, :n] = A @ B
    assert np.allclose(C, ref, rtol=1e-12, atol=1e-12), (opa, opb, m, n, k, np.abs(C - ref).max())


# ---------------------------------------------------------------------------------------------- potrf.hip
NB = 128
DLD = 130


def mma16(S, pa, lda, Sb, pb, ldb, kb, acc):
    for kq in range(4):
        k = kq * 4 + (LANES >> 4)
        a = S[pa + (LANES & 15) * lda + k]
        b = Sb[pb + (LANES & 15) * ldb + k] if kb else Sb[pb + k * ldb + (LANES & 15)]
        acc = mfma(a, b, acc)
    return acc


def emulate_diag128(Ain, nb, do_factor=True):
    A = Ain.copy()
    S = np.zeros(NB * DLD)
    Dg = np.zeros(8 * 256)
    rd = np.zeros(128)
    info = 0
    for r in range(NB):
        for c in range(NB):
            v = 1.0 if r == c else 0.0
            if r < nb and c <= r:
                v = A[r, c]
            S[r * DLD + c] = v
    if do_factor:
        for jj in range(8):
            if jj > 0:
                upd = []
                for wave in range(4):
                    for i in range(jj + wave, 8, 4):
                        acc = np.zeros((64, 4))
                        for p in range(jj):
                            acc = mma16(S, (i * 16) * DLD + p * 16, DLD, S, (jj * 16) * DLD + p * 16, DLD, True, acc)
                        upd.append((i, acc))
                for i, acc in upd:  # writes only touch column block jj, reads only blocks p < jj
                    for r in range(4):
                        S[(i * 16 + (LANES >> 4) + 4 * r) * DLD + jj * 16 + (LANES & 15)] -= acc[:, r]
            # (b) wave 0
            i = LANES & 15
            v = np.stack([S[(jj * 16 + i) * DLD + jj * 16 + c] for c in range(16)], 1)  # [lane, c]
            bad = 0
            for j in range(16):
                d = v[j, j]
                if not d > 0.0:
                    if not bad:
                        bad = j + 1
                    d = 1.0
                rs = 1.0 / np.sqrt(d)
                lij = np.where(i == j, d * rs, v[:, j] * rs)
                v[:, j] = lij
                rd[jj * 16 + j] = rs
                for c in range(j + 1, 16):
                    v[:, c] -= lij * lij[c]
            for lane in range(16):
                for c in range(16):
                    S[(jj * 16 + lane) * DLD + jj * 16 + c] = v[lane, c] if c <= lane else 0.0
            if bad and (jj * 16 + bad) <= nb and info == 0:
                info = jj * 16 + bad
            # (c)
            nrows = NB - (jj + 1) * 16
            for tid in range(nrows):
                row = (jj + 1) * 16 + tid
                pr = row * DLD + jj * 16
                Lj = (jj * 16) * DLD + jj * 16
                x = S[pr:pr + 16].copy()
                for c in range(16):
                    s = x[c]
                    for k in range(c):
                        s -= x[k] * S[Lj + c * DLD + k]
                    x[c] = s * rd[jj * 16 + c]
                S[pr:pr + 16] = x
        for r in range(nb):
            for c in range(r + 1):
                A[r, c] = S[r * DLD + c]
    else:
        for t in range(NB):
            rd[t] = 1.0 / S[t * DLD + t]
    # phase 3
    for wave in range(2):
        jj = wave * 4 + (LANES >> 4)
        c = LANES & 15
        Lj = (jj * 16) * DLD + jj * 16
        x = np.zeros((64, 16))
        for i in range(16):
            s = np.where(i == c, 1.0, 0.0)
            for k in range(i):
                s = s - S[Lj + i * DLD + k] * x[:, k]
            x[:, i] = s * rd[jj * 16 + i]
        for i in range(16):
            Dg[jj * 256 + i * 16 + c] = x[:, i]
    for idx in range(8 * 256):
        jj, i, c = idx >> 8, (idx >> 4) & 15, idx & 15
        S[(jj * 16 + i) * DLD + jj * 16 + c] = Dg[idx]
    # phase 4
    for j in range(6, -1, -1):
        for wave in range(4):
            for i in range(j + 1 + wave, 8, 4):
                acc = mma16(S, (i * 16) * DLD + j * 16, DLD, Dg, j * 256, 16, False, np.zeros((64, 4)))
                for r in range(4):
                    S[(i * 16 + (LANES >> 4) + 4 * r) * DLD + j * 16 + (LANES & 15)] = acc[:, r]
        res = []
        for wave in range(4):
            for i in range(j + 1 + wave, 8, 4):
                acc = np.zeros((64, 4))
                for k in range(j + 1, i + 1):
                    acc = mma16(S, (i * 16) * DLD + k * 16, DLD, S, (k * 16) * DLD + j * 16, DLD, False, acc)
                res.append((i, acc))
        for i, acc in res:
            for r in range(4):
                S[(i * 16 + (LANES >> 4) + 4 * r) * DLD + j * 16 + (LANES & 15)] = -acc[:, r]
    Dinv = np.zeros((NB, NB))
    for r in range(NB):
        for c in range(r + 1):
            Dinv[r, c] = S[r * DLD + c]
    return A, Dinv, info


def main():
    rng = np.random.default_rng(0)
    for opa in (KMAJOR, MNMAJOR):
        for opb in (KMAJOR, MNMAJOR):
            emulate_gemm_tile(opa, opb, 128, 128, 32, rng)
            emulate_gemm_tile(opa, opb, 100, 77, 21, rng)
    print("gemm tile emulation ok (all operand layouts, guards, conflict-free operand reads)")
    for nb in (128, 100, 16, 1):
        B = rng.standard_normal((nb, nb))
        Spd = B @ B.T + nb * np.eye(nb)
        A, Dinv, info = emulate_diag128(Spd, nb)
        L = np.linalg.cholesky(Spd)
        assert info == 0
        assert np.allclose(np.tril(A), L, rtol=1e-11, atol=1e-11), nb
        X = np.eye(NB)
        X[:nb, :nb] = np.linalg.inv(L)
        assert np.allclose(Dinv, X, rtol=1e-9, atol=1e-10), (nb, np.abs(Dinv - X).max())
        # inverse-only entry
        Lp = np.tril(L)
        _, Dinv2, _ = emulate_diag128(Lp, nb, do_factor=False)
        assert np.allclose(Dinv2, X, rtol=1e-9, atol=1e-10)
    bad = -np.eye(40)
    _, _, info = emulate_diag128(bad, 40)
    assert info == 1
    M = np.eye(40)
    M[17, 17] = -2.0
    _, _, info = emulate_diag128(M, 40)
    assert info == 18, info
    print("potrf_diag128 emulation ok (factor, inverse, padding, pivot failure index)")
    print("emulation ok")


if __name__ == "__main__":
    main()
