"""CPU distributed Groth16 prover on top of oracle/c/libzkref.so: the "port" timed as bench.py's cpu_baseline
and used as the full-size checker of the GPU proof.  TEST INFRASTRUCTURE / CPU BASELINE ONLY.

Follows groth16/examples/sha256.rs:32-129 per party (ext_wit::circom_h, then A, B-in-G1, B-in-G2, C through
d_msm) with the reference's structure: the n parties' MSMs run concurrently (one thread per party, like the
reference's tokio tasks), the king steps are serial.  In-memory hand-off between parties (no serialization, no
TCP), which favours the CPU number.
"""
import ctypes as C
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .cref import CPss, lib
from .field import Domain


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class CpuProver:
    def __init__(self, curve_name="bn254", l=2):
        self.cp = CPss(curve_name, l)
        o = self.cp.opp
        # coef_p = sum_k unpack2(e_p)[k]: the king's unpack2 + sum (dmsm/mod.rs:85-86) as one linear form
        self.coef = []
        for p in range(o.n):
            e = [0] * o.n
            e[p] = 1
            self.coef.append(sum(o.unpack2(e)) % o.p)
        self.one_q = self.cp.fq.enc([1])[0]

    # ---- small group helpers in C ----
    def _jac1(self, aff):          # affine uint64[8] -> jacobian uint64[12]
        if not aff.any():
            return np.zeros(12, dtype=np.uint64)
        return np.concatenate([aff, self.one_q])

    def _jac2(self, aff):
        if not aff.any():
            return np.zeros(24, dtype=np.uint64)
        return np.concatenate([aff, self.one_q, np.zeros(4, dtype=np.uint64)])

    def _mul(self, g2, P, k):
        out = np.zeros_like(P)
        fn = lib().zkref_g2_mul if g2 else lib().zkref_g1_mul
        ks = self.cp.fr.enc([k])[0]
        fn(C.byref(self.cp.fr.ct), C.byref(self.cp.fq.ct), _p(np.ascontiguousarray(P)), _p(ks), _p(out))
        return out

    def _add(self, g2, P, Q):
        out = np.zeros_like(P)
        fn = lib().zkref_g2_add if g2 else lib().zkref_g1_add
        fn(C.byref(self.cp.fq.ct), _p(np.ascontiguousarray(P)), _p(np.ascontiguousarray(Q)), _p(out))
        return out

    def _neg(self, g2, P):
        q = self.cp.curve.q
        vals = self.cp.fq.dec(P)
        if g2:
            vals[2], vals[3] = (-vals[2]) % q, (-vals[3]) % q
        else:
            vals[1] = (-vals[1]) % q
        return self.cp.fq.enc(vals).reshape(-1)

    # ---- prover ----
    def circom_h(self, qap, log_m, seed, fft_masks=None, degred_mask=None):
        """qap: 3 arrays [n*Lc][4] (copied); returns h shares [n*Lc][4]  (ext_wit.rs:104-181).  fft_masks: six
        (in_mask, out_mask) array pairs [n*Lc][4] or None; degred_mask: one pair or None."""
        cp = self.cp
        m = 1 << log_m
        dom = Domain(cp.curve, m)
        w2m = Domain(cp.curve, 2 * m).element(1)
        Lc = m // cp.l
        ev = []
        for k in range(3):
            x = qap[k].copy()
            mi, mf = (fft_masks[k], fft_masks[3 + k]) if fft_masks else ((None, None), (None, None))
            cp.d_fft_arrays(x, Lc, dom.group_gen_inv, dom.size_inv, w2m, True, mi[0], mi[1], seed + k)
            cp.d_fft_arrays(x, Lc, dom.group_gen, None, None, False, mf[0], mf[1], seed + 3 + k)
            ev.append(x)
        h = cp.mul_sub_arrays(ev[0], ev[1], ev[2])
        dm = degred_mask or (None, None)
        cp.deg_red_arrays(h, Lc, dm[0], dm[1], seed + 6)
        return h

    def prove(self, inp, threads=8, msm_threads=1, tuned_king=0):
        """inp: dict of numpy arrays (see bench.py); returns (A, B, C) Jacobian uint64 arrays and timing.
        threads: parties proved concurrently; msm_threads: window-parallel threads inside each G::msm; tuned_king = T > 0:
        the king's pack / unpack2 as precomputed matrices over T threads (zkref.c zkref_set_fast_king) instead of the
        reference's serial FFT-form king (same shares)."""
        from .cref import lib
        cp = self.cp
        n = cp.n
        lib().zkref_set_fast_king(1 if tuned_king else 0, max(1, tuned_king))
        t0 = time.perf_counter()
        try:
            h = self.circom_h(inp["qap"], inp["log_m"], inp["seed"], inp.get("fft_masks"), inp.get("degred_mask"))
        finally:
            lib().zkref_set_fast_king(0, 1)
        t1 = time.perf_counter()
        Lc = h.shape[0] // n

        def party(p):
            sa = inp["a_share"][p]
            mt = msm_threads
            S = cp.msm_g1_arrays(inp["s"][p], sa, sa.shape[0], mt)
            H = cp.msm_g1_arrays(inp["h"][p], sa, sa.shape[0], mt)
            V = cp.msm_g2_arrays(inp["v"][p], sa, sa.shape[0], mt)
            W = cp.msm_g1_arrays(inp["w"][p], inp["ax_share"][p], inp["ax_share"][p].shape[0], mt)
            U = cp.msm_g1_arrays(inp["u"][p], h[p * Lc:(p + 1) * Lc], Lc, mt)
            return S, H, V, W, U

        with ThreadPoolExecutor(max_workers=threads) as ex:
            res = list(ex.map(party, range(n)))
        t2 = time.perf_counter()
        # king of d_msm: unpack2 over the n points and sum (dmsm/mod.rs:85-86)
        tot = []
        for k in range(5):
            g2 = k == 2
            acc = np.zeros(24 if g2 else 12, dtype=np.uint64)
            for p in range(n):
                acc = self._add(g2, acc, self._mul(g2, res[p][k], self.coef[p]))
            tot.append(acc)
        S, H, V, W, U = tot
        r, s = inp["r"], inp["s_"]
        d1, d2 = self._jac1(inp["delta_g1"]), self._jac2(inp["delta_g2"])
        # prove.rs:40-56, 99-110, 148-158, 229-235
        A = self._add(False, self._add(False, self._add(False, self._jac1(inp["a_query0"]), self._mul(False, d1, r)), S),
                      self._jac1(inp["alpha_g1"]))
        if r % cp.curve.r == 0:
            B1 = np.zeros(12, dtype=np.uint64)
        else:
            B1 = self._add(False, self._add(False, self._add(False, self._jac1(inp["b_g1_query0"]),
                                                            self._mul(False, d1, s)), H), self._jac1(inp["beta_g1"]))
        B2 = self._add(True, self._add(True, self._add(True, self._jac2(inp["b_g2_query0"]), self._mul(True, d2, s)), V),
                       self._jac2(inp["beta_g2"]))
        Cc = self._add(False, self._mul(False, A, s), self._mul(False, B1, r))
        Cc = self._add(False, Cc, self._neg(False, self._mul(False, d1, r * s % cp.curve.r)))
        Cc = self._add(False, self._add(False, Cc, W), U)
        t3 = time.perf_counter()
        return (A, B2, Cc), {"circom_h_s": t1 - t0, "msm_s": t2 - t1, "king_assemble_s": t3 - t2, "total_s": t3 - t0}

    def circom_ref(self, a, b, c, log_m, threads=1):
        """ext_wit.rs:239-285 in C (zkref_circom_ref): a, b, c uint64 [m][4] Montgomery, natural order (copied); returns h [m][4]."""
        from .cref import _domain
        cp = self.cp
        m = 1 << log_m
        dom = Domain(cp.curve, m)
        dt = _domain(cp.fr, dom)
        w2m = Domain(cp.curve, 2 * m).element(1)
        a, b, c = (np.ascontiguousarray(x).copy() for x in (a, b, c))
        h = np.empty_like(a)
        lib().zkref_circom_ref(C.byref(cp.fr.ct), C.byref(dt), _p(a), _p(b), _p(c), cp.fr.mont(w2m), int(threads), _p(h))
        return h

    def prove_local(self, inp, threads=1):
        """BASELINE configs[0]: the LOCAL (non-distributed) prover the reference runs first (groth16/examples/sha256.rs:191-199,
        ark-groth16's create_proof_with_reduction_and_matrices with the circom reduction): h by circom_ref, then the five
        G::msm over the UNPACKED proving key -- h_query . h, l_query . aux, a_query[1..] / b_g1_query[1..] / b_g2_query[1..] .
        assignment[1..] -- and the assembly of prove.rs (restated by oracle/groth16.py create_proof_local, against which
        tests/test_oracle_c.py pins this).  inp: qap_a / qap_b / qap_c [m][4], w [nv][4] (full assignment, Montgomery), ni,
        a_query / b_g1_query / l_query / h_query [..][8], b_g2_query [..][16], the single elements, r, s (ints).
        threads: 1 = everything serial; otherwise the three transforms and then the five MSMs run concurrently, each MSM
        window-parallel over its share of the threads."""
        cp = self.cp
        t0 = time.perf_counter()
        h = self.circom_ref(inp["qap_a"], inp["qap_b"], inp["qap_c"], inp["log_m"], threads)
        t1 = time.perf_counter()
        w, ni = inp["w"], inp["ni"]
        asg, aux = np.ascontiguousarray(w[1:]), np.ascontiguousarray(w[ni:])
        jobs = [("h", False, inp["h_query"], h), ("l", False, inp["l_query"], aux), ("a", False, inp["a_query"][1:], asg),
                ("b1", False, inp["b_g1_query"][1:], asg), ("b2", True, inp["b_g2_query"][1:], asg)]
        per = max(1, threads // len(jobs))

        def run(job):
            _, g2, bases, sc = job
            ln = min(len(bases), len(sc))
            assert len(bases) == len(sc), (job[0], len(bases), len(sc))
            fn = cp.msm_g2_arrays if g2 else cp.msm_g1_arrays
            return fn(np.ascontiguousarray(bases), np.ascontiguousarray(sc), ln, per)
        if threads > 1:
            with ThreadPoolExecutor(max_workers=min(len(jobs), threads)) as ex:
                Hm, Lm, Am, B1m, B2m = list(ex.map(run, jobs))
        else:
            Hm, Lm, Am, B1m, B2m = [run(j) for j in jobs]
        t2 = time.perf_counter()
        r, s = inp["r"], inp["s_"]
        d1, d2 = self._jac1(inp["delta_g1"]), self._jac2(inp["delta_g2"])
        A = self._add(False, self._add(False, self._add(False, self._mul(False, d1, r), self._jac1(inp["a_query"][0])), Am),
                      self._jac1(inp["alpha_g1"]))
        if r % cp.curve.r == 0:
            B1 = np.zeros(12, dtype=np.uint64)
        else:
            B1 = self._add(False, self._add(False, self._add(False, self._mul(False, d1, s), self._jac1(inp["b_g1_query"][0])),
                                            B1m), self._jac1(inp["beta_g1"]))
        B2 = self._add(True, self._add(True, self._add(True, self._mul(True, d2, s), self._jac2(inp["b_g2_query"][0])), B2m),
                       self._jac2(inp["beta_g2"]))
        Cc = self._add(False, self._mul(False, A, s), self._mul(False, B1, r))
        Cc = self._add(False, Cc, self._neg(False, self._mul(False, d1, r * s % cp.curve.r)))
        Cc = self._add(False, self._add(False, Cc, Lm), Hm)
        t3 = time.perf_counter()
        return (A, B2, Cc), {"circom_h_s": t1 - t0, "msm_s": t2 - t1, "assemble_s": t3 - t2, "total_s": t3 - t0}

    def affine(self, P, g2=False):
        """Jacobian uint64 -> canonical affine ints (or None)."""
        v = self.cp.fq.dec(P)
        q = self.cp.curve.q
        if g2:
            from .curve import g2 as G2f
            G = G2f(self.cp.curve)
            return G.to_affine(((v[0], v[1]), (v[2], v[3]), (v[4], v[5])))
        from .curve import g1 as G1f
        G = G1f(self.cp.curve)
        return G.to_affine((v[0], v[1], v[2]))
