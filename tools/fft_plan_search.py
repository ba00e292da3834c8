#!/usr/bin/env python3
"""Pick radices and LDS paddings for a fused plan of a band count M = 2^a 3^b 5^c 7^d (tools/fft_plan_model.py has the
index formulas and the MI355X banking rules; this script searches them).

For every candidate factorisation R (radices the kernels have an in-register DFT for; the last one small, so that the
final pass's stores are long runs of adjacent channels) it chooses the row strides RS_i and the frame stride FS that
minimise the modelled bank conflicts of the passes' ds_read_b64 / ds_write_b64, checks the plan against numpy.fft, and
prints the FastCfg parameter list.   usage: tools/fft_plan_search.py M C NT [max_passes]
"""
import itertools
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from fft_plan_model import conflicts, run_plan, strides  # noqa: E402

RADICES = (2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16)


def factorisations(M, max_passes):
    out = []
    for n in range(2, max_passes + 1):
        for R in itertools.product(RADICES, repeat=n):
            if int(np.prod(R)) == M:
                out.append(list(R))
    return out


def pass_conflicts(M, R, i, RS_i, RS_next, FS, C, NT):
    """(read conflicts of pass i given RS_i, write conflicts of pass i given RS_next), each the sum of (ways - 1)"""
    S, K = strides(R)
    Ri, ipf = R[i], M // R[i]
    items = C * ipf
    rd = wr = 0
    for it in range((items + NT - 1) // NT):
        for wave in range((NT + 63) // 64):
            lanes = [wave * 64 + l + it * NT for l in range(64)]
            if RS_i is not None:
                for n in range(Ri):
                    ad = [None if (w >= items or w - it * NT >= NT) else (w // ipf) * FS + n * RS_i + (w % ipf) for w in lanes]
                    rd += conflicts(ad, "read") - 1
            if RS_next is not None and i + 1 < len(R):
                for k in range(Ri):
                    ad = []
                    for w in lanes:
                        if w >= items or w - it * NT >= NT:
                            ad.append(None)
                            continue
                        fc, item = divmod(w, ipf)
                        kk, rest = divmod(item, S[i])
                        n1, rest2 = divmod(rest, S[i + 1])
                        ad.append(fc * FS + n1 * RS_next + (kk + k * K[i]) * S[i + 1] + rest2)
                    wr += conflicts(ad, "write") - 1
    return rd, wr


def best_plan(M, R, C, NT, pad=10):
    n = len(R)
    ipf = [M // r for r in R]
    best = None
    for FS in range(max(R[i] * ipf[i] for i in range(n)), max(R[i] * ipf[i] for i in range(n)) + 3 * pad):
        RS, total = [], 0
        ok = True
        for i in range(n):
            cands = []
            for rs in range(ipf[i], ipf[i] + pad):
                if R[i] * rs > FS:
                    continue
                rd, _ = pass_conflicts(M, R, i, rs, None, FS, C, NT)
                wr = pass_conflicts(M, R, i - 1, None, rs, FS, C, NT)[1] if i > 0 else 0
                cands.append((rd + wr, rs))
            if not cands:
                ok = False
                break
            c, rs = min(cands)
            RS.append(rs)
            total += c
        if ok and (best is None or (total, FS) < (best[0], best[1])):
            best = (total, FS, RS)
        if best and best[0] == 0:
            break
    return best


if __name__ == "__main__":
    M, C, NT = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    max_passes = int(sys.argv[4]) if len(sys.argv) > 4 else (2 if M <= 160 else 3)
    rows = []
    for R in factorisations(M, max_passes):
        if len(R) == 3 and M <= 160:
            continue
        # in-place non-final passes: one iteration per thread
        if any(C * (M // r) > NT for r in R[:-1]) and "--pingpong" not in sys.argv:
            continue
        b = best_plan(M, R, C, NT)
        if b is None:
            continue
        total, FS, RS = b
        x = np.random.default_rng(1).standard_normal(M) + 1j * np.random.default_rng(2).standard_normal(M)
        err = abs(run_plan(x, R, RS) - np.fft.ifft(x) * M).max()
        assert err < 1e-9 * M, err
        run = M // R[-1]
        rows.append((total, -run, max(R), R, RS, FS))
    rows.sort()
    for total, nrun, _, R, RS, FS in rows[:8]:
        print(f"M={M} R={R} RS={RS} FS={FS} conflicts={total} final-pass run={-nrun} channels")
