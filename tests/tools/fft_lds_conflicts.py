#!/usr/bin/env python3
"""LDS bank-conflict model of k_ltm_acf2's passes (sla_amd/csrc/sla_kernels.hip), on the lane groups and bank functions that
/opt/skills/guides/MI355X_MICROARCH.md gives for gfx950: a ds_read_b128 is served in 4 groups of 16 lanes
({0-3,12-15,20-27}, {4-11,16-19,28-31}, and the same + 32), 64 banks of 4 bytes; a ds_write_b128 in 8 groups of 8
consecutive lanes, 32 banks.  Every slot is one 16-byte complex number at byte address 16 * sw(c).  For each pass the script
prints the extra cycles per instruction (0 = conflict-free) under a swizzle; `python fft_lds_conflicts.py search` tries the
GF(2)-linear swizzles c ^ (M . c >> 4) of the low nibble and prints the best ones.  Used in round 4 to pick acf_sw."""
import sys
import itertools

READ_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
READ_GROUPS = READ_GROUPS + [[l + 32 for l in g] for g in READ_GROUPS]
WRITE_GROUPS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]


def brev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def cost(slots, groups, nbank_slots):
    """extra cycles of one wave instruction: per lane group, (max number of distinct slots on one bank-quad) - 1"""
    extra = 0
    for g in groups:
        per = {}
        for l in g:
            s = slots[l]
            if s is None:
                continue
            per.setdefault(s % nbank_slots, set()).add(s)
        if per:
            extra += max(len(v) for v in per.values()) - 1
    return extra


def passes(L, threads, need=162):
    """yield (name, kind, [slot index c per lane or None] per wave instruction) for one job; c is the UNswizzled slot"""
    npts = 1 << L
    # first pass: R = 3, thread t < npts/8 writes slots 8 * rev(t) + m
    R = 3
    ng = npts >> R
    for it in range((ng + threads - 1) // threads):
        for w0 in range(0, threads, 64):
            for m in range(8):
                lanes = []
                for l in range(64):
                    t = it * threads + w0 + l
                    lanes.append(8 * brev(t, L - R) + m if t < ng else None)
                yield ("first_pass(write)", "w", lanes)
    def run_stages(log2h, inv, prune):
        left = L - log2h
        while left > 0:
            R = 3 if (left >= 3 and left != 4) else (2 if left >= 2 else 1)
            h = 1 << log2h
            H = h << R
            groups = npts >> R
            pr = prune and H > need
            for it in range((groups + threads - 1) // threads):
                for w0 in range(0, threads, 64):
                    cis = []
                    for l in range(64):
                        tid = w0 + l
                        if pr and h >= need:
                            g = tid + it * threads
                            if g >= (groups // h) * need:
                                cis.append(None); continue
                            blk, low = divmod(g, need)
                        else:
                            b = tid + it * threads
                            if b >= groups:
                                cis.append(None); continue
                            low, blk = b & (h - 1), b >> log2h
                        cis.append((low + (blk << (log2h + R)), low))
                    if all(c is None for c in cis):
                        continue
                    for m in range(1 << R):
                        yield ("%s pass R=%d h=%d (read)" % ("inv" if inv else "fwd", R, h), "r", [None if c is None else c[0] + m * h for c in cis])
                    for m in range(1 << R):
                        yield ("%s pass R=%d h=%d (write)" % ("inv" if inv else "fwd", R, h), "w",
                               [None if c is None or (pr and c[1] + m * h >= need) else c[0] + m * h for c in cis])
            log2h += R
            left -= R
    yield from run_stages(3, False, False)
    # middle: thread tid takes the pairs (c, npts - c), c = tid + k * threads in 1 .. npts/2 - 1 (round 4; round 3 took c = tid + 1 +
    # k * threads, a run that starts one slot off the 16-slot grid), reads both, writes both at their bit-reversed places
    K = max((npts >> 1) // threads, 1)
    for k in range(K):
        for w0 in range(0, threads, 64):
            cs = [w0 + l + k * threads for l in range(64)]
            ok = [1 <= c < npts // 2 for c in cs]
            yield ("middle (read A)", "r", [c if o else None for c, o in zip(cs, ok)])
            yield ("middle (read B)", "r", [npts - c if o else None for c, o in zip(cs, ok)])
    for k in range(K):
        for w0 in range(0, threads, 64):
            cs = [w0 + l + k * threads for l in range(64)]
            ok = [1 <= c < npts // 2 for c in cs]
            yield ("middle (write A, bit-reversed)", "w", [brev(c, L) if o else None for c, o in zip(cs, ok)])
            yield ("middle (write B, bit-reversed)", "w", [brev(npts - c, L) if o else None for c, o in zip(cs, ok)])
    yield from run_stages(0, True, True)


def evaluate(L, threads, sw, verbose=False):
    tot_extra, tot_instr = 0, 0
    by = {}
    for name, kind, lanes in passes(L, threads):
        sl = [None if c is None else sw(c) for c in lanes]
        e = cost(sl, READ_GROUPS if kind == "r" else WRITE_GROUPS, 16 if kind == "r" else 8)
        tot_extra += e
        tot_instr += 1
        a = by.setdefault(name, [0, 0])
        a[0] += e
        a[1] += 1
    if verbose:
        for name, (e, n) in by.items():
            print("  %-40s %6d instr  %6d extra cycles  %.2f per instr" % (name, n, e, e / n))
    return tot_extra, tot_instr


CONFIGS = ((11, 512), (12, 512), (13, 1024))      # the instantiations of k_ltm_acf2 (capacity 2048 / 4096 / 8192 samples per block)


def sw_round3(c):
    return c ^ (((c >> 4) ^ (c >> 8) ^ (c >> 12)) & 15)


def make_sw(rows):
    """bit j of the low nibble ^= parity((c >> 3) & rows[j]); rows[3] has bit 0 clear (bit 3 of c is not a source of itself)"""
    def sw(c):
        hi = c >> 3
        x = 0
        for j in range(4):
            x |= (bin(hi & rows[j]).count("1") & 1) << j
        return c ^ x
    return sw


ROWS4 = [630, 1013, 129, 908]       # ACF_SW_ROW0..3 of sla_kernels.hip
sw_round4 = make_sw(ROWS4)


if __name__ == "__main__":
    for L, threads in CONFIGS:
        e, n = evaluate(L, threads, sw_round3, verbose=True)
        print("L=%d threads=%d round-3 swizzle: %d extra cycles over %d LDS instructions = %.2f per instruction" % (L, threads, e, n, e / n))
    e = sum(evaluate(L, t, sw_round4)[0] for L, t in CONFIGS)
    n = sum(evaluate(L, t, sw_round4)[1] for L, t in CONFIGS)
    for L, threads in CONFIGS:
        e1, n1 = evaluate(L, threads, sw_round4, verbose=(len(sys.argv) > 1 and sys.argv[1] == "verbose"))
        print("L=%d threads=%d round-4 rows %s: %d extra cycles over %d LDS instructions = %.2f per instruction" % (L, threads, ROWS4, e1, n1, e1 / n1))
    if len(sys.argv) > 1 and sys.argv[1] == "search":
        import random
        random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)

        def score(rows):
            return sum(evaluate(L, t, make_sw(rows))[0] for L, t in CONFIGS)
        best = (list(ROWS4), score(ROWS4))
        print("start", best)
        for trial in range(4):
            cur = list(ROWS4) if trial == 0 else [random.getrandbits(10) for _ in range(3)] + [random.getrandbits(9) << 1]
            cs = score(cur)
            for step in range(600):
                cand = list(cur)
                j, b = random.randrange(4), random.randrange(10)
                if j == 3 and b == 0:
                    continue
                cand[j] ^= 1 << b
                sc = score(cand)
                if sc <= cs:
                    cur, cs = cand, sc
            print(trial, cur, cs, flush=True)
            if cs < best[1]:
                best = (cur, cs)
        print("best", best)
