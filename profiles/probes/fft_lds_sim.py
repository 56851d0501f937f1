"""LDS bank-conflict model of the fft kernel's accesses (MI355X_MICROARCH.md, LDS table): ds_read_b128 is served in 4 groups
of 16 lanes, bank = 16-byte slot mod 16; ds_write_b128 in 8 groups of 8 contiguous lanes, slot mod 8.  Cost of a group =
the largest number of distinct slots on one bank.  Prints LDS-array cycles per wave-instruction, per access kind."""
import sys
RG = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
RG += [[l+32 for l in g] for g in RG]
WG = [list(range(8*g, 8*g+8)) for g in range(8)]
def cost(slots, groups, mod):
    c = 0
    for g in groups:
        banks = {}
        for l in g:
            if slots[l] is None: continue
            banks.setdefault(slots[l] % mod, set()).add(slots[l])
        c += max((len(v) for v in banks.values()), default=0)
    return c
def fold(j, w):
    r, j2 = 0, j >> w
    while j2: r ^= j2 & ((1 << w) - 1); j2 >>= w
    return j ^ r
def brev(i, n): return int(format(i, '0%db' % n)[::-1], 2) if n else 0
def run(log2n, NT, dpad, tpad, B=1):
    N = 1 << log2n; total = N * B
    res = {"scatter_w": [0,0], "data_r": [0,0], "data_w": [0,0], "tw_r": [0,0], "final_r": [0,0]}
    def acc(kind, slots, wr):
        c = cost(slots, WG if wr else RG, 8 if wr else 16)
        res[kind][0] += c; res[kind][1] += 1
    for w0 in range(0, min(NT, 256), 64):
        for q in range(total // NT):
            idx = [w0 + l + NT*q for l in range(64)]
            acc("scatter_w", [dpad((i & ~(N-1)) + brev(i & (N-1), log2n)) for i in idx], True)
            acc("final_r", [dpad(i) for i in idx], False)
        st = 0
        while st < log2n:
            R = 3 if st + 3 <= log2n else log2n - st
            E = 1 << R; half = 1 << st; per = N >> R
            for t0 in range(w0, total // E, NT):
                ts = [t0 + l if t0 + l < total // E else None for l in range(64)]
                def base(t):
                    tb, tt = t >> (log2n - R), t & (per - 1)
                    return (tb << log2n) + ((tt >> st) << (st + R)) + (tt & (half - 1)), tt & (half - 1)
                for q in range(E):
                    sl = [None if t is None else dpad(base(t)[0] + q*half) for t in ts]
                    acc("data_r", sl, False); acc("data_w", sl, True)
                for r in range(R):
                    shift = log2n - 1 - st - r
                    for q in range(E):
                        if q & (1 << r): continue
                        sl = [None if t is None else tpad((((q & ((1 << r) - 1)) << st) + base(t)[1]) << shift) for t in ts]
                        acc("tw_r", sl, False)
            st += R
    return {k: (round(v[0] / max(v[1], 1), 2)) for k, v in res.items()}
pads = {"i+(i>>3)": lambda i: i + (i >> 3), "none": lambda i: i, "fold4": lambda i: fold(i, 4), "fold3": lambda i: fold(i, 3),
        "fold4+pad": lambda i: fold(i, 4) + (i >> 3) }
for log2n, NT, B in ((11, 256, 1), (10, 256, 1), (12, 1024, 1), (8, 512, 8), (6, 512, 32)):
    print("N", 1 << log2n, "NT", NT, "B", B, "(ideal: reads 4, writes 8)")
    for dn in ("i+(i>>3)", "fold4", "fold3"):
        for tn in ("none", "fold4", "fold3"):
            print("  data", dn, "tw", tn, run(log2n, NT, pads[dn], pads[tn], B))
