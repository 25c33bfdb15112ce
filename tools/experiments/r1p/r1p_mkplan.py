#!/usr/bin/env python3
"""Source plans for csrc/socmx_rollout1p.hip (SOCMX_R1P_PLAN0 / 1 / 2): spreads a role's stream (S) and LDS (L) blocks evenly over
its program, keeps the DPP-form blocks resident, at most one LDS block per unit (pair of blocks), checks the LDS budget.
    python3 tools/r1p_mkplan.py  R L S   Rb Lb Sb      (waves 0..6 | wave 7, the books, incl. res_0; counts of R, L, S blocks)"""
import sys

def spread(n, counts):
    """n slots, counts = {'L': a, 'S': b}; the rest 'R'; evenly interleaved."""
    out = ['R'] * n
    tot = sum(counts.values())
    # place the non-R types at evenly spaced slots, alternating types proportionally
    slots = [int((i + 0.5) * n / tot) for i in range(tot)] if tot else []
    kinds = []
    acc = {k: 0.0 for k in counts}
    for i in range(tot):
        for k in counts:
            acc[k] += counts[k] / tot
        k = max(acc, key=acc.get)
        acc[k] -= 1.0
        kinds.append(k)
    for s, k in zip(slots, kinds):
        out[s] = k
    return out

def units(role):
    # blocks in consumption order (csrc/socmx_rollout1p.h): units = sets of blocks multiplied together (one LDS landing block per unit)
    return [(1, 2), (3, 4), (5, 6), (7,), (8, 9), (10, 11), (12,), (13, 14), (15, 16), (17, 18), (19, 20)]

def make(role, R, L, S):
    n = 22 if role == 0 else 23
    fixed = {0: [0, 21], 1: [0, 21, 22]}[role]
    free = [b for b in range(n) if b not in fixed]
    assert R + L + S == n and R >= len(fixed), (role, R, L, S, n)
    assert S % 2 == 0 and S >= 2
    body = spread(len(free), {'L': L, 'S': S})
    plan = ['R'] * n
    for b, k in zip(free, body):
        plan[b] = k
    # no two LDS blocks in one unit
    for u in units(role):
        ls = [b for b in u if plan[b] == 'L']
        while len(ls) > 1:
            moved = False
            for b in free:
                if plan[b] == 'R' and not any(b in v and any(plan[c] == 'L' for c in v) for v in units(role)):
                    plan[b], plan[ls[-1]] = 'L', 'R'
                    ls.pop()
                    moved = True
                    break
            assert moved, "cannot separate the LDS blocks"
    p = ''.join(plan)
    assert p.count('R') == R and p.count('L') == L and p.count('S') == S, (p, R, L, S)
    for u in units(role):
        assert sum(plan[b] == 'L' for b in u) <= 1, p
    return p

if __name__ == "__main__":
    v = [int(x) for x in sys.argv[1:7]]
    ps = [make(r, *v[3 * r:3 * r + 3]) for r in range(2)]
    lds = 7 * v[1] + v[4]
    stream = 7 * v[2] + v[5]
    assert lds <= 31, f"LDS blocks {lds} > 31"
    print(' '.join(ps))
    print(f"# LDS {lds} blocks (<= 33), stream {stream} blocks = {stream * 4} KB per step", file=sys.stderr)
