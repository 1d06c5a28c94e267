"""Small seeded synthetic inputs for the test-suite (numpy; bench-scale data comes from libfemhost's C generator)."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = {65: 84, 67: 71, 71: 67, 84: 65}


def rand_seq(rng, n):
    return ACGT[rng.integers(0, 4, size=n)].tobytes()


def revcomp(s):
    return bytes(COMP.get(c, 78) for c in reversed(s))


def mutate(rng, s, n_err):
    """n_err random edits (60% substitution, 20% insertion, 20% deletion) at interior offsets."""
    s = bytearray(s)
    for _ in range(n_err):
        pos = int(rng.integers(1, max(2, len(s) - 1)))
        r = rng.random()
        if r < 0.6:
            s[pos] = ACGT[(np.searchsorted(ACGT, s[pos]) + 1 + rng.integers(0, 3)) % 4] if s[pos] in b"ACGT" else 65
        elif r < 0.8:
            s.insert(pos, int(ACGT[rng.integers(0, 4)]))
        else:
            del s[pos]
    return bytes(s)


def make_reads(rng, seqs, n, L, e, rev_frac=0.5, n_rate=0.0, fixed_err=None):
    """Reads sampled from `seqs` (list of bytes) with 0..e edits, truncated to L."""
    out = []
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    ok = np.nonzero(lens > L + 2 * e + 2)[0]
    for _ in range(n):
        si = int(ok[rng.integers(0, len(ok))])
        start = int(rng.integers(0, lens[si] - (L + e) - 1))
        w = seqs[si][start:start + L + e]
        k = int(rng.integers(0, e + 1)) if fixed_err is None else fixed_err
        r = mutate(rng, w, k)[:L]
        if len(r) < L:
            r = r + rand_seq(rng, L - len(r))
        if n_rate > 0:
            r = bytearray(r)
            for i in np.nonzero(rng.random(L) < n_rate)[0]:
                r[int(i)] = 78
            r = bytes(r)
        if rng.random() < rev_frac:
            r = revcomp(r)
        out.append(r)
    return out


def repeat_rich_reference(rng, n_seq=3, unit_len=400, n_units=12, copies=40, spacer=300, n_run=50):
    """Multi-sequence reference where a few units recur (with light divergence) many times, plus N runs."""
    units = [rand_seq(rng, unit_len) for _ in range(n_units)]
    seqs = []
    for _ in range(n_seq):
        parts = [rand_seq(rng, spacer)]
        for _ in range(copies):
            u = units[int(rng.integers(0, n_units))]
            u = mutate(rng, u, int(rng.integers(0, 4)))
            parts.append(u)
            if rng.random() < 0.15:
                parts.append(b"N" * n_run)
            parts.append(rand_seq(rng, int(rng.integers(5, spacer))))
        seqs.append(b"".join(parts))
    return seqs
